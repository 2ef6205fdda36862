// Internal declarations shared by the kernel translation units and neko_capi.hip.
#pragma once
#include "neko_common.h"

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  long lda, ldb;
  int M, N, K;
  float alpha;
  const float* alpha_dev; // optional device scalar multiplied into alpha (autograd grad_output) or null
  const float* bias;      // [N] or null
  const float* resid;     // f32 [M, ldr] or null
  long ldr;
  const bf16_t* act_in;   // bf16 [M, ldact] pre-activation (act == 2)
  long ldact;
  bf16_t* pre_out;        // bf16 [M, ldpre] pre-activation store (act == 1) or null
  long ldpre;
  int act;                // 0 none, 1 gelu forward, 2 multiply by gelu'(act_in)
  float* Cf;              // f32 out or null
  long ldcf;
  int accumulate;         // Cf += result
  bf16_t* Cb;             // bf16 out or null
  long ldcb;
  int splitk;             // >1: each grid.y slice handles k_per_split of K, atomicAdd into Cf
  int k_per_split;        // multiple of 64
  float* splitk_ws;       // optional [splitk, M, N] f32 workspace: slices reduced in fixed order (no atomics)
  int drop_thr;           // residual dropout on (acc*alpha + bias [act]) before the residual add: 0 = off, else round(p*256)
  unsigned drop_key;      // site key (element index = row*N + col)
  float drop_scale;       // 256/(256-thr)
  float* colsum_ws;       // optional: per-(32*TM)-row-band column sums of the result (before bf16 rounding), f32 [bands, N];
                          // written only when the launch can fold them (gemm_glds.hip), see neko_gemm_glds_colsum_bands()
  int group_m;            // row panels per rasterisation group (tile_coords); 0 = the compiled default.  Set by neko_gemm_glds_try
  int epi_lock;           // 1: the output phase runs under the CU's lock (gemm_glds.hip, co-resident workgroups).  Set by neko_gemm_glds_try
};

int neko_gemm_bf16_impl(GemmArgs a, int a_kstrided, int b_kstrided, int safe_transpose, hipStream_t s);
int neko_gemm_bf16_full(GemmArgs a, int a_kstrided, int b_kstrided, int safe_transpose, hipStream_t s);
int neko_gather_rows_bf16_impl(const bf16_t* src, const int* idx, bf16_t* dst, int n, int npad, int d, hipStream_t s);
int neko_scatter_rows_f32_impl(const float* src, const int* idx, float* dst, int n, int d, hipStream_t s);
int neko_splitk_reduce_impl(const float* ws, int S, int M, int N, float* C, long ldc, int accumulate, hipStream_t s);
int neko_gemm_glds_try(const GemmArgs& a, int a_kstrided, int b_kstrided, hipStream_t s);   // 1 = not applicable
int neko_gemm_a16_try(const GemmArgs& a, int a_kstrided, int b_kstrided, hipStream_t s);    // 1 = not applicable (gemm_a16.hip)
int neko_gemm_set_mainloop_impl(int mode);
int neko_gemm_mainloop_mode();
// gemm_b16.hip (two workgroups per CU): 1 = not applicable; mainloop_mode = neko_gemm_mainloop_mode()
int neko_gemm_b16_try(const GemmArgs& a, int a_kstrided, int b_kstrided, int mainloop_mode, int* colsum_bands, hipStream_t s);
// gemm_p16.hip (two waves per SIMD, role-alternating; round 6): 1 = not applicable; *colsum_bands = 128-row bands filled
int neko_gemm_p16_try(const GemmArgs& a, int a_kstrided, int b_kstrided, int mainloop_mode, int* colsum_bands, hipStream_t s);
// which main loop served the calling thread's last neko_gemm_bf16 launch: 0 gemm_glds (32 x 32 x 16 loop), 1 gemm_a16, 2 gemm_b16,
// 3 gemm_glds64 (8-wave loop, whole-line A slots), 4 gemm_bf16 (register-staged fallback), 5 gemm_p16; -1 = none yet
extern thread_local int g_neko_last_mainloop;
// bands of colsum_ws the last neko_gemm_glds_try() of this thread filled (0: the column sums were not folded into it)
int neko_gemm_glds_colsum_bands();
int neko_colsum_bands_reduce_impl(const float* ws, int bands, int N, float* out, hipStream_t s);   // out[N] += sum over bands, fixed order
int neko_layernorm_fwd_impl(const float* x, const float* gamma, const float* beta, bf16_t* y16, float* y32,
                            float* mean, float* rstd, int M, int d, float eps, hipStream_t s);
int neko_layernorm_bwd_blocks_impl(int M);
int neko_layernorm_bwd_impl(const void* dy, int dy_is_bf16, const float* x, const float* gamma, const float* mean,
                            const float* rstd, const float* g_in, float* dx, bf16_t* dx16, float* dgamma,
                            float* dbeta, int accumulate, float* workspace, int M, int d, int drop_thr,
                            unsigned drop_key, float drop_scale, float* dcolsum16, hipStream_t s, const int* dy_map = nullptr);
int neko_dropout_f32_impl(const float* x, float* y, long n, int thr, unsigned key, float scale, hipStream_t s);
int neko_attn_fwd_impl(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T,
                       int H, int hd, int drop_thr, unsigned drop_key, float drop_scale, uint32_t* dmask, hipStream_t s);
long neko_attn_mask_dwords_impl(int B, int T, int H, int hd);
long neko_attn_res_mask_dwords(int B, int T, int H, int hd);
int neko_attn_bwd_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* kbias, const int* kstart,
                       const float* lse, float* D, int* qflags, bf16_t* dqkv, int B, int T, int H, int hd,
                       int drop_thr, unsigned drop_key, float drop_scale, const uint32_t* dmask, hipStream_t s);
int neko_attn_set_path_impl(int mode);
int neko_attn_bwd_reproducible_impl(int on);
bool neko_attn_res_applicable(int T, int hd);
bool neko_attn_stream_applicable(int T, int hd);
int neko_attn_bwd_stream_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* kbias, const int* kstart,
                              const float* lse, float* D, bf16_t* dqkv, int B, int T, int H, int hd, int drop_thr,
                              unsigned drop_key, float drop_scale, hipStream_t s, const int* seq_off = nullptr, long rows = 0);
int neko_attn_fwd_stream_impl(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T,
                              int H, int hd, int drop_thr, unsigned drop_key, float drop_scale, hipStream_t s,
                              const int* seq_off = nullptr);
int neko_attn_fwd_res_impl(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T,
                           int H, int drop_thr, unsigned drop_key, float drop_scale, uint32_t* dmask, hipStream_t s,
                           const int* seq_off = nullptr, const long long* mask_off = nullptr);
int neko_attn_bwd_res_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* kbias, const int* kstart,
                           const float* lse, float* D, bf16_t* dqkv, int B, int T, int H, int drop_thr, unsigned drop_key,
                           float drop_scale, const uint32_t* dmask, hipStream_t s, const int* seq_off = nullptr,
                           const long long* mask_off = nullptr);
int neko_gemv_bf16_impl(const bf16_t* x, long ldx, const bf16_t* W, long ldw, int b_kstrided, int M, int N, int K,
                        const float* bias, const float* resid, long ldr, int act, float* Cf, long ldcf, bf16_t* Cb,
                        long ldcb, hipStream_t s);
int neko_attn_decode_impl(bf16_t* cache, const bf16_t* row, const int* pos, bf16_t* out, int H, int hd, int cap,
                          hipStream_t s);
int neko_ce_fwd_bwd_impl(const float* logits, long ldl, int V, int Vpad, const long long* target, const float* weight,
                         float* loss_row, bf16_t* dlogits, long ldd, int R, hipStream_t s);
int neko_ce_bf16_inplace_impl(bf16_t* z, long ld, int V, int Vpad, const long long* target, const float* weight,
                              float* loss_row, int want_grad, int R, hipStream_t s);
int neko_pack_embed_fwd_impl(const int* desc, const float* cont_vals, const int* disc_vals, const float* img_emb,
                             const float* embed, const float* pos_embed, const float* sep, float* x,
                             long long* tokens, float* tmask, float* pmask, int ntok, int d, float mu, float M,
                             int n_bins, int cont_start, int disc_start, hipStream_t s);
int neko_pack_embed_bwd_impl(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos,
                             float* d_sep, float* d_img, int ntok, int d, hipStream_t s);
int neko_tokenize_continuous_impl(const float* x, int* ids, long n, int use_mu_law, float mu, float M, int n_bins,
                                  int offset, hipStream_t s);
int neko_cast_f32_bf16_impl(const float* x, bf16_t* y, long n, hipStream_t s);
int neko_geglu_fwd_impl(bf16_t* h, const bf16_t* gate, long n, hipStream_t s);
int neko_geglu_bwd_impl(const bf16_t* dh, const bf16_t* pre, const bf16_t* gate, bf16_t* d_pre, bf16_t* d_gate, long n,
                        hipStream_t s);
int neko_colsum_bf16_impl(const bf16_t* x, long ld, int M, int N, float* out, int accumulate, hipStream_t s);
int neko_mask_bias_impl(const float* mask, float* kbias, int* kstart, int B, int T, hipStream_t s);
int neko_sqnorm_f32_impl(const float* g, long n, double* out_accum, hipStream_t s);
int neko_adamw_step_impl(float* p, const float* g, float* m, float* v, bf16_t* p16, long n, float lr, float beta1,
                         float beta2, float eps, float wd, const double* gnorm_sq, float max_norm,
                         const float* grad_scale, int* step, const int* active, const float* lr_dev, hipStream_t s);
// per-translation-unit setters of the device-side dropout key offset (neko_common.h), fanned out by neko_set_drop_salt
int neko_set_drop_salt_dropout(const uint32_t* p);
int neko_set_drop_salt_layernorm(const uint32_t* p);
int neko_set_drop_salt_gemm_bf16(const uint32_t* p);
int neko_set_drop_salt_gemm_glds(const uint32_t* p);
int neko_set_drop_salt_attention(const uint32_t* p);
int neko_set_drop_salt_attention_res(const uint32_t* p);
int neko_set_drop_salt_attention_stream(const uint32_t* p);
int neko_patch_resblock_fwd_impl(const void* images, int images_are_u8, int n, int H, int W, const float* w1,
                                 const float* b1, const float* gn_w, const float* gn_b, const float* w2,
                                 const float* b2, int mid_channels, int num_groups, bf16_t* y16, float* x_patches,
                                 hipStream_t s, float* gn_stats = nullptr);
int neko_patch_resblock_bwd_impl(const float* x_patches, const float* dy, int P, const float* w1, const float* b1,
                                 const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                                 int mid_channels, int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b,
                                 float* dw2, float* db2, float* workspace, hipStream_t s, const float* gn_stats = nullptr);
int neko_patch_resblock_bwd_blocks_impl(int P);
int neko_patch_resblock_ws_stride_impl();
int neko_patch_pos_add_impl(float* out, const int* hpos, const int* wpos, const float* row_emb, const float* col_emb,
                            int P, int d, hipStream_t s);
// segsum.hip: out[keys[i], :] += src[i, :] in index order (no atomics); keys >= NEKO_SEGSUM_KEY_NONE are skipped, key == nrows -> `extra`
#define NEKO_SEGSUM_KEY_NONE 0xFFFFFu
size_t neko_segsum_ws_bytes_impl(int n, int d);
int neko_segsum_rows_impl(const float* src, long ld_src, const unsigned* keys, int n, int d, float* out, long ld_out, int nrows,
                          float* extra, void* ws, size_t ws_bytes, hipStream_t s);
size_t neko_segsum_sorted_ws_bytes_impl(int n, int d);
int neko_segsum_rows_sorted_impl(const float* src, long ld_src, const unsigned* keys_sorted, const int* idx_sorted, int n, int d, float* out,
                                 long ld_out, int nrows, float* extra, void* ws, size_t ws_bytes, hipStream_t s);
long neko_pack_embed_bwd_sorted_ws_bytes_impl(int ntok, int d);
int neko_pack_embed_bwd_sorted_impl(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos, float* d_sep,
                                    float* d_img, int ntok, int d, int pos_rows, const unsigned* keys_sorted, const int* idx_sorted,
                                    void* ws, long ws_bytes, hipStream_t s);
long neko_patch_pos_add_bwd_sorted_ws_bytes_impl(int P, int d);
int neko_patch_pos_add_bwd_sorted_impl(const float* dout, const unsigned* hkeys, const int* hidx, const unsigned* wkeys, const int* widx,
                                       float* d_row_emb, float* d_col_emb, int P, int d, int nrows, void* ws, long ws_bytes, hipStream_t s);
long neko_pack_embed_bwd_det_ws_bytes_impl(int ntok, int d);
int neko_pack_embed_bwd_det_impl(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos, float* d_sep,
                                 float* d_img, int ntok, int d, int vocab_rows, int pos_rows, void* ws, long ws_bytes, hipStream_t s);
long neko_patch_pos_add_bwd_det_ws_bytes_impl(int P, int d);
int neko_patch_pos_add_bwd_det_impl(const float* dout, const int* hpos, const int* wpos, float* d_row_emb, float* d_col_emb, int P,
                                    int d, int nrows, void* ws, long ws_bytes, hipStream_t s);
int neko_patch_pos_add_bwd_impl(const float* dout, const int* hpos, const int* wpos, float* d_row_emb,
                                float* d_col_emb, int P, int d, hipStream_t s);
