// extern "C" surface of libneko_hip.so (declared in include/neko_hip.h).  Thin argument
// marshalling only; kernels live in the sibling .hip files.
#include "neko_kernels.h"
#include "../../include/neko_hip.h"

#define S(x) reinterpret_cast<hipStream_t>(x)

extern "C" {

int neko_abi_version(void) { return NEKO_ABI_VERSION; }

const char* neko_status_string(int code) {
  if (code == NEKO_OK) return "ok";
  if (code == NEKO_ERR_ARG) return "invalid argument (null pointer, misaligned or inconsistent size)";
  if (code == NEKO_ERR_UNSUPPORTED) return "unsupported shape for this kernel";
  if (code <= NEKO_ERR_LAUNCH) return hipGetErrorString((hipError_t)(NEKO_ERR_LAUNCH - code));
  return "unknown";
}

int neko_gemm_bf16(const uint16_t* A, long lda, int a_kstrided, const uint16_t* B, long ldb, int b_kstrided, int M,
                   int N, int K, float alpha, const float* alpha_dev, const float* bias, const float* resid, long ldr, int act,
                   const uint16_t* act_in, long ldact, uint16_t* pre_out, long ldpre, float* Cf, long ldcf,
                   int accumulate, uint16_t* Cb, long ldcb, int splitk, int k_per_split, float* splitk_ws,
                   int drop_thr, unsigned drop_key, float drop_scale, int safe_transpose, void* stream) {
  GemmArgs a{A, B, lda, ldb, M, N, K, alpha, alpha_dev, bias, resid, ldr, act_in, ldact, pre_out, ldpre, act,
             Cf, ldcf, accumulate, Cb, ldcb, splitk, k_per_split, splitk_ws, drop_thr, drop_key, drop_scale};
  if (drop_thr < 0 || drop_thr > 255) return NEKO_ERR_ARG;
  return neko_gemm_bf16_full(a, a_kstrided, b_kstrided, safe_transpose, S(stream));
}

int neko_gemm_set_mainloop(int mode) { return neko_gemm_set_mainloop_impl(mode); }
int neko_gemm_last_mainloop(void) { return g_neko_last_mainloop; }
long neko_gemm_colsum_ws_floats(int M, int N) { return (long)((M + 63) / 64) * (long)N; }
int neko_gemm_dgrad_gelu_colsum(const uint16_t* dY, long lda, const uint16_t* W, long ldb, int M, int N, int K,
                                const uint16_t* act_in, long ldact, int act_in_is_factor, uint16_t* Cb, long ldcb,
                                float* colsum_ws, float* colsum_out, void* stream) {
  if (!Cb || !act_in || !colsum_out) return NEKO_ERR_ARG;
  GemmArgs a{dY, W, lda, ldb, M, N, K, 1.0f, nullptr, nullptr, nullptr, 0, act_in, ldact, nullptr, 0, act_in_is_factor ? 4 : 2,
             nullptr, 0, 0, Cb, ldcb, 1, 0, nullptr, 0, 0u, 1.0f};
  a.colsum_ws = colsum_ws;
  (void)neko_gemm_glds_colsum_bands();
  const int rc = neko_gemm_bf16_full(a, 0, 0, 0, S(stream));
  if (rc != NEKO_OK || M <= 0 || N <= 0 || K <= 0) return rc;
  const int bands = neko_gemm_glds_colsum_bands();
  if (bands > 0) return neko_colsum_bands_reduce_impl(colsum_ws, bands, N, colsum_out, S(stream));
  return neko_colsum_bf16_impl(Cb, ldcb, M, N, colsum_out, 1, S(stream));
}

int neko_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y16, float* y32, float* mean,
                       float* rstd, int M, int d, float eps, void* stream) {
  return neko_layernorm_fwd_impl(x, gamma, beta, y16, y32, mean, rstd, M, d, eps, S(stream));
}
int neko_layernorm_bwd_blocks(int M) { return neko_layernorm_bwd_blocks_impl(M); }
int neko_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       const float* g_in, float* dx, uint16_t* dx16, float* dgamma, float* dbeta, int accumulate,
                       float* workspace, int M, int d, int drop_thr, unsigned drop_key, float drop_scale, float* dcolsum16,
                       void* stream) {
  if (drop_thr < 0 || drop_thr > 255) return NEKO_ERR_ARG;
  return neko_layernorm_bwd_impl(dy, 0, x, gamma, mean, rstd, g_in, dx, dx16, dgamma, dbeta, accumulate, workspace, M, d,
                                 drop_thr, drop_key, drop_scale, dcolsum16, S(stream));
}
int neko_layernorm_bwd_rows(const float* dy_rows, const int* dy_row_map, const float* x, const float* gamma, const float* mean,
                            const float* rstd, const float* g_in, float* dx, uint16_t* dx16, float* dgamma, float* dbeta, int accumulate,
                            float* workspace, int M, int d, int drop_thr, unsigned drop_key, float drop_scale, float* dcolsum16,
                            void* stream) {
  if (drop_thr < 0 || drop_thr > 255 || !dy_row_map) return NEKO_ERR_ARG;
  return neko_layernorm_bwd_impl(dy_rows, 0, x, gamma, mean, rstd, g_in, dx, dx16, dgamma, dbeta, accumulate, workspace, M, d,
                                 drop_thr, drop_key, drop_scale, dcolsum16, S(stream), dy_row_map);
}
int neko_layernorm_bwd_bf16dy(const uint16_t* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       const float* g_in, float* dx, uint16_t* dx16, float* dgamma, float* dbeta, int accumulate,
                       float* workspace, int M, int d, int drop_thr, unsigned drop_key, float drop_scale, float* dcolsum16,
                       void* stream) {
  if (drop_thr < 0 || drop_thr > 255) return NEKO_ERR_ARG;
  return neko_layernorm_bwd_impl(dy, 1, x, gamma, mean, rstd, g_in, dx, dx16, dgamma, dbeta, accumulate, workspace, M, d,
                                 drop_thr, drop_key, drop_scale, dcolsum16, S(stream));
}

int neko_mask_bias(const float* mask, float* kbias, int* kstart, int B, int T, void* stream) {
  return neko_mask_bias_impl(mask, kbias, kstart, B, T, S(stream));
}
int neko_attn_fwd(const uint16_t* qkv, const float* kbias, const int* kstart, uint16_t* out, float* lse, int B, int T,
                  int H, int hd, int drop_thr, unsigned drop_key, float drop_scale, uint32_t* drop_mask, void* stream) {
  return neko_attn_fwd_impl(qkv, kbias, kstart, out, lse, B, T, H, hd, drop_thr, drop_key, drop_scale, drop_mask, S(stream));
}
long neko_attn_mask_dwords(int B, int T, int H, int hd) { return neko_attn_mask_dwords_impl(B, T, H, hd); }
// packed sequences of different lengths in ONE launch: head-resident kernels (hd = 32, every length <= 1024) or the DMA-ring
// kernels (hd = 64 / 128, every length <= 4096; ABI v16)
int neko_attn_varlen_supported(int Tmax, int hd) { return (neko_attn_res_applicable(Tmax, hd) || neko_attn_stream_applicable(Tmax, hd)) ? 1 : 0; }
int neko_attn_fwd_varlen(const uint16_t* qkv, const float* kbias, const int* kstart, const int* seq_off, const long long* mask_off,
                         uint16_t* out, float* lse, int nseq, int Tmax, int H, int hd, int drop_thr, unsigned drop_key,
                         float drop_scale, uint32_t* drop_mask, void* stream) {
  if (!qkv || !kbias || !seq_off || !out || !lse || nseq <= 0 || Tmax <= 0 || H <= 0) return NEKO_ERR_ARG;
  if (drop_thr < 0 || drop_thr > 255 || (drop_mask && !mask_off)) return NEKO_ERR_ARG;
  if (neko_attn_stream_applicable(Tmax, hd)) {          // hd = 64 / 128: no stored keep masks (the backward re-hashes the decisions)
    if (drop_mask) return NEKO_ERR_ARG;
    return neko_attn_fwd_stream_impl(qkv, kbias, kstart, out, lse, nseq, Tmax, H, hd, drop_thr, drop_key, drop_scale, S(stream), seq_off);
  }
  if (!neko_attn_res_applicable(Tmax, hd)) return NEKO_ERR_UNSUPPORTED;
  return neko_attn_fwd_res_impl(qkv, kbias, kstart, out, lse, nseq, Tmax, H, drop_thr, drop_key, drop_scale, drop_mask, S(stream),
                                seq_off, mask_off);
}
int neko_attn_bwd_varlen(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* kbias, const int* kstart,
                         const int* seq_off, const long long* mask_off, const float* lse, float* D, uint16_t* dqkv, int nseq,
                         long rows, int Tmax, int H, int hd, int drop_thr, unsigned drop_key, float drop_scale,
                         const uint32_t* drop_mask, void* stream) {
  if (!qkv || !out || !dout || !kbias || !seq_off || !lse || !D || !dqkv || nseq <= 0 || Tmax <= 0 || H <= 0) return NEKO_ERR_ARG;
  if (drop_thr < 0 || drop_thr > 255 || (drop_mask && !mask_off)) return NEKO_ERR_ARG;
  if (neko_attn_stream_applicable(Tmax, hd)) {
    if (drop_mask || rows <= 0) return NEKO_ERR_ARG;
    return neko_attn_bwd_stream_impl(qkv, out, dout, kbias, kstart, lse, D, dqkv, nseq, Tmax, H, hd, drop_thr, drop_key, drop_scale,
                                     S(stream), seq_off, rows);
  }
  if (!neko_attn_res_applicable(Tmax, hd)) return NEKO_ERR_UNSUPPORTED;
  return neko_attn_bwd_res_impl(qkv, out, dout, kbias, kstart, lse, D, dqkv, nseq, Tmax, H, drop_thr, drop_key, drop_scale,
                                drop_mask, S(stream), seq_off, mask_off);
}
int neko_attn_set_path(int mode) { return neko_attn_set_path_impl(mode); }
int neko_attn_bwd_reproducible(int on) { return neko_attn_bwd_reproducible_impl(on); }
int neko_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* kbias,
                  const int* kstart, const float* lse, float* D, int* qflags, uint16_t* dqkv, int B, int T, int H,
                  int hd, int drop_thr, unsigned drop_key, float drop_scale, const uint32_t* drop_mask, void* stream) {
  return neko_attn_bwd_impl(qkv, out, dout, kbias, kstart, lse, D, qflags, dqkv, B, T, H, hd, drop_thr, drop_key, drop_scale,
                            drop_mask, S(stream));
}

int neko_gemv_bf16(const uint16_t* x, long ldx, const uint16_t* W, long ldw, int b_kstrided, int M, int N, int K,
                   const float* bias, const float* resid, long ldr, int act, float* Cf, long ldcf, uint16_t* Cb, long ldcb,
                   void* stream) {
  return neko_gemv_bf16_impl(x, ldx, W, ldw, b_kstrided, M, N, K, bias, resid, ldr, act, Cf, ldcf, Cb, ldcb, S(stream));
}
int neko_attn_decode(uint16_t* cache, const uint16_t* row, const int* pos, uint16_t* out, int H, int hd, int cap,
                     void* stream) {
  return neko_attn_decode_impl(cache, row, pos, out, H, hd, cap, S(stream));
}
int neko_ce_fwd_bwd(const float* logits, long ldl, int V, int Vpad, const long long* target, const float* weight,
                    float* loss_row, uint16_t* dlogits, long ldd, int R, void* stream) {
  return neko_ce_fwd_bwd_impl(logits, ldl, V, Vpad, target, weight, loss_row, dlogits, ldd, R, S(stream));
}
int neko_ce_bf16_inplace(uint16_t* z, long ld, int V, int Vpad, const long long* target, const float* weight,
                         float* loss_row, int want_grad, int R, void* stream) {
  return neko_ce_bf16_inplace_impl(z, ld, V, Vpad, target, weight, loss_row, want_grad, R, S(stream));
}

int neko_pack_embed_fwd(const int* desc, const float* cont_vals, const int* disc_vals, const float* img_emb,
                        const float* embed, const float* pos_embed, const float* sep, float* x, long long* tokens,
                        float* tmask, float* pmask, int ntok, int d, float mu, float M, int n_bins, int cont_start,
                        int disc_start, void* stream) {
  return neko_pack_embed_fwd_impl(desc, cont_vals, disc_vals, img_emb, embed, pos_embed, sep, x, tokens, tmask, pmask,
                                  ntok, d, mu, M, n_bins, cont_start, disc_start, S(stream));
}
int neko_pack_embed_bwd(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos,
                        float* d_sep, float* d_img, int ntok, int d, void* stream) {
  return neko_pack_embed_bwd_impl(desc, tokens, dx, d_embed, d_pos, d_sep, d_img, ntok, d, S(stream));
}
int neko_tokenize_continuous(const float* x, int* ids, long n, int use_mu_law, float mu, float M, int n_bins,
                             int offset, void* stream) {
  return neko_tokenize_continuous_impl(x, ids, n, use_mu_law, mu, M, n_bins, offset, S(stream));
}

int neko_gather_rows_bf16(const uint16_t* src, const int* idx, uint16_t* dst, int n, int npad, int d, void* stream) {
  return neko_gather_rows_bf16_impl(src, idx, dst, n, npad, d, S(stream));
}
int neko_scatter_rows_f32(const float* src, const int* idx, float* dst, int n, int d, void* stream) {
  return neko_scatter_rows_f32_impl(src, idx, dst, n, d, S(stream));
}
int neko_dropout_f32(const float* x, float* y, long n, int thr, unsigned key, float scale, void* stream) {
  return neko_dropout_f32_impl(x, y, n, thr, key, scale, S(stream));
}
int neko_cast_f32_bf16(const float* x, uint16_t* y, long n, void* stream) {
  return neko_cast_f32_bf16_impl(x, y, n, S(stream));
}
int neko_geglu_fwd(uint16_t* h, const uint16_t* gate, long n, void* stream) {
  return neko_geglu_fwd_impl(h, gate, n, S(stream));
}
int neko_geglu_bwd(const uint16_t* dh, const uint16_t* pre, const uint16_t* gate, uint16_t* d_pre, uint16_t* d_gate, long n,
                   void* stream) {
  return neko_geglu_bwd_impl(dh, pre, gate, d_pre, d_gate, n, S(stream));
}
int neko_colsum_bf16(const uint16_t* x, long ld, int M, int N, float* out, int accumulate, void* stream) {
  return neko_colsum_bf16_impl(x, ld, M, N, out, accumulate, S(stream));
}
int neko_sqnorm_f32(const float* g, long n, double* out_accum, void* stream) {
  return neko_sqnorm_f32_impl(g, n, out_accum, S(stream));
}
int neko_adamw_step(float* p, const float* g, float* m, float* v, uint16_t* p16, long n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, const double* gnorm_sq, float max_norm,
                    const float* grad_scale, int* step, const int* active, const float* lr_dev, void* stream) {
  return neko_adamw_step_impl(p, g, m, v, p16, n, lr, beta1, beta2, eps, weight_decay, gnorm_sq, max_norm, grad_scale,
                              step, active, lr_dev, S(stream));
}
int neko_set_drop_salt(const uint32_t* salt) {
  int rc = neko_set_drop_salt_dropout(salt);
  if (rc == NEKO_OK) rc = neko_set_drop_salt_layernorm(salt);
  if (rc == NEKO_OK) rc = neko_set_drop_salt_gemm_bf16(salt);
  if (rc == NEKO_OK) rc = neko_set_drop_salt_gemm_glds(salt);
  if (rc == NEKO_OK) rc = neko_set_drop_salt_attention(salt);
  if (rc == NEKO_OK) rc = neko_set_drop_salt_attention_res(salt);
  if (rc == NEKO_OK) rc = neko_set_drop_salt_attention_stream(salt);
  return rc;
}

int neko_patch_resblock_fwd(const void* images, int images_are_u8, int n, int H, int W, const float* w1,
                            const float* b1, const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                            int mid_channels, int num_groups, uint16_t* y16, float* x_patches, void* stream) {
  return neko_patch_resblock_fwd_impl(images, images_are_u8, n, H, W, w1, b1, gn_w, gn_b, w2, b2, mid_channels,
                                      num_groups, y16, x_patches, S(stream));
}
int neko_patch_resblock_bwd(const float* x_patches, const float* dy, int P, const float* w1, const float* b1,
                            const float* gn_w, const float* gn_b, const float* w2, const float* b2, int mid_channels,
                            int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b, float* dw2, float* db2,
                            float* workspace, void* stream) {
  return neko_patch_resblock_bwd_impl(x_patches, dy, P, w1, b1, gn_w, gn_b, w2, b2, mid_channels, num_groups, dw1, db1,
                                      dgn_w, dgn_b, dw2, db2, workspace, S(stream));
}
int neko_patch_resblock_fwd_stats(const void* images, int images_are_u8, int n, int H, int W, const float* w1,
                                  const float* b1, const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                                  int mid_channels, int num_groups, uint16_t* y16, float* x_patches, float* gn_stats, void* stream) {
  return neko_patch_resblock_fwd_impl(images, images_are_u8, n, H, W, w1, b1, gn_w, gn_b, w2, b2, mid_channels,
                                      num_groups, y16, x_patches, S(stream), gn_stats);
}
int neko_patch_resblock_bwd_stats(const float* x_patches, const float* gn_stats, const float* dy, int P, const float* w1, const float* b1,
                                  const float* gn_w, const float* gn_b, const float* w2, const float* b2, int mid_channels,
                                  int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b, float* dw2, float* db2,
                                  float* workspace, void* stream) {
  if (!gn_stats) return NEKO_ERR_ARG;
  return neko_patch_resblock_bwd_impl(x_patches, dy, P, w1, b1, gn_w, gn_b, w2, b2, mid_channels, num_groups, dw1, db1,
                                      dgn_w, dgn_b, dw2, db2, workspace, S(stream), gn_stats);
}
long neko_pack_embed_bwd_sorted_ws_bytes(int ntok, int d) { return neko_pack_embed_bwd_sorted_ws_bytes_impl(ntok, d); }
int neko_pack_embed_bwd_sorted(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos, float* d_sep,
                               float* d_img, int ntok, int d, int pos_rows, const int* keys_sorted, const int* idx_sorted, void* workspace,
                               long ws_bytes, void* stream) {
  return neko_pack_embed_bwd_sorted_impl(desc, tokens, dx, d_embed, d_pos, d_sep, d_img, ntok, d, pos_rows,
                                         reinterpret_cast<const unsigned*>(keys_sorted), idx_sorted, workspace, ws_bytes, S(stream));
}
long neko_patch_pos_add_bwd_sorted_ws_bytes(int P, int d) { return neko_patch_pos_add_bwd_sorted_ws_bytes_impl(P, d); }
int neko_patch_pos_add_bwd_sorted(const float* dout, const int* hkeys_sorted, const int* hidx_sorted, const int* wkeys_sorted,
                                  const int* widx_sorted, float* d_row_emb, float* d_col_emb, int P, int d, int nrows, void* workspace,
                                  long ws_bytes, void* stream) {
  return neko_patch_pos_add_bwd_sorted_impl(dout, reinterpret_cast<const unsigned*>(hkeys_sorted), hidx_sorted,
                                            reinterpret_cast<const unsigned*>(wkeys_sorted), widx_sorted, d_row_emb, d_col_emb, P, d, nrows,
                                            workspace, ws_bytes, S(stream));
}
int neko_patch_resblock_bwd_ws_floats(int P) {
  return neko_patch_resblock_bwd_blocks_impl(P) * neko_patch_resblock_ws_stride_impl();
}
int neko_patch_pos_add(float* out, const int* hpos, const int* wpos, const float* row_emb, const float* col_emb,
                       int P, int d, void* stream) {
  return neko_patch_pos_add_impl(out, hpos, wpos, row_emb, col_emb, P, d, S(stream));
}
int neko_patch_pos_add_bwd(const float* dout, const int* hpos, const int* wpos, float* d_row_emb, float* d_col_emb,
                           int P, int d, void* stream) {
  return neko_patch_pos_add_bwd_impl(dout, hpos, wpos, d_row_emb, d_col_emb, P, d, S(stream));
}
long neko_patch_pos_add_bwd_det_ws_bytes(int P, int d) { return neko_patch_pos_add_bwd_det_ws_bytes_impl(P, d); }
int neko_patch_pos_add_bwd_det(const float* dout, const int* hpos, const int* wpos, float* d_row_emb, float* d_col_emb, int P, int d,
                               int nrows, void* workspace, long ws_bytes, void* stream) {
  return neko_patch_pos_add_bwd_det_impl(dout, hpos, wpos, d_row_emb, d_col_emb, P, d, nrows, workspace, ws_bytes, S(stream));
}
long neko_pack_embed_bwd_det_ws_bytes(int ntok, int d) { return neko_pack_embed_bwd_det_ws_bytes_impl(ntok, d); }
int neko_pack_embed_bwd_det(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos, float* d_sep,
                            float* d_img, int ntok, int d, int vocab_rows, int pos_rows, void* workspace, long ws_bytes, void* stream) {
  return neko_pack_embed_bwd_det_impl(desc, tokens, dx, d_embed, d_pos, d_sep, d_img, ntok, d, vocab_rows, pos_rows, workspace, ws_bytes,
                                      S(stream));
}

}  // extern "C"
