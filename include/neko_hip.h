/*
 * neko_hip.h -- C ABI of libneko_hip.so: the MI355X (gfx950) hot path of ManifoldRG/NEKO's
 * Gato-style policy (forward + backward of gato.policy.GatoPolicy / gato.transformers.GPT2Model
 * and the optimiser tail of Trainer.train_step).
 *
 * The reference is pure Python and has NO FFI / plugin interface; the
 * seam it offers is a Python call signature.  Each entry point below therefore names the reference
 * expression it replaces (file:line relative to the reference root); the Python binding a NEKO
 * maintainer would add is shown in INTEGRATION.md and implemented in neko_amd/_lib.py (ctypes).
 *
 * Conventions
 *  - plain C: raw device pointers, explicit sizes / leading dimensions (in ELEMENTS), a HIP stream
 *    passed as void* (hipStream_t).  No torch types, no allocation: every buffer (inputs, outputs,
 *    workspaces) is owned by the caller for the duration of the enqueued work.  Process-wide state is
 *    limited to three documented knobs -- neko_gemm_set_mainloop, neko_attn_set_path (schedule
 *    selectors: they pick between kernels with the same contract) and neko_set_drop_salt (the
 *    device word a captured training step advances) -- none of which changes what a call computes.
 *  - every call only ENQUEUES kernels on `stream` and returns immediately.
 *  - return value: 0 = ok; NEKO_ERR_ARG (-1) bad argument; NEKO_ERR_UNSUPPORTED (-2) shape outside
 *    the supported set; <= -3: launch failure, value = -3 - hipError_t.  Nothing throws.
 *  - bf16 tensors are `uint16_t` bit patterns; "f32" = float; row-major everywhere.
 *  - thread model: one host thread per process, one process per GPU (the Accelerate launch model
 *    of train.py:33-41).
 */
#ifndef NEKO_HIP_H
#define NEKO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEKO_OK 0
#define NEKO_ERR_ARG (-1)
#define NEKO_ERR_UNSUPPORTED (-2)
#define NEKO_ERR_LAUNCH (-3)

#define NEKO_ABI_VERSION 19

int neko_abi_version(void);
/* human-readable text for a return code (static storage) */
const char* neko_status_string(int code);

/* ---------------------------------------------------------------------------------------------
 * GEMM with fused epilogue -- HF Conv1D `torch.addmm(bias, x.view(-1,in), W)` with W stored
 * (in,out) (gato/transformers/trajectory_gpt2.py:139-141,222,253,264-265,274,277), the exact-erf
 * GELU of MLP.forward (:274), the residual adds of Block.forward (:333,355), nn.Linear
 * predict_token / post_embedding_projection (gato/policy/gato_policy.py:122,172;
 * gato/policy/embeddings.py:24,53) and the dgrad / wgrad contractions autograd derives from them.
 *
 *   C[M,N] = alpha * opA(A)[M,K] x opB(B)[K,N]  (+ bias[N])  (act)  (+ resid[M,N])  (+ C)
 *   alpha_dev (may be null): device scalar multiplied into alpha (autograd's grad_output, no host sync)
 *   a_kstrided = 0: A[m*lda + k]        1: A[k*lda + m]   (wgrad: X^T)
 *   b_kstrided = 0: B[n*ldb + k]        1: B[k*ldb + n]   (Conv1D weight (in,out))
 *   act = 0 none | 1 GELU (pre-activation rounded to bf16 first; optionally stored to pre_out)
 *         | 2 multiply by GELU'(act_in[m,n])  (dgrad through the MLP activation)
 *         | 3 GELU as 1, but pre_out (required) receives GELU'(pre-activation) as bf16: the factor the backward needs,
 *             from the same evaluation of the erf series (ABI v14)
 *         | 4 multiply by act_in[m,n] itself (the factor stored by act = 3)
 *   outputs: Cf (f32) and/or Cb (bf16); accumulate != 0: Cf += result.
 *   splitk > 1: K is cut in `splitk` slices of k_per_split (multiple of 64).  With splitk_ws (f32
 *   [splitk*M*N], N % 4 == 0) the slices are written to the workspace and summed in a fixed order into Cf
 *   (bit-reproducible); with splitk_ws == NULL they meet in f32 atomics on Cf.  act must be 0, Cb null.
 *   drop_thr != 0: residual dropout (resid_dropout / MLP dropout, trajectory_gpt2.py:253-254,277-278) applied to
 *   (alpha*acc + bias [act]) before the residual add; see "Dropout" below for (thr, key, scale).
 *   Contract: contiguous extents and leading dims are multiples of 8 elements.
 *   safe_transpose != 0 selects the transposing-store fallback for k-strided operands (debug).
 * ------------------------------------------------------------------------------------------- */
int neko_gemm_bf16(const uint16_t* A, long lda, int a_kstrided, const uint16_t* B, long ldb, int b_kstrided,
                   int M, int N, int K, float alpha, const float* alpha_dev, const float* bias, const float* resid,
                   long ldr, int act,
                   const uint16_t* act_in, long ldact, uint16_t* pre_out, long ldpre, float* Cf, long ldcf,
                   int accumulate, uint16_t* Cb, long ldcb, int splitk, int k_per_split, float* splitk_ws,
                   int drop_thr, unsigned drop_key, float drop_scale, int safe_transpose, void* stream);

/* neko_gemm_bf16 has three main loops behind one contract.  The default one (neko_amd/csrc/gemm_glds.hip) serves every shape; the
 * long-contraction loop (neko_amd/csrc/gemm_a16.hip: 4 waves x 128 x 128 per wave on v_mfma_f32_16x16x32_bf16 with the accumulators
 * in AGPRs, instruction stream placed by hand) takes launches whose tiles are all interior 256 x 256 ones and whose contraction
 * range is a multiple of 128, where it measured faster (long K); the two-workgroups-per-CU loop (neko_amd/csrc/gemm_b16.hip, ABI v17:
 * 128 x 256 per workgroup, 64 x 128 per wave, 128 accumulators, 80 KB of LDS, so that one workgroup's output phase runs under its
 * CU-mate's main loop) takes launches with A k-contiguous, M % 128 == N % 256 == K % 384 == 0 whose output phase is long (the fp32
 * residual epilogues) or that are smaller than one round of workgroups.  neko_gemm_set_mainloop(1) sends every launch it can serve
 * to gemm_a16 (none to gemm_b16), (2) every launch it can serve to gemm_b16, (0) none to either, (-1) returns to the built-in
 * per-shape choice (also: environment NEKO_GEMM_A16 / NEKO_GEMM_B16 = 0/1); returns the previous mode.  The choice never changes
 * which products are summed into an output element; the loops add them in a different order (fp32).  (ABI v16; replaces v14's
 * neko_gemm_set_persistent, whose kernel moved to tools/probe/r05/) */
int neko_gemm_set_mainloop(int mode);
/* ABI v19 (round 6).  A fourth main loop, neko_amd/csrc/gemm_p16.hip: 256 x 256 per workgroup of EIGHT waves (128 x 64 per wave, 128
 * accumulators), the two waves of a SIMD alternating between a matrix segment (32 MFMAs) and a load segment (fragment reads + L2 -> LDS
 * requests) of hand-placed instruction streams; serves launches of interior 256 x 256 tiles whose contraction range is a multiple of 128 and
 * at least 384 (A k-contiguous: 12 k-tiles per loop trip, a tail of 4 / 8) or 128 (both operands k-strided).  neko_gemm_set_mainloop(3) sends every launch it can serve to it.
 * neko_gemm_last_mainloop(): which loop served the calling thread's last neko_gemm_bf16 / neko_gemm_dgrad_gelu_colsum launch --
 * 0 gemm_glds (32 x 32 x 16 loop, any tile configuration), 1 gemm_a16, 2 gemm_b16, 3 gemm_glds64 (the 8-wave 256 x 256 loop with
 * whole-line A slots), 4 the register-staged fallback (gemm_bf16.hip), 5 gemm_p16; -1 before the first launch.  Thread-local; exists so
 * that tests can assert WHICH loop they exercised (reference: the Conv1D / Linear products of gato/transformers/trajectory_gpt2.py:139-141). */
int neko_gemm_last_mainloop(void);

/* Backward of the MLP's first Linear + GELU in one launch (trajectory_gpt2.py:266,274: h = act(c_fc(x)); autograd's
 * d_pre = (d_h . W_proj^T) * gelu'(pre) and the c_fc bias gradient sum_rows d_pre):
 *   Cb[M,N] bf16 = (dY[M,K] . W[N,K]^T) * gelu'(act_in[M,N]);   colsum_out[N] (f32) += column sums of that product.
 * The column sums ride in the GEMM epilogue (per 128-row band partial rows in colsum_ws, added up in band order: bit-
 * reproducible) whenever M and N are multiples of the tile the launch picks; otherwise a stand-alone pass over the stored
 * bf16 result computes them (the two differ by the bf16 rounding of the summands).  colsum_ws: at least
 * neko_gemm_colsum_ws_floats(M, N) floats. */
long neko_gemm_colsum_ws_floats(int M, int N);
int neko_gemm_dgrad_gelu_colsum(const uint16_t* dY, long lda, const uint16_t* W, long ldb, int M, int N, int K,
                                const uint16_t* act_in, long ldact, int act_in_is_factor, uint16_t* Cb, long ldcb,
                                float* colsum_ws, float* colsum_out, void* stream);
/* act_in_is_factor (ABI v14): 0 = act_in holds the pre-activation (epilogue act 2), 1 = act_in holds gelu'(pre) as the
 * forward with act = 3 stored it (epilogue act 4: one multiply per element, no erf evaluation in the backward). */

/* ---------------------------------------------------------------------------------------------
 * LayerNorm -- nn.LayerNorm(d, eps) ln_1 / ln_2 / ln_f (trajectory_gpt2.py:301,303,323,353,543,779).
 * fwd: x f32 [M,d] -> y16 (bf16, may be null) and/or y32 (f32, may be null); mean/rstd f32 [M] (may be null).
 * bwd: dy f32 [M,d]; g_in (may be null) is the residual-stream gradient added to the result;
 *      dx f32 and/or dx16 bf16 (either may be null); dgamma/dbeta f32 [d] (+= when accumulate);
 *      workspace: neko_layernorm_bwd_blocks(M) * 3 * d floats.  drop_thr != 0: dx16 (only) additionally carries the
 *      dropout mask of the residual-dropout site whose Linear consumes it (element index row*d + col); with dx16 null the
 *      mask is applied to dx instead (the embedding dropout in front of the first block, trajectory_gpt2.py:707: its backward
 *      is the last thing the stack's backward does).
 *      dcolsum16 (may be null; needs dx16): f32 [d] += column sums of the dx16 values as stored = the bias gradient of
 *      the Conv1D that consumes dx16 (trajectory_gpt2.py:253,277), folded into this pass.
 * ------------------------------------------------------------------------------------------- */
int neko_layernorm_fwd(const float* x, const float* gamma, const float* beta, uint16_t* y16, float* y32,
                       float* mean, float* rstd, int M, int d, float eps, void* stream);
int neko_layernorm_bwd_blocks(int M);
int neko_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       const float* g_in, float* dx, uint16_t* dx16, float* dgamma, float* dbeta, int accumulate,
                       float* workspace, int M, int d, int drop_thr, unsigned drop_key, float drop_scale,
                       float* dcolsum16, void* stream);
/* same with the gradient given for SOME rows only (ABI v17): dy_rows f32 [n, d] compact, dy_row_map int32 [M] = index of row r's
 * gradient in dy_rows, or -1 where it is zero.  The LM head returns gradient rows for the loss positions only (the reference's
 * boolean-mask gather, gato/policy/gato_policy.py:183-185, differentiated); ln_f's backward reads them in place instead of a
 * zero-filled [M, d] expansion. */
int neko_layernorm_bwd_rows(const float* dy_rows, const int* dy_row_map, const float* x, const float* gamma, const float* mean,
                            const float* rstd, const float* g_in, float* dx, uint16_t* dx16, float* dgamma, float* dbeta,
                            int accumulate, float* workspace, int M, int d, int drop_thr, unsigned drop_key, float drop_scale,
                            float* dcolsum16, void* stream);
/* same with dy as bf16 [M,d] (the dgrad GEMM's bf16 output, as autocast leaves it in the reference: the gradient of a
 * bf16 addmm input is bf16, trajectory_gpt2.py:274-277): half the bytes written by the GEMM and read here */
int neko_layernorm_bwd_bf16dy(const uint16_t* dy, const float* x, const float* gamma, const float* mean,
                              const float* rstd, const float* g_in, float* dx, uint16_t* dx16, float* dgamma,
                              float* dbeta, int accumulate, float* workspace, int M, int d, int drop_thr,
                              unsigned drop_key, float drop_scale, float* dcolsum16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention -- Attention._attn + split_heads/merge_heads (trajectory_gpt2.py:163-201,222-226,252)
 * with the mask preparation of GPT2Model.forward (:663-679); attn_dropout (:142,179) via (drop_thr, drop_key,
 * drop_scale) on the probabilities, element index ((b*H+h)*T+q)*T+key (mod 2^32), normaliser undropped.
 *   neko_mask_bias: mask f32 [B,T] (1 real, 0 pad) -> kbias f32 [B,T] = (1-mask)*-1e4 and
 *                   kstart int32 [B] = index of the first real key (may be null).
 *   qkv  bf16 [B*T, 3*H*hd]  (q | k | v, head h at columns h*hd..) as produced by c_attn
 *   out  bf16 [B*T, H*hd]    lse f32 [B,H,T]
 *   bwd workspace: D f32 [B*H*T], qflags int32 [B*ceil(T/64)]; dqkv bf16 [B*T, 3*H*hd] fully written.
 *   hd in {32, 64, 128}.
 *   Two schedules compute the same sums: head-resident kernels (hd = 32, T <= 1024: one workgroup per (b, h) keeps
 *   the head's K/V or Q/dO in LDS) and streaming kernels (any T, hd).  neko_attn_set_path(0) = automatic (default; since
 *   round 6 the head-resident backward is the two-kernel, bit-reproducible form at EVERY length: the one-pass kernel sums dQ in arrival
 *   order and bought 0.05 ms of a 10.4 ms configs[3] step), (1) = always streaming, (2) / (3) = head-resident with the two-kernel / the
 *   one-pass backward at every length; returns the previous
 *   mode (any other argument only queries).  Process-wide tuning knob.
 *   neko_attn_bwd_reproducible(1 / 0) (ABI v18): the CALLING THREAD's backward calls use the two-kernel form at every length (0: follow
 *   the knob); returns the previous value, any other argument only queries.  What NEKO_DETERMINISTIC=1 selects -- thread-local, so it
 *   never changes the schedule of another thread's calls.
 *   drop_mask (optional, only touched when drop_thr > 0): neko_attn_mask_dwords(B, T, H, hd) uint32 of device memory
 *   (0 = the schedule in use does not exchange masks: pass null).  The forward stores its keep decisions there (scalar
 *   stores of the compares' lane masks) and the backward of the SAME forward call applies them instead of re-hashing
 *   every element in both of its kernels; with null both directions hash -- identical decisions either way.
 * ------------------------------------------------------------------------------------------- */
int neko_attn_set_path(int mode);
int neko_attn_bwd_reproducible(int on);

/* Packed sequences of DIFFERENT lengths in one launch ("varlen": SURVEY 8(f) rank 3; removes the left-pad of
 * gato/policy/gato_policy.py:408-416 without one attention launch per length bucket).  Rows seq_off[b] .. seq_off[b+1]-1 of qkv /
 * out / dout / dqkv / kbias are sequence b (seq_off: int32 [nseq + 1], device); kstart [nseq] is relative to its sequence.
 * lse and D are [rows * H] laid out [sequence][head][position] = seq_off[b] * H + h * T_b + q.  The keep masks of sequence b
 * start at mask_off[b] dwords (int64 [nseq], device): mask_off[b] = sum over b' < b of H * ceil(T_b' / 32)^2 * 32; the buffer holds
 * that sum over all sequences.  The dropout hash indexes (unique row id) * ceil(Tmax / 4) + key / 4.  Served by the head-resident
 * kernels (hd = 32, Tmax <= 1024) and, since ABI v16, by the DMA-ring kernels (hd = 64 / 128, Tmax <= 4096; these keep no stored
 * masks: drop_mask and mask_off must be null); neko_attn_varlen_supported tells, otherwise NEKO_ERR_UNSUPPORTED (callers fall
 * back to one launch per length bucket).  rows (ABI v16) = seq_off[nseq], the number of packed rows.  Arithmetic, masks and
 * dropout semantics are those of neko_attn_fwd / neko_attn_bwd.  (ABI v14) */
int neko_attn_varlen_supported(int Tmax, int hd);
int neko_attn_fwd_varlen(const uint16_t* qkv, const float* kbias, const int* kstart, const int* seq_off, const long long* mask_off,
                         uint16_t* out, float* lse, int nseq, int Tmax, int H, int hd, int drop_thr, unsigned drop_key,
                         float drop_scale, uint32_t* drop_mask, void* stream);
int neko_attn_bwd_varlen(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* kbias, const int* kstart,
                         const int* seq_off, const long long* mask_off, const float* lse, float* D, uint16_t* dqkv, int nseq,
                         long rows, int Tmax, int H, int hd, int drop_thr, unsigned drop_key, float drop_scale,
                         const uint32_t* drop_mask, void* stream);
int neko_mask_bias(const float* mask, float* kbias, int* kstart, int B, int T, void* stream);
long neko_attn_mask_dwords(int B, int T, int H, int hd);
int neko_attn_fwd(const uint16_t* qkv, const float* kbias, const int* kstart, uint16_t* out, float* lse, int B,
                  int T, int H, int hd, int drop_thr, unsigned drop_key, float drop_scale, uint32_t* drop_mask,
                  void* stream);
int neko_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* kbias,
                  const int* kstart, const float* lse, float* D, int* qflags, uint16_t* dqkv, int B, int T, int H,
                  int hd, int drop_thr, unsigned drop_key, float drop_scale, const uint32_t* drop_mask, void* stream);

/* Skinny-M (M <= 8, K <= 3072) companion of neko_gemm_bf16 for incremental decode: a pure weight stream instead of
 * the tiled MFMA loop.  y[M,N] = x[M,K] . W (+ bias[N]) (act 1 = GELU on the bf16-rounded pre-activation) (+ resid),
 * W bf16 either [K, ldw] (b_kstrided = 1, HF Conv1D (in,out) layout) or [N, ldw] (b_kstrided = 0, e.g. predict_token);
 * outputs f32 and/or bf16 like the GEMM. */
int neko_gemv_bf16(const uint16_t* x, long ldx, const uint16_t* W, long ldw, int b_kstrided, int M, int N, int K,
                   const float* bias, const float* resid, long ldr, int act, float* Cf, long ldcf, uint16_t* Cb, long ldcb,
                   void* stream);

/* Single-query attention for KV-cached decode (gato_policy.py:434-614 re-run the whole forward per generated
 * token; trajectory_gpt2.py:163-188 with a query length of 1).  cache bf16 [cap, 3*H*hd] (q|k|v rows, the layout
 * neko_attn_fwd reads), row bf16 [3*H*hd] = the freshly projected q|k|v of position *pos (device int32, so the launch
 * is shape-independent and HIP-graph capturable), out bf16 [H*hd].  Appends the row's k/v to cache[*pos]. */
int neko_attn_decode(uint16_t* cache, const uint16_t* row, const int* pos, uint16_t* out, int H, int hd, int cap,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Masked cross-entropy over a chunk of logits -- gato_policy.py:174-186 (shift, mask product,
 * boolean gather, F.cross_entropy mean) and its backward.
 *   logits f32 [R, ldl] (V valid columns), target int64 [R], weight f32 [R] (= loss_mask / N)
 *   loss_row f32 [R] = lse - logit[target] (0 where weight == 0)           (may be null)
 *   dlogits bf16 [R, ldd] = weight * (softmax - onehot), zero in columns V..Vpad-1   (may be null)
 * ------------------------------------------------------------------------------------------- */
int neko_ce_fwd_bwd(const float* logits, long ldl, int V, int Vpad, const long long* target, const float* weight,
                    float* loss_row, uint16_t* dlogits, long ldd, int R, void* stream);
/* Training path: the LM-head GEMM writes bf16 logits straight into the dlogits buffer and this call turns them into
 * the gradient IN PLACE (one read + one write of the chunk; fp32 logits never reach HBM).
 *   z bf16 [R, ld]: in = logits (columns < V valid), out = weight * (softmax - onehot), 0 in columns V..Vpad-1
 *   (untouched when want_grad == 0); loss_row as above (may be null).  Vpad <= 57344. */
int neko_ce_bf16_inplace(uint16_t* z, long ld, int V, int Vpad, const long long* target, const float* weight,
                         float* loss_row, int want_grad, int R, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Packing front-end -- GatoPolicy.tokenize_input_dicts (gato_policy.py:195-432) incl.
 * ContinuousTokenizer.encode / mu_law (gato/policy/input_tokenizers.py:5-30) and the embedding
 * lookups (gato_policy.py:117,124,149).  desc: int32 [ntok,4] = {kind, src, pos, target}:
 *   kind 0 pad | 1 token id = src | 2 continuous obs (mu-law) cont_vals[src] | 3 continuous action
 *   cont_vals[src] | 4 discrete disc_vals[src] | 5 separator | 6 image-patch row src of img_emb;
 *   pos >= 0: add pos_embed[pos]; target: 1 if the position is a prediction target.
 * fwd writes x f32 [ntok,d], tokens int64, tmask f32, pmask f32.
 * bwd scatters dx into d_embed / d_pos / d_sep (+=, f32 atomics) and copies image rows to d_img.
 * ------------------------------------------------------------------------------------------- */
int neko_pack_embed_fwd(const int* desc, const float* cont_vals, const int* disc_vals, const float* img_emb,
                        const float* embed, const float* pos_embed, const float* sep, float* x, long long* tokens,
                        float* tmask, float* pmask, int ntok, int d, float mu, float M, int n_bins, int cont_start,
                        int disc_start, void* stream);
int neko_pack_embed_bwd(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos,
                        float* d_sep, float* d_img, int ntok, int d, void* stream);
/* The same gradients WITHOUT atomics (ABI v15): the tokens are sorted by destination row (stable radix sort) and every row of
 * d_embed / d_pos / d_sep is the sum of its tokens' gradient rows IN TOKEN ORDER -- bit-identical from run to run, where the fp32
 * atomics of neko_pack_embed_bwd differ in the last bits (enough to separate two identical AdamW runs after a few steps).
 * vocab_rows = rows of d_embed, pos_rows = rows of d_pos (ABI v16): a token id / position outside its table is dropped, never
 * written; tables of 2^20 - 1 rows or more are refused (NEKO_ERR_UNSUPPORTED: the sort orders on 20 key bits).
 * workspace: neko_pack_embed_bwd_det_ws_bytes(ntok, d) bytes (256-B aligned). */
long neko_pack_embed_bwd_det_ws_bytes(int ntok, int d);
int neko_pack_embed_bwd_det(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos,
                            float* d_sep, float* d_img, int ntok, int d, int vocab_rows, int pos_rows, void* workspace,
                            long ws_bytes, void* stream);
/* ContinuousTokenizer.encode on a flat array (input_tokenizers.py:17-30) */
int neko_tokenize_continuous(const float* x, int* ids, long n, int use_mu_law, float mu, float M, int n_bins,
                             int offset, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Elementwise / optimiser tail -- Trainer.train_step (gato/training/trainer.py:181-186):
 * clip_grad_norm_(params, max_norm) then torch.optim.AdamW.step (train.py:127-133).
 *   neko_cast_f32_bf16 : weight shadow refresh (bf16 MFMA operands of fp32 master weights)
 *   neko_colsum_bf16   : out[N] (+)= column sums of bf16 [M,ld]   (Conv1D bias gradients)
 *   neko_sqnorm_f32    : *out_accum (device double) += sum(g^2)
 *   neko_adamw_step    : per contiguous parameter range; clip coefficient min(1, max_norm /
 *                        (sqrt(*gnorm_sq)+1e-6)) read on device (gnorm_sq null = no clipping);
 *                        *grad_scale (may be null) multiplies g; step = device int32 counter of the
 *                        range (bias correction); *active == 0 (may be null) skips the range the way
 *                        torch skips parameters whose grad is None; p16 (may be null) gets bf16(p);
 *                        lr_dev (may be null) = the learning rate in device memory, used instead of `lr` --
 *                        a captured (HIP-graph) step cannot change a kernel argument between replays.
 * ------------------------------------------------------------------------------------------- */
/* Dropout (nn.Dropout sites of the path: embd :541,707; attention :179; residual :254,278).  Counter-based:
 * element idx is kept iff top byte of hash(idx ^ key) >= thr with thr = round(p*256) in [0,255] (0 = off), survivors
 * scaled by scale = 256/(256-thr); forward and backward regenerate the same mask from (idx, key), no mask tensor.
 * neko_dropout_f32: y = keep ? x*scale : 0 on a flat fp32 array (embedding dropout and its backward). */
int neko_dropout_f32(const float* x, float* y, long n, int thr, unsigned key, float scale, void* stream);
/* Captured (HIP-graph) training steps freeze every kernel argument, site keys included.  neko_set_drop_salt registers ONE
 * device uint32 (or null = off, the default) that every kernel with a dropout site ADDS to its site key at entry; the
 * captured step advances it once per replay, so the masks of every site differ from step to step.  Process-wide, set
 * outside of any capture (it is a blocking copy into the library's device globals). */
int neko_set_drop_salt(const uint32_t* salt);
int neko_cast_f32_bf16(const float* x, uint16_t* y, long n, void* stream);
/* loss-position selection (gato_policy.py:183-185): dst[r,:] = r < n ? src[idx[r],:] : 0 (bf16 rows, d % 8 == 0);
 * and its adjoint dst[idx[r],:] = src[r,:] (f32 rows, dst pre-zeroed, idx unique). */
int neko_gather_rows_bf16(const uint16_t* src, const int* idx, uint16_t* dst, int n, int npad, int d, void* stream);
int neko_scatter_rows_f32(const float* src, const int* idx, float* dst, int n, int d, void* stream);
int neko_colsum_bf16(const uint16_t* x, long ld, int M, int N, float* out, int accumulate, void* stream);
int neko_sqnorm_f32(const float* g, long n, double* out_accum, void* stream);
/* GEGLU gate of the MLP (activation_fn='geglu': gato_policy.py:97-100, MLP.forward trajectory_gpt2.py:273-278,
 * h = gelu(c_fc x) * gated_layer(x)); flat bf16 arrays of n elements, 16-B aligned.
 *   neko_geglu_fwd: h *= gate in place (h holds gelu(pre) from the c_fc GEMM epilogue)
 *   neko_geglu_bwd: d_pre = dh * gate * gelu'(pre),  d_gate = dh * gelu(pre) */
int neko_geglu_fwd(uint16_t* h, const uint16_t* gate, long n, void* stream);
int neko_geglu_bwd(const uint16_t* dh, const uint16_t* pre, const uint16_t* gate, uint16_t* d_pre, uint16_t* d_gate,
                   long n, void* stream);
int neko_adamw_step(float* p, const float* g, float* m, float* v, uint16_t* p16, long n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, const double* gnorm_sq, float max_norm,
                    const float* grad_scale, int* step, const int* active, const float* lr_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image patch embedding -- ImageEmbedding.forward / ResidualBlock_V2 / PatchPosEncoding
 * (gato/policy/embeddings.py:28-61,63-110,111-131).
 *   neko_patch_resblock_fwd: images (f32 or u8, [n,3,H,W], values 0..255) -> y bf16 [P,768]
 *       (P = n*(H/16)*(W/16) patches in (b, n_h, n_w) order; y = x + conv2(GELU(GN(conv1(GELU(x)))))
 *       flattened (c p1 p2)), patch normalisation (x/255*2-1)/sqrt(16) included.
 *   neko_patch_resblock_bwd: dy f32 [P,768] -> += dW1,db1,dgamma,dbeta,dW2,db2 (recomputes the block).
 *   neko_patch_pos_add / _bwd: out[p,:] += row_emb[hpos[p]] + col_emb[wpos[p]] and its scatter.
 * The 768->d projection runs on neko_gemm_bf16.
 * ------------------------------------------------------------------------------------------- */
int neko_patch_resblock_fwd(const void* images, int images_are_u8, int n, int H, int W, const float* w1,
                            const float* b1, const float* gn_w, const float* gn_b, const float* w2,
                            const float* b2, int mid_channels, int num_groups, uint16_t* y16, float* x_patches,
                            void* stream);
int neko_patch_resblock_bwd(const float* x_patches, const float* dy, int P, const float* w1, const float* b1,
                            const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                            int mid_channels, int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b,
                            float* dw2, float* db2, float* workspace, void* stream);
/* ABI v17: the forward also leaves the GroupNorm statistics of every patch (gn_stats f32 [P, 64]: mean[32] | rstd[32] per group) and
 * the backward takes them instead of recomputing them (same values: the recomputation repeats the forward's arithmetic) */
int neko_patch_resblock_fwd_stats(const void* images, int images_are_u8, int n, int H, int W, const float* w1,
                                  const float* b1, const float* gn_w, const float* gn_b, const float* w2,
                                  const float* b2, int mid_channels, int num_groups, uint16_t* y16, float* x_patches,
                                  float* gn_stats, void* stream);
int neko_patch_resblock_bwd_stats(const float* x_patches, const float* gn_stats, const float* dy, int P, const float* w1,
                                  const float* b1, const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                                  int mid_channels, int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b,
                                  float* dw2, float* db2, float* workspace, void* stream);
/* workspace floats the backward needs for P patches (per-block partial gradient rows, reduced in fixed order) */
int neko_patch_resblock_bwd_ws_floats(int P);
int neko_patch_pos_add(float* out, const int* hpos, const int* wpos, const float* row_emb, const float* col_emb,
                       int P, int d, void* stream);
int neko_patch_pos_add_bwd(const float* dout, const int* hpos, const int* wpos, float* d_row_emb,
                           float* d_col_emb, int P, int d, void* stream);
/* ... and without atomics (ABI v15; same method as neko_pack_embed_bwd_det): rows summed in patch order.  nrows = rows of each table */
long neko_patch_pos_add_bwd_det_ws_bytes(int P, int d);
int neko_patch_pos_add_bwd_det(const float* dout, const int* hpos, const int* wpos, float* d_row_emb, float* d_col_emb,
                               int P, int d, int nrows, void* workspace, long ws_bytes, void* stream);

/* ABI v17: the contended table gradients from HOST-sorted (key, row) pairs -- no device sort, no atomics, bit-reproducible.  The host
 * builds the packing descriptors and draws the patch positions (gato/policy/gato_policy.py:195-432, gato/policy/embeddings.py:63-110), so
 * it knows these keys and sorts them (stable, ascending; entries without a destination carry the key 0xFFFFF and come last).
 *   neko_pack_embed_bwd_sorted: like neko_pack_embed_bwd, but d_pos / d_sep are fixed-order segment sums over (keys_sorted, idx_sorted)
 *       [ntok entries]: key = local position of token idx, or pos_rows for a separator token; d_embed keeps fp32 atomics.
 *   neko_patch_pos_add_bwd_sorted: like neko_patch_pos_add_bwd from (hkeys, hidx) / (wkeys, widx) [P entries each]. */
long neko_pack_embed_bwd_sorted_ws_bytes(int ntok, int d);
int neko_pack_embed_bwd_sorted(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos, float* d_sep,
                               float* d_img, int ntok, int d, int pos_rows, const int* keys_sorted, const int* idx_sorted, void* workspace,
                               long ws_bytes, void* stream);
long neko_patch_pos_add_bwd_sorted_ws_bytes(int P, int d);
int neko_patch_pos_add_bwd_sorted(const float* dout, const int* hkeys_sorted, const int* hidx_sorted, const int* wkeys_sorted,
                                  const int* widx_sorted, float* d_row_emb, float* d_col_emb, int P, int d, int nrows, void* workspace,
                                  long ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEKO_HIP_H */
