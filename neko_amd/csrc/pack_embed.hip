// Front-end of GatoPolicy.forward: tokenise + embed + interleave + left-pad in ONE pass over a
// per-position descriptor table, and its backward (scatter of d_x into the embedding tables).
// Replaces the per-example Python loop of tokenize_input_dicts (gato/policy/gato_policy.py:245-432),
// ContinuousTokenizer.encode + mu_law (gato/policy/input_tokenizers.py:5-30), embed_token /
// pos_embed_observation / separator_token lookups (gato_policy.py:117,124,149,275,300,311,321,334,343,380-385).
//
// The host builds, per batch, one int4 descriptor per (b,t) position {kind, src, pos, target}:
//   kind 0 PAD       : zeros, token 0, pad mask 0                      (left / right padding :408-431)
//   kind 1 TOKEN     : token id = src                                   (text ids are host lists :264-277)
//   kind 2 CONT_OBS  : id = bin(mu_law(cont[src])) + continuous_start   (:298-300)
//   kind 3 CONT_ACT  : id = bin(cont[src]) + continuous_start           (:319-321)
//   kind 4 DISCRETE  : id = disc[src] + discrete_start                  (:308-311, :329-334)
//   kind 5 SEP       : separator_token vector, token 0                  (:343-345)
//   kind 6 IMAGE     : row src of the patch-embedding buffer, token 0   (:282-292)
//   kind 7 DEVID     : token id = disc[src] (ids handed over as a device tensor, :268-274)
//   pos >= 0 adds pos_embed_observation[pos] (observation tokens only, :380-385); target -> target mask.
// HBM-bound gather: one wave64 per position, float4 per lane, d*4 B read + d*4 B written per token.
#include "neko_kernels.h"

namespace {

enum { K_PAD = 0, K_TOKEN = 1, K_CONT_OBS = 2, K_CONT_ACT = 3, K_DISC = 4, K_SEP = 5, K_IMAGE = 6, K_DEVID = 7 };

struct PackArgs {
  const int4* desc;
  const float* cont_vals;
  const int* disc_vals;
  const float* img_emb;     // [n_img_rows, d]
  const float* embed;       // [V, d]
  const float* pos_embed;   // [ctx, d]
  const float* sep;         // [d]
  float* x;                 // [ntok, d]
  long long* tokens;        // [ntok]
  float* tmask;             // [ntok]
  float* pmask;             // [ntok]
  int ntok, d;
  float mu, inv_log_den;    // mu-law: log(1+mu|x|) / log(1+mu*M)   (denominator passed as fp32 value)
  float log_den;
  float half_bins;          // n_bins / 2
  int cont_start, disc_start;
};

// ContinuousTokenizer.encode (input_tokenizers.py:17-30) with torch's fp32 op order:
//   t = sign(x) * log(1 + mu*|x|) / log_den ; clamp(-1,1) ; (t+1)*(n_bins/2) ; trunc to int32
// log is evaluated in fp64 and rounded once (correctly-rounded logf), mul/add kept un-fused.
__device__ __forceinline__ int tokenize_cont(float x, bool mu_law, float mu, float log_den, float half_bins) {
  float t = x;
  if (mu_law) {
    const float a = fabsf(x);
    const float arg = __fadd_rn(1.0f, __fmul_rn(mu, a));
    const float l = (float)log((double)arg);
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    t = __fdiv_rn(__fmul_rn(sgn, l), log_den);
  }
  t = fminf(fmaxf(t, -1.f), 1.f);
  t = __fmul_rn(__fadd_rn(t, 1.0f), half_bins);
  return (int)t;   // truncation toward zero (values are >= 0)
}

__global__ __launch_bounds__(256) void pack_embed_fwd_kernel(PackArgs a) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= a.ntok) return;
  const int4 ds = a.desc[tok];
  const int kind = ds.x, src = ds.y, pos = ds.z;
  long long id = 0;
  const float* row = nullptr;
  if (kind == K_TOKEN) id = src;
  else if (kind == K_CONT_OBS) id = a.cont_start + tokenize_cont(a.cont_vals[src], true, a.mu, a.log_den, a.half_bins);
  else if (kind == K_CONT_ACT) id = a.cont_start + tokenize_cont(a.cont_vals[src], false, a.mu, a.log_den, a.half_bins);
  else if (kind == K_DISC) id = (long long)a.disc_vals[src] + a.disc_start;
  else if (kind == K_DEVID) id = (long long)a.disc_vals[src];   // token ids that live on the device (text as a device tensor)
  if ((kind >= K_TOKEN && kind <= K_DISC) || kind == K_DEVID) row = a.embed + id * (long)a.d;
  else if (kind == K_SEP) row = a.sep;
  else if (kind == K_IMAGE) row = a.img_emb + (long)src * a.d;
  if (lane == 0) {
    a.tokens[tok] = id;
    a.tmask[tok] = ds.w ? 1.f : 0.f;
    a.pmask[tok] = (kind == K_PAD) ? 0.f : 1.f;
  }
  const float4* r4 = reinterpret_cast<const float4*>(row);
  const float4* p4 = (pos >= 0 && kind != K_PAD) ? reinterpret_cast<const float4*>(a.pos_embed + (long)pos * a.d) : nullptr;
  float4* o4 = reinterpret_cast<float4*>(a.x + (long)tok * a.d);
  for (int c = lane; c < (a.d >> 2); c += 64) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row) v = r4[c];
    if (p4) { const float4 p = p4[c]; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
    o4[c] = v;
  }
}

struct PackBwdArgs {
  const int4* desc;
  const long long* tokens;
  const float* dx;          // [ntok, d]
  float* d_embed;           // [V, d]      (+=, atomics)
  float* d_pos;             // [ctx, d]    (+=)
  float* d_sep;             // [d]         (+=)
  float* d_img;             // [n_img_rows, d]  (=, each row written once) or null
  int ntok, d;
};

__global__ __launch_bounds__(256) void pack_embed_bwd_kernel(PackBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= a.ntok) return;
  const int4 ds = a.desc[tok];
  const int kind = ds.x, src = ds.y, pos = ds.z;
  if (kind == K_PAD) return;
  const float* g = a.dx + (long)tok * a.d;
  float* dst = nullptr;
  if ((kind >= K_TOKEN && kind <= K_DISC) || kind == K_DEVID) dst = a.d_embed + a.tokens[tok] * (long)a.d;
  else if (kind == K_SEP) dst = a.d_sep;
  float* dpos = (pos >= 0 && a.d_pos) ? a.d_pos + (long)pos * a.d : nullptr;      // null table: its sums are formed elsewhere
  float* dimg = (kind == K_IMAGE && a.d_img) ? a.d_img + (long)src * a.d : nullptr;
  if (dimg) {
    for (int c = lane * 4; c < a.d; c += 256) *reinterpret_cast<float4*>(dimg + c) = *reinterpret_cast<const float4*>(g + c);
  }
  // one atomic wave-instruction covers 64 CONSECUTIVE floats (two 128-B lines): with four elements per lane each
  // instruction touched eight lines at a quarter of their width and the L2 atomic units saw four times the requests
  if (dst || dpos) {
    for (int c = lane; c < a.d; c += 64) {
      const float v = g[c];
      if (dst) atomicAdd(dst + c, v);
      if (dpos) atomicAdd(dpos + c, v);
    }
  }
}

// destinations of the packing backward as sort keys (segsum.hip): embedding-table row (the separator's own parameter = row `sep_key`)
// and local-position row, NEKO_SEGSUM_KEY_NONE where a token has none; image rows are copied here (each is written once)
__global__ __launch_bounds__(256) void pack_embed_bwd_keys_kernel(PackBwdArgs a, unsigned* __restrict__ key_e, unsigned* __restrict__ key_p,
                                                                  unsigned sep_key, unsigned pos_rows) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= a.ntok) return;
  const int4 ds = a.desc[tok];
  const int kind = ds.x, src = ds.y, pos = ds.z;
  unsigned ke = NEKO_SEGSUM_KEY_NONE, kp = NEKO_SEGSUM_KEY_NONE;
  if (kind != K_PAD) {
    // ids / positions outside their table get no key at all (dropped), so nothing above 20 bits reaches the sort
    if ((kind >= K_TOKEN && kind <= K_DISC) || kind == K_DEVID) {
      const unsigned long long t = (unsigned long long)a.tokens[tok];
      if (t < sep_key) ke = (unsigned)t;
    } else if (kind == K_SEP) ke = sep_key;
    if (pos >= 0 && (unsigned)pos < pos_rows) kp = (unsigned)pos;
    if (kind == K_IMAGE && a.d_img) {
      const float* g = a.dx + (long)tok * a.d;
      float* dimg = a.d_img + (long)src * a.d;
      for (int c = lane * 4; c < a.d; c += 256) *reinterpret_cast<float4*>(dimg + c) = *reinterpret_cast<const float4*>(g + c);
    }
  }
  if (lane == 0) { key_e[tok] = ke; key_p[tok] = kp; }
}

// standalone tokenizer (predict_* / tests): ids[i] = offset + bin(x[i])
__global__ void tokenize_cont_kernel(const float* __restrict__ x, int* __restrict__ ids, long n, int use_mu_law, float mu,
                                     float log_den, float half_bins, int offset) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] = offset + tokenize_cont(x[i], use_mu_law != 0, mu, log_den, half_bins);
}

}  // namespace

int neko_pack_embed_fwd_impl(const int* desc, const float* cont_vals, const int* disc_vals, const float* img_emb,
                             const float* embed, const float* pos_embed, const float* sep, float* x,
                             long long* tokens, float* tmask, float* pmask, int ntok, int d, float mu, float M,
                             int n_bins, int cont_start, int disc_start, hipStream_t s) {
  if (ntok <= 0) return NEKO_OK;
  if (!desc || !embed || !pos_embed || !sep || !x || !tokens || !tmask || !pmask || (d & 3)) return NEKO_ERR_ARG;
  PackArgs a;
  a.desc = reinterpret_cast<const int4*>(desc);
  a.cont_vals = cont_vals; a.disc_vals = disc_vals; a.img_emb = img_emb;
  a.embed = embed; a.pos_embed = pos_embed; a.sep = sep;
  a.x = x; a.tokens = tokens; a.tmask = tmask; a.pmask = pmask;
  a.ntok = ntok; a.d = d; a.mu = mu;
  a.log_den = (float)log(1.0 + (double)mu * (double)M);   // math.log(1 + mu*M) as the fp32 divisor torch uses
  a.inv_log_den = 1.0f / a.log_den;
  a.half_bins = (float)n_bins / 2.0f;
  a.cont_start = cont_start; a.disc_start = disc_start;
  hipLaunchKernelGGL(pack_embed_fwd_kernel, dim3((ntok + 3) / 4), dim3(256), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_pack_embed_bwd_impl(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos,
                             float* d_sep, float* d_img, int ntok, int d, hipStream_t s) {
  if (ntok <= 0) return NEKO_OK;
  if (!desc || !tokens || !dx || !d_embed || !d_pos || !d_sep || (d & 3)) return NEKO_ERR_ARG;
  PackBwdArgs a{reinterpret_cast<const int4*>(desc), tokens, dx, d_embed, d_pos, d_sep, d_img, ntok, d};
  hipLaunchKernelGGL(pack_embed_bwd_kernel, dim3((ntok + 3) / 4), dim3(256), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// Round 5: the position-table and separator gradients -- the two heavily contended destinations (a local position row takes ~900 adds
// per column in an m-mix step, the separator ~1500) -- as fixed-order segment sums over HOST-sorted (key, token) pairs: key = local
// position, or pos_rows for a separator token (segsum's extra row); the embedding-table rows keep their atomics (their ids are only
// known on the device for continuous values).  keys_sorted / idx_sorted hold ntok entries, NEKO_SEGSUM_KEY_NONE-padded at the end.
long neko_pack_embed_bwd_sorted_ws_bytes_impl(int ntok, int d) { return ntok > 0 ? (long)neko_segsum_sorted_ws_bytes_impl(ntok, d) : 0; }
int neko_pack_embed_bwd_sorted_impl(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos, float* d_sep,
                                    float* d_img, int ntok, int d, int pos_rows, const unsigned* keys_sorted, const int* idx_sorted,
                                    void* ws, long ws_bytes, hipStream_t s) {
  if (ntok <= 0) return NEKO_OK;
  if (!desc || !tokens || !dx || !d_embed || !d_pos || !d_sep || !keys_sorted || !idx_sorted || !ws || (d & 3) || pos_rows <= 0) return NEKO_ERR_ARG;
  if ((unsigned)pos_rows >= NEKO_SEGSUM_KEY_NONE) return NEKO_ERR_UNSUPPORTED;
  PackBwdArgs a{reinterpret_cast<const int4*>(desc), tokens, dx, d_embed, nullptr, nullptr, d_img, ntok, d};
  hipLaunchKernelGGL(pack_embed_bwd_kernel, dim3((ntok + 3) / 4), dim3(256), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return neko_segsum_rows_sorted_impl(dx, d, keys_sorted, idx_sorted, ntok, d, d_pos, d, pos_rows, d_sep, ws, (size_t)ws_bytes, s);
}

long neko_pack_embed_bwd_det_ws_bytes_impl(int ntok, int d) {
  if (ntok <= 0) return 0;
  return (long)(((size_t)ntok * 4 + 255) / 256 * 256 * 2 + neko_segsum_ws_bytes_impl(ntok, d));
}
// the gradients of neko_pack_embed_bwd without atomics: every table row is the sum of its tokens' gradient rows in token order
int neko_pack_embed_bwd_det_impl(const int* desc, const long long* tokens, const float* dx, float* d_embed, float* d_pos, float* d_sep,
                                 float* d_img, int ntok, int d, int vocab_rows, int pos_rows, void* ws, long ws_bytes, hipStream_t s) {
  if (ntok <= 0) return NEKO_OK;
  if (!desc || !tokens || !dx || !d_embed || !d_pos || !d_sep || !ws || (d & 3) || vocab_rows <= 0 || pos_rows <= 0) return NEKO_ERR_ARG;
  if ((unsigned)vocab_rows >= NEKO_SEGSUM_KEY_NONE || (unsigned)pos_rows >= NEKO_SEGSUM_KEY_NONE) return NEKO_ERR_UNSUPPORTED;
  if (ws_bytes < neko_pack_embed_bwd_det_ws_bytes_impl(ntok, d)) return NEKO_ERR_ARG;
  const size_t kb = ((size_t)ntok * 4 + 255) / 256 * 256;
  unsigned* key_e = static_cast<unsigned*>(ws);
  unsigned* key_p = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + kb);
  void* sws = static_cast<char*>(ws) + 2 * kb;
  const size_t sws_bytes = (size_t)ws_bytes - 2 * kb;
  PackBwdArgs a{reinterpret_cast<const int4*>(desc), tokens, dx, d_embed, d_pos, d_sep, d_img, ntok, d};
  hipLaunchKernelGGL(pack_embed_bwd_keys_kernel, dim3((ntok + 3) / 4), dim3(256), 0, s, a, key_e, key_p, (unsigned)vocab_rows, (unsigned)pos_rows);
  NEKO_CHECK_LAUNCH();
  int rc = neko_segsum_rows_impl(dx, d, key_e, ntok, d, d_embed, d, vocab_rows, d_sep, sws, sws_bytes, s);
  if (rc != NEKO_OK) return rc;
  return neko_segsum_rows_impl(dx, d, key_p, ntok, d, d_pos, d, pos_rows, nullptr, sws, sws_bytes, s);
}

int neko_tokenize_continuous_impl(const float* x, int* ids, long n, int use_mu_law, float mu, float M, int n_bins,
                                  int offset, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!x || !ids) return NEKO_ERR_ARG;
  const float log_den = (float)log(1.0 + (double)mu * (double)M);
  hipLaunchKernelGGL(tokenize_cont_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ids, n, use_mu_law,
                     mu, log_den, (float)n_bins / 2.0f, offset);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
