// Masked cross-entropy over a chunk of LM-head logits: loss rows + dlogits in one pass pair.
// Replaces the shift / boolean-mask gather / F.cross_entropy tail of GatoPolicy.forward
// (gato/policy/gato_policy.py:174-186) and its backward: no (N,V) gathered copy is made.
//
//   row r (a position t of the flattened batch) predicts target[r] (= tokens[t+1]); weight[r] is
//   loss_mask[r] / N (N = number of selected positions of the whole batch), 0 for unselected rows.
//   loss_row[r]  = lse(logits[r,:V]) - logits[r,target]              (0 when weight == 0)
//   dlogits[r,c] = weight[r] * (softmax(logits[r])[c] - [c==target])   bf16, 0 for c >= V (pad cols)
//
// HBM-bound: one 256-thread block per row, float4 loads, online (max,sum) in registers, block
// reduction through LDS, second sweep re-reads the row (L2-resident: 52305*4 B = 204 KiB) and writes bf16.
#include "neko_kernels.h"

namespace {

__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, long ldl, int V, int Vpad,
                                                         const long long* __restrict__ target,
                                                         const float* __restrict__ weight, float* __restrict__ loss_row,
                                                         bf16_t* __restrict__ dlogits, long ldd, int R) {
  __shared__ float red_m[4], red_s[4];
  const int row = blockIdx.x;
  if (row >= R) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float w = weight[row];
  const float* lr = logits + (long)row * ldl;
  bf16_t* dr = dlogits ? dlogits + (long)row * ldd : nullptr;

  if (w == 0.f) {   // unselected position: contributes nothing (block-uniform branch)
    if (tid == 0 && loss_row) loss_row[row] = 0.f;
    if (dr)
      for (int c = tid * 8; c < Vpad; c += 256 * 8) *reinterpret_cast<uint4*>(dr + c) = make_uint4(0, 0, 0, 0);
    return;
  }

  // pass 1: online max / sum(exp)
  float m = -INFINITY, s = 0.f;
  const int nv4 = V >> 2;
  for (int c = tid; c < nv4; c += 256) {
    const float4 v = reinterpret_cast<const float4*>(lr)[c];
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (mx > m) { s *= __expf(m - mx); m = mx; }
    s += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m));
  }
  for (int c = nv4 * 4 + tid; c < V; c += 256) {
    const float v = lr[c];
    if (v > m) { s *= __expf(m - v); m = v; }
    s += __expf(v - m);
  }
  // wave reduce (max, then rescaled sums)
  float wm = wave_max(m);
  float ws = wave_sum(s * __expf(m - wm));   // m == -inf (idle lane): s == 0, exp(-inf) = 0
  if (lane == 0) { red_m[wave] = wm; red_s[wave] = ws; }
  __syncthreads();
  float gm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  float gs = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) gs += red_s[i] * __expf(red_m[i] - gm);
  const float lse = gm + __logf(gs);
  const long long tgt = target[row];
  if (tid == 0 && loss_row) loss_row[row] = lse - lr[tgt];

  if (!dr) return;
  // pass 2: dlogits
  const int nv8 = V >> 3;
  for (int c = tid; c < (Vpad >> 3); c += 256) {
    uint4 pk = make_uint4(0, 0, 0, 0);
    const int c0 = c * 8;
    if (c < nv8) {
      const float4 a = reinterpret_cast<const float4*>(lr)[2 * c];
      const float4 b = reinterpret_cast<const float4*>(lr)[2 * c + 1];
      float p[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = __expf(p[e] - lse);
        if (c0 + e == tgt) g -= 1.f;
        p[e] = g * w;
      }
      pk.x = pack_bf16x2(p[0], p[1]); pk.y = pack_bf16x2(p[2], p[3]);
      pk.z = pack_bf16x2(p[4], p[5]); pk.w = pack_bf16x2(p[6], p[7]);
    } else if (c0 < V) {   // ragged tail chunk (V % 8 != 0)
      float p[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = 0.f;
        if (c0 + e < V) {
          g = __expf(lr[c0 + e] - lse);
          if (c0 + e == tgt) g -= 1.f;
          g *= w;
        }
        p[e] = g;
      }
      pk.x = pack_bf16x2(p[0], p[1]); pk.y = pack_bf16x2(p[2], p[3]);
      pk.z = pack_bf16x2(p[4], p[5]); pk.w = pack_bf16x2(p[6], p[7]);
    }
    *reinterpret_cast<uint4*>(dr + c0) = pk;
  }
}

// ---- bf16 in-place variant (the training path) --------------------------------------------------------------------
// z bf16 [R, ld]: on entry the logits of the chunk (columns < V valid), written by the LM-head GEMM straight into the
// dlogits buffer; on exit weight * (softmax - onehot) in the same place, zeros in columns V..Vpad-1.  One 1024-thread
// block per row holds the whole row in registers (7 x 16 B per thread for Vpad <= 57344): the logits are read ONCE
// and never exist in fp32 in HBM -- per 4096-row chunk 0.43 GB read + 0.43 GB written instead of 1.7 GB + 0.43 GB
// (and the GEMM writes half as much).  The bf16 rounding of the logits perturbs a row's loss by ~2^-9 |z| with zero
// mean (batch mean: ~1e-6 relative); dlogits are bf16 either way.
constexpr int CE_NT = 1024, CE_MAXCH = 7;

__global__ __launch_bounds__(CE_NT) void ce_bf16_inplace_kernel(bf16_t* __restrict__ z, long ld, int V, int Vpad,
                                                               const long long* __restrict__ target,
                                                               const float* __restrict__ weight,
                                                               float* __restrict__ loss_row, int want_grad, int R) {
  __shared__ float red_m[16], red_s[16];
  const int row = blockIdx.x;
  if (row >= R) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float w = weight[row];
  bf16_t* zr = z + (long)row * ld;
  const int nch = Vpad >> 3;                       // 16-B chunks of the row
  if (w == 0.f) {                                  // unselected position (block-uniform)
    if (tid == 0 && loss_row) loss_row[row] = 0.f;
    if (want_grad)
      for (int c = tid; c < nch; c += CE_NT) reinterpret_cast<uint4*>(zr)[c] = make_uint4(0, 0, 0, 0);
    return;
  }
  const long long tgt = target[row];
  const float z_t = bf16_to_f32(zr[tgt]);          // read before anything is overwritten
  // the whole row is requested with unconditional (clamped) loads before anything looks at it: written as `if (c < nch)
  // { load; mask; max }` every one of the 7 loads got its own branch with a `s_waitcnt vmcnt(0)` behind it -- seven
  // dependent HBM round trips per row with two workgroups per CU to hide them (3.8 TB/s)
  uint4 reg[CE_MAXCH];
#pragma unroll
  for (int i = 0; i < CE_MAXCH; ++i) reg[i] = reinterpret_cast<const uint4*>(zr)[min(tid + i * CE_NT, nch - 1)];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < CE_MAXCH; ++i) {
    const int c = tid + i * CE_NT;
    uint4 v = reg[i];
    if (c >= nch) v = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);      // -inf bf16 pairs
    else if (c * 8 + 8 > V) {                      // chunk straddles or lies beyond V: invalid columns -> -inf
      uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c * 8 + e >= V) wv[e >> 1] = (e & 1) ? ((wv[e >> 1] & 0x0000ffffu) | 0xff800000u) : ((wv[e >> 1] & 0xffff0000u) | 0x0000ff80u);
      v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    reg[i] = v;
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(__uint_as_float(wv[e] << 16), __uint_as_float(wv[e] & 0xffff0000u)));
  }
  // block max
  m = wave_max(m);
  if (lane == 0) red_m[wave] = m;
  __syncthreads();
  float gm = red_m[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) gm = fmaxf(gm, red_m[i]);
  // sum of exp (invalid columns are -inf -> 0)
  const float gm2 = gm * 1.4426950408889634f;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CE_MAXCH; ++i) {
    const uint32_t wv[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      s += __builtin_amdgcn_exp2f(fmaf(__uint_as_float(wv[e] << 16), 1.4426950408889634f, -gm2)) +
           __builtin_amdgcn_exp2f(fmaf(__uint_as_float(wv[e] & 0xffff0000u), 1.4426950408889634f, -gm2));
  }
  s = wave_sum(s);
  if (lane == 0) red_s[wave] = s;
  __syncthreads();
  float gs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) gs += red_s[i];
  const float lse = gm + __logf(gs);
  if (tid == 0 && loss_row) loss_row[row] = lse - z_t;
  if (!want_grad) return;
  // weight folded into the exponent: w * 2^(z - lse) = 2^(z - lse + log2 w) for w > 0 (w == 0 rows returned above; a
  // negative weight -- never produced by the policy -- takes the multiply, block-uniform)
  const bool fold = w > 0.f;
  const float lse2 = lse * 1.4426950408889634f - (fold ? __log2f(w) : 0.f);
#pragma unroll
  for (int i = 0; i < CE_MAXCH; ++i) {
    const int c = tid + i * CE_NT;
    if (c >= nch) continue;
    const uint32_t wv[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    float p[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[2 * e] = __builtin_amdgcn_exp2f(fmaf(__uint_as_float(wv[e] << 16), 1.4426950408889634f, -lse2));
      p[2 * e + 1] = __builtin_amdgcn_exp2f(fmaf(__uint_as_float(wv[e] & 0xffff0000u), 1.4426950408889634f, -lse2));
    }
    if (!fold) {
#pragma unroll
      for (int e = 0; e < 8; ++e) p[e] *= w;
    }
    const int c0 = c * 8;
    // the one-hot term lives in exactly one 16-B chunk of the row: a rare (1 in Vpad/8) thread-level branch instead of a
    // compare + select on every element
    if (tgt >= c0 && tgt < c0 + 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e == tgt) p[e] -= w;
    }
    reinterpret_cast<uint4*>(zr)[c] = make_uint4(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]),
                                                 pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7]));
  }
}

}  // namespace

int neko_ce_fwd_bwd_impl(const float* logits, long ldl, int V, int Vpad, const long long* target, const float* weight,
                         float* loss_row, bf16_t* dlogits, long ldd, int R, hipStream_t s) {
  if (R <= 0) return NEKO_OK;
  if (!logits || !target || !weight || (!loss_row && !dlogits)) return NEKO_ERR_ARG;
  if (V <= 0 || Vpad < V || (Vpad & 7) || (ldl & 3) || (dlogits && ((ldd & 7) || ldd < Vpad))) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(R), dim3(256), 0, s, logits, ldl, V, Vpad, target, weight, loss_row,
                     dlogits, ldd, R);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_ce_bf16_inplace_impl(bf16_t* z, long ld, int V, int Vpad, const long long* target, const float* weight,
                              float* loss_row, int want_grad, int R, hipStream_t s) {
  if (R <= 0) return NEKO_OK;
  if (!z || !target || !weight || (!loss_row && !want_grad)) return NEKO_ERR_ARG;
  if (V <= 0 || Vpad < V || (Vpad & 7) || (ld & 7) || ld < Vpad) return NEKO_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(z) & 15)) return NEKO_ERR_ARG;
  if ((Vpad >> 3) > CE_NT * CE_MAXCH) return NEKO_ERR_UNSUPPORTED;      // row does not fit the register file
  hipLaunchKernelGGL(ce_bf16_inplace_kernel, dim3(R), dim3(CE_NT), 0, s, z, ld, V, Vpad, target, weight, loss_row,
                     want_grad, R);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
