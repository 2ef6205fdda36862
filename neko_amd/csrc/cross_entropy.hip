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
