// Single-query attention for incremental decode: the newest row of ONE unpadded sequence
// against the per-layer [cap, 3d] bf16 q|k|v cache that neko_attn_fwd also reads (trajectory_gpt2.py:163-188 with a
// query length of 1: every cached key is visible, nothing is masked).
//
// The row index comes from DEVICE memory (*pos): the launch is shape-independent, so the whole per-token step
// (LayerNorm, GEMMs, this kernel, LM head, argmax, embedding lookup) is captured once in a HIP graph and replayed --
// at B = 1 the step is ~70 launches and purely launch-bound otherwise.  The freshly projected q|k|v row arrives in a
// staging buffer (the GEMM's output address must be static too); each head's block appends its k/v slice to the
// cache before using it, so no other launch has to touch the cache.
//
// One 256-thread block per head.  Pass 1: thread per key, fp32 dot with the scaled query, scores in LDS, block
// max / sum(exp2).  Pass 2: thread (c, g) accumulates output channel c over the keys t = g (mod 256/hd) with
// coalesced 2*hd-byte rows of V, the groups meet in LDS.  HBM/L2 traffic: 2 * (n+1) * d * 2 B per layer.
#include "neko_kernels.h"

namespace {

constexpr int DNT = 256;
constexpr int DMAX_T = 8192;          // scores kept in LDS: cap <= 8192 positions

template <int HD>
__global__ __launch_bounds__(DNT) void attn_decode_kernel(bf16_t* __restrict__ cache, const bf16_t* __restrict__ row,
                                                          const int* __restrict__ pos, bf16_t* __restrict__ out, int H,
                                                          int cap, float scale_log2e) {
  __shared__ float sc[DMAX_T];
  __shared__ float qs[HD];
  __shared__ float red[DNT / 64];
  __shared__ float part[DNT / 64][HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, d = H * HD;
  const long ld = 3L * d;
  int n = *pos;                        // index of the new row
  if (n < 0 || n >= cap || n >= DMAX_T) return;
  // append this head's k / v slice of the new row to the cache; keep the scaled query in LDS
  if (tid < HD) {
    qs[tid] = bf16_to_f32(row[h * HD + tid]) * scale_log2e;
    cache[(long)n * ld + d + h * HD + tid] = row[d + h * HD + tid];
    cache[(long)n * ld + 2 * d + h * HD + tid] = row[2 * d + h * HD + tid];
  }
  __syncthreads();
  // pass 1: scores (exp2 domain)
  float mx = -INFINITY;
  for (int t = tid; t <= n; t += DNT) {
    const bf16_t* kp = (t == n ? row : cache + (long)t * ld) + d + h * HD;      // own writes above may not be visible yet
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
      const uint4 kv = *reinterpret_cast<const uint4*>(kp + c);
      const uint32_t w[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = fmaf(qs[c + 2 * e], __uint_as_float(w[e] << 16), s);
        s = fmaf(qs[c + 2 * e + 1], __uint_as_float(w[e] & 0xffff0000u), s);
      }
    }
    sc[t] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int t = tid; t <= n; t += DNT) {
    const float p = __builtin_amdgcn_exp2f(sc[t] - mx);
    sc[t] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  // pass 2: out[c] = sum_t p_t * v_t[c].  Thread (kl, ch): key lane kl = tid / (HD/8) walks keys kl, kl + KL, ..., with one
  // 16-B load of 8 channels each (HD/8 threads cover a V row); the KL key lanes meet in LDS.
  constexpr int CH = HD / 8, KL = DNT / CH;
  const int ch = tid % CH, kl = tid / CH;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 4
  for (int t = kl; t <= n; t += KL) {
    const bf16_t* vp = (t == n ? row : cache + (long)t * ld) + 2 * d + h * HD + ch * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(vp);
    const float pt = sc[t];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[2 * e] = fmaf(pt, __uint_as_float(w[e] << 16), acc[2 * e]);
      acc[2 * e + 1] = fmaf(pt, __uint_as_float(w[e] & 0xffff0000u), acc[2 * e + 1]);
    }
  }
  // reduce over key lanes: first inside the wave (lanes with equal ch: stride CH), then across the 4 waves
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int o = CH; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
  if (lane < CH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave][lane * 8 + e] = acc[e];
  }
  __syncthreads();
  if (tid < HD) {
    const float o = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    out[h * HD + tid] = f32_to_bf16(o * inv);
  }
}

}  // namespace

int neko_attn_decode_impl(bf16_t* cache, const bf16_t* row, const int* pos, bf16_t* out, int H, int hd, int cap,
                          hipStream_t s) {
  if (!cache || !row || !pos || !out || H <= 0 || cap <= 0) return NEKO_ERR_ARG;
  if (cap > DMAX_T) return NEKO_ERR_UNSUPPORTED;
  const float sl = 1.4426950408889634f / sqrtf((float)hd);
  switch (hd) {
    case 32: hipLaunchKernelGGL((attn_decode_kernel<32>), dim3(H), dim3(DNT), 0, s, cache, row, pos, out, H, cap, sl); break;
    case 64: hipLaunchKernelGGL((attn_decode_kernel<64>), dim3(H), dim3(DNT), 0, s, cache, row, pos, out, H, cap, sl); break;
    case 128: hipLaunchKernelGGL((attn_decode_kernel<128>), dim3(H), dim3(DNT), 0, s, cache, row, pos, out, H, cap, sl); break;
    default: return NEKO_ERR_UNSUPPORTED;
  }
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
