// HBM-bound helpers of the path: fp32->bf16 weight shadow cast, bias-gradient column sums,
// attention-mask bias, and the optimiser tail (global grad norm, clip, AdamW) of Trainer.train_step
// (gato/training/trainer.py:181-186; torch.optim.AdamW set up at train.py:127-133).
// All kernels are grid-stride, 16-B vectorised, and read their scalars from device memory so the
// host never synchronises inside a step.
#include "neko_kernels.h"

namespace {

// ---- fp32 -> bf16 (weights -> MFMA operand shadow) ---------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4*>(x + i);
      const float4 b = *reinterpret_cast<const float4*>(x + i + 4);
      uint4 pk;
      pk.x = pack_bf16x2(a.x, a.y); pk.y = pack_bf16x2(a.z, a.w);
      pk.z = pack_bf16x2(b.x, b.y); pk.w = pack_bf16x2(b.z, b.w);
      *reinterpret_cast<uint4*>(y + i) = pk;
    } else {
      for (long j = i; j < n; ++j) y[j] = f32_to_bf16(x[j]);
    }
  }
}

// ---- out[N] (+)= column sums of a bf16 [M, ld] matrix (bias gradients) -------------------------------
// block = 256 threads = 4 waves; a wave covers 512 columns (8 per lane, 16-B loads); blockIdx.x = column
// group, blockIdx.y = row slice; partial sums meet in fp32 atomics (out zeroed by the host when !accumulate).
// block = 256 threads = 32 column-chunks (8 bf16 = 16 B each -> 256 columns) x 8 row lanes; rows are walked 4 at a
// time (4 independent 16-B loads in flight per thread), the 8 row lanes meet in LDS, one atomic per column per block.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long ld, int M, int N,
                                                          float* __restrict__ out) {
  __shared__ float red[8][256 + 8];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + cl) * 8;
  const bool ok = c0 < N;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int step = gridDim.y * 8;
  auto add = [&](const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[2 * e] += bf16_to_f32((bf16_t)(w[e] & 0xffff));
      acc[2 * e + 1] += bf16_to_f32((bf16_t)(w[e] >> 16));
    }
  };
  if (ok) {
    int r = blockIdx.y * 8 + rl;
    for (; r + 3 * step < M; r += 4 * step) {
      const uint4 v0 = *reinterpret_cast<const uint4*>(x + (long)r * ld + c0);
      const uint4 v1 = *reinterpret_cast<const uint4*>(x + (long)(r + step) * ld + c0);
      const uint4 v2 = *reinterpret_cast<const uint4*>(x + (long)(r + 2 * step) * ld + c0);
      const uint4 v3 = *reinterpret_cast<const uint4*>(x + (long)(r + 3 * step) * ld + c0);
      add(v0); add(v1); add(v2); add(v3);
    }
    for (; r < M; r += step) add(*reinterpret_cast<const uint4*>(x + (long)r * ld + c0));
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cl * 8 + e] = acc[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i][threadIdx.x];
    atomicAdd(out + c, s);
  }
}

// ---- additive key bias (1-mask)*-1e4 and the number of leading masked keys per sequence ---------------
// (trajectory_gpt2.py:663-679).  one block per sequence.
__global__ void mask_bias_kernel(const float* __restrict__ mask, float* __restrict__ kbias, int* __restrict__ kstart,
                                 int T) {
  __shared__ int first_valid;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) first_valid = T;
  __syncthreads();
  int fv = T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float m = mask[(long)b * T + t];
    kbias[(long)b * T + t] = (1.0f - m) * -10000.0f;
    if (m != 0.f && t < fv) fv = t;
  }
  atomicMin(&first_valid, fv);
  __syncthreads();
  if (threadIdx.x == 0 && kstart) kstart[b] = first_valid;
}

// ---- sum of squares of an fp32 range into a device double accumulator ---------------------------------
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0.0;
  // four independent 16-B loads per thread and iteration (64 KB in flight per block): with one load per iteration the
  // pass ran at 2.6 TB/s; the 16 squares are summed in fp32 (pairwise), then added to the fp64 accumulator
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthr = (long)gridDim.x * blockDim.x;
  const long n4 = n >> 2;                                   // whole float4s
  const float4* g4 = reinterpret_cast<const float4*>(g);
  long i = tid;
  for (; i + 3 * nthr < n4; i += 4 * nthr) {
    const float4 a = g4[i], b = g4[i + nthr], c = g4[i + 2 * nthr], d = g4[i + 3 * nthr];
    const float sa = (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    const float sb = (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
    const float sc = (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
    const float sd = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    acc += (double)((sa + sb) + (sc + sd));
  }
  for (; i < n4; i += nthr) {
    const float4 v = g4[i];
    acc += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
  }
  if (tid == 0)
    for (long j = n4 << 2; j < n; ++j) acc += (double)(g[j] * g[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// ---- fused clip + AdamW (+ bf16 shadow refresh) over one contiguous parameter range -------------------
// torch semantics: p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//                  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)      with g already clipped:
// clip coef = min(1, max_norm / (sqrt(*gnorm_sq) + 1e-6))  (clip_grad_norm_), read from device memory.
// state[0] = step count of this range (incremented by the first thread *after* use via a second kernel),
// `active` (device int, may be null) lets a range whose parameters took no part in the step be skipped
// exactly like torch skips grad=None parameters.
struct AdamArgs {
  float* p; const float* g; float* m; float* v; bf16_t* p16; long n;
  float lr, beta1, beta2, eps, wd, max_norm;
  const double* gnorm_sq;     // null: no clipping
  const float* grad_scale;    // optional extra multiplier on g (e.g. 1/world) or null
  int* step;                  // device step counter of this range (already incremented for this step)
  const int* active;          // null = always
  const float* lr_dev;        // optional: the learning rate in device memory (captured steps: kernel arguments are frozen)
};
__global__ void adam_step_inc_kernel(int* step, const int* active) {
  if (!active || *active) *step += 1;
}
__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  if (a.active && *a.active == 0) return;
  // g_eff = grad_scale * g (e.g. 1/world after a SUM all-reduce); the clip norm is the norm of g_eff
  const float gs = a.grad_scale ? *a.grad_scale : 1.f;
  float coef = gs;
  if (a.gnorm_sq) {
    const float nrm = (float)sqrt(*a.gnorm_sq) * gs;
    coef = fminf(1.f, a.max_norm / (nrm + 1e-6f)) * gs;
  }
  const int t = *a.step;
  if (a.lr_dev) a.lr = *a.lr_dev;
  const float bc1 = 1.f - powf(a.beta1, (float)t);
  const float bc2 = 1.f - powf(a.beta2, (float)t);
  const float step_size = a.lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const float decay = 1.f - a.lr * a.wd;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < a.n; i += stride) {
    float pv[4], gv[4], mv[4], vv[4];
    const int cnt = (i + 4 <= a.n) ? 4 : (int)(a.n - i);
    if (cnt == 4) {
      const float4 P = *reinterpret_cast<const float4*>(a.p + i), G = *reinterpret_cast<const float4*>(a.g + i);
      const float4 Mm = *reinterpret_cast<const float4*>(a.m + i), Vv = *reinterpret_cast<const float4*>(a.v + i);
      pv[0] = P.x; pv[1] = P.y; pv[2] = P.z; pv[3] = P.w;
      gv[0] = G.x; gv[1] = G.y; gv[2] = G.z; gv[3] = G.w;
      mv[0] = Mm.x; mv[1] = Mm.y; mv[2] = Mm.z; mv[3] = Mm.w;
      vv[0] = Vv.x; vv[1] = Vv.y; vv[2] = Vv.z; vv[3] = Vv.w;
    } else {
      for (int e = 0; e < cnt; ++e) { pv[e] = a.p[i + e]; gv[e] = a.g[i + e]; mv[e] = a.m[i + e]; vv[e] = a.v[i + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < cnt) {
        const float g = gv[e] * coef;
        float p = pv[e] * decay;
        const float m = a.beta1 * mv[e] + (1.f - a.beta1) * g;
        const float v = a.beta2 * vv[e] + (1.f - a.beta2) * g * g;
        p -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
        pv[e] = p; mv[e] = m; vv[e] = v;
      }
    }
    if (cnt == 4) {
      *reinterpret_cast<float4*>(a.p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(a.m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(a.v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (a.p16) {
        uint2 pk;
        pk.x = pack_bf16x2(pv[0], pv[1]); pk.y = pack_bf16x2(pv[2], pv[3]);
        *reinterpret_cast<uint2*>(a.p16 + i) = pk;
      }
    } else {
      for (int e = 0; e < cnt; ++e) {
        a.p[i + e] = pv[e]; a.m[i + e] = mv[e]; a.v[i + e] = vv[e];
        if (a.p16) a.p16[i + e] = f32_to_bf16(pv[e]);
      }
    }
  }
}


// ---- GEGLU gate (MLP.forward with config.gate, trajectory_gpt2.py:273-278): h = gelu(c_fc x) * gated_layer(x) -------
// forward : h (= gelu(pre), written by the c_fc GEMM epilogue) *= gate, in place, bf16, 8 elements (16 B) per thread.
// backward: from dh = d(loss)/d(h) (dgrad through c_proj), pre and gate:
//           d_pre = dh * gate * gelu'(pre)   (feeds the c_fc dgrad / wgrad)
//           d_gate = dh * gelu(pre)          (feeds the gated_layer dgrad / wgrad)
// gelu and gelu' share one exponential (erf_exp_parts).  HBM-bound: 3 reads + 2 writes of [M, 4d] bf16.
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __uint_as_float(w[e] << 16);
    f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 o;
  o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
  o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
  return o;
}
__global__ __launch_bounds__(256) void geglu_fwd_kernel(bf16_t* __restrict__ h, const bf16_t* __restrict__ gate, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      float a[8], g[8];
      unpack8(*reinterpret_cast<const uint4*>(h + i), a);
      unpack8(*reinterpret_cast<const uint4*>(gate + i), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] *= g[e];
      *reinterpret_cast<uint4*>(h + i) = pack8(a);
    } else {
      for (long j = i; j < n; ++j) h[j] = f32_to_bf16(bf16_to_f32(h[j]) * bf16_to_f32(gate[j]));
    }
  }
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ pre,
                                                        const bf16_t* __restrict__ gate, bf16_t* __restrict__ d_pre,
                                                        bf16_t* __restrict__ d_gate, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      float g[8], x[8], gt[8], dp[8], dg[8];
      unpack8(*reinterpret_cast<const uint4*>(dh + i), g);
      unpack8(*reinterpret_cast<const uint4*>(pre + i), x);
      unpack8(*reinterpret_cast<const uint4*>(gate + i), gt);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float er, ex;
        erf_exp_parts(x[e], er, ex);
        const float cdf = 0.5f * (1.0f + er);
        dg[e] = g[e] * (x[e] * cdf);
        dp[e] = g[e] * gt[e] * fmaf(x[e] * 0.39894228040143267794f, ex, cdf);
      }
      *reinterpret_cast<uint4*>(d_pre + i) = pack8(dp);
      *reinterpret_cast<uint4*>(d_gate + i) = pack8(dg);
    } else {
      for (long j = i; j < n; ++j) {
        const float g = bf16_to_f32(dh[j]), x = bf16_to_f32(pre[j]);
        d_gate[j] = f32_to_bf16(g * gelu_f(x));
        d_pre[j] = f32_to_bf16(g * bf16_to_f32(gate[j]) * gelu_grad_f(x));
      }
    }
  }
}

inline int grid_for(long n, int per_thread) {
  long b = (n + 256L * per_thread - 1) / (256L * per_thread);
  if (b > 2048) b = 2048;   // grid-stride the rest (256 CUs x 8 blocks)
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

int neko_cast_f32_bf16_impl(const float* x, bf16_t* y, long n, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!x || !y) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, x, y, n);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_colsum_bf16_impl(const bf16_t* x, long ld, int M, int N, float* out, int accumulate, hipStream_t s) {
  if (M <= 0 || N <= 0) return NEKO_OK;
  if (!x || !out || (ld & 7) || (N & 7)) return NEKO_ERR_ARG;
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s) != hipSuccess) return NEKO_ERR_LAUNCH;
  }
  const int gx = (N + 255) / 256;
  int gy = (M + 255) / 256;           // >= 32 rows per thread at full size
  const int cap = 768 / gx > 0 ? 768 / gx : 1;   // ~3 blocks per CU in total
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, gy), dim3(256), 0, s, x, ld, M, N, out);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_mask_bias_impl(const float* mask, float* kbias, int* kstart, int B, int T, hipStream_t s) {
  if (B <= 0 || T <= 0) return NEKO_OK;
  if (!mask || !kbias) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(mask_bias_kernel, dim3(B), dim3(256), 0, s, mask, kbias, kstart, T);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_sqnorm_f32_impl(const float* g, long n, double* out_accum, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!g || !out_accum) return NEKO_ERR_ARG;
  // at most two blocks per CU: every block ends in ONE fp64 atomic on the same address, and ~1700 of them serialised in
  // the L2 were the whole run time of a 7 M-element range (25 us for 28 MB)
  int grid = grid_for(n, 16);
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(256), 0, s, g, n, out_accum);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_adamw_step_impl(float* p, const float* g, float* m, float* v, bf16_t* p16, long n, float lr, float beta1,
                         float beta2, float eps, float wd, const double* gnorm_sq, float max_norm,
                         const float* grad_scale, int* step, const int* active, const float* lr_dev, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!p || !g || !m || !v || !step) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, s, step, active);
  NEKO_CHECK_LAUNCH();
  AdamArgs a{p, g, m, v, p16, n, lr, beta1, beta2, eps, wd, max_norm, gnorm_sq, grad_scale, step, active, lr_dev};
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_geglu_fwd_impl(bf16_t* h, const bf16_t* gate, long n, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!h || !gate || (((uintptr_t)h | (uintptr_t)gate) & 15)) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, h, gate, n);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_geglu_bwd_impl(const bf16_t* dh, const bf16_t* pre, const bf16_t* gate, bf16_t* d_pre, bf16_t* d_gate, long n,
                        hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!dh || !pre || !gate || !d_pre || !d_gate) return NEKO_ERR_ARG;
  if (((uintptr_t)dh | (uintptr_t)pre | (uintptr_t)gate | (uintptr_t)d_pre | (uintptr_t)d_gate) & 15) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(n, 8)), dim3(256), 0, s, dh, pre, gate, d_pre, d_gate, n);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
