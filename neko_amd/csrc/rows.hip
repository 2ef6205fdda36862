// Row gather / scatter for the loss positions and the deterministic split-K reduction.
//
// gather/scatter: the reference selects the loss positions with a boolean mask before the cross-entropy
// (gato/policy/gato_policy.py:183-185, `loss_logits.reshape(-1,V)[loss_masks > 0]`).  Here the selection happens
// one GEMM earlier: only the selected rows of ln_f's output go through the LM head (the logits of the other
// positions are discarded by Trainer.train_step, trainer.py:178), and the gradient rows are scattered back.
// splitk_reduce: C (+)= sum_s ws[s] -- the split-K slices of a wgrad GEMM are written to a workspace and summed
// in a fixed order (bit-reproducible gradients, no fp32 atomics).
// All HBM-bound: 16-B accesses, one wave per row / grid-stride float4.
#include "neko_kernels.h"

namespace {

// dst[r,:] = (r < n) ? src[idx[r],:] : 0      bf16 rows of d elements (d % 8 == 0), one wave per row
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx,
                                                               bf16_t* __restrict__ dst, int n, int npad, int d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= npad) return;
  uint4* o = reinterpret_cast<uint4*>(dst + (long)r * d);
  if (r < n) {
    const uint4* s = reinterpret_cast<const uint4*>(src + (long)idx[r] * d);
    for (int c = lane; c < (d >> 3); c += 64) o[c] = s[c];
  } else {
    for (int c = lane; c < (d >> 3); c += 64) o[c] = make_uint4(0, 0, 0, 0);
  }
}

// dst[idx[r],:] = src[r,:]   fp32 rows (dst pre-zeroed by the caller; idx unique)
__global__ __launch_bounds__(256) void scatter_rows_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                               float* __restrict__ dst, int n, int d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float4* s = reinterpret_cast<const float4*>(src + (long)r * d);
  float4* o = reinterpret_cast<float4*>(dst + (long)idx[r] * d);
  for (int c = lane; c < (d >> 2); c += 64) o[c] = s[c];
}

// C[m, n] (+)= sum_{s < S} ws[s][m][n]   (ws slices are dense [M, N]; N % 4 == 0)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, int M, int N,
                                                            float* __restrict__ C, long ldc, int accumulate) {
  const long total4 = (long)M * N / 4;
  const long slice = (long)M * N;
  const int n4 = N >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    // eight slices requested before the first is added (S is a run-time value: the plain loop was compiled as one
    // dependent round trip per slice); the additions keep the slice order, so the sum is the same bits as before
    float4 a = reinterpret_cast<const float4*>(ws)[i];
    int s = 1;
    for (; s + 8 <= S; s += 8) {
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = reinterpret_cast<const float4*>(ws + (s + u) * slice)[i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
    }
    if (s + 4 <= S) {
      float4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = reinterpret_cast<const float4*>(ws + (s + u) * slice)[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
      s += 4;
    }
    for (; s < S; ++s) {
      const float4 b = reinterpret_cast<const float4*>(ws + s * slice)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const long m = i / n4;
    const int c = (int)(i % n4) * 4;
    float4* dst = reinterpret_cast<float4*>(C + m * ldc + c);
    if (accumulate) {
      const float4 q = *dst;
      a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
    }
    *dst = a;
  }
}

}  // namespace

int neko_gather_rows_bf16_impl(const bf16_t* src, const int* idx, bf16_t* dst, int n, int npad, int d, hipStream_t s) {
  if (npad <= 0) return NEKO_OK;
  if (!src || !dst || (n > 0 && !idx) || (d & 7) || n > npad) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3((npad + 3) / 4), dim3(256), 0, s, src, idx, dst, n, npad, d);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_scatter_rows_f32_impl(const float* src, const int* idx, float* dst, int n, int d, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!src || !dst || !idx || (d & 3)) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3((n + 3) / 4), dim3(256), 0, s, src, idx, dst, n, d);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_splitk_reduce_impl(const float* ws, int S, int M, int N, float* C, long ldc, int accumulate, hipStream_t s) {
  if (S <= 0 || M <= 0 || N <= 0) return NEKO_OK;
  if (!ws || !C || (N & 3) || (ldc & 3)) return NEKO_ERR_ARG;
  long blocks = ((long)M * N / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ws, S, M, N, C, ldc, accumulate);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
