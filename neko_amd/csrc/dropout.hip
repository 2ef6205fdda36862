// Elementwise dropout on the fp32 residual stream: the embedding dropout of GPT2Model.forward
// (`hidden_states = self.drop(hidden_states)`, gato/transformers/trajectory_gpt2.py:541,707; p = embd_pdrop = 0.1,
// never overridden by the reference) and its backward (the same kernel on the gradient: the mask is regenerated
// from (index, key)).  The attention-probability and residual dropouts (:142,179 and :253-254,277-278) are fused
// into attention.hip, the GEMM epilogues and the LayerNorm backward; all use drop_keep() of neko_common.h.
// HBM-bound: 16-B accesses, grid-stride.
#include "neko_kernels.h"

namespace {

__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                          uint32_t thr, uint32_t key, float scale) {
  key += neko_drop_salt();
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 v = *reinterpret_cast<const float4*>(x + i);
      float e[4] = {v.x, v.y, v.z, v.w};
      drop4(e, (uint32_t)i, key, thr, scale);
      *reinterpret_cast<float4*>(y + i) = make_float4(e[0], e[1], e[2], e[3]);
    } else {
      for (long j = i; j < n; ++j) y[j] = drop_keep((uint32_t)j, key, thr) ? x[j] * scale : 0.f;
    }
  }
}

}  // namespace

int neko_dropout_f32_impl(const float* x, float* y, long n, int thr, unsigned key, float scale, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!x || !y || thr < 0 || thr > 255) return NEKO_ERR_ARG;
  long blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, n, (uint32_t)thr, key, scale);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

NEKO_DEFINE_SALT_SETTER(dropout)
