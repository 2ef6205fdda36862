// What does the MFMA skeleton of an attention-backward sub-tile cost by itself?  One workgroup of NW waves per CU (160 KiB of
// LDS requested), each wave loops over "sub-tiles": 4 MFMAs 32x32x16 bf16 into two fresh accumulators (S, dP), 16
// v_cvt_pk_bf16_f32 of their results, 4 MFMAs accumulating into two long-lived accumulators (dV, dK) with the converted
// registers as B operands -- no LDS, no memory, no other arithmetic.  Variants: 0 as described; 1 without the converts (the
// second group takes fixed operands: no MFMA -> VALU -> MFMA dependency); 2 two sub-tiles interleaved per iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_chain_probe.hip -o /tmp/mfma_chain_probe && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_v __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ bf16x8_v from_acc(const f32x16& a, int o) {
  const uint4 r = make_uint4(pk(a[o], a[o + 1]), pk(a[o + 2], a[o + 3]), pk(a[o + 4], a[o + 5]), pk(a[o + 6], a[o + 7]));
  return __builtin_bit_cast(bf16x8_v, r);
}

template <int VAR>
__global__ __launch_bounds__(1024) void chain_kernel(const uint4* in, float* out, int iters, unsigned long long* clk) {
  extern __shared__ char smem[];
  const uint4 u = in[threadIdx.x & 63];
  const bf16x8_v a0 = __builtin_bit_cast(bf16x8_v, u), a1 = __builtin_bit_cast(bf16x8_v, make_uint4(u.y, u.z, u.w, u.x));
  f32x16 dv, dk;
  for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (VAR == 2) {
      f32x16 s0, p0, s1, p1;
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; p0[r] = 0.f; s1[r] = 0.f; p1[r] = 0.f; }
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, a1, s0, 0, 0, 0);
      p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, a0, p0, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, a1, s0, 0, 0, 0);
      p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, a0, p0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, a1, s1, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, a0, p1, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, a1, s1, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, a0, p1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8_v b0 = from_acc(s0, 0), b1 = from_acc(s0, 8), c0 = from_acc(p0, 0), c1 = from_acc(p0, 8);
      const bf16x8_v e0 = from_acc(s1, 0), e1 = from_acc(s1, 8), f0 = from_acc(p1, 0), f1 = from_acc(p1, 8);
      __builtin_amdgcn_sched_barrier(0);
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, c0, dk, 0, 0, 0);
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, c1, dk, 0, 0, 0);
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, e0, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, f0, dk, 0, 0, 0);
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, e1, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, f1, dk, 0, 0, 0);
    } else {
      f32x16 s0, p0;
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; p0[r] = 0.f; }
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, a1, s0, 0, 0, 0);
      p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, a0, p0, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, a1, s0, 0, 0, 0);
      p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, a0, p0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8_v b0 = a0, b1 = a1, c0 = a1, c1 = a0;
      if (VAR == 0) { b0 = from_acc(s0, 0); b1 = from_acc(s0, 8); c0 = from_acc(p0, 0); c1 = from_acc(p0, 8); }
      else { asm volatile("" ::"v"(s0), "v"(p0)); }
      __builtin_amdgcn_sched_barrier(0);
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, c0, dk, 0, 0, 0);
      dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, c1, dk, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
  float acc = 0.f;
  for (int r = 0; r < 16; ++r) acc += dv[r] + dk[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 99999) smem[0] = 1;
}

template <int VAR>
void run(int nw, const uint4* in, float* out, unsigned long long* clk, const char* what) {
  const int iters = 4000, tiles = VAR == 2 ? 2 * iters : iters;
  CK(hipFuncSetAttribute((const void*)chain_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(chain_kernel<VAR>, dim3(256), dim3(64 * nw), 140 * 1024, 0, in, out, iters, clk);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(chain_kernel<VAR>, dim3(256), dim3(64 * nw), 140 * 1024, 0, in, out, iters, clk);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
  const double per_wave = (double)c / tiles, per_simd = per_wave / (nw / 4.0);
  printf("%-58s %2d waves/CU: %7.0f clk per sub-tile in the wave, %6.0f per SIMD (8 MFMAs = 256), clock %.2f GHz\n", what, nw,
         per_wave, per_simd, (double)c / (ms * 1e6));
}

int main() {
  uint4* in; float* out; unsigned long long* clk;
  CK(hipMalloc(&in, 64 * 16)); CK(hipMemset(in, 0x3c, 64 * 16));
  CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&clk, 8));
  for (int nw : {4, 8, 12, 16}) {
    run<0>(nw, in, out, clk, "4 MFMA -> 16 cvt_pk -> 4 MFMA");
    run<1>(nw, in, out, clk, "4 MFMA -> 4 MFMA on fixed operands (no converts)");
    run<2>(nw, in, out, clk, "two sub-tiles interleaved (8 MFMA -> 32 cvt_pk -> 8 MFMA)");
  }
  return 0;
}
