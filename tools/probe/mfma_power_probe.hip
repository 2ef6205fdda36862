// What does the matrix pipe sustain under the 1400 W board limit, by MFMA shape and by LDS operand traffic?
// 8 waves per CU (2 per SIMD), every wave loops over independent MFMAs on register operands; variants add ds_read_b128 per MFMA
// (the operand traffic of a GEMM main loop).  No global memory in the loop.  Prints TFLOP/s; run under tools/power_probe.sh for clocks.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_v __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// VAR 0: 32x32x16, 8 accumulators; 1: 16x16x32, 8 accumulators; 2/3/4: 32x32x16 with 1 / 3 / 6 ds_read_b128 per 4 MFMAs (0.25 / 0.75 / 1.5 per MFMA)
template <int VAR>
__global__ __launch_bounds__(512) void k(const uint4* in, float* out, int iters) {
  __shared__ uint4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = in[i & 63];
  __syncthreads();
  const uint4 u = in[(threadIdx.x * 7 + blockIdx.x) & 63];
  bf16x8_v a = __builtin_bit_cast(bf16x8_v, u), b = __builtin_bit_cast(bf16x8_v, make_uint4(u.y, u.z, u.w, u.x));
  f32x16 c[8];
  f32x4 d[8];
  for (int j = 0; j < 8; ++j) { for (int r = 0; r < 16; ++r) c[j][r] = 0.f; for (int r = 0; r < 4; ++r) d[j][r] = 0.f; }
  const int l = threadIdx.x & 511;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (VAR == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, d[j], 0, 0, 0);
    } else {
      constexpr int NR = VAR == 2 ? 1 : VAR == 3 ? 3 : VAR == 4 ? 6 : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint4 t[NR ? NR : 1];
#pragma unroll
        for (int q = 0; q < NR; ++q) t[q] = lds[(l + 512 * q + 64 * h + it) & 4095];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[4 * h + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[4 * h + j], 0, 0, 0);
        if (NR) {
          uint4 x = t[0];
#pragma unroll
          for (int q = 1; q < NR; ++q) { x.x ^= t[q].x; x.y ^= t[q].y; x.z ^= t[q].z; x.w ^= t[q].w; }
          a = __builtin_bit_cast(bf16x8_v, x);      // the reads feed the next MFMAs: they cannot be dropped
        }
      }
    }
  }
  float acc = 0.f;
  for (int j = 0; j < 8; ++j) { for (int r = 0; r < 16; ++r) acc += c[j][r]; for (int r = 0; r < 4; ++r) acc += d[j][r]; }
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int VAR>
void run(const char* name, const uint4* in, float* out, double flop_per_iter_wave, int iters, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(512), 0, 0, in, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(512), 0, 0, in, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = flop_per_iter_wave * iters * 8.0 * 256.0 * reps;
  printf("%-44s %8.1f ms  %7.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  uint4* in; float* out;
  CK(hipMalloc(&in, 64 * sizeof(uint4))); CK(hipMalloc(&out, 256 * 512 * 4));
  uint4 h[64];
  // random bf16 values of magnitude ~1 with random signs (power depends on operand toggling: constants understate it)
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; const unsigned lo = 0x3f00u | ((s >> 9) & 0x80ffu), hi = 0x3f00u | ((s >> 17) & 0x80ffu); return lo | (hi << 16); };
  for (int i = 0; i < 64; ++i) h[i] = make_uint4(rnd(), rnd(), rnd(), rnd());
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  const int iters = 20000;
  const double f32 = 8 * 2.0 * 32 * 32 * 16, f16 = 16 * 2.0 * 16 * 16 * 32;
  const char* only = argc > 2 ? argv[2] : "";
  if (!*only || only[0] == '0') run<0>("32x32x16 bf16, register operands", in, out, f32, iters, reps);
  if (!*only || only[0] == '1') run<1>("16x16x32 bf16, register operands", in, out, f16, iters, reps);
  if (!*only || only[0] == '2') run<2>("32x32x16 + 0.25 ds_read_b128 per MFMA", in, out, f32, iters, reps);
  if (!*only || only[0] == '3') run<3>("32x32x16 + 0.75 ds_read_b128 per MFMA", in, out, f32, iters, reps);
  if (!*only || only[0] == '4') run<4>("32x32x16 + 1.5 ds_read_b128 per MFMA", in, out, f32, iters, reps);
  return 0;
}
