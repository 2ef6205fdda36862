// Store-pattern probe: how fast can ONE 8-wave block per CU (the GEMM's occupancy) write its 256 x 256 output tile,
// as a function of which lanes of a store instruction write which bytes?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probe/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
// Patterns (bf16 output, N = 3072 columns, tiles walked like the GEMM's raster; values come from registers):
//   0  GEMM epilogue today: a wave owns 128 rows x 64 cols; one dwordx2 store = 4 rows x 128 B
//   1  a wave owns 32 rows x 256 cols; one dwordx2 store = 1 row x 512 B
//   2  a wave owns 32 rows x 256 cols; one dwordx4 store = 2 rows x 512 B
//   3  a wave owns 128 rows x 64 cols; one dwordx4 store = 8 rows x 128 B
//   4  fp32 output (N x 4 B rows): a wave owns 128 rows x 64 cols; one dwordx4 store = 4 rows x 256 B (today's f32 path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* out, int M, int N, int tiles_n, int ntiles) {
  extern __shared__ char smem[];            // 128 KiB: one block per CU
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long esz = PAT == 4 ? 4 : 2;
  const long ld = (long)N * esz;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tm = t / tiles_n, tn = t % tiles_n;
    char* base = out + (long)tm * 256 * ld + (long)tn * 256 * esz;
    if (PAT == 0) {
      char* p = base + (long)(wave >> 2) * 128 * ld + (wave & 3) * 128 + (long)(lane >> 4) * ld + (lane & 15) * 8;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) *reinterpret_cast<uint2*>(p + (long)i * 4 * ld) = make_uint2(t, i);
    } else if (PAT == 1) {
      char* p = base + (long)wave * 32 * ld + lane * 8;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) *reinterpret_cast<uint2*>(p + (long)i * ld) = make_uint2(t, i);
    } else if (PAT == 2) {
      char* p = base + (long)wave * 32 * ld + (long)(lane >> 5) * ld + (lane & 31) * 16;
#pragma unroll 8
      for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4*>(p + (long)i * 2 * ld) = make_uint4(t, i, t, i);
    } else if (PAT == 3) {
      char* p = base + (long)(wave >> 2) * 128 * ld + (wave & 3) * 128 + (long)(lane >> 3) * ld + (lane & 7) * 16;
#pragma unroll 8
      for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4*>(p + (long)i * 8 * ld) = make_uint4(t, i, t, i);
    } else {
      char* p = base + (long)(wave >> 2) * 128 * ld + (wave & 3) * 256 + (long)(lane >> 4) * ld + (lane & 15) * 16;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) *reinterpret_cast<uint4*>(p + (long)i * 4 * ld) = make_uint4(t, i, t, i);
    }
  }
  if (threadIdx.x == 9999) smem[0] = 1;
}

template <int PAT>
void run(char* out, int M, int N, const char* what) {
  const int tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
  CK(hipFuncSetAttribute((const void*)store_kernel<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {1, 8, 64, 256, ntiles}) {
    if (grid > ntiles) continue;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(store_kernel<PAT>, dim3(grid), dim3(512), 131072, 0, out, M, N, tiles_n, ntiles);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(store_kernel<PAT>, dim3(grid), dim3(512), 131072, 0, out, M, N, tiles_n, ntiles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)M * N * (PAT == 4 ? 4 : 2);
    const int per_block = (ntiles + grid - 1) / grid;
    printf("pattern %d %-50s M %6d grid %5d: %9.1f us  %6.2f TB/s  %6.2f us per tile per block (%d tiles each)\n", PAT, what, M,
           grid, us, bytes / us * 1e-6, us / per_block, per_block);
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = 3072;
  char* out; CK(hipMalloc(&out, (size_t)M * N * 4));
  run<0>(out, M, N, "bf16 4 rows x 128 B / dwordx2 (epilogue today)");
  run<1>(out, M, N, "bf16 1 row x 512 B / dwordx2");
  run<2>(out, M, N, "bf16 2 rows x 512 B / dwordx4");
  run<3>(out, M, N, "bf16 8 rows x 128 B / dwordx4");
  run<4>(out, M, N, "f32 4 rows x 256 B / dwordx4 (f32 epilogue today)");
  return 0;
}
