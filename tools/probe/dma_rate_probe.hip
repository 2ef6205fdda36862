// Round 6 probe: what rate does ONE CU's L2 -> LDS path deliver, by transport and by the number of waves that issue?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dma_rate_probe.hip -o tools/probe/dma_rate_probe.bin && tools/probe/dma_rate_probe.bin
// Every block (one per CU, 256 of them; block b runs on XCD b % 8) streams a region that its XCD's 32 blocks share (the GEMM situation: the
// tiles in flight on an XCD share operand panels), so after the first pass the lines come from the 4 MB L2.  Per wave-instruction 1 KiB:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), rows of 128 B (whole lines: 8 rows x 128 B per piece, row stride `ld` bytes)
//   mode 1  the same with rows of 64 B (half lines: 16 rows x 64 B per piece)
//   mode 2  global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging), rows of 128 B
//   mode 3  half of the pieces by LDS-DMA, half by register staging (do the two transports add up?)
//   mode 4  LDS-DMA, 2 x 512 B contiguous (k-strided operand: two k-rows of 256 columns)
// with NW = 4 / 8 / 16 waves per block and Q pieces in flight per wave.  Prints bytes per nanosecond per CU and, with the shader clock measured by
// s_memtime over the same interval, bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int Q>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q) : "memory"); }

__device__ __forceinline__ void dma_piece(const char* base_uniform, unsigned voff, unsigned lds_dst) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base_uniform), "{m0}"(lds_dst) : "memory");
}

// region: `rows` rows of `ld` bytes; a block walks it in steps of 64 k-bytes (128 for mode 0 / 2 / 3) like a GEMM's k-loop and wraps
template <int MODE, int NW, int Q>
__global__ __launch_bounds__(NW * 64) void probe(const char* __restrict__ src, long region_bytes, int ld, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const char* region = src + (long)(blockIdx.x % 8) * region_bytes;
  // per-lane source offset inside a piece
  unsigned voff;
  int rows_per_piece;
  if (MODE == 1) { voff = (unsigned)((lane >> 2) * ld + (lane & 3) * 16); rows_per_piece = 16; }
  else if (MODE == 4) { voff = (unsigned)((lane >> 5) * ld + (lane & 31) * 16); rows_per_piece = 2; }
  else { voff = (unsigned)((lane >> 3) * ld + (lane & 7) * 16); rows_per_piece = 8; }
  const int kbytes = (MODE == 1) ? 64 : (MODE == 4 ? 512 : 128);
  const int nrows = (int)(region_bytes / ld);
  // block-specific starting row panel (256 rows per "tile"), shared k walk
  int row = ((blockIdx.x / 8) * 256 + wave * rows_per_piece) % nrows;
  int kb = 0;
  const unsigned slot = lds0 + wave * (Q * 1024);       // Q KiB of LDS per wave, pieces round-robin
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  uint4 stage[(MODE == 2 || MODE == 3) ? Q : 1];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const char* p = region + (long)row * ld + kb;
      const bool use_reg = (MODE == 2) || (MODE == 3 && (q & 1));
      if (use_reg) {
        stage[q] = *reinterpret_cast<const uint4*>(p + voff);
      } else {
        const char* pu = reinterpret_cast<const char*>(__builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p))) |
                                                       ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(p) >> 32)) << 32));
        dma_piece(pu, voff, slot + q * 1024);
      }
      row += NW * rows_per_piece;
      if (row >= nrows) { row -= nrows; }
    }
    if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (MODE == 2 || (q & 1)) *reinterpret_cast<uint4*>(smem + wave * (Q * 1024) + q * 1024 + lane * 16) = stage[q];
    }
    // keep Q/2 pieces in flight across iterations for the DMA modes (the register modes wait through their data dependence)
    if (MODE != 2) wait_vm<Q / 2>();
    kb += kbytes;
    if (kb + kbytes > ld) kb = 0;
  }
  wait_vm<0>();
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = c1 - c0;
  }
  if (lane == 9999) out[0] = smem[lane];
}

template <int MODE, int NW, int Q>
int run(const char* d_src, long region_bytes, int ld, unsigned long long* d_out, const char* what) {
  const int iters = 2000 / Q * 4;
  const int lds = NW * Q * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, NW, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<MODE, NW, Q>), dim3(256), dim3(NW * 64), lds, 0, d_src, region_bytes, ld, iters, d_out);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(512);
  CK(hipMemcpy(h.data(), d_out, 512 * 8, hipMemcpyDeviceToHost));
  double ns = 0, clk = 0;
  for (int b = 0; b < 256; ++b) { ns += h[2 * b] * 10.0; clk += (double)h[2 * b + 1]; }
  ns /= 256; clk /= 256;
  const double bytes = (double)iters * Q * NW * 1024.0;
  printf("%-44s waves %2d  in flight/wave %2d  ld %5d : %7.1f B/ns/CU  %6.1f B/clk/CU  (%.2f GHz)  chip %6.2f TB/s\n", what, NW, Q, ld, bytes / ns,
         bytes / clk, clk / ns, bytes / ns * 256 / 1e3);
  return 0;
}

int main() {
  const long region = 2L << 20;          // 2 MB per XCD: L2-resident
  char* d_src;
  unsigned long long* d_out;
  CK(hipMalloc(&d_src, 8 * region + (1 << 20)));
  CK(hipMemset(d_src, 1, 8 * region + (1 << 20)));
  CK(hipMalloc(&d_out, 512 * 8));
  for (int ld : {1536, 6144}) {
    if (run<0, 4, 8>(d_src, region, ld, d_out, "LDS-DMA whole lines (8 x 128 B)")) return 1;
    if (run<0, 8, 4>(d_src, region, ld, d_out, "LDS-DMA whole lines (8 x 128 B)")) return 1;
    if (run<0, 8, 8>(d_src, region, ld, d_out, "LDS-DMA whole lines (8 x 128 B)")) return 1;
    if (run<0, 16, 4>(d_src, region, ld, d_out, "LDS-DMA whole lines (8 x 128 B)")) return 1;
    if (run<1, 4, 8>(d_src, region, ld, d_out, "LDS-DMA half lines (16 x 64 B)")) return 1;
    if (run<1, 8, 8>(d_src, region, ld, d_out, "LDS-DMA half lines (16 x 64 B)")) return 1;
    if (run<4, 4, 8>(d_src, region, ld, d_out, "LDS-DMA 2 x 512 B")) return 1;
    if (run<4, 8, 8>(d_src, region, ld, d_out, "LDS-DMA 2 x 512 B")) return 1;
    if (run<2, 4, 8>(d_src, region, ld, d_out, "global_load_dwordx4 + ds_write_b128")) return 1;
    if (run<2, 8, 8>(d_src, region, ld, d_out, "global_load_dwordx4 + ds_write_b128")) return 1;
    if (run<2, 16, 4>(d_src, region, ld, d_out, "global_load_dwordx4 + ds_write_b128")) return 1;
    if (run<3, 4, 8>(d_src, region, ld, d_out, "half LDS-DMA, half register staging")) return 1;
    if (run<3, 8, 8>(d_src, region, ld, d_out, "half LDS-DMA, half register staging")) return 1;
  }
  // a region that does NOT fit the L2 (64 MB per XCD slice of the buffer: the last-level cache / HBM serve it)
  char* d_big;
  const long big = 64L << 20;
  CK(hipMalloc(&d_big, 8 * big + (1 << 20)));
  CK(hipMemset(d_big, 1, 8 * big + (1 << 20)));
  if (run<0, 4, 8>(d_big, big, 1536, d_out, "LDS-DMA whole lines, 64 MB per XCD")) return 1;
  if (run<0, 8, 8>(d_big, big, 1536, d_out, "LDS-DMA whole lines, 64 MB per XCD")) return 1;
  if (run<2, 8, 8>(d_big, big, 1536, d_out, "register staging, 64 MB per XCD")) return 1;
  return 0;
}
