// Issue-rate microbenchmark for the VALU instructions the attention / epilogue bodies are made of (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Every kernel runs NW waves per SIMD on every CU, each wave issuing REP x 64 independent instructions of one kind;
// cycles per instruction per SIMD = s_memtime ticks of a wave / (REP * 64) / waves-per-SIMD ... reported both as the
// per-wave figure (latency-bound when 1 wave) and the per-SIMD throughput figure (4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256

#define BODY16(INS) INS INS INS INS INS INS INS INS INS INS INS INS INS INS INS INS
#define BODY64(INS) BODY16(INS) BODY16(INS) BODY16(INS) BODY16(INS)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
  unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x1234567u;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
    if (KIND == 0) {   // v_fma_f32, 8 independent chains
      asm volatile(BODY16("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if (KIND == 1) {   // v_pk_fma_f32
      asm volatile(BODY16("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
    } else if (KIND == 2) {   // v_exp_f32
      asm volatile(BODY16("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (KIND == 3) {   // v_cvt_pk_bf16_f32
      asm volatile(BODY16("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    } else if (KIND == 4) {   // v_max3_f32
      asm volatile(BODY16("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %5, %6\n v_max3_f32 %2, %2, %6, %7\n v_max3_f32 %3, %3, %7, %4\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    } else if (KIND == 5) {   // v_mul_u32_u24
      asm volatile(BODY16("v_mul_u32_u24 %0, %0, %2\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %0, %0, %3\n v_mul_u32_u24 %1, %1, %3\n")
                   : "+v"(u0), "+v"(u1) : "v"(a4), "v"(a5));
    } else if (KIND == 6) {   // v_cndmask_b32 (vcc)
      asm volatile(BODY16("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
    } else if (KIND == 7) {   // v_pk_mul_f32
      asm volatile(BODY16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
    } else if (KIND == 8) {   // v_add_f32
      asm volatile(BODY16("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
    } else if (KIND == 9) {   // v_bfe_u32 + v_cmp (dropout byte test) pair
      asm volatile(BODY16("v_bfe_u32 %0, %2, 8, 8\n v_cmp_ge_u32 vcc, %0, %3\n v_bfe_u32 %1, %2, 16, 8\n v_cmp_ge_u32 vcc, %1, %3\n")
                   : "+v"(u0), "+v"(u1) : "v"(a4), "v"(a5) : "vcc");
    } else if (KIND == 10) {   // v_rcp_f32
      asm volatile(BODY16("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else if (KIND == 12) {   // v_cmp (writes vcc) + v_cndmask (reads vcc) pairs, as a select is really issued
      asm volatile(BODY16("v_cmp_ge_u32 vcc, %4, %5\n v_cndmask_b32 %0, 0, %6, vcc\n v_cmp_ge_u32 vcc, %5, %4\n v_cndmask_b32 %1, 0, %7, vcc\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
    } else if (KIND == 13) {   // v_cndmask_b32_e64 with a constant SGPR-pair mask
      asm volatile("s_mov_b64 s[20:21], 0x55555555\n" BODY16("v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %5, %6, s[20:21]\n v_cndmask_b32_e64 %2, %6, %7, s[20:21]\n v_cndmask_b32_e64 %3, %7, %4, s[20:21]\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21");
    } else if (KIND == 14) {   // v_and_b32
      asm volatile(BODY16("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    } else if (KIND == 15) {   // byte-select compare (SDWA) writing vcc
      asm volatile(BODY16("v_cmp_ge_u32_sdwa vcc, %0, %2 src0_sel:BYTE_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %0, %3 src0_sel:BYTE_3 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %1, %3 src0_sel:BYTE_0 src1_sel:DWORD\n")
                   : "+v"(u0), "+v"(u1) : "v"(a4), "v"(a5) : "vcc");
    } else if (KIND == 16) {   // v_cmp writing an SGPR pair + v_cndmask_e64 reading it
      asm volatile(BODY16("v_cmp_ge_u32_e64 s[20:21], %4, %5\n v_cndmask_b32_e64 %0, 0, %6, s[20:21]\n v_cmp_ge_u32_e64 s[22:23], %5, %4\n v_cndmask_b32_e64 %1, 0, %7, s[22:23]\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23");
    } else if (KIND == 17) {   // v_mul_f32 (mask as 0.0 / 1.0 factor)
      asm volatile(BODY16("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %6\n v_mul_f32 %3, %3, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    } else if (KIND == 18) {   // v_mfma_f32_32x32x16_bf16 calibration: 32 cycles per SIMD each
      typedef float f16v __attribute__((ext_vector_type(16)));
      typedef float f4v __attribute__((ext_vector_type(4)));
      static_assert(sizeof(f4v) == 16, "");
      f16v acc0 = {}, acc1 = {};
      f4v A = {a0, a1, a2, a3}, Bv = {a4, a5, a6, a7};
      asm volatile(BODY16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n v_mfma_f32_32x32x16_bf16 %0, %3, %2, %0\n v_mfma_f32_32x32x16_bf16 %1, %3, %2, %1\n")
                   : "+v"(acc0), "+v"(acc1) : "v"(A), "v"(Bv));
      a0 += acc0[0] + acc1[5];
    } else if (KIND == 11) {   // v_mul_lo_u32
      asm volatile(BODY16("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %0, %0, %3\n v_mul_lo_u32 %1, %1, %3\n")
                   : "+v"(u0), "+v"(u1) : "v"(a4), "v"(a5));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + p0[0] + p1[1] + p2[0] + p3[1] + (float)(u0 + u1);
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  const int nblk = 256 * waves_per_simd, nthr = 256;      // 4 waves per block = one per SIMD; waves_per_simd blocks per CU
  float* out;
  long long* cyc;
  hipMalloc(&out, (size_t)nblk * nthr * 4);
  hipMalloc(&cyc, (size_t)nblk * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(nthr), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(nthr), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nblk * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += v;
  avg /= h.size();
  const double n = (double)REP * 64;
  printf("%-22s waves/SIMD %d: %7.2f memtime ticks per instr per wave -> %6.2f per instr per SIMD; wall %.1f us\n", name,
         waves_per_simd, avg / n, avg / n / waves_per_simd, ms * 1e3);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w : {1, 4}) {
    run<0>("v_fma_f32", w);
    run<1>("v_pk_fma_f32", w);
    run<7>("v_pk_mul_f32", w);
    run<8>("v_add_f32", w);
    run<2>("v_exp_f32", w);
    run<10>("v_rcp_f32", w);
    run<3>("v_cvt_pk_bf16_f32", w);
    run<4>("v_max3_f32", w);
    run<5>("v_mul_u32_u24", w);
    run<11>("v_mul_lo_u32", w);
    run<6>("v_cndmask_b32", w);
    run<9>("v_bfe_u32+v_cmp", w);
    run<12>("v_cmp+v_cndmask vcc", w);
    run<16>("v_cmp+v_cndmask sgpr", w);
    run<13>("v_cndmask const sgpr", w);
    run<14>("v_and_b32", w);
    run<15>("v_cmp_sdwa byte", w);
    run<17>("v_mul_f32", w);
    run<18>("mfma 32x32x16 bf16", w);
  }
  return 0;
}
