// Long-contraction main loop of neko_gemm_bf16 (same contract and reference citations as gemm_bf16.hip): a 256 x 256 output tile per
// workgroup of FOUR waves, one per SIMD, each owning 128 x 128 = 8 x 8 blocks of v_mfma_f32_16x16x32_bf16 with the 256 accumulator
// registers in AGPRs.  The instruction stream of the k-loop (which LDS read, which DMA piece and which scalar instruction sits in
// which gap between two MFMAs, every s_waitcnt count, the register numbers) is written by hand -- tools/gen_gemm_a16.py holds it and
// its design notes, gemm_a16_loop.inc is its output -- because the hipcc-scheduled form of this geometry (gemm_glds.hip, C256x256w4 /
// tools/probe/r04/gemm_mfma16_attempt.patch) spills or serialises (profiles/r03_mfma_shape_power.txt).  Around the loop everything is
// ordinary HIP: tile rasterisation, the per-lane DMA offsets and LDS read addresses, and the fast epilogue of gemm_epi.h, which
// takes the accumulators out of the AGPRs 32 rows at a time.
//
// Serves launches whose tiles are all interior (M, N multiples of 256), whose contraction range is a multiple of 128 (>= 4 k-tiles,
// 4 per loop trip) and whose epilogue is one of the compiled feature sets; everything else stays with gemm_glds.hip.
#include <cstdlib>
#include "gemm_epi.h"
#ifndef NEKO_A16_LOOP_INC
#define NEKO_A16_LOOP_INC "gemm_a16_loop.inc"      // (tools/probe/r05/gemm_loop_ablation.sh builds timing-only variants of the stream)
#endif
#include NEKO_A16_LOOP_INC

namespace {

using CA16 = Cfg<2, 2, 4, 4, 4>;      // 2 x 2 waves, 128 x 128 per wave, 4-stage ring: sizes the LDS block and the epilogue slabs

// The loop hands its accumulators to the compiler as eight 32-float AGPR tuples pinned to a[0:255] (output constraints of the asm
// statement): block (ti, tj) of the wave = a[4 (8 ti + tj) .. +3] = acc[ti >> 1][16 (ti & 1) + 4 tj .. +3], holding row 16 ti + (l & 15)
// and columns 16 tj + 4 (l >> 4) .. +3 -- so pass I of the epilogue (32 rows) is exactly tuples 2I and 2I + 1.
typedef float f32x32 __attribute__((ext_vector_type(32)));

struct ParkAgpr16 {
  static constexpr int PREFETCH = 0;
  f32x32 (&acc)[8];
  template <int I>
  __device__ __forceinline__ void park(float* slab, int lane) const {
    constexpr int SWP = FastEpi<CA16>::SWP;
    float* wbase = slab + (lane & 15) * SWP + 4 * (lane >> 4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int tj = 0; tj < 8; ++tj) {
        // each element goes through a VGPR explicitly: with the AGPR sub-registers fed straight into the ds_write, hipcc (ROCm 7.2)
        // stops with "Illegal instruction detected: Operand has incorrect register class"
        float x0 = acc[2 * I + h][4 * tj], x1 = acc[2 * I + h][4 * tj + 1], x2 = acc[2 * I + h][4 * tj + 2], x3 = acc[2 * I + h][4 * tj + 3];
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        *reinterpret_cast<float4*>(wbase + h * 16 * SWP + tj * 16) = make_float4(x0, x1, x2, x3);
      }
  }
};

__device__ __forceinline__ int kc_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }      // g = [0, 2, 3, 1] (gen_gemm_a16.py)
__device__ __forceinline__ int ks_hh(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 1) void gemm_a16_kernel(GemmArgs p) {
  if (p.drop_thr) p.drop_key += neko_drop_salt();
  using C = CA16;
  __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];     // 128 KB ring, then the epilogue slabs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn, slice;
  tile_coords<256, 256>(p, tm, tn, slice);
  const int m0 = tm * 256, n0 = tn * 256;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = slice * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  const unsigned nkt = (unsigned)(kend - kbeg) / 32u;

  // per-lane DMA source offsets (bytes from the operand's tile origin) and LDS read addresses; layouts: gen_gemm_a16.py
  constexpr bool KC64_OF[2] = {NEKO_A16_KC_MODE_A == 64, NEKO_A16_KC_MODE_B == 64};
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const int c16 = lane & 15, g4 = lane >> 4;
  const int krd = 8 * g4 + (c16 >> 2);                                       // k-row a lane's transposing read starts at
  unsigned vo[2][8], rd[2][2], hh[2], ldsw[2], step[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const bool kc = x ? B_KC : A_KC;
    const long ld = x ? p.ldb : p.lda;
    const int half = x ? wn : wm;                                            // which 128 rows / columns of the tile this wave reads
    const unsigned region = lds0 + (x ? 65536u : 0u);
    const bool KC64 = KC64_OF[x];
    hh[x] = (unsigned)ks_hh(krd);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) {
      if (kc && KC64) {                      // piece = 8 rows x 128 B, wave w requests pieces 8w .. 8w+7 of a slot
        const int row = (wave * 8 + pc) * 8 + (lane >> 3);
        vo[x][pc] = (unsigned)((row * ld + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
      } else if (kc) {                       // piece = 16 rows x 64 B, pieces 4w .. 4w+3 of a stage
        const int row = (wave * 4 + (pc & 3)) * 16 + (lane >> 2);
        vo[x][pc] = (unsigned)((row * ld + (((lane & 3) ^ kc_swz(row)) << 3)) * 2);
      } else {                               // piece = 2 k-rows x 512 B
        const int kr = (wave * 4 + (pc & 3)) * 2 + (lane >> 5);
        vo[x][pc] = (unsigned)((kr * ld + (((lane & 31) ^ (ks_hh(kr) << 1)) << 3)) * 2);
      }
    }
    if (kc && KC64) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        rd[x][h] = region + (unsigned)((half * 128 + c16) * 128 + (((g4 + 4 * h) ^ ((c16 >> 1) & 7)) << 4));
      ldsw[x] = lds0 + (unsigned)wave * 8192u;
      step[x] = 128u;
    } else if (kc) {
      rd[x][0] = rd[x][1] = region + (unsigned)((half * 128 + c16) * 64 + ((g4 ^ kc_swz(c16)) << 4));
      ldsw[x] = lds0 + (unsigned)wave * 4096u;
      step[x] = 64u;
    } else {
      rd[x][0] = rd[x][1] = region + (unsigned)(krd * 512 + (half * 16 + ((c16 & 3) >> 1)) * 16 + (c16 & 1) * 8);
      ldsw[x] = lds0 + (unsigned)wave * 4096u;
      step[x] = (unsigned)(64 * ld);
    }
  }
  const bf16_t* gA = A_KC ? p.A + (long)m0 * p.lda + kbeg : p.A + (long)kbeg * p.lda + m0;
  const bf16_t* gB = B_KC ? p.B + (long)n0 * p.ldb + kbeg : p.B + (long)kbeg * p.ldb + n0;
  const unsigned long long gAu = reinterpret_cast<unsigned long long>(gA), gBu = reinterpret_cast<unsigned long long>(gB);
  const unsigned galo = __builtin_amdgcn_readfirstlane((unsigned)gAu), gahi = __builtin_amdgcn_readfirstlane((unsigned)(gAu >> 32));
  const unsigned gblo = __builtin_amdgcn_readfirstlane((unsigned)gBu), gbhi = __builtin_amdgcn_readfirstlane((unsigned)(gBu >> 32));

  f32x32 acc[8];
#define NEKO_A16_OPERANDS                                                                                                  \
  "={a[0:31]}"(acc[0]), "={a[32:63]}"(acc[1]), "={a[64:95]}"(acc[2]), "={a[96:127]}"(acc[3]), "={a[128:159]}"(acc[4]),        \
      "={a[160:191]}"(acc[5]), "={a[192:223]}"(acc[6]), "={a[224:255]}"(acc[7])                                               \
  : [voa0] "v"(vo[0][0]), [voa1] "v"(vo[0][1]), [voa2] "v"(vo[0][2]), [voa3] "v"(vo[0][3]), [voa4] "v"(vo[0][4]),              \
    [voa5] "v"(vo[0][5]), [voa6] "v"(vo[0][6]), [voa7] "v"(vo[0][7]), [vob0] "v"(vo[1][0]), [vob1] "v"(vo[1][1]),              \
    [vob2] "v"(vo[1][2]), [vob3] "v"(vo[1][3]), [vob4] "v"(vo[1][4]), [vob5] "v"(vo[1][5]), [vob6] "v"(vo[1][6]),              \
    [vob7] "v"(vo[1][7]), [ra0] "v"(rd[0][0]), [ra1] "v"(rd[0][1]), [rb0] "v"(rd[1][0]), [rb1] "v"(rd[1][1]), [ha] "v"(hh[0]), \
    [hb] "v"(hh[1]), [galo] "s"(galo), [gahi] "s"(gahi), [gblo] "s"(gblo), [gbhi] "s"(gbhi), [sa] "s"(step[0]),                \
    [sb] "s"(step[1]), [nkt] "s"(nkt), [ldswa] "s"(ldsw[0]), [ldswb] "s"(ldsw[1])                                              \
  : NEKO_A16_CLOBBERS
  if constexpr (A_KC && B_KC) asm volatile(NEKO_A16_LOOP_KC_KC : NEKO_A16_OPERANDS);
  else if constexpr (A_KC && !B_KC) asm volatile(NEKO_A16_LOOP_KC_KS : NEKO_A16_OPERANDS);
  else if constexpr (!A_KC && B_KC) asm volatile(NEKO_A16_LOOP_KS_KC : NEKO_A16_OPERANDS);
  else asm volatile(NEKO_A16_LOOP_KS_KS : NEKO_A16_OPERANDS);
#undef NEKO_A16_OPERANDS

  // the loop ends behind a block barrier with every DMA landed: the ring is free for the slabs
  try_epilogue_fast<C>(p, ParkAgpr16{acc}, smem, m0, n0, wm, wn, wave, lane, slice);
}

// -1: per-shape choice (default), 0: never, 1: wherever it applies
int g_mainloop_mode = -1;
int env_mode() {
  static const int v = [] { const char* e = getenv("NEKO_GEMM_A16"); return e ? atoi(e) : -1; }();
  return v;
}

template <bool A_KC, bool B_KC>
int launch_a16(const GemmArgs& a, hipStream_t s) {
  const int tiles = (a.M / 256) * (a.N / 256) * (a.splitk > 1 ? a.splitk : 1);
  hipLaunchKernelGGL((gemm_a16_kernel<A_KC, B_KC>), dim3(tiles), dim3(256), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

}  // namespace

// -1: per-shape choice, 0: no hand-placed loop, 1: gemm_a16.hip wherever it applies (never gemm_b16.hip / gemm_p16.hip), 2: gemm_b16.hip
// wherever it applies (then the per-shape choice), 3: gemm_p16.hip wherever it applies (then the per-shape choice)
int neko_gemm_set_mainloop_impl(int mode) {
  const int prev = g_mainloop_mode;
  g_mainloop_mode = mode < 0 ? -1 : (mode > 3 ? 1 : mode);
  return prev;
}
int neko_gemm_mainloop_mode() { return g_mainloop_mode; }

// 1 = not applicable (the caller runs gemm_glds.hip's loop), otherwise a status code
int neko_gemm_a16_try(const GemmArgs& a, int a_kstrided, int b_kstrided, hipStream_t s) {
  const int mode = g_mainloop_mode >= 2 ? env_mode() : (g_mainloop_mode >= 0 ? g_mainloop_mode : env_mode());
  if (mode == 0) return 1;
  if ((a.M & 255) || (a.N & 255)) return 1;
  const int klen = a.splitk > 1 ? a.k_per_split : a.K;
  if ((klen & 127) || klen < 128) return 1;
  if (a.splitk > 1) {                                                         // every slice a whole number of loop trips
    const long last = (long)a.K - (long)(a.splitk - 1) * a.k_per_split;
    if (last < 128 || (last & 127)) return 1;
  }
  // the per-lane DMA offsets are 32-bit byte offsets within the tile's operand panel
  if ((a_kstrided ? 32 * a.lda : 256 * a.lda) * 2 >= (1L << 31) || (b_kstrided ? 32 * a.ldb : 256 * a.ldb) * 2 >= (1L << 31)) return 1;
  const bool to_ws = a.splitk > 1 && a.splitk_ws;
  if (a.splitk > 1 && !to_ws) return 1;
  if (to_ws && (a.bias || a.resid || a.act || a.Cb || a.drop_thr)) return 1;
  const long ldcf_out = to_ws ? a.N : a.ldcf;
  if (((ldcf_out | a.ldr | a.ldcb | a.ldact | a.ldpre) & 3)) return 1;
  if (a.colsum_ws) return 1;                                                  // band sums are laid out for gemm_glds.hip's waves
  // every slice must find a compiled epilogue (slice 0 carries bias / residual, the others do not)
  if (!fast_epi_supported(fast_epi_mask(a, true, to_ws, to_ws || a.Cf != nullptr))) return 1;
  if (to_ws && !fast_epi_supported(fast_epi_mask(a, false, to_ws, true))) return 1;
  if (mode < 0) {
    // per-shape choice (tools/gemm_bench.py at 32768 / 65536 rows, profiles/r04_gemm_a16_ab.txt): every long contraction; short ones
    // (K = 768) only with a plain bf16 store behind them and at least two rounds of tiles (forward qkv, dgrad attention out, LM-head
    // logits: +3..10 %) -- the GELU / residual epilogues are VALU- and store-bound and prefer the 8-wave kernel's two waves per SIMD
    const long tiles = (long)(a.M / 256) * (a.N / 256) * (a.splitk > 1 ? a.splitk : 1);
    if (klen < 1536) {
      const unsigned f = fast_epi_mask(a, true, to_ws, to_ws || a.Cf != nullptr);
      // NEKO_GEMM_A16_RULE bit 0 (A/B runs): also the GELU forward epilogues (c_fc)
      static const int rule = [] { const char* e = getenv("NEKO_GEMM_A16_RULE"); return e ? atoi(e) : 0; }();
      const bool plain = f == F_CB || f == (F_BIAS | F_CB);
      const bool gelu_fwd = (f & F_GELU) != 0 && (rule & 1);
      if (!(plain || gelu_fwd) || tiles < 512) return 1;
    }
    // fewer 256 x 256 tiles than ~3/4 of the CUs (README batch sizes: 7680 rows x 768 columns = 90 tiles): gemm_glds.hip's smaller
    // tiles fill the chip better than this loop's faster k-tiles pay back (c2: 6.03 -> 6.16 ms per step with 90-tile launches)
    if (tiles < 192) return 1;
  }
  g_neko_last_mainloop = 1;
  if (a_kstrided && b_kstrided) return launch_a16<false, false>(a, s);
  if (a_kstrided) return launch_a16<false, true>(a, s);
  if (b_kstrided) return launch_a16<true, false>(a, s);
  return launch_a16<true, true>(a, s);
}
