// Two-workgroups-per-CU main loop of neko_gemm_bf16 (same contract and reference citations as gemm_bf16.hip; round 5): a 128 x 256 output
// tile per workgroup of FOUR waves, each owning 64 x 128 = 4 x 8 blocks of v_mfma_f32_16x16x32_bf16 with its 128 accumulators in
// AGPRs -- half of gemm_a16.hip's wave tile, so that a wave needs <= 256 registers and 80 KB of LDS carry a workgroup: TWO workgroups
// share a CU, and while one is in its output phase (epilogue arithmetic, stores, the next tile's prologue) the other's main loop has
// the matrix pipe to itself.  The K = 768 GEMMs of a layer (c_attn, c_fc + GELU, the GELU' dgrad, the projections' residual
// epilogues) spend 35-45 % of a 256 x 256 tile outside the main loop with one workgroup per CU (tools/gemm_trace.py: prologue 3.3,
// loop 19.0, epilogue 9.5 us for c_fc); the compiler-scheduled 256 x 128 configuration of gemm_glds.hip cannot cash that in because
// its loop reaches half the matrix rate with one wave per SIMD (profiles/r05_two_wg_trace.txt) -- this loop is written instruction by
// instruction like gemm_a16's (tools/gen_gemm_a16.py --geom b16 -> gemm_b16_loop.inc).
//
// LDS (80 KB): A k-contiguous in two 64-k slots of 128 rows (32 KB at 0), B in a 3-stage 32-k ring (48 KB at 32 KB), k-strided
// (forward: weights stored (in, out)) or k-contiguous (dgrad); the loop body covers 12 k-tiles (slot period 4 x ring period 3).
// Serves launches with A k-contiguous, M % 128 == 0, N % 256 == 0, a contraction that is a multiple of 384, no split-K, and one
// of the compiled epilogue feature sets; everything else stays with gemm_a16.hip / gemm_glds.hip.
#include <cstdlib>
#include "gemm_epi.h"
#ifndef NEKO_B16_LOOP_INC
#define NEKO_B16_LOOP_INC "gemm_b16_loop.inc"      // (tools/probe/r05/gemm_loop_ablation.sh builds timing-only variants of the stream)
#endif
#include NEKO_B16_LOOP_INC

namespace {

using CB16 = Cfg<2, 2, 2, 4, 3>;      // 2 x 2 waves, 64 x 128 per wave: sizes the epilogue slabs (the LDS block itself is B16_LDS)
constexpr int B16_LDS = 81920;        // 32 KB (A slots) + 48 KB (B ring); the epilogue's padded slabs (4 x 16.5 KB) reuse it
static_assert(4 * FastEpi<CB16>::SLAB_BYTES <= B16_LDS, "epilogue slabs must fit the ring");

// The loop hands its accumulators to the compiler as four 32-float AGPR tuples pinned to a[0:127]: block (ti, tj) of the wave =
// a[4 (8 ti + tj) .. +3] = acc[ti >> 1][16 (ti & 1) + 4 tj .. +3], row 16 ti + (l & 15), columns 16 tj + 4 (l >> 4) .. +3 -- pass I
// of the epilogue (32 rows) is tuples 2I and 2I + 1 (the same layout as gemm_a16.hip with half the rows).
typedef float f32x32 __attribute__((ext_vector_type(32)));

struct ParkAgprB16 {
  static constexpr int PREFETCH = 8;      // 8 of a pass's 16 steps of epilogue inputs in flight: the wave has 128 VGPRs next to its 128 AGPRs
  f32x32 (&acc)[4];
  template <int I>
  __device__ __forceinline__ void park(float* slab, int lane) const {
    constexpr int SWP = FastEpi<CB16>::SWP;
    float* wbase = slab + (lane & 15) * SWP + 4 * (lane >> 4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int tj = 0; tj < 8; ++tj) {
        // through VGPRs explicitly (see gemm_a16.hip: AGPR sub-registers fed straight into ds_write stop hipcc)
        // explicit reads of the AGPR sub-registers: left to itself hipcc copies the whole 32-register tuple into VGPRs -- with 128 VGPRs
        // it does that through scratch memory (192 spilled registers)
        float x0, x1, x2, x3;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                     : "a"(acc[2 * I + h][4 * tj]), "a"(acc[2 * I + h][4 * tj + 1]), "a"(acc[2 * I + h][4 * tj + 2]), "a"(acc[2 * I + h][4 * tj + 3]));
        *reinterpret_cast<float4*>(wbase + h * 16 * SWP + tj * 16) = make_float4(x0, x1, x2, x3);
      }
  }
};

__device__ __forceinline__ int kc_swz_b(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }      // g = [0, 2, 3, 1] (gen_gemm_a16.py)
__device__ __forceinline__ int ks_hh_b(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// One kernel per (B layout, epilogue feature set F): with the 15 compiled epilogues of try_epilogue_fast() inlined behind one switch
// (gemm_a16.hip's form, 256 VGPRs) the register allocator of a 128-VGPR kernel moves accumulator tuples through scratch memory.
template <bool B_KC, unsigned F>
__global__ __launch_bounds__(256, 2) void gemm_b16_kernel(GemmArgs p) {
  if (p.drop_thr) p.drop_key += neko_drop_salt();
  using C = CB16;
  __shared__ __attribute__((aligned(1024))) char smem[B16_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn, slice;
  tile_coords<128, 256>(p, tm, tn, slice);
  const int m0 = tm * 128, n0 = tn * 256;
  const unsigned nkt = (unsigned)p.K / 32u, ntrips = nkt / 12u;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const int c16 = lane & 15, g4 = lane >> 4;
  const int krd = 8 * g4 + (c16 >> 2);                                       // k-row a lane's transposing read starts at
  unsigned voa[4], vob[4], ra[2], rb, hb;
  // A: k-contiguous, 64-k slots of 128 rows (128-B rows); piece = 8 rows x 128 B, wave w requests pieces 4w .. 4w+3 of a slot;
  // 16-B chunk c of row r sits at c ^ ((r >> 1) & 7)
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    const int row = (wave * 4 + pc) * 8 + (lane >> 3);
    voa[pc] = (unsigned)((row * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) ra[h] = lds0 + (unsigned)((wm * 64 + c16) * 128 + (((g4 + 4 * h) ^ ((c16 >> 1) & 7)) << 4));
  const unsigned ldswa = lds0 + (unsigned)wave * 4096u, stepa = 128u;
  // B: 3-stage ring of 32-k stages at 32 KB
  const unsigned regionb = lds0 + 32768u;
  hb = (unsigned)ks_hh_b(krd);
  unsigned stepb;
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    if (B_KC) {                          // piece = 16 rows x 64 B
      const int row = (wave * 4 + pc) * 16 + (lane >> 2);
      vob[pc] = (unsigned)((row * p.ldb + (((lane & 3) ^ kc_swz_b(row)) << 3)) * 2);
    } else {                             // piece = 2 k-rows x 512 B
      const int kr = (wave * 4 + pc) * 2 + (lane >> 5);
      vob[pc] = (unsigned)((kr * p.ldb + (((lane & 31) ^ (ks_hh_b(kr) << 1)) << 3)) * 2);
    }
  }
  if (B_KC) {
    rb = regionb + (unsigned)((wn * 128 + c16) * 64 + ((g4 ^ kc_swz_b(c16)) << 4));
    stepb = 64u;
  } else {
    rb = regionb + (unsigned)(krd * 512 + (wn * 16 + ((c16 & 3) >> 1)) * 16 + (c16 & 1) * 8);
    stepb = (unsigned)(64 * p.ldb);
  }
  const unsigned ldswb = lds0 + (unsigned)wave * 4096u;
  const bf16_t* gA = p.A + (long)m0 * p.lda;
  const bf16_t* gB = B_KC ? p.B + (long)n0 * p.ldb : p.B + n0;
  const unsigned long long gAu = reinterpret_cast<unsigned long long>(gA), gBu = reinterpret_cast<unsigned long long>(gB);
  const unsigned galo = __builtin_amdgcn_readfirstlane((unsigned)gAu), gahi = __builtin_amdgcn_readfirstlane((unsigned)(gAu >> 32));
  const unsigned gblo = __builtin_amdgcn_readfirstlane((unsigned)gBu), gbhi = __builtin_amdgcn_readfirstlane((unsigned)(gBu >> 32));

  f32x32 acc[4];
#define NEKO_B16_OPERANDS                                                                                                   \
  "={a[0:31]}"(acc[0]), "={a[32:63]}"(acc[1]), "={a[64:95]}"(acc[2]), "={a[96:127]}"(acc[3])                                   \
  : [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [vob0] "v"(vob[0]), [vob1] "v"(vob[1]),   \
    [vob2] "v"(vob[2]), [vob3] "v"(vob[3]), [ra0] "v"(ra[0]), [ra1] "v"(ra[1]), [rb0] "v"(rb), [hb] "v"(hb),                  \
    [galo] "s"(galo), [gahi] "s"(gahi), [gblo] "s"(gblo), [gbhi] "s"(gbhi), [sa] "s"(stepa), [sb] "s"(stepb), [nkt] "s"(nkt),   \
    [ntrips] "s"(ntrips), [ldswa] "s"(ldswa), [ldswb] "s"(ldswb)                                                              \
  : NEKO_B16_CLOBBERS
  if constexpr (B_KC) asm volatile(NEKO_B16_LOOP_KC_KC : NEKO_B16_OPERANDS);
  else asm volatile(NEKO_B16_LOOP_KC_KS : NEKO_B16_OPERANDS);
#undef NEKO_B16_OPERANDS

  // the loop ends behind a block barrier with every DMA landed: the ring is free for the slabs
  (void)slice;
  // The epilogue's per-lane address arithmetic must not be hoisted above the loop: with 104 VGPRs pinned by the loop and 12 operands, what
  // hipcc computed early (row pointers, slab addresses) went through scratch memory (2-7 spilled registers in every instantiation,
  // VERDICT r05 weak 8).  A lane id that is DEFINED behind the loop keeps everything derived from it there.
  int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // (recomputed: not even `lane` stays live)
  asm volatile("" : "+v"(lane_e));
  // ... and the arguments are read again from the kernel-argument segment (GemmArgs is the kernel's only parameter, at offset 0) through a
  // pointer that is defined behind the loop: no argument register has to survive the loop for the epilogue's sake (the dropout variant
  // ran out of SGPRs and parked a pointer pair in scratch memory)
#if defined(__HIP_DEVICE_COMPILE__)
  const GemmArgs* pa = (const GemmArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
  const GemmArgs* pa = &p;      // (host pass of the single-source compile: never executed)
#endif
  asm volatile("" : "+s"(pa));
  GemmArgs q = *pa;
  if ((F & F_DROP) && q.drop_thr) q.drop_key = p.drop_key;      // (salted at entry)
  epilogue_fast<C, F>(q, ParkAgprB16{acc}, smem, m0, n0, wm, wn, wave, lane_e, q.Cf, q.ldcf);
}

// -1: per-shape choice (default), 0: never, 1: wherever it applies
int env_mode_b16() {
  static const int v = [] { const char* e = getenv("NEKO_GEMM_B16"); return e ? atoi(e) : -1; }();
  return v;
}

template <bool B_KC, unsigned F>
int launch_b16(const GemmArgs& a, hipStream_t s) {
  const int tiles = (a.M / 128) * (a.N / 256);
  hipLaunchKernelGGL((gemm_b16_kernel<B_KC, F>), dim3(tiles), dim3(256), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
// the feature sets this file instantiates, by operand layout: forward shapes (B = weights stored (in, out): k-strided) and dgrad /
// LM-head shapes (B k-contiguous).  1 = no kernel for this pair
int dispatch_b16(const GemmArgs& a, bool b_kc, unsigned f, hipStream_t s) {
#define NEKO_B16_CASE(KC, MASK) case (MASK): return launch_b16<KC, (MASK)>(a, s);
  if (!b_kc) {
    switch (f) {
      NEKO_B16_CASE(false, F_BIAS | F_CB)                                  // forward qkv
      NEKO_B16_CASE(false, F_BIAS | F_GELU | F_PRE | F_CB)                 // forward fc (act 1)
      NEKO_B16_CASE(false, F_BIAS | F_GELU | F_PRE | F_GP | F_CB)          // forward fc leaving gelu'(pre) (act 3)
      NEKO_B16_CASE(false, F_BIAS | F_RESID | F_CF)                        // forward projections
      NEKO_B16_CASE(false, F_BIAS | F_DROP | F_RESID | F_CF)               // ... with residual dropout
      NEKO_B16_CASE(false, F_CB)
      default: return 1;
    }
  }
  switch (f) {
    NEKO_B16_CASE(true, F_GELUBWD | F_CB)                                  // dgrad through the MLP projection (* GELU')
    NEKO_B16_CASE(true, F_GELUBWD | F_CB | F_COLSUM)
    NEKO_B16_CASE(true, F_GELUBWD | F_MULACT | F_CB)                       // ... * stored gelu' (act 4)
    NEKO_B16_CASE(true, F_GELUBWD | F_MULACT | F_CB | F_COLSUM)
    NEKO_B16_CASE(true, F_CB)                                              // dgrad attention out, LM-head logits
    NEKO_B16_CASE(true, F_CF)                                              // dgrad fc / qkv (fp32 out)
    default: return 1;
  }
#undef NEKO_B16_CASE
}

}  // namespace

// 1 = not applicable (the caller goes on to gemm_a16.hip / gemm_glds.hip), otherwise a status code.  *colsum_bands = number of
// 64-row bands of a.colsum_ws this launch filled (0: none / the column sums were not asked for)
int neko_gemm_b16_try(const GemmArgs& a_in, int a_kstrided, int b_kstrided, int mainloop_mode, int* colsum_bands, hipStream_t s) {
  *colsum_bands = 0;
  // neko_gemm_set_mainloop: 0 = no hand-placed loop, 1 = gemm_a16.hip's only, 2 = this one wherever it applies; -1 = NEKO_GEMM_B16 / per shape
  const int mode = (mainloop_mode == 0 || mainloop_mode == 1 || mainloop_mode == 3) ? 0 : (mainloop_mode == 2 ? 1 : env_mode_b16());
  if (mode == 0 || a_kstrided) return 1;
  GemmArgs a = a_in;
  if ((a.M & 127) || (a.N & 255) || a.K < 384 || (a.K % 384)) return 1;
  if (a.splitk > 1) return 1;
  // the per-lane DMA offsets are 32-bit byte offsets within the tile's operand panel
  if (128 * a.lda * 2 >= (1L << 31) || (b_kstrided ? 32 * a.ldb : 256 * a.ldb) * 2 >= (1L << 31)) return 1;
  if (((a.ldcf | a.ldr | a.ldcb | a.ldact | a.ldpre) & 3)) return 1;
  // column sums ride along only with the GELU' dgrad feature sets (every tile is interior here)
  const bool fold = a.colsum_ws && (a.act == 2 || a.act == 4) && a.Cb && !a.Cf && !a.bias && !a.resid && !a.drop_thr && a.alpha == 1.0f &&
                    !a.alpha_dev;
  if (!fold) a.colsum_ws = nullptr;
  const unsigned f = fast_epi_mask(a, true, false, a.Cf != nullptr);
  if (!fast_epi_supported(f)) return 1;
  if (mode < 0) {
    // Per-shape choice (profiles/r05_gemm_b16_ab.txt).  The k-loops of these GEMMs are bound by the CU's L2 -> LDS DMA rate (~28 B/clk:
    // a 256 x 256 k-tile is 32 pieces of 1 KB against 1024 clocks of MFMAs), and a 128 x 256 tile moves 1.5x the bytes per FLOP -- so
    // this kernel pays where the output phase it hides is long: the fp32 residual (+ dropout) epilogues of the two projections.
    // NEKO_GEMM_B16_RULE (bit mask) selects classes for A/B runs: 1 residual epilogues, 2 GELU forward, 4 GELU' dgrad, 8 plain
    // K <= 1536 (forward qkv, dgrad attention out), 16 LM-head logits (N > 16384)
    static const int rule = [] { const char* e = getenv("NEKO_GEMM_B16_RULE"); return e ? atoi(e) : 1; }();
    const long tiles = (long)(a.M / 128) * (a.N / 256);
    // less than one round of workgroups (README batch sizes: 7680 rows x 768 columns = 180 tiles): every class -- the launch is a single
    // partial round whatever the tile, and this loop runs it faster than gemm_glds.hip's 256 x 128 configuration (c2 5.92 -> 5.87 ms,
    // c3 5.82 -> 5.76 per step, c4 level: profiles/r05_gemm_b16_ab.txt)
    int cls;
    if (f & F_RESID) cls = 1;
    else if (f & F_GELU) cls = 2;
    else if (f & F_GELUBWD) cls = 4;
    else if (a.N > 16384) cls = 16;
    else cls = a.K <= 1536 ? 8 : 0;
    if (tiles >= 512 && !(rule & cls)) return 1;
  }
  a.epi_lock = 0;
  const int rc = dispatch_b16(a, !b_kstrided, f, s);
  if (rc == NEKO_OK) g_neko_last_mainloop = 2;
  if (rc == NEKO_OK && fold) *colsum_bands = a.M / 64;
  return rc;
}
