// Two-waves-per-SIMD main loop of neko_gemm_bf16 (same contract and reference citations as gemm_bf16.hip; round 6): a 256 x 256 output
// tile per workgroup of EIGHT waves, each owning 128 x 64 = 8 x 4 blocks of v_mfma_f32_16x16x32_bf16 with its 128 accumulators in AGPRs and
// ONE fragment set of 48 VGPRs.  The two waves of a SIMD (w and w + 4) run two different hand-placed instruction streams and alternate
// roles per half k-tile: while one issues its 32 MFMAs back to back, the other issues its fragment reads and the workgroup's L2 -> LDS
// requests; block barriers separate the segments, the second-dispatched half runs at s_setprio 1 for the whole loop
// (MI355X_MICROARCH.md "Two waves per SIMD"; tools/gen_gemm_a16.py --geom p16 holds the stream and its design notes, gemm_p16_loop.inc is
// its output).  The one-wave-per-SIMD loops (gemm_a16.hip, gemm_b16.hip) spend 1385-1550 clocks on a k-tile of 1024 matrix-pipe clocks
// because one in-order wave issues the DMA pieces, the LDS reads and the MFMAs (profiles/r05_gemm_loop_ablation.txt).
//
// LDS (160 KB, one workgroup per CU): A k-contiguous in THREE 64-k slots of whole 128-B lines (96 KB at 0) or k-strided in a 4-stage
// 32-k ring (64 KB at 0); B in a 4-stage 32-k ring at 96 KB, k-strided or k-contiguous.  Serves launches of interior 256 x 256 tiles whose
// contraction range (per split-K slice) is a multiple of 128 and at least one loop trip (384 with a k-contiguous A operand, else 128), with
// one of the compiled epilogue feature sets; everything else stays with gemm_a16.hip / gemm_glds.hip.
#include <cstdlib>
#include "gemm_epi.h"
#ifndef NEKO_P16_LOOP_INC
#define NEKO_P16_LOOP_INC "gemm_p16_loop.inc"      // (tools/probe/p16_ablation.sh builds timing-only variants of the stream)
#endif
#include NEKO_P16_LOOP_INC
#ifndef NEKO_P16_TRACE
#define NEKO_P16_TRACE 0     // 1 (diagnostic builds): s_memtime around the asm loop (prologue requests included) of wave 0, behind the bands of colsum_ws
                             // (tools/gemm_bench.py --p16-trace); 2: per block 8 x u64 {s_memrealtime at entry / loop start / loop end / stores
                             // issued / stores drained, HW_ID | XCC_ID << 32, s_memtime at loop start / loop end} into the buffer set with
                             // neko_gemm_p16_trace() (tools/probe/p16_phase_trace.py)
#endif
#if NEKO_P16_TRACE == 2
__device__ unsigned long long* g_neko_p16_trace = nullptr;
extern "C" int neko_gemm_p16_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_neko_p16_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define P16_STAMP(slot, val)                                                              \
  do {                                                                                    \
    if (g_neko_p16_trace && threadIdx.x == 0) g_neko_p16_trace[(long)blockIdx.x * 8 + (slot)] = (val); \
  } while (0)
#else
#define P16_STAMP(slot, val) do { } while (0)
#endif

namespace {

using CP16 = Cfg<2, 4, 4, 2, 4>;       // 2 x 4 waves, 128 x 64 per wave: sizes the epilogue slabs (the LDS block itself is P16_LDS)
constexpr int P16_LDS = 163840;        // 96 KB (A slots / ring) + 64 KB (B ring); the epilogue's padded slabs (8 x 8.5 KB) reuse it
constexpr unsigned P16_REGION_B = 98304u;
static_assert(8 * FastEpi<CP16>::SLAB_BYTES <= P16_LDS, "epilogue slabs must fit the ring");

// The loop hands its accumulators to the compiler as four 32-float AGPR tuples pinned to a[0:127]: block (ti, tj) of the wave =
// a[4 (4 ti + tj) .. +3] = acc[ti >> 1][16 (ti & 1) + 4 tj .. +3], row 16 ti + (l & 15), columns 16 tj + 4 (l >> 4) .. +3 -- pass I of the
// epilogue (32 rows) is tuple I.
typedef float f32x32 __attribute__((ext_vector_type(32)));

struct ParkAgprP16 {
  static constexpr int PREFETCH = 8;      // 8 of a pass's 16 steps of epilogue inputs in flight: the wave has 128 VGPRs next to its 128 AGPRs
  f32x32 (&acc)[4];
  template <int I>
  __device__ __forceinline__ void park(float* slab, int lane) const {
    constexpr int SWP = FastEpi<CP16>::SWP;
    float* wbase = slab + (lane & 15) * SWP + 4 * (lane >> 4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        // explicit reads of the AGPR sub-registers (gemm_b16.hip: left to itself hipcc copies whole tuples into VGPRs, through scratch memory)
        float x0, x1, x2, x3;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                     : "a"(acc[I][16 * h + 4 * tj]), "a"(acc[I][16 * h + 4 * tj + 1]), "a"(acc[I][16 * h + 4 * tj + 2]), "a"(acc[I][16 * h + 4 * tj + 3]));
        *reinterpret_cast<float4*>(wbase + h * 16 * SWP + tj * 16) = make_float4(x0, x1, x2, x3);
      }
  }
};

__device__ __forceinline__ int kc_swz_p(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }      // g = [0, 2, 3, 1] (gen_gemm_a16.py)
__device__ __forceinline__ int ks_hh_p(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// One kernel per (operand layouts, epilogue feature set F): see gemm_b16.hip (a 128-VGPR kernel with every epilogue behind one switch
// moves accumulator tuples through scratch memory).
template <bool A_KC, bool B_KC, unsigned F>
__global__ __launch_bounds__(512, 2) void gemm_p16_kernel(GemmArgs p) {
  P16_STAMP(0, __builtin_amdgcn_s_memrealtime());
  P16_STAMP(5, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32));
#ifdef NEKO_P16_STAGGER_10NS
  // experiment (tools/probe/p16_variants.sh stagger*): every other CU slot of the first round starts late, so that the CUs of an XCD are not
  // all in their epilogue (HBM burst) at the same time
  if (blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(NEKO_P16_STAGGER_10NS)) __builtin_amdgcn_s_sleep(32);
  }
#endif
  if ((F & F_DROP) && p.drop_thr) p.drop_key += neko_drop_salt();
  using C = CP16;
  __shared__ __attribute__((aligned(1024))) char smem[P16_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn, slice;
  tile_coords<256, 256>(p, tm, tn, slice);
  const int m0 = tm * 256, n0 = tn * 256;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = slice * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  // k-tiles = whole loop trips (12 with a k-contiguous A operand, else 4) + a tail of 0 / 4 / 8 k-tiles (the first groups of the same body)
  const unsigned nkt = (unsigned)(kend - kbeg) / 32u, trip = A_KC ? (unsigned)NEKO_P16_TRIP_KC : 4u;
  const unsigned ntrips = nkt / trip, tail = (nkt % trip) / 4u;

  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const int c16 = lane & 15, g4 = lane >> 4;
  const int krd = 8 * g4 + (c16 >> 2);                                       // k-row a lane's transposing read starts at
  unsigned voa[4], vob[4], ra[4], rb[2], ha, hb, ldswa, ldswb, stepa, stepb;
  constexpr bool B_KC64 = B_KC && NEKO_P16_KC_MODE_B == 64;
  // ---- A ----
  if (A_KC) {                            // 64-k slots of 256 rows (128-B rows); piece = 8 rows x 128 B, wave w requests pieces 4w .. 4w+3 of a slot;
                                         // 16-B chunk c of row r sits at c ^ ((r >> 1) & 7)
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const int row = (wave * 4 + pc) * 8 + (lane >> 3);
      voa[pc] = (unsigned)((row * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ra[h] = lds0 + (unsigned)((wm * 128 + c16) * 128 + (((g4 + 4 * h) ^ ((c16 >> 1) & 7)) << 4));
      ra[2 + h] = ra[h] + 65536u;        // the third slot (the DS offset field has 16 bits)
    }
    ha = 0;
    ldswa = lds0 + (unsigned)wave * 4096u;
    stepa = 128u;
  } else {                               // 4-stage ring of [32 k][256 rows]; piece = 2 k-rows x 512 B, wave w requests pieces 2w, 2w+1 of a stage
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int kr = (wave * 2 + pc) * 2 + (lane >> 5);
      voa[pc] = (unsigned)((kr * p.lda + (((lane & 31) ^ (ks_hh_p(kr) << 1)) << 3)) * 2);
    }
    voa[2] = voa[3] = 0;
    ra[0] = ra[1] = ra[2] = ra[3] = lds0 + (unsigned)(krd * 512 + ((c16 & 3) >> 1) * 16 + (c16 & 1) * 8);
    ha = (unsigned)(ks_hh_p(krd) ^ (wm * 8));       // block t of the wave = 32-B unit (8 wm + t) ^ hh of the k-row
    ldswa = lds0 + (unsigned)wave * 2048u;
    stepa = (unsigned)(64 * p.lda);
  }
  // ---- B at 96 KB: two 64-k slots of whole lines (k-contiguous, the default), or a 4-stage ring of 32-k stages ----
  const unsigned regionb = lds0 + P16_REGION_B;
  if (B_KC64) {                          // as A's slots: piece = 8 rows x 128 B, wave w requests pieces 4w .. 4w+3
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const int row = (wave * 4 + pc) * 8 + (lane >> 3);
      vob[pc] = (unsigned)((row * p.ldb + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) rb[h] = regionb + (unsigned)((wn * 64 + c16) * 128 + (((g4 + 4 * h) ^ ((c16 >> 1) & 7)) << 4));
    hb = 0;
    stepb = 128u;
    ldswb = lds0 + (unsigned)wave * 4096u;
  } else {
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      if (B_KC) {                        // piece = 16 rows x 64 B
        const int row = (wave * 2 + pc) * 16 + (lane >> 2);
        vob[pc] = (unsigned)((row * p.ldb + (((lane & 3) ^ kc_swz_p(row)) << 3)) * 2);
      } else {                           // piece = 2 k-rows x 512 B
        const int kr = (wave * 2 + pc) * 2 + (lane >> 5);
        vob[pc] = (unsigned)((kr * p.ldb + (((lane & 31) ^ (ks_hh_p(kr) << 1)) << 3)) * 2);
      }
    }
    vob[2] = vob[3] = 0;
    if (B_KC) {
      rb[0] = rb[1] = regionb + (unsigned)((wn * 64 + c16) * 64 + ((g4 ^ kc_swz_p(c16)) << 4));
      hb = 0;
      stepb = 64u;
    } else {
      rb[0] = rb[1] = regionb + (unsigned)(krd * 512 + ((c16 & 3) >> 1) * 16 + (c16 & 1) * 8);
      hb = (unsigned)(ks_hh_p(krd) ^ (wn * 4));
      stepb = (unsigned)(64 * p.ldb);
    }
    ldswb = lds0 + (unsigned)wave * 2048u;
  }
  const bf16_t* gA = A_KC ? p.A + (long)m0 * p.lda + kbeg : p.A + (long)kbeg * p.lda + m0;
  const bf16_t* gB = B_KC ? p.B + (long)n0 * p.ldb + kbeg : p.B + (long)kbeg * p.ldb + n0;
  const unsigned long long gAu = reinterpret_cast<unsigned long long>(gA), gBu = reinterpret_cast<unsigned long long>(gB);
  const unsigned galo = __builtin_amdgcn_readfirstlane((unsigned)gAu), gahi = __builtin_amdgcn_readfirstlane((unsigned)(gAu >> 32));
  const unsigned gblo = __builtin_amdgcn_readfirstlane((unsigned)gBu), gbhi = __builtin_amdgcn_readfirstlane((unsigned)(gBu >> 32));
  const unsigned half = (unsigned)wm;      // waves 0-3: first half (X), 4-7: second half (Y); w and w + 4 share a SIMD
  // (every scalar operand of the loop through v_readfirstlane: an "s" constraint on a value hipcc's uniformity analysis gave up on is
  // silently assigned a VGPR and the assembler then rejects the instruction)
#define P16_U(x) ((unsigned)__builtin_amdgcn_readfirstlane((int)(x)))

#if NEKO_P16_TRACE == 1
  unsigned long long tr0 = 0, tr1 = 0;
  if (tid == 0) tr0 = __builtin_amdgcn_s_memtime();
#endif
  P16_STAMP(1, __builtin_amdgcn_s_memrealtime());
  P16_STAMP(6, __builtin_amdgcn_s_memtime());
  f32x32 acc[4];
#define NEKO_P16_OPERANDS                                                                                                     \
  "={a[0:31]}"(acc[0]), "={a[32:63]}"(acc[1]), "={a[64:95]}"(acc[2]), "={a[96:127]}"(acc[3])                                     \
  : [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [vob0] "v"(vob[0]), [vob1] "v"(vob[1]),     \
    [vob2] "v"(vob[2]), [vob3] "v"(vob[3]), [ra0] "v"(ra[0]), [ra1] "v"(ra[1]), [ra2] "v"(ra[2]), [ra3] "v"(ra[3]), [rb0] "v"(rb[0]),  \
    [rb1] "v"(rb[1]), [ha] "v"(ha), [hb] "v"(hb),                                                                               \
    [galo] "s"(galo), [gahi] "s"(gahi), [gblo] "s"(gblo), [gbhi] "s"(gbhi), [sa] "s"(P16_U(stepa)), [sb] "s"(P16_U(stepb)),       \
    [nkt] "s"(P16_U(nkt)), [ntrips] "s"(P16_U(ntrips)), [ldswa] "s"(P16_U(ldswa)), [ldswb] "s"(P16_U(ldswb)), [half] "s"(P16_U(half)), \
    [tail] "s"(P16_U(tail))                                                                                                     \
  : NEKO_P16_CLOBBERS
  if constexpr (A_KC && B_KC) asm volatile(NEKO_P16_LOOP_KC_KC : NEKO_P16_OPERANDS);
  else if constexpr (A_KC && !B_KC) asm volatile(NEKO_P16_LOOP_KC_KS : NEKO_P16_OPERANDS);
  else asm volatile(NEKO_P16_LOOP_KS_KS : NEKO_P16_OPERANDS);
#undef NEKO_P16_OPERANDS
#undef P16_U
  P16_STAMP(2, __builtin_amdgcn_s_memrealtime());
  P16_STAMP(7, __builtin_amdgcn_s_memtime());
#if NEKO_P16_TRACE == 1
  if (tid == 0 && p.colsum_ws) {
    tr1 = __builtin_amdgcn_s_memtime();
    reinterpret_cast<unsigned long long*>(p.colsum_ws + (long)(p.M / 128) * p.N)[blockIdx.x] = tr1 - tr0;      // behind the bands this kernel fills
  }
#endif

  // the loop ends behind a block barrier with every DMA landed: the ring is free for the slabs
#if NEKO_GEMM_DIAG == 4
  if (p.M != 12345) return;      // ablation (timing only): no epilogue at all
#endif
  const bool to_ws = p.splitk > 1;
  float* Cf_out = to_ws ? p.splitk_ws + (long)slice * p.M * p.N : p.Cf;
  const long ldcf_out = to_ws ? p.N : p.ldcf;
  epilogue_fast<C, F>(p, ParkAgprP16{acc}, smem, m0, n0, wm, wn, wave, lane, Cf_out, ldcf_out);
#if NEKO_P16_TRACE == 2
  P16_STAMP(3, __builtin_amdgcn_s_memrealtime());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P16_STAMP(4, __builtin_amdgcn_s_memrealtime());
#endif
}

// -1: per-shape choice (default), 0: never, 1: wherever it applies
int env_mode_p16() {
  static const int v = [] { const char* e = getenv("NEKO_GEMM_P16"); return e ? atoi(e) : -1; }();
  return v;
}

template <bool A_KC, bool B_KC, unsigned F>
int launch_p16(const GemmArgs& a, hipStream_t s) {
  const int tiles = (a.M / 256) * (a.N / 256) * (a.splitk > 1 ? a.splitk : 1);
  hipLaunchKernelGGL((gemm_p16_kernel<A_KC, B_KC, F>), dim3(tiles), dim3(512), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// the feature sets this file instantiates, by operand layouts.  1 = no kernel for this combination
int dispatch_p16(const GemmArgs& a, bool a_kc, bool b_kc, unsigned f, hipStream_t s) {
#define NEKO_P16_CASE(AKC, BKC, MASK) case (MASK): return launch_p16<AKC, BKC, (MASK)>(a, s);
  if (a_kc && !b_kc) {                                                        // forward shapes (weights stored (in, out))
    switch (f) {
      NEKO_P16_CASE(true, false, F_BIAS | F_CB)                               // forward qkv
      NEKO_P16_CASE(true, false, F_BIAS | F_GELU | F_PRE | F_CB)              // forward fc (act 1)
      NEKO_P16_CASE(true, false, F_BIAS | F_GELU | F_PRE | F_GP | F_CB)       // forward fc leaving gelu'(pre) (act 3)
      NEKO_P16_CASE(true, false, F_BIAS | F_RESID | F_CF)                     // forward projections
      NEKO_P16_CASE(true, false, F_BIAS | F_DROP | F_RESID | F_CF)            // ... with residual dropout
      NEKO_P16_CASE(true, false, F_CB)
      NEKO_P16_CASE(true, false, F_CF)                                        // split-K slices (LM-head dH)
      NEKO_P16_CASE(true, false, F_CF | F_ALPHA)
      NEKO_P16_CASE(true, false, F_BIAS | F_CF)                               // patch projection
      default: return 1;
    }
  }
  if (a_kc && b_kc) {                                                         // dgrad shapes, LM-head logits
    switch (f) {
      NEKO_P16_CASE(true, true, F_GELUBWD | F_CB)                             // dgrad through the MLP projection (* GELU')
      NEKO_P16_CASE(true, true, F_GELUBWD | F_CB | F_COLSUM)
      NEKO_P16_CASE(true, true, F_GELUBWD | F_MULACT | F_CB)                  // ... * stored gelu' (act 4)
      NEKO_P16_CASE(true, true, F_GELUBWD | F_MULACT | F_CB | F_COLSUM)
      NEKO_P16_CASE(true, true, F_CB)                                         // dgrad attention out / c_fc / c_attn, LM-head logits
      NEKO_P16_CASE(true, true, F_CF)
      default: return 1;
    }
  }
  if (!a_kc && !b_kc) {                                                       // weight gradients (split-K slices to the workspace, or one slice)
    switch (f) {
      NEKO_P16_CASE(false, false, F_CF)
      NEKO_P16_CASE(false, false, F_CF | F_ACCUM)
      NEKO_P16_CASE(false, false, F_CF | F_ACCUM | F_ALPHA)
      NEKO_P16_CASE(false, false, F_CB)
      default: return 1;
    }
  }
  return 1;
#undef NEKO_P16_CASE
}

}  // namespace

// 1 = not applicable (the caller goes on to gemm_b16.hip / gemm_a16.hip / gemm_glds.hip), otherwise a status code.  *colsum_bands = number
// of 128-row bands of a.colsum_ws this launch filled (0: none / the column sums were not asked for)
int neko_gemm_p16_try(const GemmArgs& a_in, int a_kstrided, int b_kstrided, int mainloop_mode, int* colsum_bands, hipStream_t s) {
  *colsum_bands = 0;
  // neko_gemm_set_mainloop: 3 = this loop wherever it applies; 0 / 1 / 2 = never; -1 = NEKO_GEMM_P16 / per shape
  const int mode = mainloop_mode == 3 ? 1 : (mainloop_mode >= 0 ? 0 : env_mode_p16());
  if (mode == 0) return 1;
  if (a_kstrided && !b_kstrided) return 1;
  GemmArgs a = a_in;
  if ((a.M & 255) || (a.N & 255)) return 1;
  // contraction range (per split-K slice): at least one loop trip (12 k-tiles with a k-contiguous A operand, else 4) and a multiple of
  // 128 -- what is left behind whole trips (4 or 8 k-tiles) runs as the first groups of the same loop body
  const int trip_k = a_kstrided ? 128 : 32 * NEKO_P16_TRIP_KC;
  const int klen = a.splitk > 1 ? a.k_per_split : a.K;
  if (klen < trip_k || (klen % 128)) return 1;
  if (a.splitk > 1) {
    const long last = (long)a.K - (long)(a.splitk - 1) * a.k_per_split;
    if (last < trip_k || (last % 128)) return 1;
  }
  // the per-lane DMA offsets are 32-bit byte offsets within the tile's operand panel
  if ((a_kstrided ? 32 * a.lda : 256 * a.lda) * 2 >= (1L << 31) || (b_kstrided ? 32 * a.ldb : 256 * a.ldb) * 2 >= (1L << 31)) return 1;
  const bool to_ws = a.splitk > 1 && a.splitk_ws;
  if (a.splitk > 1 && !to_ws) return 1;
  if (to_ws && (a.bias || a.resid || a.act || a.Cb || a.drop_thr)) return 1;      // (alpha scales every slice: linear, the LM-head dH uses it)
  const long ldcf_out = to_ws ? a.N : a.ldcf;
  if (((ldcf_out | a.ldr | a.ldcb | a.ldact | a.ldpre) & 3)) return 1;
  // column sums ride along only with the GELU' dgrad feature sets (every tile is interior here)
  const bool fold = a.colsum_ws && (a.act == 2 || a.act == 4) && a.Cb && !a.Cf && !a.bias && !a.resid && !a.drop_thr && a.alpha == 1.0f &&
                    !a.alpha_dev && a.splitk <= 1;
#if NEKO_P16_TRACE != 1
  if (!fold) a.colsum_ws = nullptr;
#endif
  GemmArgs am = a;
  if (!fold) am.colsum_ws = nullptr;
  const unsigned f = fast_epi_mask(am, true, to_ws, to_ws || a.Cf != nullptr);
  if (!fast_epi_supported(f)) return 1;
  if (mode < 0) {
    // per-shape choice: NEKO_GEMM_P16_RULE (bit mask) selects launch classes for A/B runs: 1 residual epilogues, 2 GELU forward,
    // 4 GELU' dgrad, 8 plain K <= 1536 (forward qkv, dgrad attention out), 16 LM-head logits (N > 16384), 32 long contractions (K > 1536,
    // A k-contiguous), 64 weight gradients (both operands k-strided)
    // Default 63 = every class with a k-contiguous A operand: per launch at 65536 rows -6 ... -11 % against gemm_a16 / gemm_b16 /
    // gemm_glds64 (profiles/r06_p16_first_bench.txt), m-mix step 36.16 -> 35.20 ms (profiles/r06_p16_step_ab.txt); the weight gradients
    // (both operands k-strided, long contraction) are level with gemm_a16 and stay there.
    static const int rule = [] { const char* e = getenv("NEKO_GEMM_P16_RULE"); return e ? atoi(e) : 63; }();
    static const int min_tiles = [] { const char* e = getenv("NEKO_GEMM_P16_MIN_TILES"); return e ? atoi(e) : 256; }();      // one full round of CUs: 32768-row steps (384 tiles at N = 768)
                                                                                    // 19.53 -> 19.23 ms with 384 / 192 against 512; README sizes level (profiles/r06_p16_min_tiles_b32.txt, r06_nt_sizes_ab.txt)
    const long tiles = (long)(a.M / 256) * (a.N / 256) * (a.splitk > 1 ? a.splitk : 1);
    int cls;
    if (a_kstrided) cls = 64;
    else if (f & F_RESID) cls = 1;
    else if (f & F_GELU) cls = 2;
    else if (f & F_GELUBWD) cls = 4;
    else if (a.N > 16384) cls = 16;
    else cls = klen <= 1536 ? 8 : 32;
    if (!(rule & cls) || tiles < min_tiles) return 1;
  }
  a.epi_lock = 0;
  const int rc = dispatch_p16(a, !a_kstrided, !b_kstrided, f, s);
  if (rc == NEKO_OK) g_neko_last_mainloop = 5;
  if (rc == NEKO_OK && fold) *colsum_bands = a.M / 128;
  return rc;
}
