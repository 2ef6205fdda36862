// Pieces of the LDS-DMA GEMM shared by its two main loops (gemm_glds.hip: hipcc-scheduled 32x32x16 loop; gemm_a16.hip: hand-placed
// 16x16x32 loop): tile rasterisation, tile configuration, and the fast epilogue compiled per feature set.  Each translation unit
// gets its own copy (anonymous namespace).
#pragma once
#include <type_traits>
#include "neko_kernels.h"

#ifndef NEKO_EPI_ABL
#define NEKO_EPI_ABL 0     // epilogue ablations for tools/gemm_trace.py (wrong results): 1 no global stores, 2 GELU = identity, 3 no slab writes
#endif
#ifndef NEKO_GEMM_DIAG
#define NEKO_GEMM_DIAG 0   // ablations for tools/gemm_bench.py: 1 no in-loop DMA, 4 no epilogue
#endif

namespace {

constexpr int BK = 32;

// block -> output tile, grouped rasterisation: consecutive logical ids (one XCD's L2, see xcd_remap) cover groups
// of 8 row panels and sweep the column panels inside a group, 8 tiles per column panel.  The tiles in flight on an
// XCD then share <= 8 A panels and a few B panels: with B small (activations x weight) A streams once as before;
// with A small and B huge (LM-head logits: 4096 rows x 52k vocabulary columns) the embedding table streams once
// per 8 row panels instead of once per row panel.
// Split-K launches are ONE flat grid of tiles x slices with the slice as the slow index of the logical id: the
// contiguous logical range an XCD gets from xcd_remap then lies inside one or two k-slices, whose tiles share their A and
// B panels pairwise.  (As a 2-D grid the tiles of a slice were dealt round-robin over all eight XCDs and every L2 fetched
// nearly every panel of every slice: the fc / proj weight gradients moved 730 MB over the fabric for 250 MB of operands.)
template <int BM, int BN>
__device__ __forceinline__ void tile_coords_of(const GemmArgs& p, int block, int& tm, int& tn, int& slice);
template <int BM, int BN>
__device__ __forceinline__ void tile_coords(const GemmArgs& p, int& tm, int& tn, int& slice) {
  tile_coords_of<BM, BN>(p, (int)blockIdx.x, tm, tn, slice);
}
// (the tile of any block of the launch: gemm_p16.hip asks for the tile of the block that follows its own on the same XCD)
template <int BM, int BN>
__device__ __forceinline__ void tile_coords_of(const GemmArgs& p, int block, int& tm, int& tn, int& slice) {
#ifndef NEKO_GEMM_GROUP_M
#define NEKO_GEMM_GROUP_M 8        // row panels per rasterisation group (4 / 16 measured in round 3: profiles/r03_step_ab.txt)
#endif
  const int GROUP_M = p.group_m > 0 ? p.group_m : NEKO_GEMM_GROUP_M;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  int bid = xcd_remap(block, gridDim.x);
  slice = bid / (nbm * nbn);
  bid -= slice * (nbm * nbn);
  const int per_group = GROUP_M * nbn;
  const int g = bid / per_group, local = bid - g * per_group;
  const int gsz = min(GROUP_M, nbm - g * GROUP_M);
  tm = g * GROUP_M + local % gsz;
  tn = local / gsz;
}

template <int WM_, int WN_, int TM_, int TN_, int NSTAGE>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int NW = WM * WN, NT = 64 * NW;
  static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int GLDS_PER_STAGE = (BM + BN) / 16 / NW;        // wave-instructions each wave issues per stage
  static constexpr int SLAB_BYTES = 32 * 32 * TN * 4;               // per wave: [32 rows][32*TN f32]
  static constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
  static constexpr int LDS_BYTES = RING_BYTES > NW * SLAB_BYTES ? RING_BYTES : NW * SLAB_BYTES;
  static constexpr int BLOCKS_PER_CU = (160 * 1024) / LDS_BYTES;
  // waves per SIMD the register budget must allow (launch bound): blocks/CU * waves/block / 4 SIMDs
  static constexpr int WAVES_PER_SIMD = (BLOCKS_PER_CU * NW + 3) / 4;
};

// ---- fast epilogue --------------------------------------------------------------------------------------------------
// The generic epilogue below decides everything per 4-element step at run time (output kinds, activation, dropout,
// bounds): ~10 uniform branches and several 64-bit multiplies per step, 10k instructions, ~23 VALU slots per output
// element even for a plain bf16 store -- at K = 768 that was 30-55 % of the GEMM (a 256x256 tile is 1024 elements
// per physical VALU lane).  This version is compiled per FEATURE SET (template mask F), takes only full interior
// tiles with 16-B aligned rows (the caller checks), parks 32 accumulator rows in a PADDED slab (row stride +4
// floats: every ds_write_b32 / ds_read_b128 address is lane base + literal offset, no swizzle arithmetic) and walks
// the output with pointers that advance by a constant row step.
enum : unsigned { F_BIAS = 1, F_GELU = 2, F_PRE = 4, F_GELUBWD = 8, F_DROP = 16, F_RESID = 32, F_CF = 64, F_ACCUM = 128,
                  F_CB = 256, F_ALPHA = 512, F_COLSUM = 1024,
                  F_GP = 2048,        // with F_GELU | F_PRE: the pre_out store holds gelu'(pre) instead of pre (act = 3)
                  F_MULACT = 4096 };  // with F_GELUBWD: act_in already IS the factor (act = 4), no gelu' evaluation

template <class C>
struct FastEpi {
  static constexpr int SW = 32 * C::TN, SWP = SW + 4;          // slab row stride (floats), padded
  static constexpr int SLAB_BYTES = 32 * SWP * 4;
  static constexpr int CPR = SW / 4, RPI = 64 / CPR;           // float4 chunks per row, rows per wave-instruction
  static_assert(C::NW * SLAB_BYTES <= C::LDS_BYTES, "padded slabs must fit the ring");
};

// Output stores of the fast epilogue.  NEKO_EPI_STORE_POLICY: 0 plain (write-back: the line stays in the XCD's L2), 1 nt (streaming
// hint; the default since round 6), 2 sc1 (the line is written through and dropped from the L2: MI355X_MICROARCH.md, stores of each
// flavour), 3 sc0 sc1.  A tile's output is never read again by the launch that writes it, and the 128-256 KB a workgroup stores per tile
// (x 32 CUs per XCD = a whole 4 MB L2 per round) compete with the operand panels the main loops of the same launch re-read: with nt the
// k-loop of the forward qkv GEMM runs at 1200 instead of 1500 clocks per k-tile (profiles/r06_p16_phase_trace.txt), per launch -4 ... -15 %
// (profiles/r06_p16_variants.txt), per step m-mix -0.16 ms, c4 -6.5 %, c2 -2 %, m-text -2 % (profiles/r06_store_policy_step_ab.txt,
// r06_nt_sizes_ab.txt); sc1 / sc0 sc1 measured level with plain.
#ifndef NEKO_EPI_STORE_POLICY
#define NEKO_EPI_STORE_POLICY 1
#endif
#ifndef NEKO_EPI_BATCH_READS
#define NEKO_EPI_BATCH_READS 1     // round 6: -1.4 ... -2.5 % per launch on the bf16-output shapes, level inside the step (profiles/r06_p16_batchreads_ab.txt); no spills
#endif
#ifndef NEKO_EPI_STORE_POLICY_CF
#define NEKO_EPI_STORE_POLICY_CF NEKO_EPI_STORE_POLICY      // the fp32 outputs (16-B stores) can take their own policy
#endif
typedef uint32_t epi_u32x2 __attribute__((ext_vector_type(2)));
typedef float epi_f32x4 __attribute__((ext_vector_type(4)));
// Policy 1 goes through the compiler's own non-temporal store (global_store_dwordx2 / x4 ... nt): hipcc then counts the stores in its
// s_waitcnt vmcnt(N) bookkeeping and pads the wide-store data hazard itself.  The first version of this round issued them from inline asm:
// (a) a 16-byte store reads its data registers for a few cycles after it issues and the next instruction overwrote them (wrong fp32 outputs
// until wait states went behind the store), (b) stores the compiler cannot see make every later vmcnt wait for a prefetched epilogue input
// also wait for the stores issued since -- their HBM acknowledgements -- which lengthened the residual / GELU' epilogues (16 -> 19 us per tile).
// Policies 2 / 3 (sc1, sc0 sc1: A/B builds only) still use inline asm, with the wait states.
__device__ __forceinline__ void epi_store8(void* dst, uint2 v) {
#if NEKO_EPI_STORE_POLICY == 0
  *reinterpret_cast<uint2*>(dst) = v;
#else
  const epi_u32x2 w = {v.x, v.y};
#if NEKO_EPI_STORE_POLICY == 1
  __builtin_nontemporal_store(w, reinterpret_cast<epi_u32x2*>(dst));
#elif NEKO_EPI_STORE_POLICY == 2
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(w) : "memory");
#else
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(w) : "memory");
#endif
#endif
}
__device__ __forceinline__ void epi_store16(void* dst, float4 v) {
#if NEKO_EPI_STORE_POLICY_CF == 0
  *reinterpret_cast<float4*>(dst) = v;
#else
  const epi_f32x4 w = {v.x, v.y, v.z, v.w};
#if NEKO_EPI_STORE_POLICY_CF == 1
  __builtin_nontemporal_store(w, reinterpret_cast<epi_f32x4*>(dst));
#elif NEKO_EPI_STORE_POLICY_CF == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
#endif
#endif
}

template <int N, int I = 0, class Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<N, I + 1>(static_cast<Fn&&>(fn));
  }
}

// Park policy: how pass I (32 output rows of the wave) gets from the accumulators into the wave's slab.  This one serves the
// 32x32x16 loops: the operands of the MFMA are swapped (see gemm_glds_kernel), so lane l holds row l&31 of a 32 x 32 block and
// its columns 8q + 4(l>>5) + 0..3 in acc[4q .. 4q+3] -- four consecutive columns per register quad, one ds_write_b128.
template <class C>
struct ParkAcc32 {
  static constexpr int PREFETCH = 0;       // 0: a whole 32-row pass of epilogue inputs in flight
  f32x16 (&acc)[C::TM][C::TN];
  template <int I>
  __device__ __forceinline__ void park(float* slab, int lane) const {
    float* wbase = slab + (lane & 31) * FastEpi<C>::SWP + 4 * (lane >> 5);             // + 32 j + 8 q
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#if NEKO_EPI_ABL == 3
        asm volatile("" ::"v"(acc[I][j][4 * q]), "v"(acc[I][j][4 * q + 1]), "v"(acc[I][j][4 * q + 2]), "v"(acc[I][j][4 * q + 3]));
#else
        *reinterpret_cast<float4*>(wbase + 32 * j + 8 * q) =
            make_float4(acc[I][j][4 * q], acc[I][j][4 * q + 1], acc[I][j][4 * q + 2], acc[I][j][4 * q + 3]);
#endif
      }
  }
};

template <class C, unsigned F, class Park>
__device__ __forceinline__ void epilogue_fast(const GemmArgs& p, const Park& parker, char* smem, int m0, int n0,
                                              int wm, int wn, int wave, int lane, float* Cf_out, long ldcf_out) {
  using E = FastEpi<C>;
  constexpr int TM = C::TM, TN = C::TN, SWP = E::SWP, CPR = E::CPR, RPI = E::RPI;
  float* slab = reinterpret_cast<float*>(smem + wave * E::SLAB_BYTES);
  const int cchunk = lane % CPR, rsub = lane / CPR;
  const float* rbase = slab + rsub * SWP + cchunk * 4;                   // + s * RPI * SWP
  const int col = n0 + wn * E::SW + cchunk * 4;
  const int row0 = m0 + wm * TM * 32 + rsub;
  float alpha = 1.0f;
  if (F & F_ALPHA) alpha = p.alpha_dev ? p.alpha * (*p.alpha_dev) : p.alpha;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (F & F_BIAS) bv = *reinterpret_cast<const float4*>(p.bias + col);
  // row pointers at row0; advanced by RPI rows per step (and 32 rows per pass by construction: 32 = 8 steps * RPI ... )
  float* pcf = (F & F_CF) ? Cf_out + (long)row0 * ldcf_out + col : nullptr;
  bf16_t* pcb = (F & F_CB) ? p.Cb + (long)row0 * p.ldcb + col : nullptr;
  bf16_t* ppre = (F & F_PRE) ? p.pre_out + (long)row0 * p.ldpre + col : nullptr;
  const bf16_t* pact = (F & F_GELUBWD) ? p.act_in + (long)row0 * p.ldact + col : nullptr;
  const float* pres = (F & F_RESID) ? p.resid + (long)row0 * p.ldr + col : nullptr;
  uint32_t didx = (F & F_DROP) ? (uint32_t)row0 * (uint32_t)p.N + (uint32_t)col : 0u;
  const long scf = (long)RPI * ldcf_out, scb = (long)RPI * p.ldcb, spre = (long)RPI * p.ldpre, sact = (long)RPI * p.ldact,
             sres = (long)RPI * p.ldr;
  const uint32_t sdrop = (uint32_t)RPI * (uint32_t)p.N;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};      // F_COLSUM: this lane's 4 columns summed over its rows of the wave's band
  static_for<TM>([&](auto pass_tag) {
    parker.template park<decltype(pass_tag)::value>(slab, lane);
    // every global INPUT of this pass (GELU' argument, residual, accumulate target) is fetched up front: inside the step
    // loop each load would sit behind the previous step's stores (possible aliasing) and cost a full HBM round trip,
    // 32 of them per tile (the dgrad through the MLP projection spent half its time there)
    // (a park policy may bound the prefetch depth -- Park::PREFETCH steps at a time -- when the wave has 128 instead of 256 VGPRs:
    // gemm_b16.hip's 16 steps x float4 of residual would spill)
    constexpr int NST_ALL = 32 / RPI;
    constexpr int NST = (Park::PREFETCH > 0 && Park::PREFETCH < NST_ALL) ? Park::PREFETCH : NST_ALL;
    static_assert(NST_ALL % NST == 0, "prefetch depth must divide the steps of a pass");
#pragma unroll
    for (int ch = 0; ch < NST_ALL / NST; ++ch) {
    uint2 pre_act[(F & F_GELUBWD) ? NST : 1];
    float4 pre_res[(F & F_RESID) ? NST : 1];
    float4 pre_acc[(F & F_ACCUM) ? NST : 1];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (F & F_GELUBWD) pre_act[st] = *reinterpret_cast<const uint2*>(pact + (long)st * sact);
      if (F & F_RESID) pre_res[st] = *reinterpret_cast<const float4*>(pres + (long)st * sres);
      if (F & F_ACCUM) pre_acc[st] = *reinterpret_cast<const float4*>(pcf + (long)st * scf);
    }
    // same-wave LDS write -> read: ordered by the LDS queue, no barrier (a slab is private to its wave)
#if NEKO_EPI_BATCH_READS
    // every slab row of the chunk is requested before the first is used (one LDS round trip per chunk instead of one per pair of steps:
    // left to itself hipcc keeps two ds_read_b128 in flight and waits for each)
    float4 slab_rows[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) slab_rows[st] = *reinterpret_cast<const float4*>(rbase + (ch * NST + st) * RPI * SWP);
    // (pinned in registers HERE, four rows per statement: the loads cannot be sunk to their uses)
#pragma unroll
    for (int st = 0; st + 3 < NST; st += 4)
      asm volatile("" : "+v"(slab_rows[st].x), "+v"(slab_rows[st].y), "+v"(slab_rows[st].z), "+v"(slab_rows[st].w),
                        "+v"(slab_rows[st + 1].x), "+v"(slab_rows[st + 1].y), "+v"(slab_rows[st + 1].z), "+v"(slab_rows[st + 1].w),
                        "+v"(slab_rows[st + 2].x), "+v"(slab_rows[st + 2].y), "+v"(slab_rows[st + 2].z), "+v"(slab_rows[st + 2].w),
                        "+v"(slab_rows[st + 3].x), "+v"(slab_rows[st + 3].y), "+v"(slab_rows[st + 3].z), "+v"(slab_rows[st + 3].w));
#endif
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#if NEKO_EPI_BATCH_READS
      const float4 a4 = slab_rows[st];
#else
      const float4 a4 = *reinterpret_cast<const float4*>(rbase + (ch * NST + st) * RPI * SWP);
#endif
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
      if (F & F_ALPHA) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= alpha;
      }
      if (F & F_BIAS) { v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
      if (F & F_GELU) {
        const uint32_t p01 = pack_bf16x2(v[0], v[1]), p23 = pack_bf16x2(v[2], v[3]);     // bf16 pre-activation
        const f32x2_v x01 = (f32x2_v){__uint_as_float(p01 << 16), __uint_as_float(p01 & 0xffff0000u)};
        const f32x2_v x23 = (f32x2_v){__uint_as_float(p23 << 16), __uint_as_float(p23 & 0xffff0000u)};
        if (F & F_GP) {
          // act = 3: one evaluation of the erf series yields gelu(pre) AND gelu'(pre); the second output is gelu' (bf16), so
          // the backward's epilogue is a plain multiply by the stored factor instead of a second erf evaluation per element
          f32x2_v g01, g23, d01, d23;
          gelu_and_grad2_f(x01, g01, d01);
          gelu_and_grad2_f(x23, g23, d23);
#if NEKO_EPI_ABL == 1
          asm volatile("" ::"v"(d01), "v"(d23));
#else
          epi_store8(ppre, make_uint2(pack_bf16x2(d01.x, d01.y), pack_bf16x2(d23.x, d23.y)));
#endif
          v[0] = g01.x; v[1] = g01.y; v[2] = g23.x; v[3] = g23.y;
        } else {
#if NEKO_EPI_ABL == 1
          if (F & F_PRE) asm volatile("" ::"v"(p01), "v"(p23));
#else
          if (F & F_PRE) epi_store8(ppre, make_uint2(p01, p23));
#endif
#if NEKO_EPI_ABL == 2
          const f32x2_v g01 = x01, g23 = x23;
#else
          const f32x2_v g01 = gelu2_f(x01), g23 = gelu2_f(x23);
#endif
          v[0] = g01.x; v[1] = g01.y; v[2] = g23.x; v[3] = g23.y;
        }
      }
      if (F & F_GELUBWD) {
        const uint2 q = pre_act[st];
        const f32x2_v a01 = (f32x2_v){__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u)};
        const f32x2_v a23 = (f32x2_v){__uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
        const f32x2_v g01 = (F & F_MULACT) ? a01 : gelu_grad2_f(a01);
        const f32x2_v g23 = (F & F_MULACT) ? a23 : gelu_grad2_f(a23);
        v[0] *= g01.x; v[1] *= g01.y; v[2] *= g23.x; v[3] *= g23.y;
      }
      if (F & F_DROP) drop4(v, didx, p.drop_key, p.drop_thr, p.drop_scale);
      if (F & F_COLSUM) { cs[0] += v[0]; cs[1] += v[1]; cs[2] += v[2]; cs[3] += v[3]; }
      if (F & F_RESID) {
        const float4 q = pre_res[st];
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      }
      if (F & F_CF) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (F & F_ACCUM) {
          const float4 q = pre_acc[st];
          o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        epi_store16(pcf, o);
      }
#if NEKO_EPI_ABL == 1
      if (F & F_CB) asm volatile("" ::"v"(pack_bf16x2(v[0], v[1])), "v"(pack_bf16x2(v[2], v[3])));
#else
      if (F & F_CB) epi_store8(pcb, make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
#endif
      // next RPI rows
      if (F & F_CF) pcf += scf;
      if (F & F_CB) pcb += scb;
      if (F & F_PRE) ppre += spre;
      if (F & F_DROP) didx += sdrop;
    }
    // the inputs were indexed from the chunk base: advance them by the chunk (NST * RPI rows; the whole pass when NST == NST_ALL)
    if (F & F_GELUBWD) pact += (long)NST * sact;
    if (F & F_RESID) pres += (long)NST * sres;
    }   // prefetch chunks
  });
  if (F & F_COLSUM) {
    // lanes that share a column chunk differ in lane / CPR: fold them in a fixed order, then one row of the band table per
    // wave band (32*TM rows): [m0 / 32 / TM + wm][N]; neko_colsum_bands_reduce_impl adds the bands up in index order
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[e] += __shfl_xor(cs[e], o, 64);
    }
    if (lane < CPR)
      *reinterpret_cast<float4*>(p.colsum_ws + (long)(m0 / (32 * TM) + wm) * p.N + col) = make_float4(cs[0], cs[1], cs[2], cs[3]);
  }
}

// feature mask of a launch (host and device): what the compiled epilogue has to do for this GemmArgs
__host__ __device__ __forceinline__ unsigned fast_epi_mask(const GemmArgs& p, bool lead, bool to_ws, bool has_cf) {
  unsigned f = 0;
  if (p.bias && lead) f |= F_BIAS;
  if (p.act == 1) f |= F_GELU | (p.pre_out ? F_PRE : 0);
  if (p.act == 2) f |= F_GELUBWD;
  if (p.act == 3) f |= F_GELU | F_PRE | F_GP;
  if (p.act == 4) f |= F_GELUBWD | F_MULACT;
  if (p.drop_thr) f |= F_DROP;
  if (p.resid && lead) f |= F_RESID;
  if (has_cf) f |= F_CF | ((!to_ws && p.accumulate) ? F_ACCUM : 0);
  if (p.Cb) f |= F_CB;
  if (p.alpha != 1.0f || p.alpha_dev) f |= F_ALPHA;
  if (p.colsum_ws) f |= F_COLSUM;           // the host only passes it on when every tile of the launch takes this path
  return f;
}
// the masks try_epilogue_fast() has a compiled epilogue for (keep in step with its switch)
__host__ __device__ __forceinline__ bool fast_epi_supported(unsigned f) {
  switch (f) {
    case F_BIAS | F_CB: case F_BIAS | F_GELU | F_PRE | F_CB: case F_BIAS | F_RESID | F_CF: case F_BIAS | F_DROP | F_RESID | F_CF:
    case F_GELUBWD | F_CB: case F_GELUBWD | F_CB | F_COLSUM: case F_BIAS | F_GELU | F_PRE | F_GP | F_CB:
    case F_GELUBWD | F_MULACT | F_CB: case F_GELUBWD | F_MULACT | F_CB | F_COLSUM: case F_CB: case F_CF: case F_CF | F_ALPHA:
    case F_CF | F_ACCUM: case F_CF | F_ACCUM | F_ALPHA: case F_BIAS | F_CF:
      return true;
    default: return false;
  }
}

// dispatch: feature mask of this launch -> a compiled fast epilogue, or false (caller runs the generic one)
template <class C, class Park>
__device__ __forceinline__ bool try_epilogue_fast(const GemmArgs& p, const Park& parker, char* smem, int m0,
                                                  int n0, int wm, int wn, int wave, int lane, int slice) {
#if NEKO_GEMM_DIAG == 4
  if (p.M != 12345) return true;      // ablation: no epilogue at all
#endif
  const bool to_ws = p.splitk > 1 && p.splitk_ws;
  if (p.splitk > 1 && !to_ws) return false;                                   // atomic split-K: generic path
  if (m0 + C::BM > p.M || n0 + C::BN > p.N) return false;                     // edge tile
  float* Cf_out = to_ws ? p.splitk_ws + (long)slice * p.M * p.N : p.Cf;
  const long ldcf_out = to_ws ? p.N : p.ldcf;
  const bool lead = !to_ws || slice == 0;                                // bias/resid only once over split-K slices
  if (to_ws && (p.bias || p.resid || p.act || p.Cb || p.drop_thr)) return false;
  if (((ldcf_out | p.ldr | p.ldcb | p.ldact | p.ldpre) & 3) || (p.N & 3)) return false;
  const unsigned f = fast_epi_mask(p, lead, to_ws, Cf_out != nullptr);
#define NEKO_FAST_EPI(MASK)                                                                          \
  case (MASK): epilogue_fast<C, (MASK)>(p, parker, smem, m0, n0, wm, wn, wave, lane, Cf_out, ldcf_out); \
    return true;
  switch (f) {
    NEKO_FAST_EPI(F_BIAS | F_CB)                                  // forward qkv
    NEKO_FAST_EPI(F_BIAS | F_GELU | F_PRE | F_CB)                 // forward fc
    NEKO_FAST_EPI(F_BIAS | F_RESID | F_CF)                        // forward proj (dropout off)
    NEKO_FAST_EPI(F_BIAS | F_DROP | F_RESID | F_CF)               // forward proj (residual dropout)
    NEKO_FAST_EPI(F_GELUBWD | F_CB)                               // dgrad through the MLP projection (* GELU')
    NEKO_FAST_EPI(F_GELUBWD | F_CB | F_COLSUM)                    // ... with the c_fc bias gradient folded in
    NEKO_FAST_EPI(F_BIAS | F_GELU | F_PRE | F_GP | F_CB)          // forward fc that leaves gelu'(pre) for the backward (act = 3)
    NEKO_FAST_EPI(F_GELUBWD | F_MULACT | F_CB)                    // dgrad through the MLP projection (* stored gelu', act = 4)
    NEKO_FAST_EPI(F_GELUBWD | F_MULACT | F_CB | F_COLSUM)
    NEKO_FAST_EPI(F_CB)                                           // dgrad attention out, LM-head logits
    NEKO_FAST_EPI(F_CF)                                           // dgrad fc / qkv, split-K slices
    NEKO_FAST_EPI(F_CF | F_ALPHA)                                 // LM-head dH (device-side grad_output)
    NEKO_FAST_EPI(F_CF | F_ACCUM)                                 // weight gradient, single slice
    NEKO_FAST_EPI(F_CF | F_ACCUM | F_ALPHA)                       // LM-head dW
    NEKO_FAST_EPI(F_BIAS | F_CF)                                  // patch projection
    default: return false;
  }
#undef NEKO_FAST_EPI
}

}  // namespace
