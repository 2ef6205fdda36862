// Streaming attention for wide heads (hd = 64 / 128), forward and backward: K / V (or Q / dO) tiles of 64 rows travel
// HBM -> LDS by DMA (global_load_lds_dwordx4) through a ring of stages, one block barrier per tile, counted vmcnt waits.
// Replaces Attention._attn + split_heads / merge_heads + the mask preparation of the reference
// (gato/transformers/trajectory_gpt2.py:163-188,190-201,222-226,252 and :663-679) and their autograd, for the widths the
// head-resident kernels of attention_res.hip do not cover (configs[4]: 2048d x 16 heads, hd = 128).
//
// Same arithmetic, same mask semantics and the same dropout index as attention.hip (whose register-staged kernels these
// replace at hd >= 64: two barriers per tile, V transposed through 2-byte LDS stores, 180-335 TFLOP/s at hd = 128):
//     s = (q.k)/sqrt(hd);  s = (key <= query) ? s : -1e4 (REPLACE);  s += (1 - mask[key]) * -1e4 (ADD)
// scores are computed transposed (S^T = K.Q^T, one lane owns one query column), probabilities leave the accumulator as
// the B operand of O^T += V^T.P^T with the key order permuted identically on the V^T side.
//
// LDS images: a tile is [64 rows][hd] bf16, row-major, 16-B chunk c of row r stored at chunk c ^ swz(r).  swz() is
// chosen so that BOTH access patterns are conflict-free:
//   * ds_read_b128 of one chunk column by 16 consecutive rows (the K / Q / V / dO "natural" A fragments), and
//   * ds_read_b64_tr_b16 of 4 consecutive rows x 4 consecutive chunks (the transposed V^T / K^T / Q^T / dO^T fragments:
//     the hardware transposes 4 x 4 blocks of 16-bit values inside each 16-lane group, no transposed copy is ever written).
// The DMA writes 1 KiB per wave instruction at consecutive LDS addresses, so the swizzle is applied on the GLOBAL side:
// lane L of a piece fetches the chunk whose swizzled position is L.
#include <atomic>
#include "neko_kernels.h"

extern int neko_attn_path_mode();

#ifndef NEKO_AS_DIAG
#define NEKO_AS_DIAG 0      // ablations for tools/attn_bench.py (wrong results): 1 no in-loop DMA, 2 no softmax arithmetic, 4 no P.V, 8 no barrier
#endif

namespace {

constexpr int KT = 64;
constexpr int TMAX = 4096;      // key-bias image of a whole sequence lives in LDS
constexpr float MASK_VAL = -10000.0f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

template <int HD> struct SC {
  static constexpr int CH = HD / 8;          // 16-B chunks per row
  static constexpr int ROWB = HD * 2;        // bytes per row
  static constexpr int TILE = KT * ROWB;     // bytes per 64-row tile
  static constexpr int KS = HD / 16;         // MFMA k-steps over the head dim
  static constexpr int IB = HD / 32;         // 32-wide blocks over the head dim
  static constexpr int PIECES = TILE / 1024; // DMA wave-instructions per tile
};
// SW = 0: tiles read BOTH ways (natural ds_read_b128 by 16 consecutive rows and ds_read_b64_tr_b16): the swizzle is injective on
//         the low 4 row bits; SW = 1: tiles read only transposed: the swizzle depends on the two row bits that separate the 4 rows
//         of one transposed read -- rows 8 and 16 apart then share it, and every fragment of a tile is one per-lane base register
//         plus an immediate offset
template <int HD, int SW>
__device__ __forceinline__ int swz(int row) {
  if constexpr (HD == 128) return SW ? ((row & 3) << 2) : (((row & 3) << 2) | ((row >> 2) & 3));
  else return SW ? (((row >> 1) & 1) << 2) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));       // hd = 64: two rows per 256-B bank sweep
}
template <int HD, int SW>
__device__ __forceinline__ int img_off(int row, int c) { return row * SC<HD>::ROWB + ((c ^ swz<HD, SW>(row)) << 4); }

typedef __attribute__((address_space(3))) const uint4 lds_u4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ bf16x8_v lds_read16(unsigned addr) {
  return __builtin_bit_cast(bf16x8_v, *reinterpret_cast<lds_u4*>((uintptr_t)addr));
}
__device__ __forceinline__ bf16x8_v lds_read_tr(unsigned addr_lo, unsigned addr_hi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((uintptr_t)addr_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((uintptr_t)addr_hi));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_v, r);
}
// natural A fragment of k-step ks: row (lane % 32) [+ 32 t by an immediate], head-dim slots 16 ks + 8 (lane / 32) + 0..7.
// Byte offset inside the tile, lane part only; rows 32 apart share the swizzle
template <int HD, int SW>
__device__ __forceinline__ unsigned nat_off(int ks, int lane) { return (unsigned)img_off<HD, SW>(lane & 31, 2 * ks + (lane >> 5)); }
// transposed A fragment: row = head-dim index 32 i + lane % 32; contraction slot j of lane half h = tile row
// 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the order in which a 32x32 accumulator leaves its rows in a lane.  Offset of the `lo`
// read at s = 0 (lane part); with SW = 1 the `hi` read is 8 rows further and step s is 16 s rows further, both immediates
template <int HD, int SW>
__device__ __forceinline__ unsigned tr_off(int i, int s, int hi, int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  const int col = 32 * i + 16 * (g & 1) + 4 * (c16 & 3);
  const int r = 16 * s + 8 * hi + 4 * (g >> 1) + (c16 >> 2);
  return (unsigned)(img_off<HD, SW>(r, col >> 3) + ((col & 7) << 1));
}
// B fragment of contraction step h2 (16 rows of a 32-row sub-tile) from a 32x32 accumulator
__device__ __forceinline__ bf16x8_v frag_from_acc(const f32x16& a, int h2) {
  const int o = 8 * h2;
  const uint4 r = make_uint4(pack_bf16x2(a[o + 0], a[o + 1]), pack_bf16x2(a[o + 2], a[o + 3]),
                             pack_bf16x2(a[o + 4], a[o + 5]), pack_bf16x2(a[o + 6], a[o + 7]));
  return __builtin_bit_cast(bf16x8_v, r);
}
// own-row B fragments (the lane's query / key row, head-dim slots 16 ks + 8 (lane / 32) ..+7) straight from HBM
template <int HD>
__device__ __forceinline__ void row_frags(const bf16_t* __restrict__ rowptr, bool valid, int lane, bf16x8_v (&f)[SC<HD>::KS]) {
#pragma unroll
  for (int ks = 0; ks < SC<HD>::KS; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid) v = *reinterpret_cast<const uint4*>(rowptr + ks * 16 + (lane >> 5) * 8);
    f[ks] = __builtin_bit_cast(bf16x8_v, v);
  }
}
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x, int i) {   // value of lane (lane & ~3) + i, i literal 0..3
  switch (i) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x00, 0xf, 0xf, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x55, 0xf, 0xf, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xAA, 0xf, 0xf, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xFF, 0xf, 0xf, true);
  }
}

// lane ^ 32 exchange on the VALU (v_permlane32_swap_b32, new on gfx950): both lanes of a pair end with the pair's maximum / sum.
// __shfl_xor(x, 32) is a ds_bpermute: an LDS crossbar round trip in the middle of the softmax chain
__device__ __forceinline__ float pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- DMA ------------------------------------------------------------------------------------------------------------
// Issued from inline asm: the compiler's wait-count pass does not see these loads, the only waits are the counted ones
// below (with the builtin it drains the queue in front of every first LDS read of a stage).  Wave-uniform 64-bit base in
// SGPRs, 32-bit per-lane byte offset, LDS destination in M0 by register constraint.
__device__ __forceinline__ void glds16_s(const void* base_uniform, unsigned byte_off, unsigned lds_dst_wave_uniform) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(byte_off), "s"(base_uniform), "{m0}"(lds_dst_wave_uniform) : "memory");
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  return reinterpret_cast<const void*>(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                                       (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const char*)p);
}
// this wave's pieces of one 64-row tile: per-lane byte offsets of the pieces relative to the tile's first row (row stride ldb
// bytes), computed once; `last` = the same for the last tile of the sequence, whose rows past T - 1 re-read row T - 1
template <int HD, int NW, int SW>
__device__ __forceinline__ void dma_offsets(unsigned ldb, int T, int wave, int lane, unsigned (&off)[SC<HD>::PIECES / NW],
                                            unsigned (&off_last)[SC<HD>::PIECES / NW]) {
  constexpr int PPW = SC<HD>::PIECES / NW;
  const int last0 = (T - 1) / KT * KT;
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave * PPW + i;
    const int row = p * (1024 / SC<HD>::ROWB) + lane / SC<HD>::CH;
    const int c = (lane % SC<HD>::CH) ^ swz<HD, SW>(row);
    off[i] = (unsigned)row * ldb + (unsigned)c * 16u;
    off_last[i] = (unsigned)(min(last0 + row, T - 1) - last0) * ldb + (unsigned)c * 16u;
  }
}
template <int HD, int NW>
__device__ __forceinline__ void dma_tile(const void* tile_base_uniform, const unsigned (&off)[SC<HD>::PIECES / NW], unsigned img_lds, int wave) {
  constexpr int PPW = SC<HD>::PIECES / NW;
#pragma unroll
  for (int i = 0; i < PPW; ++i) glds16_s(tile_base_uniform, off[i], __builtin_amdgcn_readfirstlane(img_lds + (wave * PPW + i) * 1024));
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id.  The workgroups of one (sequence, head) stream the same K / V
// tiles in the same order: ids n and n + 8 are made the same head, so the second reader of a tile finds it in its XCD's L2 (with
// the plain grid order the P query blocks of a head sat on P different XCDs and every one of them fetched K / V from HBM).
// Returns the (sequence * H + head) index and the tile pair of this workgroup; P = pairs per head.
__device__ __forceinline__ void xcd_remap(int P, int nbh, int& bh, int& pair) {
  const int n = blockIdx.x;
  const int grp = 8 * P, full = (nbh / 8) * grp;
  if (n < full) {
    const int g = n % grp;
    bh = (n / grp) * 8 + (g & 7);
    pair = g >> 3;
  } else {
    bh = (nbh / 8) * 8 + (n - full) / P;
    pair = (n - full) % P;
  }
}

// Where a (sequence, head) lives (same conventions as attention_res.hip's SeqGeom).  Uniform batches: sequence b = rows [b T, (b+1) T);
// packed batches (neko_attn_*_varlen, ABI v16 for hd = 64 / 128): rows seq_off[b] .. seq_off[b+1]-1, every sequence with its own length,
// ONE launch sized for the longest.  lse / D are [sequence][head][position] = row0 * H + h * T_b + q; the dropout hash walks that index
// with the row stride of the longest sequence.  For a uniform batch every value below is what the kernels computed before (bit-identical).
struct SeqG {
  int b, T;          // sequence index, its length
  long row0;         // first row of the sequence in the [rows, ...] matrices
  long hrow;         // index of (b, h, position 0) in lse / D, and unique row id base of the dropout hash
  uint32_t T4;       // dropout hash words per row
};
__device__ __forceinline__ SeqG seq_geom(int b, int h, int H, int T_launch, const int* __restrict__ seq_off) {
  SeqG g;
  g.b = b;
  if (seq_off) {
    g.row0 = seq_off[b];
    g.T = seq_off[b + 1] - (int)g.row0;
  } else {
    g.row0 = (long)b * T_launch;
    g.T = T_launch;
  }
  g.hrow = g.row0 * H + (long)h * g.T;
  g.T4 = (uint32_t)((T_launch + 3) >> 2);
  return g;
}

// key-bias image of the sequence + one "holds a padded key" flag per 64-key tile (both read by every wave all along)
template <int NW>
__device__ __forceinline__ void stage_kbias(const float* __restrict__ kb, int T, float* ldsKb, int* ldsPad, int tid, int lane, int wave) {
  const int ntile = (T + KT - 1) / KT;
  for (int j = wave; j < ntile; j += NW) {
    const int k = j * KT + lane;
    const float v = k < T ? kb[k] : 0.f;
    ldsKb[k] = v;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(v != 0.f);      // bit t: 32-position half t holds a padded position
    if (lane == 0) ldsPad[j] = ((uint32_t)bal ? 1 : 0) | ((uint32_t)(bal >> 32) ? 2 : 0);
  }
}

// =====================================================================================================================
// forward: NW waves x 32 queries per block
// =====================================================================================================================
template <int HD, bool DROP, int NW, int NST>
__device__ __forceinline__ void fwd_tile(char* ring, const float* ldsKb, const int* ldsPad, const bf16_t* __restrict__ qkv,
                                         const int* __restrict__ kstart, bf16_t* __restrict__ out, float* __restrict__ lse,
                                         int T, int H, float scale, uint32_t drop_thr, uint32_t drop_key, float drop_scale,
                                         const int tile, const SeqG g, const int h) {
  using C = SC<HD>;
  constexpr int QB = 32 * NW, PPW = C::PIECES / NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = H * HD;
  const long ld = 3L * d;
  const unsigned ldb = (unsigned)(ld * 2);
  const bf16_t* qbase = qkv + g.row0 * ld + h * HD;
  const char* kbase = reinterpret_cast<const char*>(qbase + d);
  const char* vbase = reinterpret_cast<const char*>(qbase + 2 * d);
  const unsigned ring_lds = lds_addr(ring);

  const int q0 = tile * QB, qw0 = q0 + wave * 32;
  const int q = qw0 + (lane & 31);
  const bool qvalid = q < T;
  bf16x8_v qf[C::KS];
  row_frags<HD>(qbase + (long)q * ld, qvalid, lane, qf);

  // a block that holds a masked (padded) query row visits every key: those rows see all of them (finite -1e4 scores)
  int full = 0;
  for (int j = q0 / KT; j < min((q0 + QB + KT - 1) / KT, (T + KT - 1) / KT); ++j) full |= ldsPad[j];      // block-uniform
  const bool wave_full = __builtin_amdgcn_ballot_w64(qvalid && ldsKb[min(q, T - 1)] != 0.f) != 0;
  const int qmax = min(q0 + QB - 1, T - 1);
  const int ntile = (T + KT - 1) / KT;
  const int kt_end = full ? ntile : qmax / KT + 1;
  const int kt_beg = full ? 0 : (kstart ? kstart[g.b] / KT : 0);

  // per-lane addressing, computed once and kept small (the loop lives at the 256-register limit of two waves per SIMD):
  //   K fragments (natural reads): ONE base, k-step ks is base ^ (ks << 5) -- the swizzle XORs the chunk index and 2 ks + h differs
  //     from h in exactly those bits -- sub-tile by immediate;
  //   V^T fragments (transposed reads): one base per 32-wide head-dim block, key step and lo / hi by immediates;
  //   DMA: one per-lane offset per operand; piece p adds a wave-uniform row offset (folded into the SGPR base) and, for K, XORs
  //     its two low chunk bits with p & 3 (rows 4 p .. 4 p + 3: the part of the swizzle that depends on the piece)
  const unsigned kfrag0 = ring_lds + nat_off<HD, 0>(0, lane);
  unsigned voff[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i) voff[i] = ring_lds + C::TILE + tr_off<HD, 1>(i, 0, 0, lane);
  constexpr int RPP = 1024 / C::ROWB;                    // rows per DMA piece
  const int prow = lane / C::CH;                          // row of this lane inside a piece
  const unsigned dmaK0 = (unsigned)prow * ldb + (unsigned)(((lane % C::CH) ^ swz<HD, 0>(prow)) << 4);      // piece 0
  const unsigned dmaV0 = (unsigned)prow * ldb + (unsigned)(((lane % C::CH) ^ swz<HD, 1>(prow)) << 4);      // every piece

  f32x16 o[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float scale2 = scale * LOG2E;

  auto issue = [&](int kt, int stage) {       // tiles past the end re-fetch the last one: the counted waits stay uniform
    const int ktc = min(kt, kt_end - 1);
    const unsigned img = ring_lds + stage * 2 * C::TILE;
    const bool ragged = (ktc + 1) * KT > T;                // last tile of a sequence whose length is not a multiple of 64
    const char* kt_k = kbase + (long)ktc * KT * ldb;
    const char* kt_v = vbase + (long)ktc * KT * ldb;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave * PPW + i;                          // wave-uniform
      const unsigned pxor = (unsigned)((swz<HD, 0>(p * RPP) & (C::CH - 1)) << 4);
      unsigned offK, offV;
      if (!ragged) {
        offK = (dmaK0 ^ pxor) + (unsigned)(p * RPP) * ldb;
        offV = dmaV0 + (unsigned)(p * RPP) * ldb;
      } else {                                               // rows past T - 1 re-read row T - 1 (finite values under zero weights)
        const unsigned row = (unsigned)min(p * RPP + prow, T - 1 - ktc * KT);
        offK = row * ldb + (((dmaK0 ^ pxor) - (unsigned)prow * ldb) & 0xffu);
        offV = row * ldb + ((dmaV0 - (unsigned)prow * ldb) & 0xffu);
      }
      glds16_s(uniform_ptr(kt_k), offK, __builtin_amdgcn_readfirstlane(img + p * 1024));
      glds16_s(uniform_ptr(kt_v), offV, __builtin_amdgcn_readfirstlane(img + C::TILE + p * 1024));
    }
  };
  if (kt_beg < kt_end) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(kt_beg + s, s);
  }
  int stage = 0;
  // barrier + refill of one tile step; returns the LDS offset of the stage that holds tile kt
  auto tile_sync = [&](int kt) -> unsigned {
    if (!(NEKO_AS_DIAG & 1)) wait_vm<(NST - 2) * 2 * PPW>();          // this wave's pieces of tile kt have landed
    if (!(NEKO_AS_DIAG & 8)) __syncthreads();                         // ... everyone's have, and everyone is done with the stage refilled next
    if (!(NEKO_AS_DIAG & 1)) {
      int st_next = stage + NST - 1;
      if (st_next >= NST) st_next -= NST;
      issue(kt + NST - 1, st_next);
    }
    const unsigned sb = (unsigned)stage * 2u * C::TILE;
    if (++stage == NST) stage = 0;
    return sb;
  };
  // the common tile: every key visible to every query of the wave, nothing padded
  auto is_fast = [&](int kt) { return !(NEKO_AS_DIAG & 16) && (kt * KT + KT - 1 <= qw0) && ldsPad[kt] == 0 && (kt * KT + KT <= T); };
  auto fast_tile = [&](int kt) __attribute__((always_inline)) {
    const int k0 = kt * KT;
    const unsigned sb = tile_sync(kt);
      // ---- the common tile: every key visible to every query of the wave, nothing padded.  Both sub-tiles in one software
      // pipeline, MFMA groups and the softmax arithmetic of the OTHER sub-tile issued alternately (a wave issues in order: behind
      // a chain of dependent MFMAs its VALU work would wait for the whole chain, and two waves of a SIMD that run the same
      // phases in lockstep -- one barrier per tile keeps them there -- only add their times up):
      //   K reads | S0 chain + V reads(0) | max(0) | S1 chain || exp/sum/pack(0) | max(1) | P.V(0) || exp/sum/pack(1) + V reads(1) | P.V(1)
      // Register budget (two waves per SIMD: 256): o 64, qf 32, s0 + s1 32, ONE set of K fragments (32: sub-tile 0, reloaded for
      // sub-tile 1 behind the S0 chain) and ONE set of V^T fragments in two halves (2 x 16: each half reloaded for sub-tile 1 as
      // soon as the MFMAs of sub-tile 0 that read it are issued).  The sched_barriers fence the phases: left alone, hipcc hoists
      // every fragment read of the tile to the top and spills.
      bf16x8_v kfr[C::KS], vA[C::IB], vB[C::IB];
      unsigned kbase[C::KS];                    // the stage's K fragment bases, once per tile: both sub-tiles read base + immediate
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) kbase[ks] = (kfrag0 + sb) ^ (unsigned)(ks << 5);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) kfr[ks] = lds_read16(kbase[ks]);
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], s0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < C::IB; ++i) {
        const unsigned a0 = voff[i] + sb, a1 = voff[i] + sb + 16 * C::ROWB;
        vA[i] = lds_read_tr(a0, a0 + 8 * C::ROWB);
        vB[i] = lds_read_tr(a1, a1 + 8 * C::ROWB);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) kfr[ks] = lds_read16(kbase[ks] + 32 * C::ROWB);
      __builtin_amdgcn_sched_barrier(0);
      // row maximum of a sub-tile and the (rare) move of the lazy reference
      auto move_reference = [&](const f32x16& st) {
        float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
        mx = pair_max(fmaxf(mx, st[15]));
        const float cand = mx * scale2;
        if (__builtin_amdgcn_ballot_w64(cand > m_run + 8.0f) != 0) {
          const float m_new = fmaxf(m_run, cand);
          const float alpha = exp2_fast(m_run - m_new);
          l_run *= alpha;
#pragma unroll
          for (int i = 0; i < C::IB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
          m_run = m_new;
        }
      };
      // probabilities of a sub-tile (in place), row sum, dropout
      auto probabilities = [&](f32x16& st, int t) {
        f32x2_v ps = {0.f, 0.f};
        const f32x2_v sc2 = {scale2, scale2}, nm2 = {-m_run, -m_run};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_v a = __builtin_elementwise_fma((f32x2_v){st[r], st[r + 1]}, sc2, nm2);
          const f32x2_v e = {exp2_fast(a.x), exp2_fast(a.y)};
          st[r] = e.x;
          st[r + 1] = e.y;
          ps += e;
        }
        l_run += ps.x + ps.y;
        if (DROP) {
          const uint32_t g0 = ((uint32_t)g.hrow + (uint32_t)q) * g.T4 +
                              (uint32_t)((k0 + t * 32 + 4 * (lane >> 5)) >> 2);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t w = drop_word(g0 + 2 * j, drop_key);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[4 * j + e] = drop_byte_keep(w, e, drop_thr) ? st[4 * j + e] : 0.f;
          }
        }
      };
      move_reference(s0);
      // S1 chain || probabilities of sub-tile 0
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], s1, 0, 0, 0);
      probabilities(s0, 0);
      const bf16x8_v p00 = frag_from_acc(s0, 0), p01 = frag_from_acc(s0, 1);
      if (!DROP) {
#pragma unroll
        for (int g = 0; g < C::KS; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 24 / C::KS, 0);
          __builtin_amdgcn_sched_group_barrier(0x400, 16 / C::KS, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      move_reference(s1);
      // P.V of sub-tile 0 || probabilities of sub-tile 1
#pragma unroll
      for (int i = 0; i < C::IB; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vA[i], p00, o[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < C::IB; ++i) {
        const unsigned a0 = voff[i] + sb + 32 * C::ROWB;
        vA[i] = lds_read_tr(a0, a0 + 8 * C::ROWB);
      }
#pragma unroll
      for (int i = 0; i < C::IB; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vB[i], p01, o[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < C::IB; ++i) {
        const unsigned a1 = voff[i] + sb + 48 * C::ROWB;
        vB[i] = lds_read_tr(a1, a1 + 8 * C::ROWB);
      }
      probabilities(s1, 1);
      const bf16x8_v p10 = frag_from_acc(s1, 0), p11 = frag_from_acc(s1, 1);
      if (!DROP) {
#pragma unroll
        for (int g = 0; g < 2 * C::IB; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 24 / (2 * C::IB), 0);
          __builtin_amdgcn_sched_group_barrier(0x400, 16 / (2 * C::IB), 0);
          if (g == C::IB - 1 || g == 2 * C::IB - 1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * C::IB, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < C::IB; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vA[i], p10, o[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < C::IB; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vB[i], p11, o[i], 0, 0, 0);
  };
  // the rare tile (diagonal, padded keys, ragged end), one sub-tile at a time
  auto slow_tile = [&](int kt) __attribute__((always_inline)) {
    const int k0 = kt * KT;
    const unsigned sb = tile_sync(kt);
    const bool has_pad = ldsPad[kt] != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!wave_full && k0 + t * 32 > qw0 + 31) continue;     // wave-uniform: nothing visible, no masked row
      // the rare tile (diagonal, padded keys, ragged end), one sub-tile at a time.  Its per-lane constants are derived from an
      // opaque copy of the lane id INSIDE the branch: derived from `lane` they are loop invariants, hipcc hoists them out of the
      // tile loop, they stay live across the common path above and the kernel spills
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      const int q_s = qw0 + (lane_s & 31);
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      {
        // every fragment of the sub-tile is requested before the first MFMA waits for one (left to itself hipcc reuses one
        // register quad: read, wait, MFMA, read, ... -- an LDS round trip in front of every MFMA)
        bf16x8_v kfr[C::KS];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) kfr[ks] = lds_read16((kfrag0 + sb + t * 32 * C::ROWB) ^ (unsigned)(ks << 5));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], st, 0, 0, 0);
      }
      // the V^T fragments of this sub-tile do not depend on the softmax: requested now, they arrive under its arithmetic
      bf16x8_v vfr[2][C::IB];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < C::IB; ++i) {
          const unsigned a = voff[i] + sb + (2 * t + h2) * 16 * C::ROWB;
          vfr[h2][i] = lds_read_tr(a, a + 8 * C::ROWB);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (!(NEKO_AS_DIAG & 2)) {
      const bool interior = (k0 + t * 32 + 31 <= qw0) && !has_pad && (k0 + KT <= T);
      if (!interior) {
        const int lim_causal = q_s - k0 - t * 32 - 4 * (lane_s >> 5);        // key <= q  <=>  c(r) <= lim_causal
        const int lim_len = T - 1 - k0 - t * 32 - 4 * (lane_s >> 5);       // key <  T  <=>  c(r) <= lim_len
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          float v = (c <= lim_causal) ? st[r] * scale2 : MASK_VAL * LOG2E;
          v = fmaf(ldsKb[k0 + t * 32 + c + 4 * (lane_s >> 5)], LOG2E, v);       // the image is padded to whole tiles
          st[r] = (c <= lim_len) ? v : -INFINITY;
        }
      }
      float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
      mx = fmaxf(mx, st[15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float sc = interior ? scale2 : 1.0f;      // interior scores are still unscaled
      // lazy rescale (attention.hip): the reference maximum moves only when some row's maximum grew by more than 2^8
      const float cand = mx * sc;
      if (__builtin_amdgcn_ballot_w64(cand > m_run + 8.0f) != 0) {
        const float m_new = fmaxf(m_run, cand);
        const float alpha = exp2_fast(m_run - m_new);   // 2^(-inf) = 0 on the first tile
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < C::IB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        m_run = m_new;
      }
      f32x2_v ps = {0.f, 0.f};
      const f32x2_v sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {       // the exponent arguments and the row sum on the packed fp32 pipe
        const f32x2_v a = __builtin_elementwise_fma((f32x2_v){st[r], st[r + 1]}, sc2, nm2);
        const f32x2_v e = {exp2_fast(a.x), exp2_fast(a.y)};
        st[r] = e.x;
        st[r + 1] = e.y;
        ps += e;
      }
      l_run += ps.x + ps.y;
      }
      if (DROP) {   // attn_dropout on the probabilities (trajectory_gpt2.py:179): the normaliser stays undropped
        const uint32_t g0 = ((uint32_t)g.hrow + (uint32_t)q_s) * g.T4 +
                            (uint32_t)((k0 + t * 32 + 4 * (lane_s >> 5)) >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // registers 4j..4j+3 = keys +8j .. +8j+3: one word
          const uint32_t w = drop_word(g0 + 2 * j, drop_key);
#pragma unroll
          for (int e = 0; e < 4; ++e) st[4 * j + e] = drop_byte_keep(w, e, drop_thr) ? st[4 * j + e] : 0.f;   // survivor scale: in the final 1/l
        }
      }
      if (!(NEKO_AS_DIAG & 4))
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const bf16x8_v pf = frag_from_acc(st, h2);
#pragma unroll
        for (int i = 0; i < C::IB; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[h2][i], pf, o[i], 0, 0, 0);
      }
    }
  };
  // Three loops instead of one loop with a branch: [padded prefix] [common tiles] [diagonal and whatever follows].  In one loop
  // the per-lane constants of the rare path are hoisted, stay live across the common path and the kernel spills (the common
  // path alone needs 220 registers, the rare one 200, both in one loop body more than 256).  Every wave passes one barrier per
  // tile whichever loop it is in.
  int kt = kt_beg;
  for (; kt < kt_end && !is_fast(kt); ++kt) slow_tile(kt);
  for (; kt < kt_end && is_fast(kt); ++kt) fast_tile(kt);
  for (; kt < kt_end; ++kt) slow_tile(kt);
  wait_vm<0>();

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qvalid) {
    const float inv = (DROP ? drop_scale : 1.0f) / l_tot;
    bf16_t* orow = out + (g.row0 + q) * d + h * HD;
#pragma unroll
    for (int i = 0; i < C::IB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
        pk.y = pack_bf16x2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
      }
    if (lane < 32) lse[g.hrow + q] = m_run * LN2 + __logf(l_tot);
  }
}

template <int HD, bool DROP, int NW, int NST>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_stream_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ kbias,
                                                                     const int* __restrict__ kstart, bf16_t* __restrict__ out,
                                                                     float* __restrict__ lse, int B, int T_launch, int H, float scale,
                                                                     uint32_t drop_thr, uint32_t drop_key, float drop_scale,
                                                                     const int* __restrict__ seq_off) {
  extern __shared__ __attribute__((aligned(1024))) char dyn_smem[];
  if (DROP) drop_key += neko_drop_salt();
  const int tid = threadIdx.x;
  // one workgroup = a heavy and a light causal tile (G-1-p and p): every workgroup walks the same number of key tiles.  T_launch is
  // the (longest) sequence length the grid and the LDS block were sized for; a packed sequence shorter than that owns fewer tiles
  const int G = (T_launch + 32 * NW - 1) / (32 * NW);
  int bh, p;
  xcd_remap((G + 1) / 2, B * H, bh, p);
  const int b = bh / H, h = bh - b * H;
  const SeqG g = seq_geom(b, h, H, T_launch, seq_off);
  const int T = g.T;
  char* ring = dyn_smem;                                                        // NST stages of [K tile | V tile]
  float* ldsKb = reinterpret_cast<float*>(dyn_smem + NST * 2 * SC<HD>::TILE);   // [round_up(T, 64)]
  int* ldsPad = reinterpret_cast<int*>(ldsKb + (T + KT - 1) / KT * KT);         // [ceil(T / 64)]
  const int first = G - 1 - p, second = p;
  const bool do_first = first * 32 * NW < T, do_second = second != first && second * 32 * NW < T;
  if (!do_first && !do_second) return;
  stage_kbias<NW>(kbias + g.row0, T, ldsKb, ldsPad, tid, tid & 63, tid >> 6);
  __syncthreads();
  if (do_first) fwd_tile<HD, DROP, NW, NST>(ring, ldsKb, ldsPad, qkv, kstart, out, lse, T, H, scale, drop_thr, drop_key, drop_scale, first, g, h);
  if (do_second) {
    __syncthreads();
    fwd_tile<HD, DROP, NW, NST>(ring, ldsKb, ldsPad, qkv, kstart, out, lse, T, H, scale, drop_thr, drop_key, drop_scale, second, g, h);
  }
}

// =====================================================================================================================
// backward.  D = sum(dO * O) / s and nothing else comes from the prep kernel of attention.hip; lse and D of the sequence
// (dK/dV) and the key bias live in LDS for the whole workgroup.
// =====================================================================================================================
// one operand stream: 64-row tiles of a [rows, ld] bf16 matrix, read both ways (swizzle kind 0)
template <int HD, int NW>
struct TileStream {
  const char* base;       // (sequence, head) origin of the operand
  unsigned ldb, lane_off; // row stride in bytes; per-lane offset of DMA piece 0
  int prow;
  __device__ __forceinline__ void init(const void* b, long ld_elems, int lane) {
    base = reinterpret_cast<const char*>(b);
    ldb = (unsigned)(ld_elems * 2);
    prow = lane / SC<HD>::CH;
    lane_off = (unsigned)prow * ldb + (unsigned)(((lane % SC<HD>::CH) ^ swz<HD, 0>(prow)) << 4);
  }
  // this wave's pieces of the tile whose first row is row0 (rows past nrows - 1 re-read the last row)
  __device__ __forceinline__ void issue(int row0, int nrows, unsigned img_lds, int wave) const {
    constexpr int PPW = SC<HD>::PIECES / NW, RPP = 1024 / SC<HD>::ROWB;
    const char* tb = base + (long)row0 * ldb;
    const bool ragged = row0 + KT > nrows;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave * PPW + i;
      const unsigned pxor = (unsigned)((swz<HD, 0>(p * RPP) & (SC<HD>::CH - 1)) << 4);
      unsigned off;
      if (!ragged) off = (lane_off ^ pxor) + (unsigned)(p * RPP) * ldb;
      else off = (unsigned)min(p * RPP + prow, nrows - 1 - row0) * ldb + (((lane_off ^ pxor) - (unsigned)prow * ldb) & 0xffu);
      glds16_s(uniform_ptr(tb), off, __builtin_amdgcn_readfirstlane(img_lds + p * 1024));
    }
  }
};
// transposed fragment of a kind-0 tile: lo address from tr_off<HD, 0>(i, 0, 0, lane) + 16 s rows; the hi read is 8 rows further,
// where the swizzle differs in chunk bit 1
__device__ __forceinline__ bf16x8_v lds_read_tr0(unsigned lo, int rowb) { return lds_read_tr(lo, (lo ^ 32u) + 8u * rowb); }

// ---- dQ: lanes own queries (the forward's geometry) -------------------------------------------------------------------------
template <int HD, bool DROP, int NW, int NST>
__device__ __forceinline__ void dq_tile(char* ring, const float* ldsKb, const int* ldsPad, const bf16_t* __restrict__ qkv,
                                        const bf16_t* __restrict__ dout, const int* __restrict__ kstart,
                                        const float* __restrict__ lse, const float* __restrict__ Dv, bf16_t* __restrict__ dqkv,
                                        int T, int H, float scale, uint32_t drop_thr, uint32_t drop_key, float drop_scale,
                                        const int tile, const SeqG g, const int h) {
  using C = SC<HD>;
  constexpr int QB = 32 * NW, PPW = C::PIECES / NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = H * HD;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + g.row0 * ld + h * HD;
  const unsigned ring_lds = lds_addr(ring);
  TileStream<HD, NW> srcK, srcV;
  srcK.init(qbase + d, ld, lane);
  srcV.init(qbase + 2 * d, ld, lane);

  const int q0 = tile * QB, qw0 = q0 + wave * 32;
  const int q = qw0 + (lane & 31);
  const bool qvalid = q < T;
  bf16x8_v qf[C::KS], dof[C::KS];
  row_frags<HD>(qbase + (long)q * ld, qvalid, lane, qf);
  row_frags<HD>(dout + (g.row0 + q) * d + h * HD, qvalid, lane, dof);
  const float my_lse = (qvalid ? lse[g.hrow + q] : 0.f) * LOG2E;
  const float my_D = qvalid ? Dv[g.hrow + q] : 0.f;
  const float scale2 = scale * LOG2E;

  int full = 0;
  for (int j = q0 / KT; j < min((q0 + QB + KT - 1) / KT, (T + KT - 1) / KT); ++j) full |= ldsPad[j];      // block-uniform
  const bool wave_full = __builtin_amdgcn_ballot_w64(qvalid && ldsKb[min(q, T - 1)] != 0.f) != 0;
  const int qmax = min(q0 + QB - 1, T - 1);
  const int ntile = (T + KT - 1) / KT;
  const int kt_end = full ? ntile : qmax / KT + 1;
  const int kt_beg = full ? 0 : (kstart ? kstart[g.b] / KT : 0);

  const unsigned nat0 = ring_lds + nat_off<HD, 0>(0, lane);      // K (and, + TILE, V) natural fragment of k-step 0; k-step ks: ^ (ks << 5)
  unsigned ktr[C::IB], ktr_hi[C::IB];                            // K^T fragment of head-dim block i, key step 0: lo / hi read
#pragma unroll
  for (int i = 0; i < C::IB; ++i) { ktr[i] = ring_lds + tr_off<HD, 0>(i, 0, 0, lane); ktr_hi[i] = ring_lds + tr_off<HD, 0>(i, 0, 1, lane); }

  f32x16 dq[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  auto issue = [&](int kt, int stage) {
    const int k0 = min(kt, kt_end - 1) * KT;
    srcK.issue(k0, T, ring_lds + stage * 2 * C::TILE, wave);
    srcV.issue(k0, T, ring_lds + stage * 2 * C::TILE + C::TILE, wave);
  };
  if (kt_beg < kt_end) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(kt_beg + s, s);
  }
  int stage = 0;
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const int k0 = kt * KT;
    wait_vm<(NST - 2) * 2 * PPW>();
    __syncthreads();
    {
      int st_next = stage + NST - 1;
      if (st_next >= NST) st_next -= NST;
      issue(kt + NST - 1, st_next);
    }
    const unsigned sb = (unsigned)stage * 2u * C::TILE;
    const bool has_pad = ldsPad[kt] != 0;
    // the stage's fragment bases, once per tile: every read below is base + immediate (computed per read, the XOR / add pairs were
    // a third of the kernel's VALU instructions)
    unsigned nb[C::KS], tlo[C::IB], thi[C::IB];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) nb[ks] = (nat0 + sb) ^ (unsigned)(ks << 5);
#pragma unroll
    for (int i = 0; i < C::IB; ++i) { tlo[i] = ktr[i] + sb; thi[i] = ktr_hi[i] + sb; }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!wave_full && k0 + t * 32 > qw0 + 31) continue;     // wave-uniform: nothing visible, no masked row
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
      {
        // the two score chains one after the other, V fragments into the registers the K fragments leave (two waves per SIMD: 256
        // registers; both fragment sets live at once need 300 and one wave per SIMD)
        bf16x8_v fr[C::KS];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) fr[ks] = lds_read16(nb[ks] + (t * 32 * C::ROWB));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks], qf[ks], st, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) fr[ks] = lds_read16(nb[ks] + (C::TILE + t * 32 * C::ROWB));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks], dof[ks], dpt, 0, 0, 0);
      }
      // K^T fragments of the first key half of the sub-tile: requested now, they arrive under the elementwise work (the second
      // half is requested behind the first half's MFMAs, into the same registers)
      bf16x8_v ktf[C::IB];
#pragma unroll
      for (int i = 0; i < C::IB; ++i) ktf[i] = lds_read_tr(tlo[i] + ((2 * t) * 16 * C::ROWB), thi[i] + ((2 * t) * 16 * C::ROWB));
      __builtin_amdgcn_sched_barrier(0);
      // dS^T = P^T o (keep * dP^T - D / s); zero where the score was REPLACED by the causal constant
      const uint32_t g0 = ((uint32_t)g.hrow + (uint32_t)q) * g.T4 +
                          (uint32_t)((k0 + t * 32 + 4 * (lane >> 5)) >> 2);
      const bool interior = (k0 + t * 32 + 31 <= qw0) && !has_pad && (k0 + KT <= T);
      if (interior) {
        const float nlse = -my_lse;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = DROP ? drop_word(g0 + 2 * j, drop_key) : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * j + e;
            const float pv = exp2_fast(fmaf(st[r], scale2, nlse));
            float dpe = dpt[r];
            if (DROP) dpe = drop_byte_keep(w, e, drop_thr) ? dpe : 0.f;
            st[r] = pv * (dpe - my_D);
          }
        }
      } else {
        const int lim_causal = q - k0 - t * 32 - 4 * (lane >> 5);
        const int lim_len = T - 1 - k0 - t * 32 - 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = DROP ? drop_word(g0 + 2 * j, drop_key) : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * j + e;
            const int c = (r & 3) + 8 * (r >> 2);
            const bool causal_ok = c <= lim_causal;
            const float sv = fmaf(ldsKb[k0 + t * 32 + c + 4 * (lane >> 5)], LOG2E, causal_ok ? st[r] * scale2 : MASK_VAL * LOG2E);
            const float pv = exp2_fast((c <= lim_len) ? sv - my_lse : -INFINITY);      // select, not a branch: 2^-inf = 0
            float dpe = dpt[r];
            if (DROP) dpe = drop_byte_keep(w, e, drop_thr) ? dpe : 0.f;
            st[r] = causal_ok ? pv * (dpe - my_D) : 0.f;
          }
        }
      }
      // dQ^T += K^T . dS^T
      {
        const bf16x8_v df0 = frag_from_acc(st, 0), df1 = frag_from_acc(st, 1);
#pragma unroll
        for (int i = 0; i < C::IB; ++i) dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[i], df0, dq[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < C::IB; ++i) ktf[i] = lds_read_tr(tlo[i] + ((2 * t + 1) * 16 * C::ROWB), thi[i] + ((2 * t + 1) * 16 * C::ROWB));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < C::IB; ++i) dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[i], df1, dq[i], 0, 0, 0);
      }
    }
    if (++stage == NST) stage = 0;
  }
  wait_vm<0>();

  // dS was formed as P o (keep * dP - D / s): the dropout survivor scale s multiplies the result once, here
  const float qs = scale * (DROP ? drop_scale : 1.0f);
  if (qvalid) {
    bf16_t* orow = dqkv + (g.row0 + q) * ld + h * HD;
#pragma unroll
    for (int i = 0; i < C::IB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dq[i][4 * g + 0] * qs, dq[i][4 * g + 1] * qs);
        pk.y = pack_bf16x2(dq[i][4 * g + 2] * qs, dq[i][4 * g + 3] * qs);
        *reinterpret_cast<uint2*>(orow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
      }
  }
}

template <int HD, bool DROP, int NW, int NST>
// two waves per SIMD where the body fits 256 registers: the dropout variant at hd = 128 does (5 spilled registers, 273 -> 254 us for the
// whole backward), the plain one does not (118 spilled: 96 -> 158 us) and keeps one wave per SIMD
__global__ __launch_bounds__(64 * NW, (HD <= 64 || DROP) ? 2 : 1) void attn_dq_stream_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                    const float* __restrict__ kbias, const int* __restrict__ kstart,
                                                                    const float* __restrict__ lse, const float* __restrict__ Dv,
                                                                    bf16_t* __restrict__ dqkv, int B, int T_launch, int H, float scale,
                                                                    uint32_t drop_thr, uint32_t drop_key, float drop_scale,
                                                                    const int* __restrict__ seq_off) {
  extern __shared__ __attribute__((aligned(1024))) char dyn_smem[];
  if (DROP) drop_key += neko_drop_salt();
  const int tid = threadIdx.x;
  const int G = (T_launch + 32 * NW - 1) / (32 * NW);
  int bh, p;
  xcd_remap((G + 1) / 2, B * H, bh, p);
  const int b = bh / H, h = bh - b * H;
  const SeqG g = seq_geom(b, h, H, T_launch, seq_off);
  const int T = g.T;
  char* ring = dyn_smem;
  float* ldsKb = reinterpret_cast<float*>(dyn_smem + NST * 2 * SC<HD>::TILE);
  int* ldsPad = reinterpret_cast<int*>(ldsKb + (T + KT - 1) / KT * KT);
  const int first = G - 1 - p, second = p;
  const bool do_first = first * 32 * NW < T, do_second = second != first && second * 32 * NW < T;
  if (!do_first && !do_second) return;
  stage_kbias<NW>(kbias + g.row0, T, ldsKb, ldsPad, tid, tid & 63, tid >> 6);
  __syncthreads();
  if (do_first) dq_tile<HD, DROP, NW, NST>(ring, ldsKb, ldsPad, qkv, dout, kstart, lse, Dv, dqkv, T, H, scale, drop_thr, drop_key, drop_scale, first, g, h);
  if (do_second) {
    __syncthreads();
    dq_tile<HD, DROP, NW, NST>(ring, ldsKb, ldsPad, qkv, dout, kstart, lse, Dv, dqkv, T, H, scale, drop_thr, drop_key, drop_scale, second, g, h);
  }
}

// ---- dK / dV: lanes own keys, 64-query tiles of Q and dO stream through the ring -------------------------------------------
template <int HD, bool DROP, int NW, int NST>
__device__ __forceinline__ void dkv_tile(char* ring, const float* ldsKb, const int* ldsPad, const float* ldsLse, const float* ldsD,
                                         const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                         int T, int H, float scale, uint32_t drop_thr, uint32_t drop_key, float drop_scale,
                                         const int tile, const SeqG g, const int h) {
  using C = SC<HD>;
  constexpr int KB = 32 * NW, PPW = C::PIECES / NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = H * HD;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + g.row0 * ld + h * HD;
  const unsigned ring_lds = lds_addr(ring);
  TileStream<HD, NW> srcQ, srcO;
  srcQ.init(qbase, ld, lane);
  srcO.init(dout + g.row0 * d + h * HD, d, lane);

  const int k0 = tile * KB, kw0 = k0 + wave * 32;
  const int key = kw0 + (lane & 31);
  const bool kvalid = key < T;
  const float my_kb = ldsKb[min(key, T - 1)] * LOG2E * (kvalid ? 1.f : 0.f);
  const float scale2 = scale * LOG2E;
  bf16x8_v kf[C::KS], vf[C::KS];
  row_frags<HD>(qbase + d + (long)key * ld, kvalid, lane, kf);
  row_frags<HD>(qbase + 2 * d + (long)key * ld, kvalid, lane, vf);

  const unsigned nat0 = ring_lds + nat_off<HD, 0>(0, lane);
  unsigned qtr[C::IB], qtr_hi[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i) { qtr[i] = ring_lds + tr_off<HD, 0>(i, 0, 0, lane); qtr_hi[i] = ring_lds + tr_off<HD, 0>(i, 0, 1, lane); }

  f32x16 dk[C::IB], dv[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  const int nqt = (T + KT - 1) / KT;
  const int qt_causal = k0 / KT;      // first query tile that sees this key block causally; earlier ones only through masked query rows
  auto next_tile = [&](int t) {       // next query tile >= t that has to be visited (block-uniform)
    while (t < nqt && t < qt_causal && !ldsPad[t]) ++t;
    return t;
  };
  // the visiting order is data dependent: the tiles in flight are kept in a small queue
  int tq[NST];
  tq[0] = next_tile(0);
#pragma unroll
  for (int s = 1; s < NST; ++s) tq[s] = next_tile(min(tq[s - 1] + 1, nqt));
  auto issue = [&](int qt, int stage) {
    const int q0 = min(qt, nqt - 1) * KT;
    srcQ.issue(q0, T, ring_lds + stage * 2 * C::TILE, wave);
    srcO.issue(q0, T, ring_lds + stage * 2 * C::TILE + C::TILE, wave);
  };
  if (tq[0] < nqt) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(tq[s], s);
  }
  int stage = 0;
  // barrier + refill of one tile step; returns the LDS offset of the stage that holds the current tile tq[0]
  auto tile_sync = [&]() -> unsigned {
    wait_vm<(NST - 2) * 2 * PPW>();
    __syncthreads();
    {
      int st_next = stage + NST - 1;
      if (st_next >= NST) st_next -= NST;
      issue(tq[NST - 1], st_next);
    }
    const unsigned sb = (unsigned)stage * 2u * C::TILE;
    if (++stage == NST) stage = 0;
    return sb;
  };
  auto advance = [&]() {
#pragma unroll
    for (int s = 0; s + 1 < NST; ++s) tq[s] = tq[s + 1];
    tq[NST - 1] = next_tile(min(tq[NST - 1] + 1, nqt));
  };
  const bool keys_plain = (kw0 + 31 < T) && __builtin_amdgcn_ballot_w64(my_kb != 0.f) == 0;      // no padded / invalid key in this wave
  // the common tile: every query of the tile sees every key of this wave, nothing padded or ragged
  auto is_fast = [&](int qt) { return !(NEKO_AS_DIAG & 16) && keys_plain && (qt * KT >= kw0 + 31) && (qt * KT + KT - 1 < T); };
  // elementwise part of a common sub-tile: st := dropped P, dpt := dS
  auto elementwise_plain = [&](f32x16& st, f32x16& dpt, int q0, int t) {
    uint32_t mine[4] = {0u, 0u, 0u, 0u};
    const int ksh = 8 * (lane & 3);
    if (DROP) {
      const uint32_t T4 = g.T4;
      const uint32_t gq = ((uint32_t)g.hrow + (uint32_t)(q0 + t * 32 + 4 * (lane >> 5) + (lane & 3))) * T4 +
                          (uint32_t)(key >> 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[j] = drop_word(gq + (uint32_t)(8 * j) * T4, drop_key);
    }
    // rows (r & 3) + 8 (r >> 2) + 4 h: four consecutive floats per register quad -> one 16-B read each
    float lq[16], dq_[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(ldsLse + q0 + t * 32 + 4 * (lane >> 5) + 8 * j);
      const float4 g = *reinterpret_cast<const float4*>(ldsD + q0 + t * 32 + 4 * (lane >> 5) + 8 * j);
      lq[4 * j] = a.x; lq[4 * j + 1] = a.y; lq[4 * j + 2] = a.z; lq[4 * j + 3] = a.w;
      dq_[4 * j] = g.x; dq_[4 * j + 1] = g.y; dq_[4 * j + 2] = g.z; dq_[4 * j + 3] = g.w;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = r;
      const float pv = exp2_fast(fmaf(st[r], scale2, -lq[c]));
      float pd = pv, dpe = dpt[r];
      if (DROP) {
        const bool keep = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr;
        pd = keep ? pv : 0.f;
        dpe = keep ? dpe : 0.f;
      }
      st[r] = pd;
      dpt[r] = pv * (dpe - dq_[c]);
    }
  };
  // Both sub-tiles in one software pipeline (this kernel runs one wave per SIMD: nothing else covers a wave's own chains):
  //   reads(0) | S0, dP0 chains | S1, dP1 chains || elementwise(0) | dV, dK += (0) || elementwise(1) | dV, dK += (1)
  auto fast_tile = [&]() __attribute__((always_inline)) {
    const int q0 = tq[0] * KT;
    const unsigned sb = tile_sync();
    unsigned nb[C::KS], tlo[C::IB], thi[C::IB];      // the stage's fragment bases: every read below is base + immediate
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) nb[ks] = (nat0 + sb) ^ (unsigned)(ks << 5);
#pragma unroll
    for (int i = 0; i < C::IB; ++i) { tlo[i] = qtr[i] + sb; thi[i] = qtr_hi[i] + sb; }
    f32x16 s0, p0, s1, p1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; p0[r] = 0.f; s1[r] = 0.f; p1[r] = 0.f; }
    {
      bf16x8_v qfr[C::KS], ofr[C::KS];
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        qfr[ks] = lds_read16(nb[ks]);
        ofr[ks] = lds_read16(nb[ks] + (C::TILE));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], s0, 0, 0, 0);
        p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr[ks], vf[ks], p0, 0, 0, 0);
      }
    }
    bf16x8_v qtf[2][C::IB], otf[2][C::IB];
    {
      bf16x8_v qfr[C::KS], ofr[C::KS];
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        qfr[ks] = lds_read16(nb[ks] + (32 * C::ROWB));
        ofr[ks] = lds_read16(nb[ks] + (C::TILE + 32 * C::ROWB));
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < C::IB; ++i) {
          qtf[h2][i] = lds_read_tr(tlo[i] + (h2 * 16 * C::ROWB), thi[i] + (h2 * 16 * C::ROWB));
          otf[h2][i] = lds_read_tr(tlo[i] + (C::TILE + h2 * 16 * C::ROWB), thi[i] + (C::TILE + h2 * 16 * C::ROWB));
        }
      __builtin_amdgcn_sched_barrier(0);
      // S1, dP1 chains || elementwise(0)
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], s1, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr[ks], vf[ks], p1, 0, 0, 0);
      }
      elementwise_plain(s0, p0, q0, 0);
      if (!DROP) {
#pragma unroll
        for (int g = 0; g < 2 * C::KS; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const bf16x8_v pf0 = frag_from_acc(s0, 0), pf1 = frag_from_acc(s0, 1), df0 = frag_from_acc(p0, 0), df1 = frag_from_acc(p0, 1);
      // dV, dK += sub-tile 0 || elementwise(1)
#pragma unroll
      for (int i = 0; i < C::IB; ++i) {
        dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(otf[0][i], pf0, dv[i], 0, 0, 0);
        dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[0][i], df0, dk[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < C::IB; ++i) {
        dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(otf[1][i], pf1, dv[i], 0, 0, 0);
        dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[1][i], df1, dk[i], 0, 0, 0);
      }
      elementwise_plain(s1, p1, q0, 1);
      if (!DROP) {
#pragma unroll
        for (int g = 0; g < 4 * C::IB; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      bf16x8_v qtg[2][C::IB], otg[2][C::IB];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < C::IB; ++i) {
          qtg[h2][i] = lds_read_tr(tlo[i] + ((2 + h2) * 16 * C::ROWB), thi[i] + ((2 + h2) * 16 * C::ROWB));
          otg[h2][i] = lds_read_tr(tlo[i] + (C::TILE + (2 + h2) * 16 * C::ROWB), thi[i] + (C::TILE + (2 + h2) * 16 * C::ROWB));
        }
      const bf16x8_v pf0 = frag_from_acc(s1, 0), pf1 = frag_from_acc(s1, 1), df0 = frag_from_acc(p1, 0), df1 = frag_from_acc(p1, 1);
#pragma unroll
      for (int i = 0; i < C::IB; ++i) {
        dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(otg[0][i], pf0, dv[i], 0, 0, 0);
        dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtg[0][i], df0, dk[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < C::IB; ++i) {
        dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(otg[1][i], pf1, dv[i], 0, 0, 0);
        dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtg[1][i], df1, dk[i], 0, 0, 0);
      }
    }
    advance();
  };
  auto slow_tile = [&]() __attribute__((always_inline)) {
    const int qt = tq[0], q0 = qt * KT;
    const unsigned sb = tile_sync();
    const int qfl = ldsPad[qt];
    unsigned nb[C::KS], tlo[C::IB], thi[C::IB];      // the stage's fragment bases: every read below is base + immediate
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) nb[ks] = (nat0 + sb) ^ (unsigned)(ks << 5);
#pragma unroll
    for (int i = 0; i < C::IB; ++i) { tlo[i] = qtr[i] + sb; thi[i] = qtr_hi[i] + sb; }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // wave-uniform skip: no query of the sub-tile sees a key of this wave causally and none is a masked row
      if (q0 + t * 32 + 31 < kw0 && !((qfl >> t) & 1)) continue;
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
      {
        bf16x8_v qfr[C::KS], ofr[C::KS];
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          qfr[ks] = lds_read16(nb[ks] + (t * 32 * C::ROWB));
          ofr[ks] = lds_read16(nb[ks] + (C::TILE + t * 32 * C::ROWB));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ofr[ks], vf[ks], dpt, 0, 0, 0);
        }
      }
      bf16x8_v qtf[2][C::IB], otf[2][C::IB];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int i = 0; i < C::IB; ++i) {
          qtf[h2][i] = lds_read_tr(tlo[i] + ((2 * t + h2) * 16 * C::ROWB), thi[i] + ((2 * t + h2) * 16 * C::ROWB));
          otf[h2][i] = lds_read_tr(tlo[i] + (C::TILE + (2 * t + h2) * 16 * C::ROWB), thi[i] + (C::TILE + (2 * t + h2) * 16 * C::ROWB));
        }
      __builtin_amdgcn_sched_barrier(0);
      const int lim_causal = q0 + t * 32 + 4 * (lane >> 5) - key;    // key <= query  <=>  -c(r) <= lim_causal
      const int lim_len = T - 1 - q0 - t * 32 - 4 * (lane >> 5);     // query < T     <=>   c(r) <= lim_len
      // dropout words: rows of this sub-tile are registers, the 4 lanes of a quad own the 4 keys of one group -> lane
      // (key & 3) = i hashes rows 4j + i and the quad shares the 16 words by DPP
      uint32_t mine[4] = {0u, 0u, 0u, 0u};
      const int ksh = 8 * (lane & 3);
      if (DROP) {
        const uint32_t T4 = g.T4;
        const uint32_t gq = ((uint32_t)g.hrow + (uint32_t)(q0 + t * 32 + 4 * (lane >> 5) + (lane & 3))) * T4 +
                            (uint32_t)(key >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j] = drop_word(gq + (uint32_t)(8 * j) * T4, drop_key);    // row c = (lane&3) + 8j
      }
      const float* lq = ldsLse + q0 + t * 32 + 4 * (lane >> 5);
      const float* dq_ = ldsD + q0 + t * 32 + 4 * (lane >> 5);
      const bool interior = (q0 + t * 32 >= kw0 + 31) && (q0 + t * 32 + 31 < T) && (kw0 + 31 < T) &&
                            __builtin_amdgcn_ballot_w64(my_kb != 0.f) == 0;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const float lse_q = lq[c], d_q = dq_[c];
          const float pv = exp2_fast(fmaf(st[r], scale2, -lse_q));
          float pd = pv, dpe = dpt[r];
          if (DROP) {
            const bool keep = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr;
            pd = keep ? pv : 0.f;
            dpe = keep ? dpe : 0.f;
          }
          st[r] = pd;
          dpt[r] = pv * (dpe - d_q);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const bool causal_ok = (-c) <= lim_causal;
          const float sv = (causal_ok ? st[r] * scale2 : MASK_VAL * LOG2E) + my_kb;
          const float lse_q = lq[c], d_q = dq_[c];            // the arrays are padded to whole tiles
          const float pv = exp2_fast((c <= lim_len && kvalid) ? sv - lse_q : -INFINITY);        // select: 2^-inf = 0
          float pd = pv, dpe = dpt[r];
          if (DROP) {
            const bool keep = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr;   // word of row r
            pd = keep ? pv : 0.f;
            dpe = keep ? dpe : 0.f;
          }
          st[r] = pd;                                              // dropped P (for dV)
          dpt[r] = causal_ok ? pv * (dpe - d_q) : 0.f;             // dS        (for dK)
        }
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const bf16x8_v pf = frag_from_acc(st, h2), df = frag_from_acc(dpt, h2);
#pragma unroll
        for (int i = 0; i < C::IB; ++i) {
          dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(otf[h2][i], pf, dv[i], 0, 0, 0);
          dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[h2][i], df, dk[i], 0, 0, 0);
        }
      }
    }
    advance();
  };
  while (tq[0] < nqt && !is_fast(tq[0])) slow_tile();
  while (tq[0] < nqt && is_fast(tq[0])) fast_tile();
  while (tq[0] < nqt) slow_tile();
  wait_vm<0>();

  // P and dP were masked but not scaled in the loop (D holds D / s): the survivor scale s is applied once, here
  const float vsc = DROP ? drop_scale : 1.0f, ksc = scale * vsc;
  if (kvalid) {
    bf16_t* krow = dqkv + (g.row0 + key) * ld + d + h * HD;
    bf16_t* vrow = krow + d;
#pragma unroll
    for (int i = 0; i < C::IB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dk[i][4 * g + 0] * ksc, dk[i][4 * g + 1] * ksc);
        pk.y = pack_bf16x2(dk[i][4 * g + 2] * ksc, dk[i][4 * g + 3] * ksc);
        *reinterpret_cast<uint2*>(krow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
        pk.x = pack_bf16x2(dv[i][4 * g + 0] * vsc, dv[i][4 * g + 1] * vsc);
        pk.y = pack_bf16x2(dv[i][4 * g + 2] * vsc, dv[i][4 * g + 3] * vsc);
        *reinterpret_cast<uint2*>(vrow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
      }
  }
}

template <int HD, bool DROP, int NW, int NST>
__global__ __launch_bounds__(64 * NW, 1) void attn_dkv_stream_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                     const float* __restrict__ kbias, const float* __restrict__ lse,
                                                                     const float* __restrict__ Dv, bf16_t* __restrict__ dqkv, int B,
                                                                     int T_launch, int H, float scale, uint32_t drop_thr, uint32_t drop_key,
                                                                     float drop_scale, const int* __restrict__ seq_off) {
  extern __shared__ __attribute__((aligned(1024))) char dyn_smem[];
  const int tid = threadIdx.x;
  const int G = (T_launch + 32 * NW - 1) / (32 * NW);
  int bh, p;
  xcd_remap((G + 1) / 2, B * H, bh, p);
  const int b = bh / H, h = bh - b * H;
  const SeqG g = seq_geom(b, h, H, T_launch, seq_off);
  const int T = g.T;
  const int Tp = (T + KT - 1) / KT * KT;
  char* ring = dyn_smem;                                                        // NST stages of [Q tile | dO tile]
  float* ldsKb = reinterpret_cast<float*>(dyn_smem + NST * 2 * SC<HD>::TILE);   // [Tp]
  float* ldsLse = ldsKb + Tp;                                                   // [Tp]  lse * log2(e)
  float* ldsD = ldsLse + Tp;                                                    // [Tp]  D / s
  int* ldsPad = reinterpret_cast<int*>(ldsD + Tp);                              // [Tp / 64]  bit t: 32-row half t holds a padded position
  if (DROP) drop_key += neko_drop_salt();
  const int first = p, second = G - 1 - p;
  const bool do_first = first * 32 * NW < T, do_second = second != first && second * 32 * NW < T;
  if (!do_first && !do_second) return;
  stage_kbias<NW>(kbias + g.row0, T, ldsKb, ldsPad, tid, tid & 63, tid >> 6);
  for (int i = tid; i < Tp; i += 64 * NW) {
    ldsLse[i] = i < T ? lse[g.hrow + i] * LOG2E : 0.f;
    ldsD[i] = i < T ? Dv[g.hrow + i] : 0.f;
  }
  __syncthreads();
  if (do_first) dkv_tile<HD, DROP, NW, NST>(ring, ldsKb, ldsPad, ldsLse, ldsD, qkv, dout, dqkv, T, H, scale, drop_thr, drop_key, drop_scale, first, g, h);
  if (do_second) {
    __syncthreads();
    dkv_tile<HD, DROP, NW, NST>(ring, ldsKb, ldsPad, ldsLse, ldsD, qkv, dout, dqkv, T, H, scale, drop_thr, drop_key, drop_scale, second, g, h);
  }
}

// The > 64 KiB dynamic-LDS limit is a per-device property of the kernel: one bit per device ordinal, set after the attribute call
// succeeded on that device (setting it twice from two threads is harmless, skipping it is a launch failure)
typedef std::atomic<unsigned long long> LdsLimitDone;
template <typename K>
int set_lds_limit(K kernel, LdsLimitDone& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NEKO_ERR_LAUNCH;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return NEKO_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return NEKO_ERR_LAUNCH;
  done.fetch_or(bit, std::memory_order_release);
  return NEKO_OK;
}
template <int HD, int NW, int NST>
int fwd_launch(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T, int H,
               int thr, unsigned key, float dscale, hipStream_t s, const int* seq_off) {
  const float scale = 1.0f / sqrtf((float)HD);
  const int G = (T + 32 * NW - 1) / (32 * NW);
  dim3 grid((unsigned)((long)((G + 1) / 2) * H * B), 1, 1);      // (sequence, head, tile pair) decoded by xcd_remap
  const int ntile = (T + KT - 1) / KT;
  const size_t lds = (size_t)NST * 2 * SC<HD>::TILE + (size_t)ntile * KT * 4 + (size_t)ntile * 4;
  static LdsLimitDone attr_set[2];
  if ((thr ? set_lds_limit(&attn_fwd_stream_kernel<HD, true, NW, NST>, attr_set[1])
           : set_lds_limit(&attn_fwd_stream_kernel<HD, false, NW, NST>, attr_set[0])) != NEKO_OK)
    return NEKO_ERR_LAUNCH;
  if (thr)
    hipLaunchKernelGGL((attn_fwd_stream_kernel<HD, true, NW, NST>), grid, dim3(64 * NW), lds, s, qkv, kbias, kstart, out, lse,
                       B, T, H, scale, (uint32_t)thr, key, dscale, seq_off);
  else
    hipLaunchKernelGGL((attn_fwd_stream_kernel<HD, false, NW, NST>), grid, dim3(64 * NW), lds, s, qkv, kbias, kstart, out, lse,
                       B, T, H, scale, 0u, key, dscale, seq_off);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// D[b, h, q] = sum_hd dO * O / s: 8 elements per lane, the HD / 8 lanes of a (row, head) reduce among themselves; every load
// is a coalesced 16-B piece of the flat [B T, d] arrays (the one-thread-per-(row, head) form of attention.hip walks 256 B per
// lane at a 256-B lane stride: 82 us at B = 8, H = 16, T = 1024, more than the forward kernel)
template <int HD>
__global__ __launch_bounds__(256) void attn_D_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ D,
                                                    long npieces, int T, int H, float inv_drop_scale,
                                                    const int* __restrict__ seq_off, int nseq) {
  constexpr int LPH = HD / 8;                 // lanes per (row, head)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (i < npieces) {
    const uint4 a = reinterpret_cast<const uint4*>(o)[i], g = reinterpret_cast<const uint4*>(dout)[i];
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc = fmaf(__uint_as_float(aw[e] << 16), __uint_as_float(gw[e] << 16), acc);
      acc = fmaf(__uint_as_float(aw[e] & 0xffff0000u), __uint_as_float(gw[e] & 0xffff0000u), acc);
    }
  }
#pragma unroll
  for (int m = 1; m < LPH; m <<= 1) acc += __shfl_xor(acc, m, 64);
  if (i < npieces && (threadIdx.x & (LPH - 1)) == 0) {
    const long rh = i / LPH;                  // row * H + head
    const long row = rh / H;
    const int h = (int)(rh - row * H);
    if (seq_off) {                            // packed sequences: the sequence of this row by bisection, D at row0 * H + h * T_b + q
      int lo = 0, hi = nseq;                  // invariant: seq_off[lo] <= row < seq_off[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((long)seq_off[mid] <= row) lo = mid; else hi = mid;
      }
      const long row0 = seq_off[lo];
      const int Tb = seq_off[lo + 1] - (int)row0;
      D[row0 * H + (long)h * Tb + (row - row0)] = acc * inv_drop_scale;
    } else {
      const long b = row / T;
      const int q = (int)(row - b * T);
      D[(b * H + h) * T + q] = acc * inv_drop_scale;
    }
  }
}

template <int HD, int NWQ, int NSTQ, int NWK, int NSTK>
int bwd_launch(const bf16_t* qkv, const bf16_t* dout, const float* kbias, const int* kstart, const float* lse, const float* D,
               bf16_t* dqkv, int B, int T, int H, int thr, unsigned key, float dscale, hipStream_t s, const int* seq_off) {
  const float scale = 1.0f / sqrtf((float)HD);
  const int ntile = (T + KT - 1) / KT;
  {
    const int G = (T + 32 * NWQ - 1) / (32 * NWQ);
    dim3 grid((unsigned)((long)((G + 1) / 2) * H * B), 1, 1);
    const size_t lds = (size_t)NSTQ * 2 * SC<HD>::TILE + (size_t)ntile * KT * 4 + (size_t)ntile * 4;
    static LdsLimitDone done[2];
    if (thr) {
      if (set_lds_limit(&attn_dq_stream_kernel<HD, true, NWQ, NSTQ>, done[1]) != NEKO_OK) return NEKO_ERR_LAUNCH;
      hipLaunchKernelGGL((attn_dq_stream_kernel<HD, true, NWQ, NSTQ>), grid, dim3(64 * NWQ), lds, s, qkv, dout, kbias, kstart, lse, D,
                         dqkv, B, T, H, scale, (uint32_t)thr, key, dscale, seq_off);
    } else {
      if (set_lds_limit(&attn_dq_stream_kernel<HD, false, NWQ, NSTQ>, done[0]) != NEKO_OK) return NEKO_ERR_LAUNCH;
      hipLaunchKernelGGL((attn_dq_stream_kernel<HD, false, NWQ, NSTQ>), grid, dim3(64 * NWQ), lds, s, qkv, dout, kbias, kstart, lse, D,
                         dqkv, B, T, H, scale, 0u, key, dscale, seq_off);
    }
    NEKO_CHECK_LAUNCH();
  }
  {
    const int G = (T + 32 * NWK - 1) / (32 * NWK);
    dim3 grid((unsigned)((long)((G + 1) / 2) * H * B), 1, 1);
    const size_t lds = (size_t)NSTK * 2 * SC<HD>::TILE + (size_t)ntile * KT * 4 * 3 + (size_t)ntile * 4;
    static LdsLimitDone done[2];
    if (thr) {
      if (set_lds_limit(&attn_dkv_stream_kernel<HD, true, NWK, NSTK>, done[1]) != NEKO_OK) return NEKO_ERR_LAUNCH;
      hipLaunchKernelGGL((attn_dkv_stream_kernel<HD, true, NWK, NSTK>), grid, dim3(64 * NWK), lds, s, qkv, dout, kbias, lse, D, dqkv, B,
                         T, H, scale, (uint32_t)thr, key, dscale, seq_off);
    } else {
      if (set_lds_limit(&attn_dkv_stream_kernel<HD, false, NWK, NSTK>, done[0]) != NEKO_OK) return NEKO_ERR_LAUNCH;
      hipLaunchKernelGGL((attn_dkv_stream_kernel<HD, false, NWK, NSTK>), grid, dim3(64 * NWK), lds, s, qkv, dout, kbias, lse, D, dqkv, B,
                         T, H, scale, 0u, key, dscale, seq_off);
    }
    NEKO_CHECK_LAUNCH();
  }
  return NEKO_OK;
}

}  // namespace

bool neko_attn_stream_applicable(int T, int hd) { return (hd == 64 || hd == 128) && T >= 1 && T <= TMAX; }

// seq_off null: the uniform (B, T) batch; otherwise B packed sequences (rows seq_off[b] .. seq_off[b+1]-1) whose longest has T rows
int neko_attn_fwd_stream_impl(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T,
                              int H, int hd, int drop_thr, unsigned drop_key, float drop_scale, hipStream_t s, const int* seq_off) {
  // 4 waves x 2 workgroups per CU measured ahead of 8 waves x 1 (hd = 128: 59 vs 63 us at B = 8, 291 vs 312 at B = 32; hd = 64: 70 vs 81)
  static const int nw = [] { const char* e = getenv("NEKO_ATTN_STREAM_WAVES"); return e ? atoi(e) : 4; }();
  if (hd == 128) {
    if (nw == 4) return fwd_launch<128, 4, 2>(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, s, seq_off);
    return fwd_launch<128, 8, 3>(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, s, seq_off);
  }
  if (hd == 64) {
    if (nw == 4) return fwd_launch<64, 4, 3>(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, s, seq_off);
    return fwd_launch<64, 8, 3>(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, s, seq_off);
  }
  return NEKO_ERR_UNSUPPORTED;
}

// D: fp32 workspace [B, H, T]
int neko_attn_bwd_stream_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* kbias, const int* kstart,
                              const float* lse, float* D, bf16_t* dqkv, int B, int T, int H, int hd, int drop_thr,
                              unsigned drop_key, float drop_scale, hipStream_t s, const int* seq_off, long rows) {
  const long npieces = (seq_off ? rows : (long)B * T) * H * hd / 8;
  const float inv = drop_thr ? 1.0f / drop_scale : 1.0f;
  if (hd == 128)
    hipLaunchKernelGGL((attn_D_kernel<128>), dim3((unsigned)((npieces + 255) / 256)), dim3(256), 0, s, out, dout, D, npieces, T, H, inv, seq_off, B);
  else
    hipLaunchKernelGGL((attn_D_kernel<64>), dim3((unsigned)((npieces + 255) / 256)), dim3(256), 0, s, out, dout, D, npieces, T, H, inv, seq_off, B);
  NEKO_CHECK_LAUNCH();
  if (hd == 128) return bwd_launch<128, 4, 2, 4, 3>(qkv, dout, kbias, kstart, lse, D, dqkv, B, T, H, drop_thr, drop_key, drop_scale, s, seq_off);
  if (hd == 64) return bwd_launch<64, 4, 3, 4, 3>(qkv, dout, kbias, kstart, lse, D, dqkv, B, T, H, drop_thr, drop_key, drop_scale, s, seq_off);
  return NEKO_ERR_UNSUPPORTED;
}

NEKO_DEFINE_SALT_SETTER(attention_stream)
