// Persistent 256 x 192 bf16 GEMM whose epilogue is drained under the next tile's k-loop (see the note below).  Same operand
// contract as gemm_glds.hip (A k-contiguous; B k-contiguous or k-strided), bf16 outputs only; neko_gemm_glds_try() asks
// neko_gemm_pers_try() first and falls back to its own tiles when this kernel does not apply.
#include <cstdlib>
#include <type_traits>
#include "neko_kernels.h"

#ifdef NEKO_PERS_TRACE
// phase trace (diagnostic builds, tools/gemm_pers_trace.py): per block, wave (0 = a DMA wave, 4 = a store wave) and tile 8 x
// s_memrealtime (100 MHz): tile start | k-tile 0 | park 0 | 8 output units | park 1 | 8 output units | rest of the k-loop | boundary
__device__ unsigned long long* g_neko_pers_trace = nullptr;
extern "C" int neko_gemm_pers_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_neko_pers_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define PTRACE(jt, slot)                                                                                              \
  do {                                                                                                                \
    if (g_neko_pers_trace && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 3) == 0 && (jt) < 16)                   \
      g_neko_pers_trace[(((long)blockIdx.x * 2 + (threadIdx.x >> 8)) * 16 + (jt)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define PTRACE(jt, slot) do { } while (0)
#endif

namespace {

constexpr int BK = 32;
// epilogue feature bits (the subset of gemm_glds.hip's that this kernel compiles)
enum : unsigned { F_BIAS = 1, F_GELU = 2, F_PRE = 4, F_CB = 256, F_GP = 2048 };

// LDS-DMA piece: wave-uniform 64-bit base in SGPRs + per-lane byte offset, LDS destination handed over in M0 (gemm_glds.hip)
__device__ __forceinline__ void glds16_s(const bf16_t* base_uniform, unsigned byte_off, unsigned lds_dst_wave_uniform) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(byte_off), "s"(base_uniform), "{m0}"(lds_dst_wave_uniform)
               : "memory");
}
// a pointer the compiler can prove wave-uniform (an "s" asm operand must be)
__device__ __forceinline__ const bf16_t* uniform_ptr(const bf16_t* p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  return reinterpret_cast<const bf16_t*>(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                                         (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}
// k-contiguous tile [rows][32 k] (64-B rows): 16-B piece p of row r holds global piece p ^ ((r >> 2) & 3)
__device__ __forceinline__ bf16x8_v frag_kc(const char* lds, int rowbase, int ks, int lane) {
  const int row = rowbase + (lane & 31);
  const int piece = (ks * 2 + (lane >> 5)) ^ ((row >> 2) & 3);
  const uint4 v = *reinterpret_cast<const uint4*>(lds + row * 64 + piece * 16);
  return __builtin_bit_cast(bf16x8_v, v);
}

// =====================================================================================================================
// Persistent 256 x 256 kernel (r03): HALF of a tile's epilogue is drained under the k-loop of the next tile, and the stream of
// k-tiles never stops between tiles
// =====================================================================================================================
// Why.  At K = 768 a 256 x 256 tile of gemm_glds.hip spends 2.2 us in its prologue, ~18.5 us in its k-loop and 3.6 us (bf16
// store) to 9-12 us (GELU with two outputs) in its epilogue, with ONE block per CU and the matrix pipe idle outside the loop:
// 20-35 % of the forward qkv / fc, GELU' dgrad and LM-head logits GEMMs (profiles/r02_gemm_phase_trace.txt).
// Two blocks per CU (256 x 128 tiles) lose more on the L2->LDS path than the overlap returns (profiles/r03_tile_ab1.txt), so
// the second tile in flight lives INSIDE the block:
//   * the block is persistent: it walks tiles b, b + G, b + 2G, ... as ONE stream of k-tiles, so the LDS-DMA ring never drains
//     (the first stages of tile i+1 are requested under the last k-tiles of tile i: no prologue after the first tile);
//   * when the k-loop of tile i ends, rows 0..63 of every wave's 128 x 64 block (+ bias) are rounded to bf16 -- what autocast
//     leaves in the reference for these outputs anyway -- and PARKED in 32 VGPRs per lane; rows 64..127 go out at once through
//     the wave's 4 KB LDS slab (the classic epilogue, half as long); then the accumulators belong to tile i+1;
//   * the parked half leaves in 10 units on the next tile's first 10 k-tiles: 2 x { one unit that writes 32 parked rows into
//     the slab, four units that each read 64 x 16 B back row-major, apply GELU (+ the gelu' factor) where asked and store
//     full 128-byte rows }.  A unit's instructions sit between the MFMAs of its k-tile (pinned like the DMA pieces): the
//     serial chain "slab write -> LDS round trip -> math -> store" that makes the epilogue slow is latency, and latency hides
//     under a k-loop that runs at 82-85 % matrix-pipe utilisation.
// Why only half: registers.  Two waves per SIMD = 256 per lane: 128 accumulators + 48 operand fragments (two k-steps) + ~35 of
// addressing leave 32-45 for a parked tile, and a whole 128 x 64 block is 64.  A 256 x 192 tile (96 + 48 parked) was built
// first and fits entirely, but its k-loop is ~15 % slower (1/6 more L2->LDS bytes and LDS reads per MFMA): net loss
// (profiles/r03_gemm_pers_ab.txt).  LDS: 4 stages x 32 KB + 8 x 4 KB slabs = 160 KB, all of it.
// Stores of the drain share vmcnt with the DMA queue; loads return in order among themselves, so "k-tile g+1 has landed" is
// still "at most the four pieces requested after it are outstanding" (a store in flight only makes the wait conservative).
constexpr int BM = 256, BN = 256, TM = 4, TN = 2, NSTAGE = 4, NW = 8, NT = 512;
constexpr int TMD = 2;                                    // row blocks (of 32) per wave whose output is deferred
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES, RING_BYTES = NSTAGE * STAGE_BYTES;
constexpr int SLAB_BYTES = 32 * 64 * 2;                   // per wave: 32 rows x 64 bf16 (128-B rows)
constexpr int LDS_BYTES = RING_BYTES + NW * SLAB_BYTES;   // 163840 = all of a CU's LDS
constexpr int NKT_MIN = 19;                               // k-tiles a tile needs to hold the drain schedule: 1 + 2 x (1 park + 8 output units)
static_assert(LDS_BYTES == 160 * 1024, "ring + slabs fill the LDS exactly");

// k-strided B tile [32 k][256 cols] (512-B rows, 32 pieces of 16 B): piece p of k-row r holds global piece p ^ ((r & 3) << 2)
__device__ __forceinline__ bf16x8_v frag_ks256(const char* lds, int colbase, int ks, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int col = colbase + 16 * (g & 1) + 4 * (c & 3);
  const int krow = ks * 16 + 8 * (g >> 1) + (c >> 2);          // krow & 3 == (krow + 4) & 3
  const int off = (((col >> 3) ^ ((krow & 3) << 2)) << 4) + ((col & 7) << 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + krow * 512 + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (krow + 4) * 512 + off));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_v, r);
}
// byte offset (from the tile's first element) of the 16 bytes lane `lane` of 1-KiB chunk `chunk` copies
__device__ __forceinline__ unsigned off_kc(long ld, int chunk, int lane) {          // [rows][32 k], 16 rows per chunk
  const int row = chunk * 16 + (lane >> 2);
  const int piece = (lane & 3) ^ ((row >> 2) & 3);
  return (unsigned)((row * ld + piece * 8) * 2);
}
__device__ __forceinline__ unsigned off_ks256(long ld, int chunk, int lane) {       // [32 k][256 cols], 2 k-rows per chunk
  const int kr = chunk * 2 + (lane >> 5);
  const int piece = (lane & 31) ^ ((kr & 3) << 2);
  return (unsigned)((kr * ld + piece * 8) * 2);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

enum : int { D_NONE = 0, D_PARK0 = 1, D_PARK1 = 2, D_OUT = 3 };
#ifndef NEKO_PERS_ABL
#define NEKO_PERS_ABL 0     // ablations for tools/gemm_bench.py (WRONG results): 1 no immediate half-epilogue at the tile boundary,
#endif                      // 2 no deferred drain inside the k-loop, 4 no stores in the drain units (math and LDS traffic stay),
                            // 8 every tile READS the operands of tile (0, 0) (all operand requests hit the L2),
                            // 16 every tile WRITES the output rows / columns of tile (0, 0) (the stores stay in the L2)

template <bool B_KC, unsigned F>
__global__ __launch_bounds__(NT, 2) void gemm_pers_kernel(GemmArgs p, int ntiles) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves of 128 x 64
  const int G = gridDim.x;
  const int nbm = p.M / BM, nbn = p.N / BN;
  const int nkt = p.K / BK;
  // ---- tile ownership.  Block b runs on XCD b % 8 (observed placement; only speed depends on it).  Every XCD OWNS a set of
  // operand panels -- groups of 8 row panels (wide-M launches) or single column panels (the LM head: few rows, 205 column
  // panels) -- so that no A (resp. B) panel is ever requested by two L2s, and its 32 blocks walk the XCD's tile list in
  // lock-step (tile q of the list on block q % 32, time slice q / 32): concurrent tiles share panels 8 x 4, and because
  // persistent blocks advance through k together, a panel's k-slice is fetched once and hit by its other readers.
  // (gemm_glds gives an XCD a contiguous id range too, but its blocks drift apart: 3.7x the operand bytes cross the fabric
  // on the forward qkv GEMM, profiles/r02_s11_mmix_counters.txt, and every GEMM of the step sits at 3.0-3.5 TB/s of it.)
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = (G + 7 - xcd) >> 3;     // this XCD's blocks: bi = 0 .. nbx-1
  const bool by_rows = nbm >= 64;
  int txcd;                                                 // tiles this XCD owns
  if (by_rows) {
    const int ngr = (nbm + 7) >> 3;                         // row-panel groups; XCD x owns groups x, x + 8, ...
    const int mine = (ngr + 7 - xcd) >> 3;
    int rows = mine * 8;
    if (mine > 0 && ((ngr - 1) & 7) == xcd) rows -= ngr * 8 - nbm;      // the ragged last group
    txcd = rows * nbn;
  } else {
    txcd = ((nbn + 7 - xcd) >> 3) * nbm;                    // column panels x, x + 8, ...
  }
  const int nmine = (txcd - bi + nbx - 1) / nbx;
  if (nmine <= 0) return;
  const int ktot = nmine * nkt;
  auto coords = [&](int L, int& m0, int& n0) {              // L = this block's L-th tile = entry bi + L * nbx of the XCD's list
    const int q = bi + L * nbx;
    if (by_rows) {
      const int per_group = 8 * nbn;
      const int k = q / per_group, local = q - k * per_group;
      const int gid = xcd + 8 * k;
      const int gsz = min(8, nbm - gid * 8);
      // (a ragged group is only ever the XCD's last one, so q / per_group above still finds it)
      m0 = (gid * 8 + local % gsz) * BM;
      n0 = (local / gsz) * BN;
    } else {
      const int c = q / nbm, r = q - c * nbm;
      m0 = r * BM;
      n0 = (xcd + 8 * c) * BN;
    }
  };

#ifdef NEKO_PERS_STAGGER
  // experiment: every other block of an XCD starts NEKO_PERS_STAGGER x ~4 us late, so that tile boundaries (store bursts) of
  // half of the CUs fall into the k-loops of the other half
  if ((blockIdx.x >> 3) & 1) {
#pragma unroll 1
    for (int i = 0; i < NEKO_PERS_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  // ---- accumulators, parked half tile, slabs (every wave: same in both roles) ----
  f32x16 acc[TM][TN];
  uint32_t parked[TMD][TN][8];
#pragma unroll
  for (int i = 0; i < TMD; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) parked[i][j][e] = 0u;
  char* slab = smem + RING_BYTES + wave * SLAB_BYTES;
  int pm0 = 0, pn0 = 0;                      // coordinates of the parked tile
  // slab write of one register quad pair: lane l holds row l & 31 and, per 32-column block j and quad q, the 4 columns
  // 32j + 8q + 4(l >> 5) + 0..3 = 8 bytes at 16-B chunk 4j + q (XOR (row >> 2) & 3), half l >> 5
  auto slab_write = [&](uint32_t lo, uint32_t hi, int j, int q) {
    const int row = lane & 31;
    const int chunk = (4 * j + q) ^ ((row >> 2) & 3);
    *reinterpret_cast<uint2*>(slab + row * 128 + chunk * 16 + 8 * (lane >> 5)) = make_uint2(lo, hi);
  };
  auto park_write = [&](auto itag, int j, int q) {
    constexpr int i = decltype(itag)::value;
    slab_write(parked[i][j][2 * q], parked[i][j][2 * q + 1], j, q);
  };
  auto barrier_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // slab writes visible to the other waves

  // ---- operand fragments: read one k-step ahead ----
  bf16x8_v a0[TM], b0[TN], a1[TM], b1[TN];
  auto frag_a = [&](int g, int i, int ks) {
    return frag_kc(smem + (g & (NSTAGE - 1)) * STAGE_BYTES, (wm * TM + i) * 32, ks, lane);
  };
  auto frag_b = [&](int g, int j, int ks) {
    const char* lb = smem + (g & (NSTAGE - 1)) * STAGE_BYTES + A_BYTES;
    return B_KC ? frag_kc(lb, (wn * TN + j) * 32, ks, lane) : frag_ks256(lb, (wn * TN + j) * 32, ks, lane);
  };
  // bias of the wave's 64 columns through the scalar cache (see tile_boundary)
  auto load_bias = [&](int n0, float4 (&bv)[TN][4]) {
    typedef __attribute__((address_space(4))) const float const_f32;
    const const_f32* bp = reinterpret_cast<const const_f32*>(reinterpret_cast<uintptr_t>(p.bias + n0 + wn * 64));
    const bool hi = lane >= 32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr ((F & F_BIAS) != 0) {
          const int c = 32 * j + 8 * q;
          float s[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s[e] = bp[c + e];
            asm volatile("" : "+s"(s[e]));         // keep the value in an SGPR: the select below must not fold into a per-lane load
          }
          bv[j][q] = make_float4(hi ? s[4] : s[0], hi ? s[5] : s[1], hi ? s[6] : s[2], hi ? s[7] : s[3]);
        }
      }
  };

  // =================================================================================================================
  // The two roles.  gfx9 has ONE vmcnt for loads and stores: a wave that waits for its LDS-DMA pieces with a counted vmcnt
  // also waits for every store it has in flight, and a store takes microseconds to retire (profiles/r03_gemm_pers_ab.txt:
  // with all waves doing both, the stores cost 54-125 us per launch although they sat between MFMAs).  So the work is split:
  //   waves 0-3  request ALL LDS-DMA pieces (8 per wave and k-tile) and are the only ones that wait on vmcnt -- nothing else of
  //              theirs touches that counter (their bias comes through the scalar cache);
  //   waves 4-7  issue ALL global stores -- their own slab's and the slab of wave w - 4 (the wave above them in the tile) -- and
  //              never wait on vmcnt: the block barrier of every k-tile tells them that the DMA waves have seen the data land.
  // Every wave runs the same MFMAs and the same barriers.  One DMA wave and one store wave share a SIMD (waves w and w + 4).
  // =================================================================================================================
  auto run = [&](auto role_tag) {
    constexpr bool DMA = decltype(role_tag)::value;

    // ---- DMA role: the stream of k-tiles, 3 k-tiles ahead of the MFMAs; chunks 4w .. 4w+3 of A and of B ----
    unsigned voffA[4], voffB[4];
    unsigned ldsA = 0, ldsB = 0;
    int s_tile = 0, s_kt = 0, s_g = 0;
    const bf16_t* sA = p.A; const bf16_t* sB = p.B;
    const long stepB = B_KC ? (long)BK : (long)BK * p.ldb;
    auto cursor_tile = [&](int j) {
      int m0, n0;
      coords(j, m0, n0);
      if (NEKO_PERS_ABL & 8) m0 = n0 = 0;
      sA = p.A + (long)m0 * p.lda;
      sB = B_KC ? p.B + (long)n0 * p.ldb : p.B + n0;
    };
    if constexpr (DMA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        voffA[i] = off_kc(p.lda, wave * 4 + i, lane);
        voffB[i] = B_KC ? off_kc(p.ldb, wave * 4 + i, lane) : off_ks256(p.ldb, wave * 4 + i, lane);
      }
      const unsigned lds0 = __builtin_amdgcn_readfirstlane(
          (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
      ldsA = lds0 + wave * 4096;
      ldsB = lds0 + A_BYTES + wave * 4096;
      cursor_tile(0);
    }
    auto stage_piece = [&](int pc) {          // piece pc (0..3: A; 4..7: B) of k-tile s_g into ring slot s_g % NSTAGE
      const unsigned slot = (unsigned)(s_g & (NSTAGE - 1)) * STAGE_BYTES;
      if (pc < 4) glds16_s(uniform_ptr(sA), voffA[pc], ldsA + slot + pc * 1024);
      else glds16_s(uniform_ptr(sB), voffB[pc - 4], ldsB + slot + (pc - 4) * 1024);
    };
    // The cursor SATURATES at the stream's last k-tile: behind the end the loop keeps requesting that k-tile again, into ring
    // slots whose tiles are already consumed -- every k-tile of the loop issues exactly eight pieces per DMA wave and "k-tile
    // g+1 has landed" is always "at most the eight pieces requested after it are outstanding": no tail cases.
    auto cursor_advance = [&]() {
      ++s_g;
      if (s_tile * nkt + s_kt + 1 < ktot) {
        if (++s_kt == nkt) {
          s_kt = 0;
          ++s_tile;
          cursor_tile(s_tile);
        } else {
          sA += BK;
          sB += stepB;
        }
      }
    };

    // ---- store role: one output unit = 64 x 16 B of a slab row-major (chunk id = 64 it + lane -> row id >> 3, chunk id & 7) ----
    uint4 rd = make_uint4(0, 0, 0, 0);          // the unit's slab read
    f32x2_v gl[4], gf[4];                       // GELU results (and factors) of its 8 elements
    auto out_read = [&](int sel, int it) {      // sel 0: the wave's own slab, 1: the slab of wave - 4
      const char* sl = smem + RING_BYTES + (wave - 4 * sel) * SLAB_BYTES;
      const int id = it * 64 + lane;
      const int row = id >> 3, cc = id & 7;
      rd = *reinterpret_cast<const uint4*>(sl + row * 128 + ((cc ^ ((row >> 2) & 3)) << 4));
    };
    auto out_math = [&](int e) {                // pair e (0..3) of the 8 elements
      if constexpr ((F & F_GELU) != 0) {
        const uint32_t w = e == 0 ? rd.x : e == 1 ? rd.y : e == 2 ? rd.z : rd.w;
        const f32x2_v x = (f32x2_v){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
        if constexpr ((F & F_GP) != 0) gelu_and_grad2_f(x, gl[e], gf[e]);
        else gl[e] = gelu2_f(x);
      }
    };
    auto out_store = [&](int m0t, int n0t, int sel, int rowblk, int it, int which) {   // which 0: main output; 1: the second (pre / gelu')
      const int id = it * 64 + lane;
      const int row = id >> 3, cc = id & 7;
      const long grow = (long)m0t + (wm - sel) * 128 + rowblk * 32 + row;
      const int gcol = n0t + wn * 64 + cc * 8;
      if (NEKO_PERS_ABL & 4) {
        asm volatile("" ::"v"(rd.x), "v"(gl[0].x), "v"(gl[1].x), "v"(gl[2].x), "v"(gl[3].x), "v"(gf[0].x), "v"(gf[3].y));
        return;
      }
      if (which == 0) {
        uint4 o = rd;
        if constexpr ((F & F_GELU) != 0)
          o = make_uint4(pack_bf16x2(gl[0].x, gl[0].y), pack_bf16x2(gl[1].x, gl[1].y), pack_bf16x2(gl[2].x, gl[2].y), pack_bf16x2(gl[3].x, gl[3].y));
        *reinterpret_cast<uint4*>(p.Cb + grow * p.ldcb + gcol) = o;
      } else if constexpr ((F & F_PRE) != 0) {
        uint4 o = rd;                            // the bf16 pre-activation as parked
        if constexpr ((F & F_GP) != 0)
          o = make_uint4(pack_bf16x2(gf[0].x, gf[0].y), pack_bf16x2(gf[1].x, gf[1].y), pack_bf16x2(gf[2].x, gf[2].y), pack_bf16x2(gf[3].x, gf[3].y));
        *reinterpret_cast<uint4*>(p.pre_out + grow * p.ldpre + gcol) = o;
      }
    };
    auto out_unit_now = [&](int m0t, int n0t, int sel, int rowblk, int it) {     // an output unit outside the k-loop
      out_read(sel, it);
#pragma unroll
      for (int e = 0; e < 4; ++e) out_math(e);
      out_store(m0t, n0t, sel, rowblk, it, 0);
      out_store(m0t, n0t, sel, rowblk, it, 1);
    };
    // drain hooks of one k-tile: step 0 / step 1, after MFMA m (0..7)
    auto drain_hook = [&](auto ktag, int step, int m, int pass, int sel, int it) {
      constexpr int KIND = decltype(ktag)::value;
      if constexpr (KIND == D_PARK0 || KIND == D_PARK1) {
        using I = std::integral_constant<int, KIND == D_PARK0 ? 0 : 1>;
        if (step == 0) {                        // 8 slab writes: one behind each MFMA of the first k-step
          __builtin_amdgcn_sched_barrier(0);
          park_write(I{}, m >> 2, m & 3);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (KIND == D_OUT && !DMA) {
        __builtin_amdgcn_sched_barrier(0);
        if (step == 0) {
          if (m == 0) out_read(sel, it);
          if (m >= 2 && m < 6) out_math(m - 2);  // pairs 0..3 behind MFMAs 2..5
        } else {
          if (m == 2) out_store(pm0, pn0, sel, pass, it, 0);
          if (m == 6) out_store(pm0, pn0, sel, pass, it, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // ---- one k-tile: [MFMAs k-step 0 | barrier: k-tile g+1 visible | MFMAs k-step 1 (+ the DMA of k-tile g+3)] ----
    auto kstep = [&](auto ktag, int step, const bf16x8_v (&a)[TM], const bf16x8_v (&b)[TN], bf16x8_v (&an)[TM], bf16x8_v (&bn)[TN],
                     int g_load, int ks_load, auto stage_tag, int pass, int sel, int it) {
      constexpr bool STAGE = decltype(stage_tag)::value && DMA;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          const int m = i * TN + j;
          if (m < 6) {                            // next k-step's operands, one read per MFMA: a0, b0, b1, a1, a2, a3
            __builtin_amdgcn_sched_barrier(0);
            if (m == 0) an[0] = frag_a(g_load, 0, ks_load);
            else if (m <= 2) bn[m - 1] = frag_b(g_load, m - 1, ks_load);
            else an[m - 2] = frag_a(g_load, m - 2, ks_load);
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (STAGE) {                  // the eight pieces of k-tile g+3, one behind every MFMA
            __builtin_amdgcn_sched_barrier(0);
            stage_piece(m);
            __builtin_amdgcn_sched_barrier(0);
          }
          drain_hook(ktag, step, m, pass, sel, it);
        }
    };
    auto ktile = [&](auto ktag, int g, int pass, int sel, int it) {
      constexpr int KIND = decltype(ktag)::value;
      kstep(ktag, 0, a0, b0, a1, b1, g, 1, std::false_type{}, pass, sel, it);
      if constexpr (DMA) wait_vm<8>();                      // k-tile g+1 has landed (k-tile g+2's pieces may be in flight)
      if constexpr (KIND == D_PARK0 || KIND == D_PARK1) barrier_lds();      // ... and this wave's slab is visible to its store wave
      else asm volatile("s_barrier" ::: "memory");
      kstep(ktag, 1, a1, b1, a0, b0, g + 1, 0, std::true_type{}, pass, sel, it);
      if constexpr (DMA) cursor_advance();
    };
    using K_NONE = std::integral_constant<int, D_NONE>;
    using K_P0 = std::integral_constant<int, D_PARK0>;
    using K_P1 = std::integral_constant<int, D_PARK1>;
    using K_OUT = std::integral_constant<int, (DMA ? D_NONE : D_OUT)>;

    // ---- prologue: the first NSTAGE-1 k-tiles of the stream ----
    if constexpr (DMA) {
#pragma unroll 1
      for (int t = 0; t < NSTAGE - 1; ++t) {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) stage_piece(pc);
        cursor_advance();
      }
      wait_vm<16>();
    }
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int j = 0; j < TN; ++j) b0[j] = frag_b(0, j, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) a0[i] = frag_a(0, i, 0);

    // ---- the stream of tiles ----
    int g = 0;
#pragma unroll 1
    for (int jt = 0; jt < nmine; ++jt) {
      int m0, n0;
      coords(jt, m0, n0);
      if (NEKO_PERS_ABL & 16) m0 = n0 = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      PTRACE(jt, 0);
      if (jt == 0 || (NEKO_PERS_ABL & 2)) {
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt, ++g) ktile(K_NONE{}, g, 0, 0, 0);
        PTRACE(jt, 6);
      } else {
        // the parked half of the previous tile leaves on k-tiles 1..18 -- a SEQUENCE of loops (one switch inside one loop made
        // the register allocator spill 170-200 registers at the merge of the variants):
        //   k-tile 0: nothing (its barrier separates the boundary's slab reads from the writes below)
        //   1: park row block 0 into the slabs | 2..9: eight output units (own slab, then the slab of wave - 4) | 10: park row block 1 | 11..18: eight units
        ktile(K_NONE{}, g, 0, 0, 0);
        ++g;
        PTRACE(jt, 1);
        ktile(K_P0{}, g, 0, 0, 0);
        ++g;
        PTRACE(jt, 2);
#pragma unroll 1
        for (int u = 0; u < 8; ++u, ++g) ktile(K_OUT{}, g, 0, u >> 2, u & 3);
        PTRACE(jt, 3);
        ktile(K_P1{}, g, 1, 0, 0);
        ++g;
        PTRACE(jt, 4);
#pragma unroll 1
        for (int u = 0; u < 8; ++u, ++g) ktile(K_OUT{}, g, 1, u >> 2, u & 3);
        PTRACE(jt, 5);
#pragma unroll 1
        for (int kt = NKT_MIN; kt < nkt; ++kt, ++g) ktile(K_NONE{}, g, 0, 0, 0);
        PTRACE(jt, 6);
      }
      // ---- tile boundary: (acc + bias) -> bf16; row blocks 0, 1 are parked, row blocks 2, 3 leave now through the slabs ----
      pm0 = m0;
      pn0 = n0;
      float4 bv[TN][4];
      load_bias(n0, bv);
#pragma unroll
      for (int i = 0; i < TMD; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            parked[i][j][2 * q] = pack_bf16x2(acc[i][j][4 * q] + bv[j][q].x, acc[i][j][4 * q + 1] + bv[j][q].y);
            parked[i][j][2 * q + 1] = pack_bf16x2(acc[i][j][4 * q + 2] + bv[j][q].z, acc[i][j][4 * q + 3] + bv[j][q].w);
          }
      if (!(NEKO_PERS_ABL & 1)) {
#pragma unroll
        for (int i = TMD; i < TM; ++i) {
          if (i > TMD) asm volatile("s_barrier" ::: "memory");          // the store waves are done with the previous row block's slabs
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              slab_write(pack_bf16x2(acc[i][j][4 * q] + bv[j][q].x, acc[i][j][4 * q + 1] + bv[j][q].y),
                         pack_bf16x2(acc[i][j][4 * q + 2] + bv[j][q].z, acc[i][j][4 * q + 3] + bv[j][q].w), j, q);
          barrier_lds();
          if constexpr (!DMA) {
#pragma unroll 1
            for (int u = 0; u < 8; ++u) out_unit_now(m0, n0, u >> 2, i, u & 3);
          }
        }
      }
      PTRACE(jt, 7);
    }
    // ---- the last tile's parked half has nothing to hide under: drain it here ----
#pragma unroll
    for (int pass = 0; pass < TMD; ++pass) {
      asm volatile("s_barrier" ::: "memory");                              // the store waves are done with the slabs
      if (pass == 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) park_write(std::integral_constant<int, 0>{}, m >> 2, m & 3);
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) park_write(std::integral_constant<int, 1>{}, m >> 2, m & 3);
      }
      barrier_lds();
      if constexpr (!DMA) {
#pragma unroll 1
        for (int u = 0; u < 8; ++u) out_unit_now(pm0, pn0, u >> 2, pass, u & 3);
      }
    }
    if constexpr (DMA) wait_vm<0>();       // (the saturated cursor's last requests)
  };
  if (wave < 4) run(std::true_type{});
  else run(std::false_type{});
}

// Off unless NEKO_GEMM_PERS=1: measured level with gemm_glds on the step's shapes (profiles/r03_gemm_pers_ab.txt: -1..-3 % on the
// N = 768 dgrads, -10 % on 32768 x 1024 x 768, +1 % forward qkv, +6 % forward fc) -- the K = 768 GEMMs turned out to be bound by
// the bytes they move over the fabric (every GEMM of the step sits at 3.0-3.5 TB/s), not by the latency this kernel hides.
int g_pers_mode = -1;      // -1: NEKO_GEMM_PERS (default off), 0 / 1: forced by neko_gemm_set_persistent (tests, A/B runs)
int pers_enabled() {
  static const int v = [] { const char* e = getenv("NEKO_GEMM_PERS"); return e ? atoi(e) : 0; }();
  return g_pers_mode >= 0 ? g_pers_mode : v;
}
int num_cus() {
  static const int v = [] {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
    return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }();
  return v;
}
// the launches this kernel takes: A k-contiguous, bf16 output only, every tile interior, K long enough to hold the 14 drain
// units, at least two rounds of tiles (with fewer there is no "next tile" to hide an epilogue under)
template <bool B_KC>
int try_launch(const GemmArgs& a, hipStream_t s) {
  if (!pers_enabled()) return 1;
  if (a.splitk > 1 || a.Cf || !a.Cb || a.resid || a.drop_thr || a.alpha != 1.0f || a.alpha_dev || a.accumulate) return 1;
  if (a.M % BM || a.N % BN || a.K % BK || a.K / BK < NKT_MIN) return 1;
  if ((a.ldcb & 7) || (reinterpret_cast<uintptr_t>(a.Cb) & 15)) return 1;
  const int ntiles = (a.M / BM) * (a.N / BN);
  const int G = num_cus();
  if (ntiles < 2 * G) return 1;
  unsigned f = F_CB;
  if (a.bias) f |= F_BIAS;
  if (a.act == 1) f |= F_GELU | (a.pre_out ? F_PRE : 0);
  else if (a.act == 3) f |= F_GELU | F_PRE | F_GP;
  else if (a.act != 0) return 1;
  if ((f & F_PRE) && ((a.ldpre & 7) || (reinterpret_cast<uintptr_t>(a.pre_out) & 15))) return 1;
#define NEKO_PERS(MASK)                                                                                                  \
  case (MASK): hipLaunchKernelGGL((gemm_pers_kernel<B_KC, (MASK)>), dim3(G), dim3(NT), 0, s, a, ntiles); break;
  switch (f) {
#ifdef NEKO_PERS_ONLY
    NEKO_PERS(NEKO_PERS_ONLY)
#else
    NEKO_PERS(F_CB)
    NEKO_PERS(F_BIAS | F_CB)
    NEKO_PERS(F_BIAS | F_GELU | F_CB)
    NEKO_PERS(F_BIAS | F_GELU | F_PRE | F_CB)
    NEKO_PERS(F_BIAS | F_GELU | F_PRE | F_GP | F_CB)
#endif
    default: return 1;
  }
#undef NEKO_PERS
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}


}  // namespace

int neko_gemm_set_persistent_impl(int mode) {
  const int prev = g_pers_mode;
  g_pers_mode = mode < 0 ? -1 : (mode ? 1 : 0);
  return prev;
}
// 1 = this kernel does not take the launch (the caller uses its own tiles), otherwise a neko status code
int neko_gemm_pers_try(const GemmArgs& a, int b_kstrided, hipStream_t s) {
  return b_kstrided ? try_launch<false>(a, s) : try_launch<true>(a, s);
}
