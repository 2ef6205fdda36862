// Fast path of neko_gemm_bf16: same contract as gemm_bf16.hip (see there for the reference citations),
// restricted to K-ranges that are multiples of 64; everything else falls back to the register-staged kernel.
//
// gfx950 design
//  * HBM/L2 -> LDS by direct DMA: global_load_lds_dwordx4 (16 B/lane, 1 KiB per wave-instruction), no staging
//    VGPRs and no ds_write pass.  The LDS image of a wave-instruction is lane-linear, so the bank-conflict
//    swizzle is applied to the per-lane SOURCE address and undone by the fragment reads:
//      k-contiguous tile [rows][32 k] (64-B rows)   : 16-B piece index ^= (row>>2)&3  (ds_read_b128)
//      k-strided   tile [32 k][cols] (2*cols-B rows): 16-B piece index ^= (k&3)<<2    (ds_read_b64_tr_b16)
//  * block tile (32*TM*WM) x (32*TN*WN) x 32 with WM x WN waves, each wave TM x TN v_mfma_f32_32x32x16_bf16
//    tiles.  The tile shape is a template parameter because the CU's L2->LDS path (64 B/clk) is the binding
//    resource: a 128x128 tile needs 64 B/clk at full MFMA rate (caps near 35 % of peak), 256x128 needs 48,
//    256x256 needs 32.  launch() picks per shape (tile quantisation vs intensity), see pick_config().
//  * NSTAGE-deep LDS ring, NSTAGE-1 tiles in flight, counted s_waitcnt vmcnt(N) + one raw s_barrier per k-tile
//    (the DMA queue is never drained inside the loop).
//  * epilogue through LDS: each wave parks 32 rows of its accumulators at a time in a private slab
//    (XOR-swizzled float4 chunks), re-reads them row-major and applies bias / GELU / GELU' / dropout / residual /
//    accumulate with 16-B loads and stores (bf16 outputs 8 B per lane).
//  * out-of-range rows / columns are clamped to the last valid one (their products only reach outputs that are
//    never stored); the contraction range itself is exact (K % 64 == 0 is the precondition of this path).
#include <cstdlib>
#include "neko_kernels.h"

#ifndef NEKO_GEMM_DIAG
#define NEKO_GEMM_DIAG 0   // ablations for tools/gemm_bench.py: 1 no in-loop DMA, 2 fragments read once, 3 no MFMA (ping-pong kernel)
#endif

namespace {

constexpr int BK = 32;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

// LDS-DMA issued from inline asm on purpose: with the builtin, hipcc cannot prove that the ring slot being
// filled is not the one being read and drains the queue (s_waitcnt vmcnt(0)) before every first ds_read, which
// defeats the multi-stage ring.  Issued this way the DMA is invisible to the compiler's wait-count pass; the only
// waits are the counted ones in wait_dma_and_barrier().  M0 (the LDS destination base) is compiler-reserved: it is
// saved and restored inside the same statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_dst_wave_uniform) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(
      (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)lds_dst_wave_uniform));
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(dst)
      : "memory");
}

// ---- staging: EXT/16/NW wave-instructions (1 KiB each) per operand per wave ---------------------------------
// k-contiguous tile [EXT rows][32 k] (64-B rows): piece p of row r holds global piece p ^ ((r>>2)&3)
template <int EXT, int NW>
__device__ __forceinline__ void stage_kc(const bf16_t* __restrict__ P, long ld, int r0, int nrows, int k0, char* lds,
                                         int wave, int lane) {
  constexpr int PER = EXT / 16 / NW;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int chunk = wave * PER + i;               // 1 KiB = 16 rows x 64 B
    const int row = chunk * 16 + (lane >> 2);
    const int piece = (lane & 3) ^ ((row >> 2) & 3);
    const int gr = min(r0 + row, nrows - 1);
    glds16(P + (long)gr * ld + k0 + piece * 8, lds + chunk * 1024);
  }
}
// k-strided tile [32 k][EXT cols] (2*EXT-B rows): piece p of k-row r holds global piece p ^ ((r&3)<<2)
template <int EXT, int NW>
__device__ __forceinline__ void stage_ks(const bf16_t* __restrict__ P, long ld, int c0, int ncols, int k0, char* lds,
                                         int wave, int lane) {
  constexpr int PER = EXT / 16 / NW;
  constexpr int PPR = EXT / 8;                      // 16-B pieces per k-row
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int chunk = wave * PER + i;               // 1 KiB = 512/EXT k-rows x 2*EXT B
    const int kr = chunk * (64 / PPR) + lane / PPR;
    const int piece = (lane % PPR) ^ ((kr & 3) << 2);
    const int gc = min(c0 + piece * 8, ncols - 8);
    glds16(P + (long)(k0 + kr) * ld + gc, lds + chunk * 1024);
  }
}

// one 1-KiB piece (index i of this wave's EXT/16/NW) of the tiles above, for issue interleaved with MFMAs
template <int EXT, int NW, bool KC>
__device__ __forceinline__ void stage_piece(const bf16_t* __restrict__ P, long ld, int e0, int next, int k0, char* lds,
                                            int wave, int lane, int i) {
  constexpr int PER = EXT / 16 / NW;
  const int chunk = wave * PER + i;
  if (KC) {
    const int row = chunk * 16 + (lane >> 2);
    const int piece = (lane & 3) ^ ((row >> 2) & 3);
    const int gr = min(e0 + row, next - 1);
    glds16(P + (long)gr * ld + k0 + piece * 8, lds + chunk * 1024);
  } else {
    constexpr int PPR = EXT / 8;
    const int kr = chunk * (64 / PPR) + lane / PPR;
    const int piece = (lane % PPR) ^ ((kr & 3) << 2);
    const int gc = min(e0 + piece * 8, next - 8);
    glds16(P + (long)(k0 + kr) * ld + gc, lds + chunk * 1024);
  }
}

__device__ __forceinline__ bf16x8_v frag_kc(const char* lds, int rowbase, int ks, int lane) {
  const int row = rowbase + (lane & 31);
  const int piece = (ks * 2 + (lane >> 5)) ^ ((row >> 2) & 3);
  const uint4 v = *reinterpret_cast<const uint4*>(lds + row * 64 + piece * 16);
  return __builtin_bit_cast(bf16x8_v, v);
}
template <int EXT>
__device__ __forceinline__ bf16x8_v frag_ks(const char* lds, int colbase, int ks, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int col = colbase + 16 * (g & 1) + 4 * (c & 3);
  const int krow = ks * 16 + 8 * (g >> 1) + (c >> 2);          // krow & 3 == (krow+4) & 3
  const int off = (((col >> 3) ^ ((krow & 3) << 2)) << 4) + ((col & 7) << 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + krow * (2 * EXT) + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (krow + 4) * (2 * EXT) + off));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_v, r);
}

// wait until at most `n` of this wave's LDS-DMA instructions are outstanding, then block barrier.
// One asm statement with a memory clobber: neither the DMA issue nor the ds_reads may cross it.
template <int N>
__device__ __forceinline__ void wait_dma_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// block -> output tile, grouped rasterisation: consecutive logical ids (one XCD's L2, see xcd_remap) cover groups
// of 8 row panels and sweep the column panels inside a group, 8 tiles per column panel.  The tiles in flight on an
// XCD then share <= 8 A panels and a few B panels: with B small (activations x weight) A streams once as before;
// with A small and B huge (LM-head logits: 4096 rows x 52k vocabulary columns) the embedding table streams once
// per 8 row panels instead of once per row panel.
template <int BM, int BN>
__device__ __forceinline__ void tile_coords(const GemmArgs& p, int& tm, int& tn) {
  constexpr int GROUP_M = 8;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
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

// ---- epilogue (shared by both kernels): the caller has passed a block barrier after the last ring read -------
template <class C>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x16 (&acc)[C::TM][C::TN], char* smem, int m0, int n0,
                                         int wm, int wn, int wave, int lane) {
  constexpr int TM = C::TM, TN = C::TN;
  // ---- epilogue through this wave's private slab, TM passes of 32 rows ---------------------------------------
  constexpr int SW = 32 * TN;              // slab row length (f32)
  constexpr int CPR = SW / 4;              // float4 chunks per slab row (16 or 32)
  constexpr int RPI = 64 / CPR;            // slab rows one wave-instruction covers
  float* slab = reinterpret_cast<float*>(smem + wave * C::SLAB_BYTES);   // float4 chunk index ^= row & 15
  // same-wave LDS write -> read: the compiler's lgkmcnt wait orders it (no cross-wave sharing of a slab)
  const float alpha = p.alpha_dev ? p.alpha * (*p.alpha_dev) : p.alpha;
  const bool to_ws = p.splitk > 1 && p.splitk_ws;   // split-K slices go to a workspace, reduced in fixed order afterwards
  const bool atomic = p.splitk > 1 && !to_ws;
  float* const Cf_out = to_ws ? p.splitk_ws + (long)blockIdx.y * p.M * p.N : p.Cf;
  const long ldcf_out = to_ws ? p.N : p.ldcf;
  const int acc_out = to_ws ? 0 : p.accumulate;
  const bool lead = !atomic || blockIdx.y == 0;
  const int cchunk = lane % CPR;
  const int col = n0 + wn * SW + cchunk * 4;
  const bool col_ok = col < p.N;
  const bool vec = col + 4 <= p.N;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias && lead && col_ok) {
    if (vec) bv = *reinterpret_cast<const float4*>(p.bias + col);
    else {
      bv.x = p.bias[col];
      if (col + 1 < p.N) bv.y = p.bias[col + 1];
      if (col + 2 < p.N) bv.z = p.bias[col + 2];
    }
  }
  const bool cf_vec = vec && ((ldcf_out & 3) == 0);
  const bool r_vec = vec && ((p.ldr & 3) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int lc = j * 32 + (lane & 31);
        slab[lr * SW + ((((lc >> 2) ^ (lr & 15)) << 2) | (lc & 3))] = acc[i][j][r];
      }
#pragma unroll 4
  for (int s = 0; s < 32 / RPI; ++s) {
    const int lrow = s * RPI + lane / CPR;
    const int row = m0 + (wm * TM + i) * 32 + lrow;
    const float4 a4 = *reinterpret_cast<const float4*>(slab + lrow * SW + ((cchunk ^ (lrow & 15)) << 2));
    if (row >= p.M || !col_ok) continue;
    float v[4] = {a4.x * alpha + bv.x, a4.y * alpha + bv.y, a4.z * alpha + bv.z, a4.w * alpha + bv.w};
    const int nv = vec ? 4 : (p.N - col);
    if (p.act == 1) {
      uint32_t pk[2];
      bf16_t pb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pb[e] = f32_to_bf16(v[e]);
        v[e] = gelu_f(bf16_to_f32(pb[e]));
      }
      if (p.pre_out) {
        bf16_t* dst = p.pre_out + (long)row * p.ldpre + col;
        if (vec) {
          pk[0] = (uint32_t)pb[0] | ((uint32_t)pb[1] << 16);
          pk[1] = (uint32_t)pb[2] | ((uint32_t)pb[3] << 16);
          *reinterpret_cast<uint2*>(dst) = make_uint2(pk[0], pk[1]);
        } else {
          for (int e = 0; e < nv; ++e) dst[e] = pb[e];
        }
      }
    } else if (p.act == 2) {
      const bf16_t* src = p.act_in + (long)row * p.ldact + col;
      if (vec) {
        const uint2 q = *reinterpret_cast<const uint2*>(src);
        v[0] *= gelu_grad_f(bf16_to_f32((bf16_t)(q.x & 0xffff)));
        v[1] *= gelu_grad_f(bf16_to_f32((bf16_t)(q.x >> 16)));
        v[2] *= gelu_grad_f(bf16_to_f32((bf16_t)(q.y & 0xffff)));
        v[3] *= gelu_grad_f(bf16_to_f32((bf16_t)(q.y >> 16)));
      } else {
        for (int e = 0; e < nv; ++e) v[e] *= gelu_grad_f(bf16_to_f32(src[e]));
      }
    }
    if (p.drop_thr) {
      const uint32_t base = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
      drop4(v, base, p.drop_key, p.drop_thr, p.drop_scale);
    }
    if (p.resid && lead) {
      const float* rs = p.resid + (long)row * p.ldr + col;
      if (r_vec) {
        const float4 q = *reinterpret_cast<const float4*>(rs);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      } else {
        for (int e = 0; e < nv; ++e) v[e] += rs[e];
      }
    }
    if (Cf_out) {
      float* dst = Cf_out + (long)row * ldcf_out + col;
      if (atomic) {
        for (int e = 0; e < nv; ++e) atomicAdd(dst + e, v[e]);
      } else if (cf_vec) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (acc_out) {
          const float4 q = *reinterpret_cast<const float4*>(dst);
          o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        *reinterpret_cast<float4*>(dst) = o;
      } else {
        for (int e = 0; e < nv; ++e) dst[e] = acc_out ? dst[e] + v[e] : v[e];
      }
    }
    if (p.Cb) {
      bf16_t* dst = p.Cb + (long)row * p.ldcb + col;
      if (vec && ((p.ldcb & 3) == 0)) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      } else {
        for (int e = 0; e < nv; ++e) dst[e] = f32_to_bf16(v[e]);
      }
    }
  }
  }   // 32-row passes
}

template <bool A_KC, bool B_KC, class C>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD) void gemm_glds_kernel(GemmArgs p) {
  constexpr int NSTAGE = C::RING_BYTES / C::STAGE_BYTES, BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN;
  constexpr int GLDS_PER_STAGE = C::GLDS_PER_STAGE;
  __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];   // ring of [A|B] stages, then epilogue slabs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  int tm, tn;
  tile_coords<BM, BN>(p, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = blockIdx.y * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  const int nkt = (kend - kbeg) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    char* la = smem + (kt % NSTAGE) * C::STAGE_BYTES;
    char* lb = la + C::A_BYTES;
    if (A_KC) stage_kc<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    else stage_ks<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    if (B_KC) stage_kc<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
    else stage_ks<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
  };

  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t)
    if (t < nkt) stage(t);

  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt must have landed: the tiles issued after it (at most NSTAGE-2) may stay in flight
    const int later = min(NSTAGE - 2, nkt - 1 - kt);
    if (NSTAGE >= 4 && later >= 2) wait_dma_and_barrier<2 * GLDS_PER_STAGE>();
    else if (later >= 1) wait_dma_and_barrier<1 * GLDS_PER_STAGE>();
    else wait_dma_and_barrier<0>();
    // every wave is past tile kt-1: its ring slot is free for tile kt+NSTAGE-1
#if NEKO_GEMM_DIAG != 1
    if (kt + NSTAGE - 1 < nkt) stage(kt + NSTAGE - 1);
#endif
    const char* la = smem + (kt % NSTAGE) * C::STAGE_BYTES;
    const char* lb = la + C::A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_v a[TM], b[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = B_KC ? frag_kc(lb, (wn * TN + j) * 32, ks, lane) : frag_ks<BN>(lb, (wn * TN + j) * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = A_KC ? frag_kc(la, (wm * TM + i) * 32, ks, lane) : frag_ks<BM>(la, (wm * TM + i) * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();   // all waves done with the ring before the slabs overwrite it
  epilogue<C>(p, acc, smem, m0, n0, wm, wn, wave, lane);
}

// ---- ping-pong variant (WM == 2, one wave of each M-half per SIMD) ------------------------------------------
// The two M-halves of the block run one barrier apart: while one half issues its 16 MFMAs of k-tile t, the other
// half reads its fragments of k-tile t (12 ds_read_b128) and issues its share of the DMA for tile t+NSTAGE-1, then
// they swap.  Interval n (between barriers n and n+1):  half 0: LOAD(t) at n = 2t, MFMA(t) at 2t+1;
//                                                       half 1: LOAD(t) at n = 2t+1, MFMA(t) at 2t+2.
//   RAW (DMA -> ds_read): every wave waits for its own pieces of tile t+1 (counted vmcnt) at the end of its LOAD(t),
//       i.e. in an interval <= 2t+1; tile t+1 is first read in interval 2t+2.
//   WAR (ds_read -> DMA): tile t+NSTAGE-1 reuses the slot of tile t-1, last read by half 1 in interval 2t-1 (retired
//       by lgkmcnt(0) before barrier 2t); it is issued in LOAD(t), interval >= 2t.
template <bool A_KC, bool B_KC, class C>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD) void gemm_pp_kernel(GemmArgs p) {
  static_assert(C::WM == 2, "ping-pong needs exactly two M-halves");
  constexpr int NSTAGE = C::RING_BYTES / C::STAGE_BYTES, BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN;
  constexpr int GL = C::GLDS_PER_STAGE, D = NSTAGE - 1;
  static_assert(NSTAGE == 4, "wait immediates below assume 3 tiles ahead");
  __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  int tm, tn;
  tile_coords<BM, BN>(p, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = blockIdx.y * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  const int nkt = (kend - kbeg) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    char* la = smem + (kt % NSTAGE) * C::STAGE_BYTES;
    char* lb = la + C::A_BYTES;
    if (A_KC) stage_kc<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    else stage_ks<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    if (B_KC) stage_kc<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
    else stage_ks<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
  };

#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nkt) stage(t);
  {
    const int later = min(D - 1, nkt - 1);          // tiles issued after tile 0
    if (later >= 2) wait_dma_and_barrier<2 * GL>();
    else if (later == 1) wait_dma_and_barrier<GL>();
    else wait_dma_and_barrier<0>();
  }
  if (wm == 1) __builtin_amdgcn_s_barrier();        // half 1 runs one interval behind

#if NEKO_GEMM_DIAG == 2
  bf16x8_v a[BK / 16][TM] = {}, b[BK / 16][TN] = {};
#endif
  for (int t = 0; t < nkt; ++t) {
    // ---- LOAD(t)
    const char* la = smem + (t % NSTAGE) * C::STAGE_BYTES;
    const char* lb = la + C::A_BYTES;
#if NEKO_GEMM_DIAG != 2
    bf16x8_v a[BK / 16][TM], b[BK / 16][TN];
#else
    if (t == 0 || p.K == 12345)
#endif
    {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[ks][j] = B_KC ? frag_kc(lb, (wn * TN + j) * 32, ks, lane) : frag_ks<BN>(lb, (wn * TN + j) * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[ks][i] = A_KC ? frag_kc(la, (wm * TM + i) * 32, ks, lane) : frag_ks<BM>(la, (wm * TM + i) * 32, ks, lane);
    }
    }
#if NEKO_GEMM_DIAG != 1
    if (t + D < nkt) stage(t + D);
#endif
    {
      const int later = min(D - 1, nkt - 2 - t);    // tiles issued after tile t+1 (negative: nothing left to wait for)
      if (later >= 2) wait_dma_and_barrier<2 * GL>();
      else if (later == 1) wait_dma_and_barrier<GL>();
      else wait_dma_and_barrier<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA(t)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#if NEKO_GEMM_DIAG == 3
          acc[i][j][0] += __builtin_bit_cast(float, (int)a[ks][i][0] ^ (int)b[ks][j][1]);
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
#endif
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();        // match half 1's extra barrier: all ring reads are retired after it
  epilogue<C>(p, acc, smem, m0, n0, wm, wn, wave, lane);
}

// ---- ping-pong variant 2: as above, but the DMA pieces of tile t+NSTAGE-1 are issued BETWEEN the MFMA groups of
// MFMA(t) (cheapest issue slot) and the LOAD phase carries only the 12 ds_reads; lgkmcnt(0) is taken after the barrier.
// RAW: tile t+1 (issued in MFMA(t-2)) is waited for at the end of LOAD(t) with one later tile (t+2) allowed in flight.
// The two M-halves of the block run one barrier apart: while one half issues its 16 MFMAs of k-tile t, the other
// half reads its fragments of k-tile t (12 ds_read_b128) and issues its share of the DMA for tile t+NSTAGE-1, then
// they swap.  Interval n (between barriers n and n+1):  half 0: LOAD(t) at n = 2t, MFMA(t) at 2t+1;
//                                                       half 1: LOAD(t) at n = 2t+1, MFMA(t) at 2t+2.
//   RAW (DMA -> ds_read): every wave waits for its own pieces of tile t+1 (counted vmcnt) at the end of its LOAD(t),
//       i.e. in an interval <= 2t+1; tile t+1 is first read in interval 2t+2.
//   WAR (ds_read -> DMA): tile t+NSTAGE-1 reuses the slot of tile t-1, last read by half 1 in interval 2t-1 (retired
//       by lgkmcnt(0) before barrier 2t); it is issued in LOAD(t), interval >= 2t.
template <bool A_KC, bool B_KC, class C>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD) void gemm_pp2_kernel(GemmArgs p) {
  static_assert(C::WM == 2, "ping-pong needs exactly two M-halves");
  constexpr int NSTAGE = C::RING_BYTES / C::STAGE_BYTES, BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN;
  constexpr int GL = C::GLDS_PER_STAGE, D = NSTAGE - 1;
  static_assert(NSTAGE == 4, "wait immediates below assume 3 tiles ahead");
  __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  int tm, tn;
  tile_coords<BM, BN>(p, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = blockIdx.y * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  const int nkt = (kend - kbeg) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    char* la = smem + (kt % NSTAGE) * C::STAGE_BYTES;
    char* lb = la + C::A_BYTES;
    if (A_KC) stage_kc<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    else stage_ks<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    if (B_KC) stage_kc<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
    else stage_ks<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
  };

#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nkt) stage(t);
  {
    const int later = min(D - 1, nkt - 1);          // tiles issued after tile 0
    if (later >= 2) wait_dma_and_barrier<2 * GL>();
    else if (later == 1) wait_dma_and_barrier<GL>();
    else wait_dma_and_barrier<0>();
  }
  if (wm == 1) __builtin_amdgcn_s_barrier();        // half 1 runs one interval behind

#if NEKO_GEMM_DIAG == 2
  bf16x8_v a[BK / 16][TM] = {}, b[BK / 16][TN] = {};
#endif
  for (int t = 0; t < nkt; ++t) {
    // ---- LOAD(t)
    const char* la = smem + (t % NSTAGE) * C::STAGE_BYTES;
    const char* lb = la + C::A_BYTES;
#if NEKO_GEMM_DIAG != 2
    bf16x8_v a[BK / 16][TM], b[BK / 16][TN];
#else
    if (t == 0 || p.K == 12345)
#endif
    {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[ks][j] = B_KC ? frag_kc(lb, (wn * TN + j) * 32, ks, lane) : frag_ks<BN>(lb, (wn * TN + j) * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[ks][i] = A_KC ? frag_kc(la, (wm * TM + i) * 32, ks, lane) : frag_ks<BM>(la, (wm * TM + i) * 32, ks, lane);
    }
    }
    {
      // issued so far: tiles <= t+2; tile t+1 must have landed
      if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::"n"(GL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool issue = t + D < nkt;
    const int k0n = kbeg + (t + D) * BK;
    char* lan = smem + ((t + D) % NSTAGE) * C::STAGE_BYTES;
    char* lbn = lan + C::A_BYTES;
    // ---- MFMA(t)
    __builtin_amdgcn_s_setprio(1);
    static_assert(TM == 4 && TN == 2 && GL == 4, "interleave below is written for 128x64 wave tiles");
#pragma unroll
    for (int g = 0; g < 4; ++g) {            // g = (ks, i-pair): 4 MFMAs, then one DMA piece
      const int ks = g >> 1, i0 = (g & 1) * 2;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i0 + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i0 + ii], b[ks][j], acc[i0 + ii][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#if NEKO_GEMM_DIAG != 1
      if (issue) {
        if (g < 2) stage_piece<BM, C::NW, A_KC>(p.A, p.lda, m0, p.M, k0n, lan, wave, lane, g);
        else stage_piece<BN, C::NW, B_KC>(p.B, p.ldb, n0, p.N, k0n, lbn, wave, lane, g - 2);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();        // match half 1's extra barrier: all ring reads are retired after it
  epilogue<C>(p, acc, smem, m0, n0, wm, wn, wave, lane);
}

// 128x128: 4 waves x (64x64), 3 blocks/CU (3-stage) or 2 (4-stage)      -- short K / ragged or small outputs
// 256x128: 4 waves x (128x64), 3-stage 72 KB ring -> 2 blocks/CU          -- wide-M shapes with N a multiple of 128
// 256x256: 8 waves x (128x64), 4-stage 128 KB ring -> 1 block/CU          -- long-K shapes (weight gradients, LM-head dH)
using C128s3 = Cfg<2, 2, 2, 2, 3>;
using C128s4 = Cfg<2, 2, 2, 2, 4>;
using C256x128 = Cfg<2, 2, 4, 2, 3>;
using C256x256 = Cfg<2, 4, 4, 2, 4>;

template <bool A_KC, bool B_KC, class C>
int launch_cfg(const GemmArgs& a, hipStream_t s) {
  const int nbm = (a.M + C::BM - 1) / C::BM, nbn = (a.N + C::BN - 1) / C::BN;
  dim3 grid(nbm * nbn, a.splitk > 1 ? a.splitk : 1);
  hipLaunchKernelGGL((gemm_glds_kernel<A_KC, B_KC, C>), grid, dim3(C::NT), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

template <bool A_KC, bool B_KC, class C>
int launch_pp(const GemmArgs& a, hipStream_t s) {
  const int nbm = (a.M + C::BM - 1) / C::BM, nbn = (a.N + C::BN - 1) / C::BN;
  dim3 grid(nbm * nbn, a.splitk > 1 ? a.splitk : 1);
  hipLaunchKernelGGL((gemm_pp_kernel<A_KC, B_KC, C>), grid, dim3(C::NT), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

template <bool A_KC, bool B_KC, class C>
int launch_pp2(const GemmArgs& a, hipStream_t s) {
  const int nbm = (a.M + C::BM - 1) / C::BM, nbn = (a.N + C::BN - 1) / C::BN;
  dim3 grid(nbm * nbn, a.splitk > 1 ? a.splitk : 1);
  hipLaunchKernelGGL((gemm_pp2_kernel<A_KC, B_KC, C>), grid, dim3(C::NT), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// tile configuration: NEKO_GEMM_TILE=0..3 forces one (benchmarking); default heuristic below
int forced_tile() {
  static const int v = [] { const char* e = getenv("NEKO_GEMM_TILE"); return e ? atoi(e) : -1; }();
  return v;
}

// Per-shape choice, from tools/gemm_bench.py on the shapes of the 768d step (MI355X, random operands):
//   long contraction (weight gradients incl. split-K slices, LM-head dH/dW): 256x256, +5..15 % over 128x128;
//   A k-contiguous x B k-strided (forward) with N >= 2048: 256x256 (+6..8 %);
//   both k-contiguous (dgrad) with K >= 2048: 256x128 (+7..9 %); K = 768 dgrad keeps 128x128 (GELU' epilogue
//   overlaps better at 3 blocks/CU);
//   everything else 128x128 (3-stage, 3 blocks/CU).
template <bool A_KC, bool B_KC>
int launch(const GemmArgs& a, hipStream_t s) {
  const int klen = a.splitk > 1 ? a.k_per_split : a.K;
  int cfg = forced_tile();
  if (cfg < 0) {
    const bool big_out = (long)a.M * a.N >= (long)256 * 256 * 64;      // enough 256x256 tiles to matter
    if (a.splitk > 1 || (klen >= 4096 && (long)a.M * a.N >= (long)256 * 256 * 8)) cfg = 3;
    else if (A_KC && !B_KC && a.N >= 2048 && big_out) cfg = 3;
    else if (A_KC && B_KC && klen >= 2048 && big_out && a.N % 128 == 0) cfg = 2;
    else cfg = klen >= 16384 ? 1 : 0;
  }
  switch (cfg) {
    case 1: return launch_cfg<A_KC, B_KC, C128s4>(a, s);
    case 2: return launch_cfg<A_KC, B_KC, C256x128>(a, s);
    case 3: return launch_cfg<A_KC, B_KC, C256x256>(a, s);
    case 4: return launch_pp<A_KC, B_KC, C256x256>(a, s);
    case 5: return launch_pp2<A_KC, B_KC, C256x256>(a, s);
    default: return launch_cfg<A_KC, B_KC, C128s3>(a, s);
  }
}

}  // namespace

// returns 1 if the fast path does not apply (caller falls back), otherwise a neko status code
int neko_gemm_glds_try(const GemmArgs& a, int a_kstrided, int b_kstrided, hipStream_t s) {
  if (a.K % 64) return 1;
  if (a.splitk > 1 && (a.k_per_split % 64)) return 1;
  // 16-B aligned operands / outputs (all neko_amd buffers are; guards foreign callers)
  if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.B)) & 15) return 1;
  if (a_kstrided && a.M < 8) return 1;
  if (b_kstrided && a.N < 8) return 1;
  if (a.Cf && (reinterpret_cast<uintptr_t>(a.Cf) & 15)) return 1;
  if (a.Cb && (reinterpret_cast<uintptr_t>(a.Cb) & 7)) return 1;
  if (a.resid && (reinterpret_cast<uintptr_t>(a.resid) & 15)) return 1;
  if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return 1;
  if (a.act_in && ((reinterpret_cast<uintptr_t>(a.act_in) & 7) || (a.ldact & 3))) return 1;
  if (a.pre_out && ((reinterpret_cast<uintptr_t>(a.pre_out) & 7) || (a.ldpre & 3))) return 1;
  if (a_kstrided && b_kstrided) return launch<false, false>(a, s);
  if (a_kstrided) return launch<false, true>(a, s);
  if (b_kstrided) return launch<true, false>(a, s);
  return launch<true, true>(a, s);
}
