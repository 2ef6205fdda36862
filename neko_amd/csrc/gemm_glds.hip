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
#include "gemm_epi.h"

#if NEKO_GEMM_DIAG == 9
// phase trace (diagnostic builds only): per block 4 x s_memrealtime (100 MHz) = start, first tile landed, k-loop done,
// epilogue done (and, for the first 32768 blocks, s_memtime at the same points in the second half of the buffer); buffer set with neko_gemm_diag_trace()
__device__ unsigned long long* g_neko_gemm_trace = nullptr;
extern "C" int neko_gemm_diag_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_neko_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define NEKO_TRACE(slot)                                                                                       \
  do {                                                                                                         \
    if (g_neko_gemm_trace && threadIdx.x == 0) {                                                               \
      g_neko_gemm_trace[(long)blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memrealtime();                     \
      if (blockIdx.x < 32768) g_neko_gemm_trace[131072 + (long)blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memtime(); \
    }                                                                                                          \
  } while (0)
#else
#define NEKO_TRACE(slot) do { } while (0)
#endif

// Per-CU output-phase lock (round 5): two co-resident workgroups of a CU may not be in their output phase (epilogue arithmetic +
// stores) at the same time, so one's output phase falls under the other's main loop instead of both idling the matrix pipe together.
// Indexed by the hardware CU id (XCC id, shader engine, shader array, CU); only timing depends on it, never results.
__device__ unsigned g_neko_cu_lock[4096 * 16];

namespace {

__device__ __forceinline__ unsigned* cu_lock_ptr() {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID [3:0]
  return &g_neko_cu_lock[((((hw >> 8) & 0xffu) | ((xcc & 0xfu) << 8)) & 4095u) * 16u];
}

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

// The in-loop form: wave-uniform 64-bit base in SGPRs + a loop-invariant 32-bit per-lane byte offset, LDS destination
// handed over IN M0 (register constraint: the compiler writes M0 itself, one s_add, and knows it is live) -- three
// instructions per piece (s_add m0 / s_nop / global_load_lds) where the generic-pointer form above costs eleven
// (64-bit VALU address add, generic->LDS cast with its null check, M0 save and restore).  In the 4-wave / 8-wave main
// loops a wave issues 4-8 of these per k-tile between 16-32 MFMAs: the saved issue slots go to the matrix pipe.
__device__ __forceinline__ void glds16_s(const bf16_t* base_uniform, unsigned byte_off, unsigned lds_dst_wave_uniform) {
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(byte_off), "s"(base_uniform), "{m0}"(lds_dst_wave_uniform)
               : "memory");
}
template <int EXT, int NW>
__device__ __forceinline__ unsigned piece_off_kc(long ld, int r0, int nrows, int wave, int lane, int i) {
  constexpr int PER = EXT / 16 / NW;
  const int row = (wave * PER + i) * 16 + (lane >> 2);
  const int piece = (lane & 3) ^ ((row >> 2) & 3);
  const int gr = min(r0 + row, nrows - 1) - r0;
  return (unsigned)((gr * ld + piece * 8) * 2);
}
template <int EXT, int NW>
__device__ __forceinline__ unsigned piece_off_ks(long ld, int c0, int ncols, int wave, int lane, int i) {
  constexpr int PER = EXT / 16 / NW;
  constexpr int PPR = EXT / 8;
  const int kr = (wave * PER + i) * (64 / PPR) + lane / PPR;
  const int piece = (lane % PPR) ^ ((kr & 3) << 2);
  const int gc = min(c0 + piece * 8, ncols - 8);
  return (unsigned)((kr * ld + gc) * 2);
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

// one wave-instruction (piece i of this wave's PER) of the two stagers above: the main loop issues a stage piece by
// piece between its MFMA groups
template <int EXT, int NW>
__device__ __forceinline__ void stage_kc_piece(const bf16_t* __restrict__ P, long ld, int r0, int nrows, int k0, char* lds,
                                               int wave, int lane, int i) {
  constexpr int PER = EXT / 16 / NW;
  const int chunk = wave * PER + i;
  const int row = chunk * 16 + (lane >> 2);
  const int piece = (lane & 3) ^ ((row >> 2) & 3);
  const int gr = min(r0 + row, nrows - 1);
  glds16(P + (long)gr * ld + k0 + piece * 8, lds + chunk * 1024);
}
template <int EXT, int NW>
__device__ __forceinline__ void stage_ks_piece(const bf16_t* __restrict__ P, long ld, int c0, int ncols, int k0, char* lds,
                                               int wave, int lane, int i) {
  constexpr int PER = EXT / 16 / NW;
  constexpr int PPR = EXT / 8;
  const int chunk = wave * PER + i;
  const int kr = chunk * (64 / PPR) + lane / PPR;
  const int piece = (lane % PPR) ^ ((kr & 3) << 2);
  const int gc = min(c0 + piece * 8, ncols - 8);
  glds16(P + (long)(k0 + kr) * ld + gc, lds + chunk * 1024);
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
// same without draining this wave's LDS reads: the pipelined loop keeps operand reads in flight across the barrier
template <int N>
__device__ __forceinline__ void wait_dma_only_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- epilogue (shared by both kernels): the caller has passed a block barrier after the last ring read -------
template <class C>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x16 (&acc)[C::TM][C::TN], char* smem, int m0, int n0,
                                         int wm, int wn, int wave, int lane, int slice) {
  constexpr int TM = C::TM, TN = C::TN;
#if NEKO_GEMM_DIAG == 4
  if (p.M != 12345) return;      // ablation: no epilogue at all
#endif
  // ---- epilogue through this wave's private slab, TM passes of 32 rows ---------------------------------------
  constexpr int SW = 32 * TN;              // slab row length (f32)
  constexpr int CPR = SW / 4;              // float4 chunks per slab row (16 or 32)
  constexpr int RPI = 64 / CPR;            // slab rows one wave-instruction covers
  float* slab = reinterpret_cast<float*>(smem + wave * C::SLAB_BYTES);   // float4 chunk index ^= row & 15
  // same-wave LDS write -> read: the compiler's lgkmcnt wait orders it (no cross-wave sharing of a slab)
  const float alpha = p.alpha_dev ? p.alpha * (*p.alpha_dev) : p.alpha;
  const bool to_ws = p.splitk > 1 && p.splitk_ws;   // split-K slices go to a workspace, reduced in fixed order afterwards
  const bool atomic = p.splitk > 1 && !to_ws;
  float* const Cf_out = to_ws ? p.splitk_ws + (long)slice * p.M * p.N : p.Cf;
  const long ldcf_out = to_ws ? p.N : p.ldcf;
  const int acc_out = to_ws ? 0 : p.accumulate;
  const bool lead = !atomic || slice == 0;
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
        const int lr = lane & 31;                                            // transposed accumulators, see epilogue_fast
        const int lc = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
    } else if (p.act == 3) {            // GELU forward; pre_out receives gelu'(pre) (the backward multiplies by it, act = 4)
      for (int e = 0; e < nv; ++e) {
        const float x = bf16_to_f32(f32_to_bf16(v[e]));
        p.pre_out[(long)row * p.ldpre + col + e] = f32_to_bf16(gelu_grad_f(x));
        v[e] = gelu_f(x);
      }
    } else if (p.act == 2 || p.act == 4) {
      const bf16_t* src = p.act_in + (long)row * p.ldact + col;
      for (int e = 0; e < nv; ++e) {
        const float a = bf16_to_f32(src[e]);
        v[e] *= (p.act == 4) ? a : gelu_grad_f(a);
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
  if (p.drop_thr) p.drop_key += neko_drop_salt();
  constexpr int NSTAGE = C::RING_BYTES / C::STAGE_BYTES, BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN;
  constexpr int GLDS_PER_STAGE = C::GLDS_PER_STAGE;
  __shared__ __attribute__((aligned(1024))) char smem[C::LDS_BYTES];   // ring of [A|B] stages, then epilogue slabs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  int tm, tn, slice;
  tile_coords<BM, BN>(p, tm, tn, slice);
  const int m0 = tm * BM, n0 = tn * BN;

  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = slice * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  const int nkt = (kend - kbeg) / BK;

  // The MFMAs below take (B fragment, A fragment): both fragments have the same register layout (index l&31, k =
  // 8(l>>5)..+7), so swapping them yields each 32 x 32 block TRANSPOSED -- lane l then holds row l&31 and FOUR CONSECUTIVE
  // columns per register quad, and the epilogue parks a quad with one ds_write_b128 where the natural layout (a column per
  // lane, rows down the registers) took four ds_write_b32: 32 -> 8 LDS writes per 32-row pass, a third of the instructions
  // of a plain bf16 epilogue, which is issue-bound (tools/gemm_trace.py with -DNEKO_EPI_ABL: no stores 3.2 us, no slab
  // writes 3.5 us, everything 3.9 us per 256 x 256 tile).
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

  NEKO_TRACE(0);
  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t)
    if (t < nkt) stage(t);

  // one piece (wave-instruction) of tile kt's stage: pieces 0..PER_A-1 fill A, the rest B
  constexpr int PER_A = BM / 16 / C::NW, NP = GLDS_PER_STAGE, NM = (BK / 16) * TM * TN;
#ifdef NEKO_GEMM_FATPIECE
  auto stage_piece = [&](int kt, int pc) {
    const int k0 = kbeg + kt * BK;
    char* la = smem + (kt % NSTAGE) * C::STAGE_BYTES;
    char* lb = la + C::A_BYTES;
    if (pc < PER_A) {
      if (A_KC) stage_kc_piece<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane, pc);
      else stage_ks_piece<BM, C::NW>(p.A, p.lda, m0, p.M, k0, la, wave, lane, pc);
    } else {
      if (B_KC) stage_kc_piece<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane, pc - PER_A);
      else stage_ks_piece<BN, C::NW>(p.B, p.ldb, n0, p.N, k0, lb, wave, lane, pc - PER_A);
    }
  };
#else
  constexpr int PER_B = BN / 16 / C::NW;
  unsigned voffA[PER_A], voffB[PER_B];
#pragma unroll
  for (int i = 0; i < PER_A; ++i)
    voffA[i] = A_KC ? piece_off_kc<BM, C::NW>(p.lda, m0, p.M, wave, lane, i) : piece_off_ks<BM, C::NW>(p.lda, m0, p.M, wave, lane, i);
#pragma unroll
  for (int i = 0; i < PER_B; ++i)
    voffB[i] = B_KC ? piece_off_kc<BN, C::NW>(p.ldb, n0, p.N, wave, lane, i) : piece_off_ks<BN, C::NW>(p.ldb, n0, p.N, wave, lane, i);
  const bf16_t* const gA0 = A_KC ? p.A + (long)m0 * p.lda + kbeg : p.A + (long)kbeg * p.lda;
  const bf16_t* const gB0 = B_KC ? p.B + (long)n0 * p.ldb + kbeg : p.B + (long)kbeg * p.ldb;
  const long gstepA = A_KC ? (long)BK : (long)BK * p.lda, gstepB = B_KC ? (long)BK : (long)BK * p.ldb;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const unsigned ldsA = lds0 + wave * PER_A * 1024, ldsB = lds0 + C::A_BYTES + wave * PER_B * 1024;
  auto stage_piece = [&](int kt, int pc) {
    const unsigned slot = (unsigned)(kt % NSTAGE) * C::STAGE_BYTES;
    if (pc < PER_A) glds16_s(gA0 + kt * gstepA, voffA[pc], ldsA + slot + pc * 1024);
    else glds16_s(gB0 + kt * gstepB, voffB[pc - PER_A], ldsB + slot + (pc - PER_A) * 1024);
  };
#endif
  // k-tile body.  With STAGE the NP DMA pieces of tile kt+NSTAGE-1 are issued BETWEEN the MFMA groups instead of in one
  // burst behind the barrier: a burst makes all waves of the block queue NP 1-KiB requests at once on the CU's single
  // L2->LDS path (16 cycles each), every wave sits in its VMEM issue for up to NP*NW*16 cycles with no MFMA queued, and
  // the matrix pipe idles for that long every k-tile (ablation without the in-loop DMA: +19..27 % throughput).
  auto body = [&](int kt, auto stage_tag) {
    constexpr bool STAGE = decltype(stage_tag)::value;
    const char* la = smem + (kt % NSTAGE) * C::STAGE_BYTES;
    const char* lb = la + C::A_BYTES;
    int pc = 0;
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
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          const int m = (ks * TM + i) * TN + j;
          // piece pc goes after MFMA index ((pc+1)*NM)/NP - 1
          if (STAGE && pc < NP && m == ((pc + 1) * NM) / NP - 1) {
#if NEKO_GEMM_DIAG != 1
            __builtin_amdgcn_sched_barrier(0);
            stage_piece(kt + NSTAGE - 1, pc);
            __builtin_amdgcn_sched_barrier(0);
#endif
            ++pc;
          }
        }
    }
  };

#ifndef NEKO_GEMM_PIPE
#define NEKO_GEMM_PIPE 1
#endif
#ifndef NEKO_GEMM_PIPE_TN
#define NEKO_GEMM_PIPE_TN 1   // both operands k-strided too: with the fragment reads interleaved it wins there as well (wgrad -2..6 %)
#endif
#ifndef NEKO_GEMM_PIPE_MIN_STAGES
#define NEKO_GEMM_PIPE_MIN_STAGES 4   // r03: 3 = the 3-stage rings take the pipelined loop too (correct, wait counts scale with NSTAGE); measured +-2 %
                                      // on 256x128 (profiles/r03_tile_ab1.txt), not measured on 128x128: left off
#endif
  if constexpr (NSTAGE >= NEKO_GEMM_PIPE_MIN_STAGES && BK == 32 && (A_KC || B_KC || NEKO_GEMM_PIPE_TN) && NEKO_GEMM_PIPE) {
    // Pipelined loop (4-stage rings).  The block barrier of tile kt+1 sits in the MIDDLE of tile kt, between its two
    // k-steps, and the operands of a k-step are requested one k-step ahead:
    //     [F1 <- tile kt, k-step 1] [MFMAs k-step 0 on F0] [barrier: tile kt+1 visible, slot of tile kt-1 free]
    //     [F0 <- tile kt+1, k-step 0] [MFMAs k-step 1 on F1, DMA pieces of tile kt+3 between them]
    // so no MFMA group starts behind a barrier AND an LDS round trip (the plain loop's first MFMA of every k-tile did:
    // ~150-250 idle matrix-pipe cycles out of ~1600 per k-tile), and a wave reaches the barrier with its k-step-0 MFMAs
    // queued.  A ring slot is refilled half a tile later than in the plain loop (2 tiles of lead instead of 3).
    // Measured against the plain loop on one box: LM-head dH -8 %, 8192^3 NT -2.6 %, forward qkv -4 %, K = 768 GELU shapes
    // +-1 %; with BOTH operands k-strided (weight gradients: two ds_read_b64_tr_b16 per fragment, twice the LDS
    // instructions in flight) it WAS 4-7 % slower with the reads issued as a burst; with the reads interleaved (step_il below) it is 2-6 % faster
    // there too (weight gradients, LM-head dW), so every 4-stage config now takes this loop.
    auto load_frags = [&](int kt, int ks, bf16x8_v (&a)[TM], bf16x8_v (&b)[TN]) {
      const char* la = smem + (kt % NSTAGE) * C::STAGE_BYTES;
      const char* lb = la + C::A_BYTES;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = B_KC ? frag_kc(lb, (wn * TN + j) * 32, ks, lane) : frag_ks<BN>(lb, (wn * TN + j) * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = A_KC ? frag_kc(la, (wm * TM + i) * 32, ks, lane) : frag_ks<BM>(la, (wm * TM + i) * 32, ks, lane);
    };
    auto mfma_step = [&](const bf16x8_v (&a)[TM], const bf16x8_v (&b)[TN], int kt_stage, bool stage_now) {
      int pc = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          const int m = i * TN + j;
          if (stage_now && pc < NP && m == ((pc + 1) * TM * TN) / NP - 1) {
#if NEKO_GEMM_DIAG != 1
            __builtin_amdgcn_sched_barrier(0);
            stage_piece(kt_stage, pc);
            __builtin_amdgcn_sched_barrier(0);
#endif
            ++pc;
          }
        }
    };
#ifndef NEKO_GEMM_BURST
    // the TM + TN fragment reads of the NEXT k-step are spread between this step's MFMAs (one read after every
    // TM*TN/(TM+TN)-th MFMA; a[0] and every b[j] first: the next step's first MFMAs need them) instead of being issued
    // as one burst ahead of the step, which made the first MFMAs of every step wait for an LDS round trip (+3..5 %;
    // -DNEKO_GEMM_BURST rebuilds the burst form for A/B runs)
    auto step_il = [&](const bf16x8_v (&a)[TM], const bf16x8_v (&b)[TN], bf16x8_v (&an)[TM], bf16x8_v (&bn)[TN],
                       int kt_load, int ks_load, bool do_load, int kt_stage, bool stage_now) {
      const char* la = smem + (kt_load % NSTAGE) * C::STAGE_BYTES;
      const char* lb = la + C::A_BYTES;
      int pc = 0, lc = 0;
      constexpr int NL = TM + TN;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          const int m = i * TN + j;
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            if (l == lc && do_load && m == (l * TM * TN) / NL) {
              __builtin_amdgcn_sched_barrier(0);
              // order: a0, b0 .. b(TN-1), a1 .. a(TM-1)
              if (l == 0) an[0] = A_KC ? frag_kc(la, (wm * TM) * 32, ks_load, lane) : frag_ks<BM>(la, (wm * TM) * 32, ks_load, lane);
              else if (l <= TN) bn[l - 1] = B_KC ? frag_kc(lb, (wn * TN + l - 1) * 32, ks_load, lane)
                                                  : frag_ks<BN>(lb, (wn * TN + l - 1) * 32, ks_load, lane);
              else an[l - TN] = A_KC ? frag_kc(la, (wm * TM + l - TN) * 32, ks_load, lane)
                                     : frag_ks<BM>(la, (wm * TM + l - TN) * 32, ks_load, lane);
              __builtin_amdgcn_sched_barrier(0);
              ++lc;
            }
          }
          if (stage_now && pc < NP && m == ((pc + 1) * TM * TN) / NP - 1) {
            __builtin_amdgcn_sched_barrier(0);
            stage_piece(kt_stage, pc);
            __builtin_amdgcn_sched_barrier(0);
            ++pc;
          }
        }
    };
#endif
    bf16x8_v a0[TM], b0[TN], a1[TM], b1[TN];
    if (nkt > 0) {
      // tile 0 visible (tiles 1 .. NSTAGE-2 may still be in flight)
      if (nkt >= NSTAGE - 1) wait_dma_only_and_barrier<(NSTAGE - 2) * GLDS_PER_STAGE>();
      else wait_dma_only_and_barrier<0>();
      NEKO_TRACE(1);
      load_frags(0, 0, a0, b0);
    }
    const int nmain = max(0, nkt - (NSTAGE - 1));
#ifndef NEKO_GEMM_BURST
    for (int kt = 0; kt < nmain; ++kt) {
      step_il(a0, b0, a1, b1, kt, 1, true, 0, false);
      wait_dma_only_and_barrier<(NSTAGE - 3) * GLDS_PER_STAGE>();
      step_il(a1, b1, a0, b0, kt + 1, 0, true, kt + NSTAGE - 1, true);
    }
#else
    for (int kt = 0; kt < nmain; ++kt) {
      load_frags(kt, 1, a1, b1);
      mfma_step(a0, b0, 0, false);
      // tile kt+1 landed everywhere (only tile kt+2 of this wave's requests may be outstanding); every wave has consumed
      // tile kt-1, whose slot the pieces below refill with tile kt+NSTAGE-1
#if NEKO_GEMM_DIAG == 11      // ablation (wrong results): block barrier only on every other k-tile, the DMA wait stays
      if (kt & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 3) * GLDS_PER_STAGE) : "memory");
      else wait_dma_only_and_barrier<(NSTAGE - 3) * GLDS_PER_STAGE>();
#elif NEKO_GEMM_DIAG == 12    // ablation (wrong results): no block barrier at all in the main loop
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 3) * GLDS_PER_STAGE) : "memory");
#else
      wait_dma_only_and_barrier<(NSTAGE - 3) * GLDS_PER_STAGE>();
#endif
      load_frags(kt + 1, 0, a0, b0);
      mfma_step(a1, b1, kt + NSTAGE - 1, true);
    }
#endif
    for (int kt = nmain; kt < nkt; ++kt) {      // drain: nothing left to request
      load_frags(kt, 1, a1, b1);
      mfma_step(a0, b0, 0, false);
      if (kt + 1 < nkt) {
        wait_dma_only_and_barrier<0>();
        load_frags(kt + 1, 0, a0, b0);
      }
      mfma_step(a1, b1, 0, false);
    }
  } else {
  // main part: tile kt must have landed, the NSTAGE-2 tiles issued after it stay in flight; every wave is past tile
  // kt-1 after the barrier, so its ring slot is free for tile kt+NSTAGE-1
  const int nmain = max(0, nkt - (NSTAGE - 1));
  for (int kt = 0; kt < nmain; ++kt) {
    wait_dma_and_barrier<(NSTAGE - 2) * GLDS_PER_STAGE>();
#if NEKO_GEMM_DIAG == 9
    if (kt == 0) NEKO_TRACE(1);
#endif
    body(kt, std::true_type{});
  }
  // drain: no tile left to request
  for (int kt = nmain; kt < nkt; ++kt) {
    const int later = nkt - 1 - kt;
    if (NSTAGE >= 4 && later >= 2) wait_dma_and_barrier<2 * GLDS_PER_STAGE>();
    else if (later >= 1) wait_dma_and_barrier<1 * GLDS_PER_STAGE>();
    else wait_dma_and_barrier<0>();
    body(kt, std::false_type{});
  }
  }
  __syncthreads();   // all waves done with the ring before the slabs overwrite it
  NEKO_TRACE(2);
#if NEKO_GEMM_DIAG == 9
  if (!try_epilogue_fast<C>(p, ParkAcc32<C>{acc}, smem, m0, n0, wm, wn, wave, lane, slice)) epilogue<C>(p, acc, smem, m0, n0, wm, wn, wave, lane, slice);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  NEKO_TRACE(3);
  return;
#endif
  unsigned* lock = nullptr;
  if (p.epi_lock) {
    lock = cu_lock_ptr();
    if (tid == 0) {
      while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(16);
    }
    __syncthreads();
  }
  if (!try_epilogue_fast<C>(p, ParkAcc32<C>{acc}, smem, m0, n0, wm, wn, wave, lane, slice))
    epilogue<C>(p, acc, smem, m0, n0, wm, wn, wave, lane, slice);
  if (p.epi_lock) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) atomicExch(lock, 0u);
  }
}

// 128x128: 4 waves x (64x64), 3 blocks/CU (3-stage) or 2 (4-stage)      -- short K / ragged or small outputs
// 256x128: 4 waves x (128x64), 3-stage 72 KB ring -> 2 blocks/CU          -- wide-M shapes with N a multiple of 128
// 256x256: 8 waves x (128x64), 4-stage 128 KB ring -> 1 block/CU          -- long-K shapes (weight gradients, LM-head dH)
using C128s3 = Cfg<2, 2, 2, 2, 3>;
using C128s4 = Cfg<2, 2, 2, 2, 4>;
using C256x128 = Cfg<2, 2, 4, 2, 3>;
using C256x256 = Cfg<2, 4, 4, 2, 4>;
using C256x256w4 = Cfg<2, 2, 4, 4, 4>;     // 4 waves x (128 x 128), one wave per SIMD, 256 accumulator AGPRs (the vendor kernel's geometry)


// ---- A k-contiguous in 64-k slots of WHOLE 128-B lines (round 5; NEKO_GEMM_KC64=0 returns to 32-k stages) ---------------------------
// The 8-wave 256 x 256 loop above requests a k-contiguous A tile as 64-B half lines (32 k per stage); gemm_a16.hip measured 4-10 % from
// requesting whole lines for that operand (two 64-k slots).  Same here (profiles/r05_gemm_kc64_ab.txt: -1...-6 % per launch, bit-identical): A lives in THREE slots of [256 rows][64 k] (96 KB: a slot is requested
// three k-tiles before its first use), B keeps its four 32-k stages (64 KB) -- 160 KB, one workgroup per CU as before.  A slot row is 128 B
// and its 16-B chunk index is XORed with row & 7 on the DMA source side; fragments read chunk (4 (kt & 1) + 2 ks + lane / 32) ^ (row & 7).
// Per k-tile a wave issues its two B pieces, and on even k-tiles also the four A pieces of the slot two ahead; the counted waits follow that
// order (entry 8, middle of an even k-tile 2, of an odd one 6 -- or 2 when no slot was requested before it).
template <bool B_KC>
__global__ __launch_bounds__(512, 2) void gemm_glds64_kernel(GemmArgs p) {
  using C = Cfg<2, 4, 4, 2, 4>;
  if (p.drop_thr) p.drop_key += neko_drop_salt();
  constexpr int BM = 256, BN = 256, TM = 4, TN = 2, NW = 8;
  constexpr int A_SLOT = 32768, A_REGION = 3 * A_SLOT, B_STAGE = 16384;
  __shared__ __attribute__((aligned(1024))) char smem[A_REGION + 4 * B_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  int tm, tn, slice;
  tile_coords<BM, BN>(p, tm, tn, slice);
  const int m0 = tm * BM, n0 = tn * BN;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = slice * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  const int nkt = (kend - kbeg) / BK;            // even, >= 4 (the launcher checks)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA sources: A piece q of this wave = rows (4 wave + q) * 8 .. + 7 of a slot, lane -> (row = lane / 8, chunk = lane & 7)
  unsigned voffA[4], voffB[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (wave * 4 + q) * 8 + (lane >> 3);
    const int gr = min(m0 + row, p.M - 1) - m0;
    voffA[q] = (unsigned)((gr * p.lda + (((lane & 7) ^ (row & 7)) << 3)) * 2);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
    voffB[i] = B_KC ? piece_off_kc<BN, NW>(p.ldb, n0, p.N, wave, lane, i) : piece_off_ks<BN, NW>(p.ldb, n0, p.N, wave, lane, i);
  const bf16_t* const gA0 = p.A + (long)m0 * p.lda + kbeg;
  const bf16_t* const gB0 = B_KC ? p.B + (long)n0 * p.ldb + kbeg : p.B + (long)kbeg * p.ldb;
  const long gstepB = B_KC ? (long)BK : (long)BK * p.ldb;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const unsigned ldsA = lds0 + wave * 4 * 1024, ldsB = lds0 + A_REGION + wave * 2 * 1024;
  auto issue_a = [&](int S, int q) { glds16_s(gA0 + (long)S * 64, voffA[q], ldsA + (unsigned)(S % 3) * A_SLOT + q * 1024); };
  auto issue_b = [&](int kt, int i) { glds16_s(gB0 + kt * gstepB, voffB[i], ldsB + (unsigned)(kt & 3) * B_STAGE + i * 1024); };
  auto frag_a = [&](int kt, int ks, int i) {
    const int row = (wm * TM + i) * 32 + (lane & 31);
    const int c = ((kt & 1) << 2) + ks * 2 + (lane >> 5);
    const uint4 v = *reinterpret_cast<const uint4*>(smem + ((kt >> 1) % 3) * A_SLOT + row * 128 + ((c ^ (row & 7)) << 4));
    return __builtin_bit_cast(bf16x8_v, v);
  };
  auto frag_b = [&](int kt, int ks, int j) {
    const char* lb = smem + A_REGION + (kt & 3) * B_STAGE;
    return B_KC ? frag_kc(lb, (wn * TN + j) * 32, ks, lane) : frag_ks<BN>(lb, (wn * TN + j) * 32, ks, lane);
  };

  // prologue: slot 0, B0, slot 1, B1, B2 (14 pieces; the first six must have landed at the loop entry)
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_a(0, q);
  issue_b(0, 0); issue_b(0, 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_a(1, q);
  issue_b(1, 0); issue_b(1, 1);
  issue_b(2, 0); issue_b(2, 1);

  auto load_frags = [&](int kt, int ks, bf16x8_v (&a)[TM], bf16x8_v (&b)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = frag_b(kt, ks, j);
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = frag_a(kt, ks, i);
  };
  auto mfma_step = [&](const bf16x8_v (&a)[TM], const bf16x8_v (&b)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
  };
  // one k-step: TM * TN MFMAs with the TM + TN fragment reads of the NEXT k-step and NPC DMA pieces spread between them (as step_il above)
  auto step = [&](const bf16x8_v (&a)[TM], const bf16x8_v (&b)[TN], bf16x8_v (&an)[TM], bf16x8_v (&bn)[TN], int kt_load, int ks_load,
                  auto npc_tag, int kt_now) {
    constexpr int NPC = decltype(npc_tag)::value;      // 0: no DMA, 2: B pieces of tile kt_now + 3, 6: + the A slot kt_now / 2 + 2
    constexpr int NL = TM + TN;
    int pc = 0, lc = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        const int m = i * TN + j;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          if (l == lc && m == (l * TM * TN) / NL) {
            __builtin_amdgcn_sched_barrier(0);
            if (l == 0) an[0] = frag_a(kt_load, ks_load, 0);
            else if (l <= TN) bn[l - 1] = frag_b(kt_load, ks_load, l - 1);
            else an[l - TN] = frag_a(kt_load, ks_load, l - TN);
            __builtin_amdgcn_sched_barrier(0);
            ++lc;
          }
        }
        if constexpr (NPC > 0) {
          if (pc < NPC && m == ((pc + 1) * TM * TN) / NPC - 1) {
            __builtin_amdgcn_sched_barrier(0);
            if (NPC == 6 && pc < 4) issue_a(kt_now / 2 + 2, pc);
            else issue_b(kt_now + 3, NPC == 6 ? pc - 4 : pc);
            __builtin_amdgcn_sched_barrier(0);
            ++pc;
          }
        }
      }
  };

  bf16x8_v a0[TM], b0[TN], a1[TM], b1[TN];
  wait_dma_only_and_barrier<8>();        // slot 0 and B0 visible
  load_frags(0, 0, a0, b0);
  // k-tiles in (even, odd) pairs -- a static schedule, no branch in the loop: the even tile requests the A slot two ahead (tiles kt + 4,
  // kt + 5) and B of tile kt + 3, the odd one B of tile kt + 4; what may still be in flight at the middle of a tile follows from that order
  int kt = 0;
  for (; kt + 4 < nkt; kt += 2) {
    step(a0, b0, a1, b1, kt, 1, std::integral_constant<int, 0>{}, kt);
    wait_dma_only_and_barrier<2>();                                       // B of tile kt + 2 may be in flight
    step(a1, b1, a0, b0, kt + 1, 0, std::integral_constant<int, 6>{}, kt);
    step(a0, b0, a1, b1, kt + 1, 1, std::integral_constant<int, 0>{}, kt + 1);
    wait_dma_only_and_barrier<6>();                                       // the slot and the B tile just requested may be in flight
    step(a1, b1, a0, b0, kt + 2, 0, std::integral_constant<int, 2>{}, kt + 1);
  }
  // kt == nkt - 4: the last B tile (nkt - 1) is still to be requested, every A slot has been
  step(a0, b0, a1, b1, kt, 1, std::integral_constant<int, 0>{}, kt);
  wait_dma_only_and_barrier<2>();
  step(a1, b1, a0, b0, kt + 1, 0, std::integral_constant<int, 2>{}, kt);
  for (int kd = nkt - 3; kd < nkt; ++kd) {      // drain: nothing left to request
    load_frags(kd, 1, a1, b1);
    mfma_step(a0, b0);
    if (kd + 1 < nkt) {
      wait_dma_only_and_barrier<0>();
      load_frags(kd + 1, 0, a0, b0);
    }
    mfma_step(a1, b1);
  }
  __syncthreads();   // all waves done with the ring before the slabs overwrite it
  if (!try_epilogue_fast<C>(p, ParkAcc32<C>{acc}, smem, m0, n0, wm, wn, wave, lane, slice))
    epilogue<C>(p, acc, smem, m0, n0, wm, wn, wave, lane, slice);
}

thread_local int t_colsum_bands = 0;

template <bool A_KC, bool B_KC, class C>
int launch_cfg(const GemmArgs& a_in, hipStream_t s) {
  GemmArgs a = a_in;
  const int nbm = (a.M + C::BM - 1) / C::BM, nbn = (a.N + C::BN - 1) / C::BN;
  // column sums ride along only when EVERY tile takes the compiled fast epilogue that carries them (interior tiles of a
  // plain GELU' dgrad); otherwise the caller runs the stand-alone column-sum kernel on the stored result
  const bool fold = a.colsum_ws && a.M % C::BM == 0 && a.N % C::BN == 0 && a.splitk <= 1 && (a.act == 2 || a.act == 4) && a.Cb && !a.Cf &&
                    !a.bias && !a.resid && !a.drop_thr && a.alpha == 1.0f && !a.alpha_dev &&
                    !(((a.ldcb | a.ldact) & 3) || (a.N & 3));
  if (!fold) a.colsum_ws = nullptr;
  t_colsum_bands = fold ? a.M / (32 * C::TM) : 0;
  dim3 grid(nbm * nbn * (a.splitk > 1 ? a.splitk : 1));
  if constexpr (A_KC && std::is_same<C, Cfg<2, 4, 4, 2, 4>>::value) {
    static const int kc64 = [] { const char* e = getenv("NEKO_GEMM_KC64"); return e ? atoi(e) : 1; }();      // 0: the 32-k stages for A too (A/B runs)
    const int klen = a.splitk > 1 ? a.k_per_split : a.K;
    const long last = a.splitk > 1 ? (long)a.K - (long)(a.splitk - 1) * a.k_per_split : a.K;
    // whole 64-k slots in every slice, at least two of them, 32-bit per-lane byte offsets
    if (kc64 && klen % 64 == 0 && klen >= 128 && last % 64 == 0 && last >= 128 && 256L * a.lda * 2 < (1L << 31)) {
      hipLaunchKernelGGL((gemm_glds64_kernel<B_KC>), grid, dim3(512), 0, s, a);
      NEKO_CHECK_LAUNCH();
      g_neko_last_mainloop = 3;
      return NEKO_OK;
    }
  }
  hipLaunchKernelGGL((gemm_glds_kernel<A_KC, B_KC, C>), grid, dim3(C::NT), 0, s, a);
  NEKO_CHECK_LAUNCH();
  g_neko_last_mainloop = 0;
  return NEKO_OK;
}

// tile configuration: NEKO_GEMM_TILE=0..4 forces one (benchmarking); default heuristic below
int forced_tile() {
  static const int v = [] { const char* e = getenv("NEKO_GEMM_TILE"); return e ? atoi(e) : -1; }();
  return v;
}

// Per-shape choice, from tools/gemm_bench.py on the shapes of the 768d step (MI355X, random operands):
//   long contraction (weight gradients incl. split-K slices, LM-head dH/dW): 256x256, +5..15 % over 128x128;
//   A k-contiguous x B k-strided (forward) with N >= 2048: 256x256 (+6..8 %);
//   both k-contiguous (dgrad): K >= 2048 or N = 768: 256x128 (+7..9 %); K = 768 with N >= 2048 (GELU' dgrad): 256x256;
//   LM-head logits (both k-contiguous, N = 52352): 256x256 (+4 % and fewer table re-reads);
//   everything else 128x128 (3-stage, 3 blocks/CU).
template <bool A_KC, bool B_KC>
int launch(const GemmArgs& a, hipStream_t s) {
  const int klen = a.splitk > 1 ? a.k_per_split : a.K;
  int cfg = forced_tile();
  if (cfg < 0) {
    const bool big_out = (long)a.M * a.N >= (long)256 * 256 * 64;      // enough 256x256 tiles to matter
    if (a.splitk > 1 || (klen >= 4096 && (long)a.M * a.N >= (long)256 * 256 * 8)) cfg = 3;
    else if (A_KC && !B_KC && a.N >= 2048 && big_out) cfg = 3;
    else if (A_KC && !B_KC && klen >= 2048 && big_out && a.N % 256 == 0) cfg = 3;    // forward MLP projection (K = 3072, N = 768): +12 % over 128x128
    else if (A_KC && B_KC && klen >= 2048 && big_out && a.N % 128 == 0 && a.N <= 4096) cfg = 2;   // (not the LM-head logits at d = 2048)
    else if (A_KC && B_KC && a.N >= 2048 && big_out) cfg = 3;       // K = 768 dgrad through the MLP, LM-head logits
    else if (A_KC && B_KC && big_out && a.N % 128 == 0) cfg = 2;    // K = 768, N = 768 dgrad (attention out)
    else cfg = klen >= 16384 ? 1 : 0;
    // The rules above were measured at B*T = 32768 rows.  At the README batch sizes (32 x 240 = 7680 rows, 8 x 240,
    // 32 x 494) the same shapes are ONE partial round of 256x256 tiles and most CUs idle; tools/gemm_tile_sweep.sh
    // (profiles/r01_step19_gemm_tile_sweep.txt): with fewer than half a round, 128x128 (wide outputs: up to 2.1x) or
    // 256x128 (N = 768) wins; between half and 1.5 rounds wide outputs prefer 256x128 (+9..16 %), while N = 768 dgrads
    // prefer ONE 8-wave 256x256 block per CU over two 4-wave 256x128 blocks on some CUs (+16..19 %).
    // (the 2048d geometry agrees where it overlaps: N = 2048 dgrads at 8192 rows are exactly one round of 256x256 tiles
    // and gain 13..15 % on them; its K >= 2048 forward shapes keep 256x256 even at one round)
    // Round 2, 65536 rows (64 x 1024 tokens per step; profiles/r02_gemm_tile_sweep_65536.txt): the N = 768 dgrads are
    // 768 tiles of 256x256 = exactly three rounds of 256 CUs there, and one 8-wave 256x256 block per CU then beats two
    // 4-wave 256x128 blocks by 15-18 % (dgrad fc 363 -> 315 us, dgrad qkv 278 -> 238, dgrad o 92 -> 85); at 32768 rows the
    // same choice is 1.5 rounds and loses.  Likewise the forward attention projection (N = K = 768) on 256x128: three full
    // rounds of 512 slots at 65536 rows (161 -> 140 us), 1.5 at 32768.  Hence: fill of the last round decides.
    auto fill = [&](int bm, int bn, int slots) {
      const long t = (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * (a.splitk > 1 ? a.splitk : 1);
      return (double)t / (double)(((t + slots - 1) / slots) * slots);
    };
    if (A_KC && B_KC && cfg == 2 && a.N % 256 == 0 && a.N <= 1024 && fill(256, 256, 256) >= 0.9) cfg = 3;
    if (A_KC && !B_KC && cfg == 0 && big_out && a.N % 128 == 0 && a.N <= 1024 && fill(256, 128, 512) >= 0.9) cfg = 2;
    // round 4 (two row panels per rasterisation group): with whole rounds of 256 x 256 tiles the 8-wave block is ahead again for this
    // shape, inside the step 138.6 -> 131.7 us at 65536 rows (tools/probe/r04/step_kernel_ab.sh)
    if (A_KC && !B_KC && cfg == 2 && klen < 1536 && a.N % 256 == 0 && a.N <= 1024 && fill(256, 256, 256) >= 0.9) cfg = 3;
    const long t3n = (long)((a.M + 255) / 256) * ((a.N + 255) / 256) * (a.splitk > 1 ? a.splitk : 1);
    const bool wide = a.N > 2048;
    if (A_KC) {
      if (cfg == 3 && t3n < 128) cfg = wide ? 0 : 2;
      else if (cfg == 3 && t3n < 384 && wide && klen <= 1024) cfg = 2;
      else if (cfg == 2 && t3n >= 128 && t3n < 384 && !wide) cfg = 3;
    } else if (cfg == 3 && t3n < 128) {
      cfg = 0;                                  // weight gradients with a handful of output tiles
    }
  }
  switch (cfg) {
    case 1: return launch_cfg<A_KC, B_KC, C128s4>(a, s);
    case 2: return launch_cfg<A_KC, B_KC, C256x128>(a, s);
    case 3: return launch_cfg<A_KC, B_KC, C256x256>(a, s);
    case 4: return launch_cfg<A_KC, B_KC, C256x256w4>(a, s);
    default: return launch_cfg<A_KC, B_KC, C128s3>(a, s);
  }
}

}  // namespace

int neko_gemm_glds_colsum_bands() { const int b = t_colsum_bands; t_colsum_bands = 0; return b; }   // read-and-clear

// out[N] += sum over the bands of ws[bands][N], bands added in index order (bit-reproducible): 64 columns per block,
// 4 interleaved band groups per column summed sequentially, the 4 partials combined in a fixed order through LDS
namespace {
// 32 columns x 32 band groups per block: a thread adds every 32nd band with 8 loads in flight (the first form walked 128 bands per thread one
// dependent load at a time on 48 blocks: 54 us for a 6 MB workspace); the order of the additions is fixed, so the result is run-to-run identical
__global__ __launch_bounds__(1024) void colsum_bands_reduce_kernel(const float* __restrict__ ws, int bands, int N,
                                                                    float* __restrict__ out) {
  __shared__ float part[32][33];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float acc = 0.f;
  if (c < N) {
    int b = g;
    for (; b + 7 * 32 < bands; b += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(long)(b + u * 32) * N + c];
      acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; b < bands; b += 32) acc += ws[(long)b * N + c];
  }
  part[g][cl] = acc;
  __syncthreads();
  if (g == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 32; ++u) t += part[u][cl];
    out[c] += t;
  }
}
}  // namespace
int neko_colsum_bands_reduce_impl(const float* ws, int bands, int N, float* out, hipStream_t s) {
  if (bands <= 0 || N <= 0) return NEKO_OK;
  hipLaunchKernelGGL(colsum_bands_reduce_kernel, dim3((N + 31) / 32), dim3(1024), 0, s, ws, bands, N, out);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// returns 1 if the fast path does not apply (caller falls back), otherwise a neko status code
namespace {
// Row panels per rasterisation group (tile_coords walks a group's tiles row-panel-fastest).  An XCD runs 32 consecutive tiles at a time:
// with GROUP_M x nbn <= 32 the tiles that share an A row panel run in the same round and the panel is fetched once.
// Measured inside the m-mix step (profiles/r04_group_m_ab.txt): 2 panels per group for every class -0.3...0.5 ms per step against 8 (the
// product until round 4), each class contributing about a third; 1 and 4 are level with 8.  The isolated per-shape timings of the same
// file do not predict this (wide outputs measure 1-3 % slower alone): inside the step the operands were just written by the
// previous kernel and what matters is how much of them is still in the 256 MB last-level cache when their tiles come up.
// NEKO_GEMM_GM_NARROW / _WIDE / _SPLITK override the three classes.
int group_m_for(const GemmArgs& a) {
  static const int env_n = [] { const char* e = getenv("NEKO_GEMM_GM_NARROW"); return e ? atoi(e) : 2; }();
  static const int env_w = [] { const char* e = getenv("NEKO_GEMM_GM_WIDE"); return e ? atoi(e) : 2; }();
  static const int env_s = [] { const char* e = getenv("NEKO_GEMM_GM_SPLITK"); return e ? atoi(e) : 2; }();
  static const int env_h = [] { const char* e = getenv("NEKO_GEMM_GM_HUGE"); return e ? atoi(e) : 2; }();
  const int nbn = (a.N + 255) / 256;
  if (a.splitk > 1) return env_s;
  if (nbn <= 4) return env_n;
  if (nbn > 64) return env_h;          // LM-head logits (205 column panels): its own class since round 5
  return env_w;
}
}  // namespace

int neko_gemm_glds_try(const GemmArgs& a_in, int a_kstrided, int b_kstrided, hipStream_t s) {
  GemmArgs a = a_in;
  a.group_m = group_m_for(a);
  static const int env_lock = [] { const char* e = getenv("NEKO_GEMM_EPILOCK"); return e ? atoi(e) : 0; }();
  a.epi_lock = env_lock;
  t_colsum_bands = 0;
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
  {                         // two waves per SIMD in alternating roles, hand-placed (gemm_p16.hip), where it applies
    int bands = 0;
    const int rc = neko_gemm_p16_try(a, a_kstrided, b_kstrided, neko_gemm_mainloop_mode(), &bands, s);
    if (rc != 1) {
      t_colsum_bands = bands;
      return rc;
    }
  }
  {                         // two workgroups per CU, hand-placed 64 x 128 wave tiles (gemm_b16.hip), where it applies
    int bands = 0;
    const int rc = neko_gemm_b16_try(a, a_kstrided, b_kstrided, neko_gemm_mainloop_mode(), &bands, s);
    if (rc != 1) {
      t_colsum_bands = bands;
      return rc;
    }
  }
  {                         // hand-placed long-contraction main loop (gemm_a16.hip), where it applies
    const int rc = neko_gemm_a16_try(a, a_kstrided, b_kstrided, s);
    if (rc != 1) return rc;
  }
  if (a_kstrided && b_kstrided) return launch<false, false>(a, s);
  if (a_kstrided) return launch<false, true>(a, s);
  if (b_kstrided) return launch<true, false>(a, s);
  return launch<true, true>(a, s);
}

NEKO_DEFINE_SALT_SETTER(gemm_glds)
