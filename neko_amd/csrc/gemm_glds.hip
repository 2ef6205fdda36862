// Fast path of neko_gemm_bf16: same contract as gemm_bf16.hip (see there for the reference citations),
// restricted to K-ranges that are multiples of 64; everything else falls back to the register-staged kernel.
//
// gfx950 design
//  * HBM -> LDS by direct DMA: global_load_lds_dwordx4 (16 B/lane, 1 KiB per wave-instruction), no staging
//    VGPRs and no ds_write pass.  The LDS image of a wave-instruction is lane-linear, so the bank-conflict
//    swizzle is applied to the per-lane SOURCE address and undone by the fragment reads:
//      k-contiguous tile [128 rows][64 k]  : 16-B piece index ^= (row>>1)&7   (conflict-free ds_read_b128)
//      k-strided   tile [64 k][128 cols]   : 16-B piece index ^= (k&3)<<2     (conflict-free ds_read_b64_tr_b16)
//  * 128x128x64 block tile, 4 waves (2x2) x (2x2) v_mfma_f32_32x32x16_bf16; double-buffered 2 x 32 KiB LDS
//    -> 2 blocks per CU; the DMA of tile t+1 is issued before the MFMAs of tile t, one barrier per tile.
//  * epilogue through LDS: each wave parks its 64x64 fp32 accumulators in its own 16 KiB slab (XOR-swizzled
//    float4 chunks), re-reads them row-major and applies bias / GELU / GELU' / residual / accumulate with 16-B
//    loads and stores (bf16 outputs 8 B per lane) instead of 64 scalar stores per lane.
//  * out-of-range rows / columns are clamped to the last valid one (their products only reach outputs that are
//    never stored); the contraction range itself is exact (K % 64 == 0 is the precondition of this path).
#include "neko_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
// LDS ring depth NSTAGE (template): NSTAGE-1 tiles in flight while one is consumed.
//   3 stages = 48 KB -> 3 blocks per CU: best for short K (768..3072), where prologue/epilogue overlap matters;
//   4 stages = 64 KB -> 2 blocks per CU: best for long K (LM-head dH, wgrad), where the deeper ring matters.
constexpr int OP_BYTES = 128 * BK * 2;          // 8 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;       // 16 KiB per stage
constexpr int GLDS_PER_STAGE = 4;               // wave-instructions each wave issues per stage (2 A + 2 B)

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

// ---- staging: 2 wave-instructions (1 KiB each) per operand per wave ------------------------------------
// k-contiguous tile [128 rows][32 k] (64-B rows): piece p of row r holds global piece p ^ ((r>>2)&3)
__device__ __forceinline__ void stage_kc(const bf16_t* __restrict__ P, long ld, int r0, int nrows, int k0, char* lds,
                                         int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int chunk = wave * 2 + i;                 // 1 KiB = 16 rows x 64 B
    const int row = chunk * 16 + (lane >> 2);
    const int piece = (lane & 3) ^ ((row >> 2) & 3);
    const int gr = min(r0 + row, nrows - 1);
    glds16(P + (long)gr * ld + k0 + piece * 8, lds + chunk * 1024);
  }
}
// k-strided tile [32 k][128 cols] (256-B rows): piece p of k-row r holds global piece p ^ ((r&3)<<2)
__device__ __forceinline__ void stage_ks(const bf16_t* __restrict__ P, long ld, int c0, int ncols, int k0, char* lds,
                                         int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int chunk = wave * 2 + i;                 // 1 KiB = 4 k-rows x 256 B
    const int kr = chunk * 4 + (lane >> 4);
    const int piece = (lane & 15) ^ ((kr & 3) << 2);
    const int gc = min(c0 + piece * 8, ncols - 8);
    glds16(P + (long)(k0 + kr) * ld + gc, lds + chunk * 1024);
  }
}

__device__ __forceinline__ bf16x8_v frag_kc(const char* lds, int rowbase, int ks, int lane) {
  const int row = rowbase + (lane & 31);
  const int piece = (ks * 2 + (lane >> 5)) ^ ((row >> 2) & 3);
  const uint4 v = *reinterpret_cast<const uint4*>(lds + row * 64 + piece * 16);
  return __builtin_bit_cast(bf16x8_v, v);
}
__device__ __forceinline__ bf16x8_v frag_ks(const char* lds, int colbase, int ks, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int col = colbase + 16 * (g & 1) + 4 * (c & 3);
  const int krow = ks * 16 + 8 * (g >> 1) + (c >> 2);          // krow & 3 == (krow+4) & 3
  const int off = (((col >> 3) ^ ((krow & 3) << 2)) << 4) + ((col & 7) << 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + krow * 256 + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (krow + 4) * 256 + off));
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

template <bool A_KC, bool B_KC, int NSTAGE>
__global__ __launch_bounds__(NT, (160 * 1024) / (NSTAGE * STAGE_BYTES)) void gemm_glds_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE_BYTES];   // ring of [A|B] stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = bid / nbn, tn = bid % nbn;
  const int m0 = tm * BM, n0 = tn * BN;

  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    kbeg = blockIdx.y * p.k_per_split;
    kend = min(p.K, kbeg + p.k_per_split);
  }
  const int nkt = (kend - kbeg) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    char* la = smem + (kt % NSTAGE) * STAGE_BYTES;
    char* lb = la + OP_BYTES;
    if (A_KC) stage_kc(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    else stage_ks(p.A, p.lda, m0, p.M, k0, la, wave, lane);
    if (B_KC) stage_kc(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
    else stage_ks(p.B, p.ldb, n0, p.N, k0, lb, wave, lane);
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
    if (kt + NSTAGE - 1 < nkt) stage(kt + NSTAGE - 1);
    const char* la = smem + (kt % NSTAGE) * STAGE_BYTES;
    const char* lb = la + OP_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = A_KC ? frag_kc(la, wm * 64 + i * 32, ks, lane) : frag_ks(la, wm * 64 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = B_KC ? frag_kc(lb, wn * 64 + j * 32, ks, lane) : frag_ks(lb, wn * 64 + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();   // all waves done with the ring before the slabs overwrite it

  // ---- epilogue through this wave's private 8 KiB slab, two passes of 32 rows ---------------------------------
  float* slab = reinterpret_cast<float*>(smem + wave * 8192);   // [32 rows][64 f32], float4 chunk ^= row&15
  // same-wave LDS write -> read: the compiler's lgkmcnt wait orders it (no cross-wave sharing of a slab)
  const float alpha = p.alpha_dev ? p.alpha * (*p.alpha_dev) : p.alpha;
  const bool to_ws = p.splitk > 1 && p.splitk_ws;   // split-K slices go to a workspace, reduced in fixed order afterwards
  const bool atomic = p.splitk > 1 && !to_ws;
  float* const Cf_out = to_ws ? p.splitk_ws + (long)blockIdx.y * p.M * p.N : p.Cf;
  const long ldcf_out = to_ws ? p.N : p.ldcf;
  const int acc_out = to_ws ? 0 : p.accumulate;
  const bool lead = !atomic || blockIdx.y == 0;
  const int cchunk = lane & 15;
  const int col = n0 + wn * 64 + cchunk * 4;
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
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int lc = j * 32 + (lane & 31);
        slab[lr * 64 + ((((lc >> 2) ^ (lr & 15)) << 2) | (lc & 3))] = acc[i][j][r];
      }
#pragma unroll 4
  for (int s = 0; s < 8; ++s) {
    const int lrow = s * 4 + (lane >> 4);
    const int row = m0 + wm * 64 + i * 32 + lrow;
    const float4 a4 = *reinterpret_cast<const float4*>(slab + lrow * 64 + ((cchunk ^ (lrow & 15)) << 2));
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
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = drop_keep(base + e, p.drop_key, p.drop_thr) ? v[e] * p.drop_scale : 0.f;
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

template <bool A_KC, bool B_KC>
int launch(const GemmArgs& a, hipStream_t s) {
  const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
  dim3 grid(nbm * nbn, a.splitk > 1 ? a.splitk : 1);
  const int klen = a.splitk > 1 ? a.k_per_split : a.K;
  if (klen >= 16384) hipLaunchKernelGGL((gemm_glds_kernel<A_KC, B_KC, 4>), grid, dim3(NT), 0, s, a);
  else hipLaunchKernelGGL((gemm_glds_kernel<A_KC, B_KC, 3>), grid, dim3(NT), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
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
