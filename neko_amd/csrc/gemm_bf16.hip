// bf16 MFMA GEMM with fused epilogues for the Conv1D / Linear contractions of the path
// (HF Conv1D addmm: gato/transformers/trajectory_gpt2.py:139-141,222,253,264-265,274,277;
//  predict_token Linear: gato/policy/gato_policy.py:122,172; their autograd dgrad/wgrad).
//
//   C[M,N] = alpha * opA(A)[M,K] * opB(B)[K,N]   (+bias[N]) (gelu | *gelu'(pre)) (+resid) (+C)
//
// Operand storage (bf16, row-major, leading dims in elements):
//   A k-contiguous : A[m*lda + k]      A k-strided : A[k*lda + m]   (wgrad: X^T)
//   B k-contiguous : B[n*ldb + k]      B k-strided : B[k*ldb + n]   (Conv1D weight (in,out))
// Contract: every *contiguous* extent (K for k-contiguous, M/N for k-strided operands) and all
// leading dims are multiples of 8 elements (16 B); row counts are arbitrary (edges predicated).
//
// gfx950 design: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
// accumulators (64 acc VGPRs).  Register-staged global->LDS double buffering, one barrier per
// k-tile.  k-contiguous tiles sit in LDS as [row][64] with the 16-B chunk index XOR-swizzled by
// (row>>1)&7 (conflict-free ds_read_b128 over a 256-B bank row); k-strided tiles sit as
// [k][128 (+32 pad)] and MFMA fragments are gathered with ds_read_b64_tr_b16 (hardware 4x16
// transpose), so the transposed contractions (wgrad, Conv1D forward) never transpose in HBM.
// Split-K (grid.y) with f32 atomics feeds the skinny wgrad shapes (768x2304 output, K = B*T).
#include "neko_kernels.h"


namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int KC_TILE_BYTES = 128 * 128;        // [128 rows][64 k] bf16
constexpr int KS_STRIDE = 320;                  // bytes per k row: 128 cols * 2 + 64 pad
constexpr int KS_TILE_BYTES = 64 * KS_STRIDE;   // [64 k][160] bf16
constexpr int OP_BYTES = KS_TILE_BYTES;         // per operand per buffer (max of the two)

__device__ __forceinline__ int kc_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// ---- global -> registers -----------------------------------------------------------------
__device__ __forceinline__ void load_kc(const bf16_t* __restrict__ P, long ld, int r0, int nrows,
                                        int k0, int kend, int tid, uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NT * i;
    const int row = c >> 3, ch = c & 7;
    const int gr = r0 + row, gk = k0 + ch * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < nrows && gk < kend) v = *reinterpret_cast<const uint4*>(P + (long)gr * ld + gk);
    reg[i] = v;
  }
}
__device__ __forceinline__ void load_ks(const bf16_t* __restrict__ P, long ld, int r0, int nrows,
                                        int k0, int kend, int tid, uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NT * i;
    const int krow = c >> 4, ch = c & 15;
    const int gk = k0 + krow, gr = r0 + ch * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gk < kend && gr < nrows) v = *reinterpret_cast<const uint4*>(P + (long)gk * ld + gr);
    reg[i] = v;
  }
}

// ---- registers -> LDS ----------------------------------------------------------------------
__device__ __forceinline__ void store_kc(char* lds, int tid, const uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NT * i;
    *reinterpret_cast<uint4*>(lds + kc_off(c >> 3, c & 7)) = reg[i];
  }
}
__device__ __forceinline__ void store_ks(char* lds, int tid, const uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NT * i;
    *reinterpret_cast<uint4*>(lds + (c >> 4) * KS_STRIDE + (c & 15) * 16) = reg[i];
  }
}
// k-strided data written transposed into the k-contiguous image (2-byte stores): the
// no-transpose-read fallback used to cross-check ds_read_b64_tr_b16.
__device__ __forceinline__ void store_ks_as_kc(char* lds, int tid, const uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NT * i;
    const int k = c >> 4, ch = c & 15;
    const uint32_t w[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = ch * 8 + e;
      const uint16_t val = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
      *reinterpret_cast<uint16_t*>(lds + kc_off(row, k >> 3) + (k & 7) * 2) = val;
    }
  }
}

// ---- LDS -> MFMA fragments -----------------------------------------------------------------
// k-contiguous image: lane l holds row (rowbase + l%32), k = ks*16 + 8*(l/32) .. +7
__device__ __forceinline__ bf16x8_v frag_kc(const char* lds, int rowbase, int ks, int lane) {
  const int row = rowbase + (lane & 31);
  const uint4 v = *reinterpret_cast<const uint4*>(lds + kc_off(row, ks * 2 + (lane >> 5)));
  return __builtin_bit_cast(bf16x8_v, v);
}
// k-strided image [k][col]: each 16-lane group gathers a [4 k][16 col] block transposed, so
// lane l receives column (rowbase + l%32) at k = ks*16 + 8*(l/32) + {0..3} and {4..7}.
__device__ __forceinline__ bf16x8_v frag_ks(const char* lds, int rowbase, int ks, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int col = rowbase + 16 * (g & 1) + 4 * (c & 3);
  const int krow = ks * 16 + 8 * (g >> 1) + (c >> 2);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const lds_s16x4* p0 = (const lds_s16x4*)(lds + krow * KS_STRIDE + col * 2);
  const lds_s16x4* p1 = (const lds_s16x4*)(lds + (krow + 4) * KS_STRIDE + col * 2);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_v, r);
}

template <bool A_KC, bool B_KC, bool SAFE_T>
__global__ __launch_bounds__(NT) void gemm_bf16_kernel(GemmArgs p) {
  if (p.drop_thr) p.drop_key += neko_drop_salt();
  __shared__ __attribute__((aligned(16))) char smem[4 * OP_BYTES];
  auto ldsA = [&](int buf) -> char* { return smem + buf * 2 * OP_BYTES; };
  auto ldsB = [&](int buf) -> char* { return smem + (buf * 2 + 1) * OP_BYTES; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int nkt = (kend - kbeg + BK - 1) / BK;

  constexpr bool A_IMG_KC = A_KC || SAFE_T;
  constexpr bool B_IMG_KC = B_KC || SAFE_T;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (A_KC) load_kc(p.A, p.lda, m0, p.M, k0, kend, tid, ra);
    else load_ks(p.A, p.lda, m0, p.M, k0, kend, tid, ra);
    if (B_KC) load_kc(p.B, p.ldb, n0, p.N, k0, kend, tid, rb);
    else load_ks(p.B, p.ldb, n0, p.N, k0, kend, tid, rb);
  };
  auto lstore = [&](int buf) {
    if (A_KC) store_kc(ldsA(buf), tid, ra);
    else if (SAFE_T) store_ks_as_kc(ldsA(buf), tid, ra);
    else store_ks(ldsA(buf), tid, ra);
    if (B_KC) store_kc(ldsB(buf), tid, rb);
    else if (SAFE_T) store_ks_as_kc(ldsB(buf), tid, rb);
    else store_ks(ldsB(buf), tid, rb);
  };

  if (nkt > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const char* la = ldsA(cur);
    const char* lb = ldsB(cur);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = A_IMG_KC ? frag_kc(la, wm * 64 + i * 32, ks, lane) : frag_ks(la, wm * 64 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = B_IMG_KC ? frag_kc(lb, wn * 64 + j * 32, ks, lane) : frag_ks(lb, wn * 64 + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------
  const float alpha = p.alpha_dev ? p.alpha * (*p.alpha_dev) : p.alpha;
  const bool to_ws = p.splitk > 1 && p.splitk_ws;   // split-K slices go to a workspace, reduced in fixed order afterwards
  const bool atomic = p.splitk > 1 && !to_ws;
  float* const Cf_out = to_ws ? p.splitk_ws + (long)blockIdx.y * p.M * p.N : p.Cf;
  const long ldcf_out = to_ws ? p.N : p.ldcf;
  const int acc_out = to_ws ? 0 : p.accumulate;
  const bool lead = !atomic || blockIdx.y == 0;   // bias/resid added once under split-K
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= p.N) continue;
      const float bv = (p.bias && lead) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= p.M) continue;
        float v = acc[i][j][r] * alpha + bv;
        if (p.act == 1) {
          const bf16_t pb = f32_to_bf16(v);
          if (p.pre_out) p.pre_out[(long)row * p.ldpre + col] = pb;
          v = gelu_f(bf16_to_f32(pb));
        } else if (p.act == 3) {
          const float x = bf16_to_f32(f32_to_bf16(v));
          p.pre_out[(long)row * p.ldpre + col] = f32_to_bf16(gelu_grad_f(x));
          v = gelu_f(x);
        } else if (p.act == 2) {
          v *= gelu_grad_f(bf16_to_f32(p.act_in[(long)row * p.ldact + col]));
        } else if (p.act == 4) {
          v *= bf16_to_f32(p.act_in[(long)row * p.ldact + col]);
        }
        if (p.drop_thr) v = drop_keep((uint32_t)row * (uint32_t)p.N + (uint32_t)col, p.drop_key, p.drop_thr) ? v * p.drop_scale : 0.f;
        if (p.resid && lead) v += p.resid[(long)row * p.ldr + col];
        if (Cf_out) {
          float* dst = Cf_out + (long)row * ldcf_out + col;
          if (atomic) atomicAdd(dst, v);
          else if (acc_out) *dst += v;
          else *dst = v;
        }
        if (p.Cb) p.Cb[(long)row * p.ldcb + col] = f32_to_bf16(v);
      }
    }
  }
}

template <bool A_KC, bool B_KC, bool SAFE_T>
int launch(const GemmArgs& a, hipStream_t s) {
  const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
  dim3 grid(nbm * nbn, a.splitk > 1 ? a.splitk : 1);
  hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC, SAFE_T>), grid, dim3(NT), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

}  // namespace

thread_local int g_neko_last_mainloop = -1;

// Host-side entry used by neko_capi.hip.  a_kstrided / b_kstrided select the operand storage.
int neko_gemm_bf16_impl(GemmArgs a, int a_kstrided, int b_kstrided, int safe_transpose, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return NEKO_OK;
  if (!a.A || !a.B || (!a.Cf && !a.Cb)) return NEKO_ERR_ARG;
  if ((a.lda & 7) || (a.ldb & 7)) return NEKO_ERR_ARG;
  if (!a_kstrided && (a.K & 7)) return NEKO_ERR_ARG;
  if (!b_kstrided && (a.K & 7)) return NEKO_ERR_ARG;
  if (a_kstrided && (a.M & 7)) return NEKO_ERR_ARG;
  if (b_kstrided && (a.N & 7)) return NEKO_ERR_ARG;
  if ((a.act == 2 || a.act == 4) && !a.act_in) return NEKO_ERR_ARG;
  if (a.act == 3 && !a.pre_out) return NEKO_ERR_ARG;
  if (a.act < 0 || a.act > 4) return NEKO_ERR_ARG;
  if (a.splitk > 1) {
    if (!a.Cf || a.Cb || a.act != 0) return NEKO_ERR_ARG;  // atomics need a linear f32 epilogue
    if (a.k_per_split <= 0 || (a.k_per_split % BK)) return NEKO_ERR_ARG;
  }
  // safe_transpose: 0 = fastest available, 1 = transposing-store fallback, 2 = register-staged kernel only
  if (safe_transpose == 0) {
    const int rc = neko_gemm_glds_try(a, a_kstrided, b_kstrided, s);   // direct-to-LDS fast path (gemm_glds.hip)
    if (rc != 1) return rc;
  }
  safe_transpose = (safe_transpose == 1);
  g_neko_last_mainloop = 4;
  if (a_kstrided && b_kstrided)
    return safe_transpose ? launch<false, false, true>(a, s) : launch<false, false, false>(a, s);
  if (a_kstrided)
    return safe_transpose ? launch<false, true, true>(a, s) : launch<false, true, false>(a, s);
  if (b_kstrided)
    return safe_transpose ? launch<true, false, true>(a, s) : launch<true, false, false>(a, s);
  return launch<true, true, false>(a, s);
}

int neko_gemm_bf16_full(GemmArgs a, int a_kstrided, int b_kstrided, int safe_transpose, hipStream_t s) {
  if (a.splitk > 1 && a.splitk_ws && (a.N & 3)) a.splitk_ws = nullptr;   // reduce kernel wants N % 4 == 0
  if (a.splitk <= 1) a.splitk_ws = nullptr;
  const int rc = neko_gemm_bf16_impl(a, a_kstrided, b_kstrided, safe_transpose, s);
  if (rc != NEKO_OK || !a.splitk_ws || a.M <= 0 || a.N <= 0 || a.K <= 0) return rc;
  return neko_splitk_reduce_impl(a.splitk_ws, a.splitk, a.M, a.N, a.Cf, a.ldcf, a.accumulate, s);
}

NEKO_DEFINE_SALT_SETTER(gemm_bf16)
