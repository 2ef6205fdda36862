// Skinny-M companion of neko_gemm_bf16 for incremental decode: y[M<=8, N] = x[M,K] . op(W)
// (+bias)(GELU)(+resid), bf16 operands, fp32 accumulation.  At M = 1 the tiled MFMA kernel still walks its k-loop at
// ~1 us per 32-deep k-tile per block (DMA -> LDS -> fragments -> MFMA on 127 clamped rows): 10-100 us per call, 0.85 ms
// per decoded token.  A row-vector product is a pure weight stream (2*N*K bytes), so this kernel just streams:
//   W k-strided [K][ldw] (HF Conv1D (in,out) weights): block = 32 columns x 32 k-lanes, 8-B loads, x staged in LDS
//       as fp32, the k-lanes meet in LDS, fused epilogue;
//   W k-contiguous [N][ldw] (the LM-head table): one wave per output column block of rows, 16-B loads along K,
//       wave reduction.
#include "neko_kernels.h"
#include <type_traits>

namespace {

constexpr int GV_NT = 256, GV_MAXM = 8, GV_MAXK = 3072;     // x staged in LDS as fp32: 8 x 3072 x 4 B = 96 KB

struct GemvArgs {
  const bf16_t* x; long ldx;
  const bf16_t* W; long ldw;
  int M, N, K;
  const float* bias;
  const float* resid; long ldr;
  int act;                 // 0 none, 1 GELU (pre-activation rounded to bf16 first, like the GEMM)
  float* Cf; long ldcf;
  bf16_t* Cb; long ldcb;
};

__device__ __forceinline__ void gv_store(const GemvArgs& p, int m, int n, float v) {
  if (p.bias) v += p.bias[n];
  if (p.act == 1) v = gelu_f(bf16_to_f32(f32_to_bf16(v)));
  if (p.resid) v += p.resid[(long)m * p.ldr + n];
  if (p.Cf) p.Cf[(long)m * p.ldcf + n] = v;
  if (p.Cb) p.Cb[(long)m * p.ldcb + n] = f32_to_bf16(v);
}

// ---- W [K][N] (k-strided): 32 columns per block, thread (cg = tid & 7 -> 4 columns, kl = tid >> 3 -> k mod 32) --------
// Narrow column tiles give N/32 blocks (24 for N = 768, 96 for N = 3072) without splitting the contraction across
// blocks -- a split needs an agent-scope fence (L2 write-back) per launch, measured at ~10 us, more than the product.
// A wave covers 8 k-rows x 64 B per load instruction; the 32 k-lanes meet in LDS in fixed order (deterministic).
template <int M>
__global__ __launch_bounds__(GV_NT) void gemv_ks_kernel(GemvArgs p) {
  __shared__ float xs[M][GV_MAXK];
  __shared__ float part[32][M][32];
  const int tid = threadIdx.x, cg = tid & 7, kl = tid >> 3;
  const int n0 = blockIdx.x * 32 + cg * 4;
  for (int i = tid; i < M * p.K; i += GV_NT) {
    const int m = i / p.K, k = i - m * p.K;
    xs[m][k] = m < p.M ? bf16_to_f32(p.x[(long)m * p.ldx + k]) : 0.f;      // template M >= p.M (3 -> 4, 5..7 -> 8)
  }
  __syncthreads();
  float acc[M][4];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[m][e] = 0.f;
  if (n0 < p.N) {                                   // N is a multiple of 8 (ABI contract): whole 4-column groups
    const bf16_t* wp = p.W + n0;
#pragma unroll 8
    for (int k = kl; k < p.K; k += 32) {
      const uint2 w = *reinterpret_cast<const uint2*>(wp + (long)k * p.ldw);
      const float w0 = __uint_as_float(w.x << 16), w1 = __uint_as_float(w.x & 0xffff0000u);
      const float w2 = __uint_as_float(w.y << 16), w3 = __uint_as_float(w.y & 0xffff0000u);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float xv = xs[m][k];
        acc[m][0] = fmaf(xv, w0, acc[m][0]);
        acc[m][1] = fmaf(xv, w1, acc[m][1]);
        acc[m][2] = fmaf(xv, w2, acc[m][2]);
        acc[m][3] = fmaf(xv, w3, acc[m][3]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[kl][m][cg * 4 + e] = acc[m][e];
  __syncthreads();
  for (int i = tid; i < M * 32; i += GV_NT) {
    const int m = i >> 5, c = i & 31, n = blockIdx.x * 32 + c;
    if (n >= p.N || m >= p.M) continue;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) v += part[q][m][c];       // fixed order: deterministic
    gv_store(p, m, n, v);
  }
}

// ---- W [N][K] (k-contiguous): one wave per output column, 16-B loads along K ---------------------------------------------
template <int M>
__global__ __launch_bounds__(GV_NT) void gemv_kc_kernel(GemvArgs p) {
  __shared__ float xs[M][GV_MAXK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < M * p.K; i += GV_NT) {
    const int m = i / p.K, k = i - m * p.K;
    xs[m][k] = m < p.M ? bf16_to_f32(p.x[(long)m * p.ldx + k]) : 0.f;      // template M >= p.M (3 -> 4, 5..7 -> 8)
  }
  __syncthreads();
  constexpr int COLS_PER_WAVE = 8;
  const int nb = (blockIdx.x * 4 + wave) * COLS_PER_WAVE;
  // The weight rows of a batch of columns are requested with unconditional (clamped) loads before the first one is used:
  // as `for column: for k-chunk: load, use` every 16-byte load was a dependent HBM round trip of its own (8 columns x
  // 2..6 chunks per wave) in a kernel that is nothing but a weight stream.  K <= 1024 (d = 768 / 1024: qkv, projections,
  // LM head): all 8 columns x 2 chunks at once; longer rows (the MLP projection): 2 columns x 6 chunks.
  auto columns = [&](auto cb_tag, auto nkc_tag, int j0) {
    constexpr int CB = decltype(cb_tag)::value, NKC = decltype(nkc_tag)::value;
    uint4 w[CB][NKC];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const bf16_t* wr = p.W + (long)min(nb + j0 + j, p.N - 1) * p.ldw;
#pragma unroll
      for (int c = 0; c < NKC; ++c) w[j][c] = *reinterpret_cast<const uint4*>(wr + min(lane * 8 + c * 512, p.K - 8));
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int n = nb + j0 + j;
      if (n >= p.N) break;                                   // wave-uniform
      float acc[M];
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] = 0.f;
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        const int k = lane * 8 + c * 512;
        if (k < p.K) {
          const uint32_t ww[4] = {w[j][c].x, w[j][c].y, w[j][c].z, w[j][c].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(ww[e] << 16), hi = __uint_as_float(ww[e] & 0xffff0000u);
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m] = fmaf(xs[m][k + 2 * e + 1], hi, fmaf(xs[m][k + 2 * e], lo, acc[m]));
          }
        }
      }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float v = wave_sum(acc[m]);
        if (lane == 0 && m < p.M) gv_store(p, m, n, v);
      }
    }
  };
  if (p.K <= 1024) {
    columns(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{}, 0);
  } else {
    static_assert(GV_MAXK <= 6 * 512, "k-chunks per row");
    for (int j0 = 0; j0 < COLS_PER_WAVE; j0 += 2) columns(std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{}, j0);
  }
}

template <int M>
int gv_launch(const GemvArgs& a, int b_kstrided, hipStream_t s) {
  if (b_kstrided) hipLaunchKernelGGL((gemv_ks_kernel<M>), dim3((a.N + 31) / 32), dim3(GV_NT), 0, s, a);
  else hipLaunchKernelGGL((gemv_kc_kernel<M>), dim3((a.N + 31) / 32), dim3(GV_NT), 0, s, a);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

}  // namespace

int neko_gemv_bf16_impl(const bf16_t* x, long ldx, const bf16_t* W, long ldw, int b_kstrided, int M, int N, int K,
                        const float* bias, const float* resid, long ldr, int act, float* Cf, long ldcf, bf16_t* Cb,
                        long ldcb, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return NEKO_OK;
  if (!x || !W || (!Cf && !Cb) || act < 0 || act > 1) return NEKO_ERR_ARG;
  if (M > GV_MAXM || K > GV_MAXK) return NEKO_ERR_UNSUPPORTED;
  if ((K & 7) || (ldw & 7) || (b_kstrided && (N & 7))) return NEKO_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(W) & 15)) return NEKO_ERR_ARG;
  GemvArgs a{x, ldx, W, ldw, M, N, K, bias, resid, ldr, act, Cf, ldcf, Cb, ldcb};
  switch (M) {
    case 1: return gv_launch<1>(a, b_kstrided, s);
    case 2: return gv_launch<2>(a, b_kstrided, s);
    case 3: case 4: return gv_launch<4>(a, b_kstrided, s);
    default: return gv_launch<8>(a, b_kstrided, s);
  }
}
