// LayerNorm forward / backward over rows of the fp32 residual stream
// (nn.LayerNorm eps 1e-5 affine: gato/transformers/trajectory_gpt2.py:301,303,323,353,543,779).
//
// HBM-bound: one wave64 per row, the row lives in registers (float4 per lane, d <= 4096),
// statistics by wave shuffles in fp32 (two-pass mean / centred variance like ATen).
// forward : x f32 [M,d] -> y bf16 (GEMM operand) and/or y f32, mean/rstd f32 [M]
// backward: dy f32 [M,d], x, mean, rstd, gamma (+ g_in f32 residual-stream grad)
//           -> dx f32 = g_in + LN'(dy), dx16 bf16 copy (next dgrad/wgrad operand),
//              per-block partial dgamma/dbeta -> reduced by a second kernel into the grads.
#include <cstdlib>
#include "neko_kernels.h"

namespace {

constexpr int LN_MAXV = 16;  // float4 per lane -> d <= 4096

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y16,
                                                     float* __restrict__ y32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nvec = d >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (long)row * d);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
      q += (a * a + b * b) + (cc * cc + dd * dd);
    }
  }
  const float var = wave_sum(q) / (float)d;
  const float rs = rsqrtf(var + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      const float4 g = g4[c], b = b4[c];
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x;
      o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z;
      o.w = (v[i].w - mu) * rs * g.w + b.w;
      if (y32) reinterpret_cast<float4*>(y32 + (long)row * d)[c] = o;
      if (y16) {
        uint2 pk;
        pk.x = pack_bf16x2(o.x, o.y);
        pk.y = pack_bf16x2(o.z, o.w);
        reinterpret_cast<uint2*>(y16 + (long)row * d)[c] = pk;
      }
    }
  }
}

// Backward. Block = 4 waves; each wave walks rows (grid-stride) and keeps per-lane partial
// dgamma/dbeta for its columns; the 4 waves are combined through LDS into one partial row per block.
// DY16: dy arrives as bf16 (the dgrad GEMM's output as the reference's autocast produces it: the gradient of a bf16
// addmm input is bf16, trajectory_gpt2.py:274-277 under train.py:33-40) -- 2 instead of 4 bytes per element read here
// and written by the GEMM.
// MAP: dy holds only SOME rows, dy_map[row] = index of row's gradient in dy or -1 (that row's gradient is zero): the LM head hands
// back gradient rows of the loss positions only (gato_policy.py:183-185), and expanding them into a zero [M, d] buffer first cost a
// 201 MB fill, a scatter and 130 MB of zeros read here at 65536 rows.
template <int NV, bool DY16, bool MAP = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_any, const int* __restrict__ dy_map, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ g_in,
                                                     float* __restrict__ dx, bf16_t* __restrict__ dx16,
                                                     float* __restrict__ part, int M, int d, uint32_t drop_thr,
                                                     uint32_t drop_key, float drop_scale, int want_colsum) {
  if (drop_thr) drop_key += neko_drop_salt();
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][3][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = d >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  // dc: column sums of the bf16 copy this kernel emits (dx16, after its dropout mask) = the bias gradient of the Linear
  // that consumes it (Conv1D c_proj of the MLP / of the attention): folded in here, the separate colsum pass over the
  // [M, d] bf16 matrix (12 launches and 2 x 50 MB of reads per layer pair at B*T = 32768) is gone
  float4 gm[NV], dg[NV], db[NV], dc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    gm[i] = (c < nvec) ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    dc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_d = 1.0f / (float)d;
  // Software-pipelined over rows: the three input rows (x, dy, residual-stream gradient) of row r + stride are requested
  // before row r is reduced and stored, so a wave always has a full row of loads in flight.  (The first version loaded
  // g_in only after the wave reduction -- a second serialized HBM round trip per row -- and ran at 3.6 TB/s; the
  // forward, which has nothing between load and store, runs at 6.1 TB/s.)
  const int stride = gridDim.x * 4;
  float4 nx[NV], nd[NV], ng[NV];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](int row, int src) {
    nmu = mean[row];
    nrs = rstd[row];
    const bool have = !MAP || src >= 0;           // wave-uniform
    if (MAP && !have) src = 0;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * d);
    const float4* dr = reinterpret_cast<const float4*>(static_cast<const float*>(dy_any) + (long)src * d);
    const uint2* dr16 = reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(dy_any) + (long)src * d);
    const float4* gr = g_in ? reinterpret_cast<const float4*>(g_in + (long)row * d) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < nvec;
      nx[i] = ok ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (DY16) {
        const uint2 q = ok ? dr16[c] : make_uint2(0u, 0u);
        nd[i] = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                            __uint_as_float(q.y & 0xffff0000u));
      } else {
        nd[i] = (ok && have) ? dr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ng[i] = (ok && gr) ? gr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int row = blockIdx.x * 4 + wave;
  // the map entry of a row is requested one fetch ahead of the row itself, so the row's loads never wait for it
  auto map_of = [&](int r) { return (MAP && r < M) ? dy_map[r] : r; };
  int src_next = 0;
  if (row < M) {
    fetch(row, map_of(row));
    src_next = map_of(row + stride);
  }
  for (; row < M; row += stride) {
    const float mu = nmu, rs = nrs;
    float4 xh[NV], dv[NV], gv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { xh[i] = nx[i]; dv[i] = nd[i]; gv[i] = ng[i]; }
    if (row + stride < M) {                               // next row in flight during this row's reduction
      const int src = src_next;
      src_next = map_of(row + 2 * stride);
      fetch(row + stride, src);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        xh[i].x = (xh[i].x - mu) * rs; xh[i].y = (xh[i].y - mu) * rs;
        xh[i].z = (xh[i].z - mu) * rs; xh[i].w = (xh[i].w - mu) * rs;
        dg[i].x += dv[i].x * xh[i].x; dg[i].y += dv[i].y * xh[i].y;
        dg[i].z += dv[i].z * xh[i].z; dg[i].w += dv[i].w * xh[i].w;
        db[i].x += dv[i].x; db[i].y += dv[i].y; db[i].z += dv[i].z; db[i].w += dv[i].w;
        dv[i].x *= gm[i].x; dv[i].y *= gm[i].y; dv[i].z *= gm[i].z; dv[i].w *= gm[i].w;
        s1 += (dv[i].x + dv[i].y) + (dv[i].z + dv[i].w);
        s2 += (dv[i].x * xh[i].x + dv[i].y * xh[i].y) + (dv[i].z * xh[i].z + dv[i].w * xh[i].w);
      } else {
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        float4 o;
        o.x = rs * (dv[i].x - m1 - xh[i].x * m2);
        o.y = rs * (dv[i].y - m1 - xh[i].y * m2);
        o.z = rs * (dv[i].z - m1 - xh[i].z * m2);
        o.w = rs * (dv[i].w - m1 - xh[i].w * m2);
        if (g_in) { o.x += gv[i].x; o.y += gv[i].y; o.z += gv[i].z; o.w += gv[i].w; }
        if (dx) {
          float4 of = o;
          if (!dx16 && drop_thr) {     // no bf16 copy asked for: the mask belongs to the fp32 result (embedding dropout behind layer 0)
            float e[4] = {o.x, o.y, o.z, o.w};
            drop4(e, (uint32_t)row * (uint32_t)d + (uint32_t)c * 4u, drop_key, drop_thr, drop_scale);
            of = make_float4(e[0], e[1], e[2], e[3]);
          }
          reinterpret_cast<float4*>(dx + (long)row * d)[c] = of;
        }
        if (dx16) {
          // the bf16 copy feeds the dgrad/wgrad of the Linear that sits behind a residual dropout: it carries that
          // site's mask (the fp32 residual-stream gradient above does not)
          if (drop_thr) {
            const uint32_t base = (uint32_t)row * (uint32_t)d + (uint32_t)c * 4u;
            float e[4] = {o.x, o.y, o.z, o.w};
            drop4(e, base, drop_key, drop_thr, drop_scale);
            o = make_float4(e[0], e[1], e[2], e[3]);
          }
          uint2 pk;
          pk.x = pack_bf16x2(o.x, o.y);
          pk.y = pack_bf16x2(o.z, o.w);
          reinterpret_cast<uint2*>(dx16 + (long)row * d)[c] = pk;
          if (want_colsum) {       // sums of the ROUNDED values: what the separate pass over dx16 added up
            dc[i].x += __uint_as_float(pk.x << 16); dc[i].y += __uint_as_float(pk.x & 0xffff0000u);
            dc[i].z += __uint_as_float(pk.y << 16); dc[i].w += __uint_as_float(pk.y & 0xffff0000u);
          }
        }
      }
    }
  }
  // combine the 4 waves: [wave][2 or 3 rows][d] floats of dynamic LDS (the column-sum row only when it is asked for)
  float4* l4 = reinterpret_cast<float4*>(lds);
  const int nrow = want_colsum ? 3 : 2;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      l4[(wave * nrow + 0) * nvec + c] = dg[i];
      l4[(wave * nrow + 1) * nvec + c] = db[i];
      if (want_colsum) l4[(wave * nrow + 2) * nvec + c] = dc[i];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nrow * nvec; idx += 256) {
    float4 a = l4[idx];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 b = l4[w * nrow * nvec + idx];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(part + (long)blockIdx.x * 3 * d)[idx] = a;   // [block][3][d]
  }
}

// dgamma[d] (+)= sum_b part[b][0][:], dbeta likewise.  One thread per column, coalesced over b rows.
// block = 64 columns x 16 row slices (1024 threads), 4 independent loads in flight per thread; the 16 slice sums meet in LDS and are added
// in a fixed order (bit-reproducible).  (Round 5: 4 slices x 256 threads took 12-14 us for 512 partial rows on 36 blocks -- pure latency,
// twelve launches per step.)
constexpr int LNR_SLICES = 16;
__global__ __launch_bounds__(64 * LNR_SLICES) void ln_param_reduce_kernel(const float* __restrict__ part, int nblk, int d,
                                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                          float* __restrict__ dcol, int accumulate) {
  __shared__ float red[LNR_SLICES][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int ncol = (dcol ? 3 : 2) * d;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < ncol) {
    const long ld = 3L * d;
    int b = sl;
    for (; b + 3 * LNR_SLICES < nblk; b += 4 * LNR_SLICES) {
      s0 += part[(long)b * ld + c];
      s1 += part[(long)(b + LNR_SLICES) * ld + c];
      s2 += part[(long)(b + 2 * LNR_SLICES) * ld + c];
      s3 += part[(long)(b + 3 * LNR_SLICES) * ld + c];
    }
    for (; b < nblk; b += LNR_SLICES) s0 += part[(long)b * ld + c];
  }
  red[sl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && c < ncol) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < LNR_SLICES; u += 4) s += (red[u][cl] + red[u + 1][cl]) + (red[u + 2][cl] + red[u + 3][cl]);
    if (c >= 2 * d) {
      dcol[c - 2 * d] += s;          // a bias gradient: always accumulated, like neko_colsum_bf16(accumulate = 1)
    } else {
      float* dst = (c < d) ? (dgamma + c) : (dbeta + (c - d));
      *dst = accumulate ? (*dst + s) : s;
    }
  }
}

template <int NV>
int fwd_launch(const float* x, const float* g, const float* b, bf16_t* y16, float* y32, float* mean, float* rstd,
               int M, int d, float eps, hipStream_t s) {
  hipLaunchKernelGGL((ln_fwd_kernel<NV>), dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, y16, y32, mean, rstd, M, d, eps);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
template <int NV>
int bwd_launch(const void* dy, int dy16, const int* dy_map, const float* x, const float* g, const float* mean, const float* rstd,
               const float* g_in, float* dx, bf16_t* dx16, float* part, int nblk, int M, int d, int thr, unsigned key,
               float scale, int want_colsum, hipStream_t s) {
  const size_t lds_bytes = (size_t)(want_colsum ? 12 : 8) * d * sizeof(float);     // 4 waves x (2 or 3) rows x d
  if (lds_bytes > 160 * 1024) return NEKO_ERR_UNSUPPORTED;                          // one workgroup's LDS limit on gfx950
  if (dy16 && dy_map) return NEKO_ERR_UNSUPPORTED;
  if (dy16)
    hipLaunchKernelGGL((ln_bwd_kernel<NV, true>), dim3(nblk), dim3(256), lds_bytes, s, dy, dy_map, x, g, mean,
                       rstd, g_in, dx, dx16, part, M, d, (uint32_t)thr, key, scale, want_colsum);
  else if (dy_map)
    hipLaunchKernelGGL((ln_bwd_kernel<NV, false, true>), dim3(nblk), dim3(256), lds_bytes, s, dy, dy_map, x, g, mean,
                       rstd, g_in, dx, dx16, part, M, d, (uint32_t)thr, key, scale, want_colsum);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<NV, false>), dim3(nblk), dim3(256), lds_bytes, s, dy, dy_map, x, g, mean,
                       rstd, g_in, dx, dx16, part, M, d, (uint32_t)thr, key, scale, want_colsum);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

}  // namespace

int neko_layernorm_fwd_impl(const float* x, const float* gamma, const float* beta, bf16_t* y16, float* y32,
                            float* mean, float* rstd, int M, int d, float eps, hipStream_t s) {
  if (M <= 0) return NEKO_OK;
  if (!x || !gamma || !beta || (!y16 && !y32)) return NEKO_ERR_ARG;
  if ((d & 3) || d > 256 * LN_MAXV) return NEKO_ERR_UNSUPPORTED;
  const int nv = (d / 4 + 63) / 64;
  if (nv <= 1) return fwd_launch<1>(x, gamma, beta, y16, y32, mean, rstd, M, d, eps, s);
  if (nv <= 2) return fwd_launch<2>(x, gamma, beta, y16, y32, mean, rstd, M, d, eps, s);
  if (nv <= 3) return fwd_launch<3>(x, gamma, beta, y16, y32, mean, rstd, M, d, eps, s);
  if (nv <= 4) return fwd_launch<4>(x, gamma, beta, y16, y32, mean, rstd, M, d, eps, s);
  if (nv <= 8) return fwd_launch<8>(x, gamma, beta, y16, y32, mean, rstd, M, d, eps, s);
  return fwd_launch<16>(x, gamma, beta, y16, y32, mean, rstd, M, d, eps, s);
}

// number of partial rows the backward may write for a given M (workspace = that many x 3 x d floats): the larger of the two
// variants' block counts
static int ln_bwd_cap(int dy16) {
  // one block per CU measured best with fp32 dy (tools/ln_bench.py: 88 us vs 99 at 512 blocks, 32768 rows); the bf16-dy variant has
  // 8-byte loads on that operand, fewer bytes in flight per wave, and wants two blocks per CU (65536 rows: 220 us at 256 blocks,
  // 157 us at 512; fp32 dy: 163 / 173)
  static const int env = [] { const char* e = getenv("NEKO_LN_BWD_BLOCKS"); return e ? atoi(e) : 0; }();
  return env > 0 ? env : (dy16 ? 512 : 256);
}
static int ln_bwd_blocks(int M, int dy16) {
  const int cap = ln_bwd_cap(dy16), nb = (M + 3) / 4;
  return nb < cap ? (nb < 1 ? 1 : nb) : cap;
}
int neko_layernorm_bwd_blocks_impl(int M) {
  const int a = ln_bwd_blocks(M, 0), b = ln_bwd_blocks(M, 1);
  return a > b ? a : b;
}

int neko_layernorm_bwd_impl(const void* dy, int dy16, const float* x, const float* gamma, const float* mean,
                            const float* rstd, const float* g_in, float* dx, bf16_t* dx16, float* dgamma,
                            float* dbeta, int accumulate, float* workspace, int M, int d, int drop_thr,
                            unsigned drop_key, float drop_scale, float* dcolsum16, hipStream_t s, const int* dy_map) {
  if (M <= 0) return NEKO_OK;
  if (!dy || !x || !gamma || !mean || !rstd || !workspace || !dgamma || !dbeta) return NEKO_ERR_ARG;
  if (dcolsum16 && !dx16) return NEKO_ERR_ARG;
  const int wc = dcolsum16 ? 1 : 0;
  if ((d & 3) || d > 256 * LN_MAXV) return NEKO_ERR_UNSUPPORTED;
  const int nblk = ln_bwd_blocks(M, dy16);
  const int nv = (d / 4 + 63) / 64;
  int rc;
  if (nv <= 1) rc = bwd_launch<1>(dy, dy16, dy_map, x, gamma, mean, rstd, g_in, dx, dx16, workspace, nblk, M, d, drop_thr, drop_key, drop_scale, wc, s);
  else if (nv <= 2) rc = bwd_launch<2>(dy, dy16, dy_map, x, gamma, mean, rstd, g_in, dx, dx16, workspace, nblk, M, d, drop_thr, drop_key, drop_scale, wc, s);
  else if (nv <= 3) rc = bwd_launch<3>(dy, dy16, dy_map, x, gamma, mean, rstd, g_in, dx, dx16, workspace, nblk, M, d, drop_thr, drop_key, drop_scale, wc, s);
  else if (nv <= 4) rc = bwd_launch<4>(dy, dy16, dy_map, x, gamma, mean, rstd, g_in, dx, dx16, workspace, nblk, M, d, drop_thr, drop_key, drop_scale, wc, s);
  else if (nv <= 8) rc = bwd_launch<8>(dy, dy16, dy_map, x, gamma, mean, rstd, g_in, dx, dx16, workspace, nblk, M, d, drop_thr, drop_key, drop_scale, wc, s);
  else rc = bwd_launch<16>(dy, dy16, dy_map, x, gamma, mean, rstd, g_in, dx, dx16, workspace, nblk, M, d, drop_thr, drop_key, drop_scale, wc, s);
  if (rc != NEKO_OK) return rc;
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(((wc ? 3 : 2) * d + 63) / 64), dim3(64 * LNR_SLICES), 0, s, workspace, nblk, d, dgamma,
                     dbeta, dcolsum16, accumulate);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

NEKO_DEFINE_SALT_SETTER(layernorm)
