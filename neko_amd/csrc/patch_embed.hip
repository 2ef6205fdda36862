// Image patch embedding: patchify + normalise + ResidualBlock_V2 (forward, and a recomputing backward)
// and the patch position encoding add / scatter.
// Replaces ImageEmbedding.forward (gato/policy/embeddings.py:28-61), ResidualBlock_V2 (:111-131:
// x + conv3x3(3<-C)(GELU(GroupNorm(conv3x3(C<-3)(GELU(x)))))), PatchPosEncoding's lookup/add (:101-110)
// and their autograd.  The 768->d projection (:53) runs on the bf16 GEMM.
//
// One 256-thread block = one 16x16 patch, one thread = one pixel.  The C=128 mid channels of the
// pixel live in registers (fully unrolled), conv weights are wave-uniform (scalar loads), GroupNorm
// statistics are two-pass block reductions, and the second conv walks the channels through a haloed
// LDS tile 32 at a time.  Nothing but the normalised patch (768 floats) is kept for backward: the
// backward kernel recomputes the block and produces all six parameter gradients in one pass
// (weight gradients by "one output per thread" sweeps over LDS tiles, accumulated in registers over
// the patches a block walks, then f32 atomics).  <2 % of the model FLOPs: plain
// fp32 VALU, no MFMA reshaping.
#include "neko_kernels.h"

namespace {

constexpr int C = 128;        // mid channels (train.py:94 always passes 128)
constexpr int G = 32;         // GroupNorm groups  -> 4 channels per group
constexpr int CPG = C / G;
constexpr int PS = 16;        // patch size
constexpr int HALO = PS + 2;  // 18
constexpr int CHUNK = 32;     // channels per LDS chunk
constexpr float GN_EPS = 1e-5f;

struct Smem {
  float gx[3][HALO][HALO];        // GELU(x) with zero halo              (3.9 KB)
  float dh3[3][HALO][HALO];       // d(conv2 out) with zero halo (bwd)
  float tile[CHUNK][HALO][HALO];  // haloed channel chunk / scratch        (41.5 KB)
  float red[4][2 * G];            // cross-wave reductions
  float stat[2 * G];              // mean / rstd per group
  float chan[2 * C];              // per-patch per-channel sums (bwd): [0,C) sum(du*xhat), [C,2C) sum(du)
  float acc_gn[2 * C];            // block accumulators for dgamma / dbeta
  float acc_b2[4];
};

// reduce NV per-thread values over the 256 threads; result broadcast through s.red -> out[NV] (LDS)
template <int NV>
__device__ __forceinline__ void block_reduce(const float (&v)[NV], float* out, Smem& s, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float r = wave_sum(v[i]);
    if (lane == 0) s.red[wave][i] = r;
  }
  __syncthreads();
  if (tid < NV) out[tid] = (s.red[0][tid] + s.red[1][tid]) + (s.red[2][tid] + s.red[3][tid]);
  __syncthreads();
}

// conv1 + GroupNorm statistics for this thread's pixel.  xh[c] returns the normalised value
// xhat = (h1 - mean_g) * rstd_g ; s.stat holds mean/rstd.
__device__ __forceinline__ void conv1_groupnorm(const Smem& cs, Smem& s, const float* __restrict__ w1,
                                                const float* __restrict__ b1, int py, int px, int tid,
                                                float (&xh)[C]) {
  float nb[27];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) nb[i * 9 + dy * 3 + dx] = cs.gx[i][py + dy][px + dx];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float a = b1[c];
#pragma unroll
    for (int k = 0; k < 27; ++k) a = fmaf(w1[c * 27 + k], nb[k], a);
    xh[c] = a;
  }
  float part[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < CPG; ++j) t += xh[g * CPG + j];
    part[g] = t;
  }
  block_reduce<G>(part, s.stat, s, tid);
  const float inv_n = 1.0f / (float)(CPG * PS * PS);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float mean = s.stat[g] * inv_n;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < CPG; ++j) {
      xh[g * CPG + j] -= mean;
      t += xh[g * CPG + j] * xh[g * CPG + j];
    }
    part[g] = t;
  }
  __syncthreads();
  block_reduce<G>(part, s.stat + G, s, tid);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float rstd = rsqrtf(s.stat[G + g] * inv_n + GN_EPS);
#pragma unroll
    for (int j = 0; j < CPG; ++j) xh[g * CPG + j] *= rstd;
  }
}

__device__ __forceinline__ void zero_halos(Smem& s, int tid) {
  float* z = &s.gx[0][0][0];
  for (int i = tid; i < 3 * HALO * HALO; i += 256) { z[i] = 0.f; (&s.dh3[0][0][0])[i] = 0.f; }
  float* t = &s.tile[0][0][0];
  for (int i = tid; i < CHUNK * HALO * HALO; i += 256) t[i] = 0.f;
}

template <bool U8>
__global__ __launch_bounds__(256) void resblock_fwd_kernel(const void* __restrict__ images, int n, int H, int W,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ gw, const float* __restrict__ gb,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           bf16_t* __restrict__ y16, float* __restrict__ xp) {
  __shared__ Smem s;
  const int tid = threadIdx.x, py = tid >> 4, px = tid & 15;
  const int nh = H / PS, nw = W / PS, P = n * nh * nw;
  zero_halos(s, tid);
  for (int p = blockIdx.x; p < P; p += gridDim.x) {
    const int b = p / (nh * nw), ph = (p / nw) % nh, pw = p % nw;
    float xv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const long off = (((long)b * 3 + i) * H + ph * PS + py) * W + pw * PS + px;
      const float raw = U8 ? (float)reinterpret_cast<const unsigned char*>(images)[off]
                           : reinterpret_cast<const float*>(images)[off];
      // embeddings.py:40-42: x = (x / 255.0 * 2) - 1 ; x = x / sqrt(patch_size)
      xv[i] = __fsub_rn(__fmul_rn(__fdiv_rn(raw, 255.0f), 2.0f), 1.0f) * 0.25f;
      if (xp) xp[(long)p * 768 + i * 256 + tid] = xv[i];
    }
    __syncthreads();   // previous patch finished with the LDS tiles
#pragma unroll
    for (int i = 0; i < 3; ++i) s.gx[i][py + 1][px + 1] = gelu_f(xv[i]);
    __syncthreads();
    float h[C];
    conv1_groupnorm(s, s, w1, b1, py, px, tid, h);
#pragma unroll
    for (int c = 0; c < C; ++c) h[c] = gelu_f(fmaf(h[c], gw[c], gb[c]));
    float o[3] = {b2[0], b2[1], b2[2]};
#pragma unroll
    for (int k = 0; k < C / CHUNK; ++k) {
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < CHUNK; ++cc) s.tile[cc][py + 1][px + 1] = h[k * CHUNK + cc];
      __syncthreads();
#pragma unroll 4
      for (int cc = 0; cc < CHUNK; ++cc) {
        float nb[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) nb[dy * 3 + dx] = s.tile[cc][py + dy][px + dx];
#pragma unroll
        for (int oc = 0; oc < 3; ++oc)
#pragma unroll
          for (int t = 0; t < 9; ++t) o[oc] = fmaf(w2[(oc * C + k * CHUNK + cc) * 9 + t], nb[t], o[oc]);
      }
    }
#pragma unroll
    for (int oc = 0; oc < 3; ++oc) y16[(long)p * 768 + oc * 256 + tid] = f32_to_bf16(xv[oc] + o[oc]);
  }
}

// Backward over patches (grid-stride).  dy f32 [P,768] is the gradient of the block output
// (= gradient wrt conv2 output; the identity branch reaches only the input image).
__global__ __launch_bounds__(256) void resblock_bwd_kernel(const float* __restrict__ xp, const float* __restrict__ dy,
                                                           int P, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ gw,
                                                           const float* __restrict__ gb, const float* __restrict__ w2,
                                                           float* __restrict__ dw1, float* __restrict__ db1,
                                                           float* __restrict__ dgw, float* __restrict__ dgb,
                                                           float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ Smem s;
  const int tid = threadIdx.x, py = tid >> 4, px = tid & 15;
  zero_halos(s, tid);
  for (int i = tid; i < 2 * C; i += 256) s.acc_gn[i] = 0.f;
  if (tid < 4) s.acc_b2[tid] = 0.f;
  // register accumulators: output slot j of chunk k is (tid + 256*j) within the chunk's output list
  float aw2[C / CHUNK][4], aw1[C / CHUNK][4];
#pragma unroll
  for (int k = 0; k < C / CHUNK; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) { aw2[k][j] = 0.f; aw1[k][j] = 0.f; }

  for (int p = blockIdx.x; p < P; p += gridDim.x) {
    float xv[3], g3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      xv[i] = xp[(long)p * 768 + i * 256 + tid];
      g3[i] = dy[(long)p * 768 + i * 256 + tid];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      s.gx[i][py + 1][px + 1] = gelu_f(xv[i]);
      s.dh3[i][py + 1][px + 1] = g3[i];
    }
    __syncthreads();
    float xh[C];
    conv1_groupnorm(s, s, w1, b1, py, px, tid, xh);

    // ---- db2 -----------------------------------------------------------------------------------
    block_reduce<3>(g3, s.chan, s, tid);   // s.chan is free until the GroupNorm backward below
    if (tid < 3) s.acc_b2[tid] += s.chan[tid];

    // ---- dW2 (needs h2 = GELU(GN out) neighbourhoods) and d_h2 ------------------------------------
    // d_h2[c] = sum_o sum_{dy,dx} w2[o][c][dy][dx] * dh3[o][py-dy+2][px-dx+2]
    float nb3[27];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
        for (int dx_ = 0; dx_ < 3; ++dx_) nb3[o * 9 + dy_ * 3 + dx_] = s.dh3[o][py - dy_ + 2][px - dx_ + 2];
    float du[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float a = 0.f;
#pragma unroll
      for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int t = 0; t < 9; ++t) a = fmaf(w2[(o * C + c) * 9 + t], nb3[o * 9 + t], a);
      const float u = fmaf(xh[c], gw[c], gb[c]);
      du[c] = a * gelu_grad_f(u);     // gradient wrt the GroupNorm output
    }
#pragma unroll
    for (int k = 0; k < C / CHUNK; ++k) {
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < CHUNK; ++cc)
        s.tile[cc][py + 1][px + 1] = gelu_f(fmaf(xh[k * CHUNK + cc], gw[k * CHUNK + cc], gb[k * CHUNK + cc]));
      __syncthreads();
      // outputs of this chunk: idx = (o*CHUNK + cc)*9 + t, 864 of them
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        if (idx < 3 * CHUNK * 9) {
          const int t = idx % 9, cc = (idx / 9) % CHUNK, o = idx / (9 * CHUNK);
          const int dy_ = t / 3, dx_ = t % 3;
          float a = 0.f;
          for (int yy = 0; yy < PS; ++yy)
#pragma unroll
            for (int xx = 0; xx < PS; ++xx) a = fmaf(s.dh3[o][yy + 1][xx + 1], s.tile[cc][yy + dy_][xx + dx_], a);
          aw2[k][j] += a;
        }
      }
    }

    // ---- GroupNorm backward ------------------------------------------------------------------------
    // per-patch per-channel sums over pixels: chan[c] = sum du*xhat, chan[C+c] = sum du
#pragma unroll
    for (int k = 0; k < 2 * C / CHUNK; ++k) {   // 8 chunks of 32 values
      __syncthreads();
      float* t = &s.tile[0][0][0];              // [32][256] scratch
#pragma unroll
      for (int v = 0; v < CHUNK; ++v) {
        const int idx = k * CHUNK + v;
        const float val = (idx < C) ? du[idx < C ? idx : 0] * xh[idx < C ? idx : 0] : du[idx >= C ? idx - C : 0];
        t[v * 256 + tid] = val;
      }
      __syncthreads();
      // 8 threads per value row
      const int row = tid >> 3, sub = tid & 7;
      float a = 0.f;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) a += t[row * 256 + sub * 32 + i];
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 4, 64);
      if (sub == 0) s.chan[k * CHUNK + row] = a;
    }
    __syncthreads();
    if (tid < C) {
      s.acc_gn[tid] += s.chan[tid];             // dgamma
      s.acc_gn[C + tid] += s.chan[C + tid];     // dbeta
    }
    // d_h1[c] = rstd_g * (du*gamma - A_g/N - xhat * B_g/N),  A_g = sum_c gamma*sum(du), B_g = sum_c gamma*sum(du*xhat)
    const float inv_n = 1.0f / (float)(CPG * PS * PS);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float A = 0.f, Bv = 0.f;
#pragma unroll
      for (int j = 0; j < CPG; ++j) {
        A = fmaf(gw[g * CPG + j], s.chan[C + g * CPG + j], A);
        Bv = fmaf(gw[g * CPG + j], s.chan[g * CPG + j], Bv);
      }
      const float rstd = rsqrtf(s.stat[G + g] * inv_n + GN_EPS);
#pragma unroll
      for (int j = 0; j < CPG; ++j) {
        const int c = g * CPG + j;
        du[c] = rstd * (du[c] * gw[c] - A * inv_n - xh[c] * Bv * inv_n);   // now d_h1
      }
    }

    // ---- dW1 / db1 -----------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < C / CHUNK; ++k) {
      __syncthreads();
      float* t = &s.tile[0][0][0];              // [32][256]
#pragma unroll
      for (int cc = 0; cc < CHUNK; ++cc) t[cc * 256 + tid] = du[k * CHUNK + cc];
      __syncthreads();
      // outputs: idx < 864: (cc, i, dy, dx) ; 864 <= idx < 896: bias of channel idx-864
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j;
        if (idx < CHUNK * 27) {
          const int t9 = idx % 9, i = (idx / 9) % 3, cc = idx / 27;
          const int dy_ = t9 / 3, dx_ = t9 % 3;
          float a = 0.f;
          for (int yy = 0; yy < PS; ++yy)
#pragma unroll
            for (int xx = 0; xx < PS; ++xx) a = fmaf(t[cc * 256 + yy * 16 + xx], s.gx[i][yy + dy_][xx + dx_], a);
          aw1[k][j] += a;
        } else if (idx < CHUNK * 28) {
          const int cc = idx - CHUNK * 27;
          float a = 0.f;
#pragma unroll 8
          for (int q = 0; q < 256; ++q) a += t[cc * 256 + q];
          aw1[k][j] += a;
        }
      }
    }
    __syncthreads();
    // restore the zero halo of the scratch tile (the [32][256] scratch use overwrote it)
    {
      float* t = &s.tile[0][0][0];
      for (int i = tid; i < CHUNK * HALO * HALO; i += 256) t[i] = 0.f;
    }
  }

  // ---- flush block accumulators ------------------------------------------------------------------------
  __syncthreads();
#pragma unroll
  for (int k = 0; k < C / CHUNK; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j;
      if (idx < 3 * CHUNK * 9) {
        const int t = idx % 9, cc = (idx / 9) % CHUNK, o = idx / (9 * CHUNK);
        atomicAdd(dw2 + (o * C + k * CHUNK + cc) * 9 + t, aw2[k][j]);
      }
      if (idx < CHUNK * 27) {
        const int t9 = idx % 9, i = (idx / 9) % 3, cc = idx / 27;
        atomicAdd(dw1 + ((k * CHUNK + cc) * 3 + i) * 9 + t9, aw1[k][j]);
      } else if (idx < CHUNK * 28) {
        atomicAdd(db1 + k * CHUNK + (idx - CHUNK * 27), aw1[k][j]);
      }
    }
  if (tid < C) {
    atomicAdd(dgw + tid, s.acc_gn[tid]);
    atomicAdd(dgb + tid, s.acc_gn[C + tid]);
  }
  if (tid < 3) atomicAdd(db2 + tid, s.acc_b2[tid]);
}

__global__ void patch_pos_add_kernel(float* __restrict__ out, const int* __restrict__ hpos, const int* __restrict__ wpos,
                                     const float* __restrict__ row_emb, const float* __restrict__ col_emb, int P, int d) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const float4* r = reinterpret_cast<const float4*>(row_emb + (long)hpos[p] * d);
  const float4* c = reinterpret_cast<const float4*>(col_emb + (long)wpos[p] * d);
  float4* o = reinterpret_cast<float4*>(out + (long)p * d);
  for (int i = lane; i < (d >> 2); i += 64) {
    float4 v = o[i];
    const float4 a = r[i], b = c[i];
    // reference order: x + (h_emb + w_emb)   (embeddings.py:57,109)
    v.x += a.x + b.x; v.y += a.y + b.y; v.z += a.z + b.z; v.w += a.w + b.w;
    o[i] = v;
  }
}
__global__ void patch_pos_add_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ hpos,
                                         const int* __restrict__ wpos, float* __restrict__ d_row, float* __restrict__ d_col,
                                         int P, int d) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  float* r = d_row + (long)hpos[p] * d;
  float* c = d_col + (long)wpos[p] * d;
  const float* g = dout + (long)p * d;
  for (int i = lane; i < d; i += 64) {
    const float v = g[i];
    atomicAdd(r + i, v);
    atomicAdd(c + i, v);
  }
}

}  // namespace

int neko_patch_resblock_fwd_impl(const void* images, int images_are_u8, int n, int H, int W, const float* w1,
                                 const float* b1, const float* gn_w, const float* gn_b, const float* w2,
                                 const float* b2, int mid_channels, int num_groups, bf16_t* y16, float* x_patches,
                                 hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!images || !w1 || !b1 || !gn_w || !gn_b || !w2 || !b2 || !y16) return NEKO_ERR_ARG;
  if (H <= 0 || W <= 0 || (H % PS) || (W % PS)) return NEKO_ERR_ARG;   // "Image dimensions must be divisible by patch size"
  if (mid_channels != C || num_groups != G) return NEKO_ERR_UNSUPPORTED;
  const int P = n * (H / PS) * (W / PS);
  const int grid = P < 2048 ? P : 2048;
  if (images_are_u8)
    hipLaunchKernelGGL((resblock_fwd_kernel<true>), dim3(grid), dim3(256), 0, s, images, n, H, W, w1, b1, gn_w, gn_b,
                       w2, b2, y16, x_patches);
  else
    hipLaunchKernelGGL((resblock_fwd_kernel<false>), dim3(grid), dim3(256), 0, s, images, n, H, W, w1, b1, gn_w, gn_b,
                       w2, b2, y16, x_patches);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_patch_resblock_bwd_impl(const float* x_patches, const float* dy, int P, const float* w1, const float* b1,
                                 const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                                 int mid_channels, int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b,
                                 float* dw2, float* db2, hipStream_t s) {
  (void)b2;
  if (P <= 0) return NEKO_OK;
  if (!x_patches || !dy || !w1 || !b1 || !gn_w || !gn_b || !w2 || !dw1 || !db1 || !dgn_w || !dgn_b || !dw2 || !db2)
    return NEKO_ERR_ARG;
  if (mid_channels != C || num_groups != G) return NEKO_ERR_UNSUPPORTED;
  const int grid = P < 512 ? P : 512;
  hipLaunchKernelGGL(resblock_bwd_kernel, dim3(grid), dim3(256), 0, s, x_patches, dy, P, w1, b1, gn_w, gn_b, w2, dw1,
                     db1, dgn_w, dgn_b, dw2, db2);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

int neko_patch_pos_add_impl(float* out, const int* hpos, const int* wpos, const float* row_emb, const float* col_emb,
                            int P, int d, hipStream_t s) {
  if (P <= 0) return NEKO_OK;
  if (!out || !hpos || !wpos || !row_emb || !col_emb || (d & 3)) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(patch_pos_add_kernel, dim3((P + 3) / 4), dim3(256), 0, s, out, hpos, wpos, row_emb, col_emb, P, d);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
int neko_patch_pos_add_bwd_impl(const float* dout, const int* hpos, const int* wpos, float* d_row_emb, float* d_col_emb,
                                int P, int d, hipStream_t s) {
  if (P <= 0) return NEKO_OK;
  if (!dout || !hpos || !wpos || !d_row_emb || !d_col_emb) return NEKO_ERR_ARG;
  hipLaunchKernelGGL(patch_pos_add_bwd_kernel, dim3((P + 3) / 4), dim3(256), 0, s, dout, hpos, wpos, d_row_emb,
                     d_col_emb, P, d);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
