// Image patch embedding: patchify + normalise + ResidualBlock_V2 (forward, and a recomputing backward)
// and the patch position encoding add / scatter.
// Replaces ImageEmbedding.forward (gato/policy/embeddings.py:28-61), ResidualBlock_V2 (:111-131:
// x + conv3x3(3<-C)(GELU(GroupNorm(conv3x3(C<-3)(GELU(x)))))), PatchPosEncoding's lookup/add (:101-110)
// and their autograd.  The 768->d projection (:53) runs on the bf16 GEMM.
//
// One 256-thread block = one 16x16 patch, one thread = one pixel.  The C=128 mid channels are processed in
// 4 chunks of 32 (GroupNorm groups of 4 channels are chunk-local): conv1 writes the chunk's raw activations
// into a haloed LDS tile, the group statistics are two-pass block reductions, GELU(GN(.)) is applied in
// place and the second conv accumulates its 3 outputs over the chunk.  Conv/GN parameters are staged in LDS
// once per block and read as wave-uniform broadcasts; every inner loop is dynamic with a small body (an
// earlier fully-unrolled register-resident version spilled thousands of SGPRs/VGPRs).  Nothing but the
// normalised patch (768 floats) is kept for backward: the backward kernel recomputes the block chunk by
// chunk and produces all six parameter gradients in one pass (weight gradients by "one output per thread"
// sweeps over LDS tiles, accumulated in registers over the patches a block walks, then f32 atomics).
// <2 % of the model FLOPs: fp32 VALU, no MFMA reshaping.
#include "neko_kernels.h"

namespace {

constexpr int C = 128;        // mid channels (train.py:94 always passes 128)
constexpr int G = 32;         // GroupNorm groups  -> 4 channels per group
constexpr int CPG = C / G;
constexpr int PS = 16;        // patch size
constexpr int HALO = PS + 2;  // 18
constexpr int CHUNK = 32;     // channels per LDS chunk
constexpr float GN_EPS = 1e-5f;
// layout of one per-block partial-gradient row of the backward kernel
constexpr int OFF_W1 = 0, OFF_B1 = 128 * 27, OFF_GW = OFF_B1 + 128, OFF_GB = OFF_GW + 128, OFF_W2 = OFF_GB + 128,
              OFF_B2 = OFF_W2 + 3 * 128 * 9, PART_USED = OFF_B2 + 3, PART_STRIDE = (PART_USED + 63) / 64 * 64;

// BWD = false drops the two backward-only [32][256] tiles: 83 KB -> two forward blocks per CU
template <bool BWD>
struct SmemT {
  float gx[3][HALO][HALO];        // GELU(x) with zero halo
  float dh3[BWD ? 3 : 1][BWD ? HALO : 1][BWD ? HALO : 1];   // d(conv2 out) with zero halo (bwd)
  float tile[CHUNK][HALO][HALO];  // haloed channel chunk: raw h1, then h2 = GELU(GN(h1))
  float xh[BWD ? CHUNK : 1][PS * PS];       // bwd: xhat of the chunk
  float du[BWD ? CHUNK : 1][PS * PS];       // bwd: d(GN out), then d(h1)
  float red[4][2 * CHUNK];        // cross-wave partials
  float mean[CHUNK / CPG], rstd[CHUNK / CPG];
  float chan[BWD ? 2 * CHUNK : 1];   // bwd per-patch per-channel sums: [cc] sum(du*xhat), [CHUNK+cc] sum(du)
  float acc_gn[BWD ? 2 * C : 1];     // bwd block accumulators for dgamma / dbeta
  float acc_b2[4];
  // conv / GroupNorm parameters staged once per block (wave-uniform LDS broadcast reads)
  // rows padded to 28 floats (112 B): a channel's 27 weights are 7 aligned ds_read_b128 broadcasts
  __attribute__((aligned(16))) float w1[C * 28];   // [c][i*9 + tap]
  __attribute__((aligned(16))) float w2[C * 28];   // [c][o*9 + tap]  (regrouped per mid channel)
  float b1[C];
  float gw[C];
  float gb[C];
};

template <class Smem>
__device__ __forceinline__ void stage_params(Smem& s, const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ gw, const float* __restrict__ gb,
                                             const float* __restrict__ w2, int tid) {
  for (int i = tid; i < C * 27; i += 256) {
    const int c = i / 27, t = i % 27;
    s.w1[c * 28 + t] = w1[i];                                  // conv1.weight [c][3][3][3]
    s.w2[c * 28 + t] = w2[((t / 9) * C + c) * 9 + (t % 9)];     // conv2.weight [o][c][3][3] -> [c][o][3][3]
  }
  if (tid < C) { s.w1[tid * 28 + 27] = 0.f; s.w2[tid * 28 + 27] = 0.f; }
  if (tid < C) { s.b1[tid] = b1[tid]; s.gw[tid] = gw[tid]; s.gb[tid] = gb[tid]; }
}

// dot of a channel's 27 (padded to 28) LDS-resident weights with 27 per-thread values
__device__ __forceinline__ float dot27(const float* __restrict__ w, const float (&v)[27], float acc) {
  const float4* w4 = reinterpret_cast<const float4*>(w);
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float4 a = w4[q];
    acc = fmaf(a.x, v[4 * q + 0], acc);
    acc = fmaf(a.y, v[4 * q + 1], acc);
    acc = fmaf(a.z, v[4 * q + 2], acc);
    if (q < 6) acc = fmaf(a.w, v[4 * q + 3], acc);
  }
  return acc;
}

template <class Smem>
__device__ __forceinline__ void zero_halos(Smem& s, int tid) {
  float* z = &s.gx[0][0][0];
  for (int i = tid; i < 3 * HALO * HALO; i += 256) z[i] = 0.f;
  for (int i = tid; i < (int)(sizeof(s.dh3) / sizeof(float)); i += 256) (&s.dh3[0][0][0])[i] = 0.f;
  float* t = &s.tile[0][0][0];
  for (int i = tid; i < CHUNK * HALO * HALO; i += 256) t[i] = 0.f;
}

// out[cc] = sum over the 256 pixels of f(cc, px) for the chunk's 32 channels: 8 threads per channel sum 32 pixels
// each (rotated start so the lanes of a half-wave hit distinct banks), then 3 in-row xor steps.  Replaces per-channel
// wave_sum chains (6 dependent cross-lane steps each), which dominated the kernels at one wave per SIMD.
template <class F>
__device__ __forceinline__ void chan_reduce32(F f, float* out, int tid) {
  const int cc = tid >> 3, sub = tid & 7;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
  for (int i = 0; i < 32; i += 2) {
    a0 += f(cc, sub * 32 + ((i + tid) & 31));
    a1 += f(cc, sub * 32 + ((i + 1 + tid) & 31));
  }
  float a = a0 + a1;
  a += __shfl_xor(a, 1, 64);
  a += __shfl_xor(a, 2, 64);
  a += __shfl_xor(a, 4, 64);
  if (sub == 0) out[cc] = a;
}

// conv1 of channel chunk k for this thread's pixel -> raw h1 into the haloed tile, then the GroupNorm
// statistics of the chunk's 8 groups (two-pass: mean, centred variance) into s.mean / s.rstd.
// nb = the 27 GELU(x) neighbours of the pixel.  All loops are dynamic on purpose (small live ranges).
template <class Smem>
__device__ __forceinline__ void conv1_stats_chunk(Smem& s, int k, const float (&nb)[27], int py, int px, int tid) {
  constexpr int NG = CHUNK / CPG;
#pragma unroll 1
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int j = 0; j < CPG; ++j) {
      const int cc = g * CPG + j, c = k * CHUNK + cc;
      s.tile[cc][py + 1][px + 1] = dot27(&s.w1[c * 28], nb, s.b1[c]);
    }
  }
  __syncthreads();
  float* csum = &s.red[0][0];      // 32 per-channel sums
  chan_reduce32([&](int cc, int p) { return s.tile[cc][(p >> 4) + 1][(p & 15) + 1]; }, csum, tid);
  __syncthreads();
  const float inv_n = 1.0f / (float)(CPG * PS * PS);
  if (tid < NG) s.mean[tid] = ((csum[4 * tid] + csum[4 * tid + 1]) + (csum[4 * tid + 2] + csum[4 * tid + 3])) * inv_n;
  __syncthreads();
  chan_reduce32([&](int cc, int p) { const float d = s.tile[cc][(p >> 4) + 1][(p & 15) + 1] - s.mean[cc / CPG]; return d * d; },
                csum, tid);
  __syncthreads();
  if (tid < NG)
    s.rstd[tid] = rsqrtf(((csum[4 * tid] + csum[4 * tid + 1]) + (csum[4 * tid + 2] + csum[4 * tid + 3])) * inv_n + GN_EPS);
  __syncthreads();
}

// ---- MFMA forward -------------------------------------------------------------------------------------------------
// Both 3x3 convolutions are dense contractions over (input channel, tap) and run on v_mfma_f32_32x32x16_bf16 with bf16
// operands and fp32 accumulation (the reference runs them in bf16 under autocast):
//   conv1  H1^T[c][px] = W1[c][k] * im2col[k][px],  k = (in, dy, dx) padded 27 -> 32; columns 27/28 of the im2col are
//          1.0 and carry the bias as a bf16 hi + lo pair.  With the CHANNEL as the accumulator row, lane (px, h) holds
//          rows (r&3) + 8(r>>2) + 4h: every 4 consecutive registers are one GroupNorm group of its pixel.
//   GN     two-pass statistics from the fp32 accumulators: in-lane over the 4 channels, xor-shuffles over the 32 pixel
//          lanes, one LDS exchange across the 4 waves (each wave owns 64 of the 256 pixels).
//   conv2  Z^T[q][px] = W2[q][c] * H2[c][px], q = (out, dy, dx) padded 27 -> 32: the contraction index (channel) may be
//          permuted freely, so the B fragment of k-step (t, s) is simply bf16(acc[t][.][8s .. 8s+7]) -- no LDS round
//          trip, no shuffles -- and W2 is staged with the matching channel order.  out[o][p] = b2[o] +
//          sum_taps Z[(o,tap)][p + tap - 1] is a 27-term shift-sum through a zero-haloed LDS buffer.
struct FwdSmem {
  float gx[3][HALO][HALO];                         // GELU(x), zero halo
  float z[27][HALO][HALO];                         // conv2 partial products per (out, tap), zero halo
  __attribute__((aligned(16))) float red[2][G][4]; // GroupNorm partials [pass][group][wave]
  __attribute__((aligned(16))) bf16_t w1[C * 32];  // [c][k] k-contiguous, 16-B piece ^= (c>>2)&3
  __attribute__((aligned(16))) bf16_t w2[32 * C];  // [q][permuted c], 16-B piece ^= q&15
  float gw[C], gb[C];
  __attribute__((aligned(16))) bf16_t im[PS * PS * 32];   // im2col [pixel][k], 16-B piece ^= (pixel>>2)&3
};

// one pixel's im2col row: k = in*9 + dy*3 + dx from the haloed tile t3[3][18][18] at (py+dy, px+dx), then ones
__device__ __forceinline__ void write_im2col_row(bf16_t* im, const float* t3, int pix, float one_cols) {
  const float* base = t3 + (pix >> 4) * HALO + (pix & 15);
  float v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k)
    v[k] = k < 27 ? base[(k / 9) * HALO * HALO + ((k % 9) / 3) * HALO + (k % 3)] : (k < 29 ? one_cols : 0.f);
  uint4* row = reinterpret_cast<uint4*>(im + pix * 32);
#pragma unroll
  for (int pc = 0; pc < 4; ++pc)
    row[pc ^ ((pix >> 2) & 3)] = make_uint4(pack_bf16x2(v[8 * pc], v[8 * pc + 1]), pack_bf16x2(v[8 * pc + 2], v[8 * pc + 3]),
                                            pack_bf16x2(v[8 * pc + 4], v[8 * pc + 5]), pack_bf16x2(v[8 * pc + 6], v[8 * pc + 7]));
}
// B fragment (pixel = column) of k-step sidx from an im2col tile: 8 consecutive k of pixel `pix`
__device__ __forceinline__ bf16x8_v im2col_frag(const bf16_t* im, int pix, int sidx, int h) {
  return *reinterpret_cast<const bf16x8_v*>(im + pix * 32 + (((2 * sidx + h) ^ ((pix >> 2) & 3)) << 3));
}

// channel held by lane-half h in register 8s+j of channel tile t (== MFMA accumulator row of H1^T)
__device__ __forceinline__ int acc_channel(int t, int s, int h, int j) { return 32 * t + (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * h; }

__device__ __forceinline__ void stage_params_mfma(FwdSmem& s, const float* __restrict__ w1, const float* __restrict__ b1,
                                                  const float* __restrict__ gw, const float* __restrict__ gb,
                                                  const float* __restrict__ w2, int tid) {
  for (int i = tid; i < C * 32; i += 256) {
    const int c = i >> 5, k = i & 31;
    float v = 0.f;
    if (k < 27) v = w1[c * 27 + k];
    else if (k == 27) v = bf16_to_f32(f32_to_bf16(b1[c]));
    else if (k == 28) v = b1[c] - bf16_to_f32(f32_to_bf16(b1[c]));
    s.w1[c * 32 + ((((k >> 3) ^ ((c >> 2) & 3)) << 3) | (k & 7))] = f32_to_bf16(v);
  }
  for (int i = tid; i < 32 * C; i += 256) {
    const int q = i >> 7, slot = i & 127;          // slot = ((t*2+s)*2+h)*8 + j
    const int j = slot & 7, h = (slot >> 3) & 1, sidx = (slot >> 4) & 1, t = slot >> 5;
    const int c = acc_channel(t, sidx, h, j);
    const float v = q < 27 ? w2[((q / 9) * C + c) * 9 + (q % 9)] : 0.f;     // conv2.weight [o][c][3][3]
    s.w2[q * C + ((((slot >> 3) ^ (q & 15)) << 3) | j)] = f32_to_bf16(v);
  }
  if (tid < C) { s.gw[tid] = gw[tid]; s.gb[tid] = gb[tid]; }
}

template <bool U8>
__global__ __launch_bounds__(256, 2) void resblock_fwd_kernel(const void* __restrict__ images, int n, int H, int W,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ gw, const float* __restrict__ gb,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           bf16_t* __restrict__ y16, float* __restrict__ xp) {
  __shared__ FwdSmem s;
  const int tid = threadIdx.x, py = tid >> 4, px = tid & 15, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int nh = H / PS, nw = W / PS, P = n * nh * nw;
  for (int i = tid; i < 3 * HALO * HALO; i += 256) (&s.gx[0][0][0])[i] = 0.f;
  for (int i = tid; i < 27 * HALO * HALO; i += 256) (&s.z[0][0][0])[i] = 0.f;
  stage_params_mfma(s, w1, b1, gw, gb, w2, tid);
  const float bias2[3] = {b2[0], b2[1], b2[2]};
  const float inv_n = 1.0f / (float)(CPG * PS * PS);

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
    __syncthreads();   // previous patch finished reading gx / z
#pragma unroll
    for (int i = 0; i < 3; ++i) s.gx[i][py + 1][px + 1] = gelu_f(xv[i]);
    __syncthreads();
    write_im2col_row(s.im, &s.gx[0][0][0], tid, 1.0f);      // rows 64w .. 64w+63 are written and read by wave w only

    // ---- conv1: this wave's 2 pixel tiles x 4 channel tiles -----------------------------------------------------
    f32x16 acc[4][2];
    {
      bf16x8_v bfr[2][2];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) bfr[pt][sidx] = im2col_frag(s.im, 64 * wave + 32 * pt + l32, sidx, h);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bf16x8_v afr[2];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int c = 32 * t + l32;
          afr[sidx] = *reinterpret_cast<const bf16x8_v*>(&s.w1[c * 32 + (((2 * sidx + h) ^ ((c >> 2) & 3)) << 3)]);
        }
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][pt][r] = 0.f;
#pragma unroll
          for (int sidx = 0; sidx < 2; ++sidx)
            acc[t][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[sidx], bfr[pt][sidx], acc[t][pt], 0, 0, 0);
        }
      }
    }
    // ---- GroupNorm statistics: group (t, qq, h) = channels 32t + 8qq + 4h + {0..3}, all 256 pixels ------------------
    // pass 0: subtract the group mean in place; pass 1: scale by rstd in place (acc becomes xhat)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          float v = 0.f;
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = acc[t][pt][4 * qq + e];
              v = pass == 0 ? v + x : fmaf(x, x, v);
            }
#pragma unroll
          for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
          if (l32 == 0) s.red[pass][8 * t + 2 * qq + h][wave] = v;
          if (qq == 3) __builtin_amdgcn_sched_barrier(0);     // bound the live shuffle chains (register pressure)
        }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float4 r4 = *reinterpret_cast<const float4*>(&s.red[pass][8 * t + 2 * qq + h][0]);
          const float tot = ((r4.x + r4.y) + (r4.z + r4.w)) * inv_n;
          const float k = pass == 0 ? tot : rsqrtf(tot + GN_EPS);
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[t][pt][4 * qq + e] = pass == 0 ? acc[t][pt][4 * qq + e] - k : acc[t][pt][4 * qq + e] * k;
          if (qq == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- h2 = GELU(GN(h1)) in registers -> conv2 partial products Z^T[q][px] -----------------------------------------
    f32x16 zacc[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) zacc[pt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        float gwv[8], gbv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = acc_channel(t, sidx, h, j);
          gwv[j] = s.gw[c];
          gbv[j] = s.gb[c];
        }
        const int q = l32;
        const bf16x8_v afr = *reinterpret_cast<const bf16x8_v*>(&s.w2[q * C + (((((t * 2 + sidx) * 2 + h)) ^ (q & 15)) << 3)]);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = gelu_f(fmaf(acc[t][pt][8 * sidx + j], gwv[j], gbv[j]));
            if (j == 3) __builtin_amdgcn_sched_barrier(0);     // 4 GELUs in flight at a time (register pressure)
          }
          zacc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, pack8_bf16(v), zacc[pt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- shift-sum: Z^T rows q = (r&3) + 8(r>>2) + 4h of this lane's pixel -> haloed LDS, then 27 taps per output -----
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int pix = 64 * wave + 32 * pt + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (q < 27) s.z[q][(pix >> 4) + 1][(pix & 15) + 1] = zacc[pt][r];
      }
    }
    __syncthreads();
    float o[3] = {bias2[0], bias2[1], bias2[2]};
#pragma unroll
    for (int oc = 0; oc < 3; ++oc)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) o[oc] += s.z[oc * 9 + dy * 3 + dx][py + dy][px + dx];
    y16[(long)p * 768 + 0 * 256 + tid] = f32_to_bf16(xv[0] + o[0]);
    y16[(long)p * 768 + 1 * 256 + tid] = f32_to_bf16(xv[1] + o[1]);
    y16[(long)p * 768 + 2 * 256 + tid] = f32_to_bf16(xv[2] + o[2]);
  }
}

// Backward.  dy f32 [P,768] is the gradient of the block output (= gradient wrt conv2 output; the identity
// branch reaches only the input image).  Loop order: channel chunk k OUTER, patches (grid-stride) INNER, so only the
// chunk's weight-gradient accumulators are live in registers; everything per patch is recomputed per chunk (cheap
// next to the sweeps).  Weight-gradient sweeps are register-blocked: thread (cp, sl) owns channels {2cp, 2cp+1} of
// the chunk and pixel row sl, and accumulates all (output, tap) / (input, tap) combinations -> 54 FMAs per 21 / 29
// LDS reads; the 16 row-slices meet in xor-shuffles once per chunk, then one fp32 atomic per weight per block.
__global__ __launch_bounds__(256) void resblock_bwd_kernel(const float* __restrict__ xp, const float* __restrict__ dy,
                                                           int P, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ gw,
                                                           const float* __restrict__ gb, const float* __restrict__ w2,
                                                           float* __restrict__ part) {
  // every block writes one partial row [PART_STRIDE] = dw1 | db1 | dgamma | dbeta | dw2 | db2 (plain stores);
  // resblock_param_reduce_kernel sums the rows in a fixed order.  (fp32 atomics from 512 blocks onto the same 54
  // cache lines serialised in L2 and cost 8x the kernel's compute.)
  float* const dw1 = part + (long)blockIdx.x * PART_STRIDE + OFF_W1;
  float* const db1 = part + (long)blockIdx.x * PART_STRIDE + OFF_B1;
  float* const dgw = part + (long)blockIdx.x * PART_STRIDE + OFF_GW;
  float* const dgb = part + (long)blockIdx.x * PART_STRIDE + OFF_GB;
  float* const dw2 = part + (long)blockIdx.x * PART_STRIDE + OFF_W2;
  float* const db2 = part + (long)blockIdx.x * PART_STRIDE + OFF_B2;
  __shared__ SmemT<true> s;
  const int tid = threadIdx.x, py = tid >> 4, px = tid & 15, lane = tid & 63, wave = tid >> 6;
  const int cp = tid >> 4, sl = tid & 15;          // sweep role: channel pair / pixel row
  zero_halos(s, tid);
  stage_params(s, w1, b1, gw, gb, w2, tid);
  for (int i = tid; i < 2 * C; i += 256) s.acc_gn[i] = 0.f;
  if (tid < 4) s.acc_b2[tid] = 0.f;
  const float inv_n = 1.0f / (float)(CPG * PS * PS);

#pragma unroll 1
  for (int k = 0; k < C / CHUNK; ++k) {
    float a2[27][2], a1[27][2], ab[2];
#pragma unroll
    for (int i = 0; i < 27; ++i) { a2[i][0] = a2[i][1] = 0.f; a1[i][0] = a1[i][1] = 0.f; }
    ab[0] = ab[1] = 0.f;

#pragma unroll 1
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
      float g3[3];
      {
        float xv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          xv[i] = xp[(long)p * 768 + i * 256 + tid];
          g3[i] = dy[(long)p * 768 + i * 256 + tid];
        }
        __syncthreads();      // previous (chunk, patch) is done with every LDS tile
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          s.gx[i][py + 1][px + 1] = gelu_f(xv[i]);
          s.dh3[i][py + 1][px + 1] = g3[i];
        }
      }
      if (k == 0) {           // db2 once per patch
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float r = wave_sum(g3[i]);
          if (lane == 0) s.red[wave][i] = r;
        }
      }
      __syncthreads();
      if (k == 0 && tid < 3) s.acc_b2[tid] += (s.red[0][tid] + s.red[1][tid]) + (s.red[2][tid] + s.red[3][tid]);
      float nb[27], nb3[27];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
          for (int dx_ = 0; dx_ < 3; ++dx_) {
            nb[i * 9 + dy_ * 3 + dx_] = s.gx[i][py + dy_][px + dx_];
            nb3[i * 9 + dy_ * 3 + dx_] = s.dh3[i][py - dy_ + 2][px - dx_ + 2];
          }
      __syncthreads();
      conv1_stats_chunk(s, k, nb, py, px, tid);
      // xhat -> s.xh ; h2 -> haloed tile ; du = d(h2)*GELU'(u) -> s.du ; per-channel sums
#pragma unroll 1
      for (int g = 0; g < CHUNK / CPG; ++g) {
        const float m = s.mean[g], rs = s.rstd[g];
#pragma unroll
        for (int j = 0; j < CPG; ++j) {
          const int cc = g * CPG + j, c = k * CHUNK + cc;
          const float xh = (s.tile[cc][py + 1][px + 1] - m) * rs;
          const float u = fmaf(xh, s.gw[c], s.gb[c]);
          s.xh[cc][tid] = xh;
          s.tile[cc][py + 1][px + 1] = gelu_f(u);
          // d_h2[c] = sum_o sum_taps w2[o][c][tap] * dh3[o][pixel - tap + 1]
#ifndef NEKO_PATCH_DIAG_NODH2
          const float a = dot27(&s.w2[c * 28], nb3, 0.f);
#else
          const float a = nb3[cc % 27];
#endif
          const float du = a * gelu_grad_f(u);
          s.du[cc][tid] = du;
        }
      }
      __syncthreads();
      // per-channel sums over the patch: chan[cc] = sum du*xhat, chan[32+cc] = sum du
      chan_reduce32([&](int cc, int p) { return s.du[cc][p] * s.xh[cc][p]; }, &s.chan[0], tid);
      chan_reduce32([&](int cc, int p) { return s.du[cc][p]; }, &s.chan[CHUNK], tid);
      __syncthreads();
      if (tid < 2 * CHUNK) {
        const int c = k * CHUNK + (tid & (CHUNK - 1));
        s.acc_gn[(tid < CHUNK ? 0 : C) + c] += s.chan[tid];      // [0,C) dgamma, [C,2C) dbeta
      }
      // ---- dW2 sweep: a2[o*9+tap][e] += dh3[o][px] * h2[2cp+e][px + tap - 1] over pixel row sl ------------------
#ifndef NEKO_PATCH_DIAG_NOSWEEP
      {
        const int c0 = 2 * cp;
#pragma unroll 2
        for (int xx = 0; xx < PS; ++xx) {
          const float d0 = s.dh3[0][sl + 1][xx + 1], d1 = s.dh3[1][sl + 1][xx + 1], d2 = s.dh3[2][sl + 1][xx + 1];
#pragma unroll
          for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
            for (int dx_ = 0; dx_ < 3; ++dx_) {
              const int t = dy_ * 3 + dx_;
              const float h0 = s.tile[c0][sl + dy_][xx + dx_], h1 = s.tile[c0 + 1][sl + dy_][xx + dx_];
              a2[t][0] = fmaf(d0, h0, a2[t][0]);      a2[t][1] = fmaf(d0, h1, a2[t][1]);
              a2[9 + t][0] = fmaf(d1, h0, a2[9 + t][0]);  a2[9 + t][1] = fmaf(d1, h1, a2[9 + t][1]);
              a2[18 + t][0] = fmaf(d2, h0, a2[18 + t][0]); a2[18 + t][1] = fmaf(d2, h1, a2[18 + t][1]);
            }
        }
      }
#endif
      __syncthreads();
      // ---- d_h1 = rstd_g * (du*gamma - A_g/N - xhat*B_g/N) in place of du (own pixel) -------------------------
#pragma unroll 1
      for (int g = 0; g < CHUNK / CPG; ++g) {
        float A = 0.f, Bv = 0.f;
#pragma unroll
        for (int j = 0; j < CPG; ++j) {
          const int cc = g * CPG + j;
          A = fmaf(s.gw[k * CHUNK + cc], s.chan[CHUNK + cc], A);
          Bv = fmaf(s.gw[k * CHUNK + cc], s.chan[cc], Bv);
        }
        const float rs = s.rstd[g];
#pragma unroll
        for (int j = 0; j < CPG; ++j) {
          const int cc = g * CPG + j;
          s.du[cc][tid] = rs * (s.du[cc][tid] * s.gw[k * CHUNK + cc] - A * inv_n - s.xh[cc][tid] * Bv * inv_n);
        }
      }
      __syncthreads();
      // ---- dW1 / db1 sweep: a1[i*9+tap][e] += d_h1[2cp+e][px] * gx[i][px + tap - 1] over pixel row sl -----------
#ifndef NEKO_PATCH_DIAG_NOSWEEP
      {
        const int c0 = 2 * cp;
#pragma unroll 2
        for (int xx = 0; xx < PS; ++xx) {
          const float u0 = s.du[c0][sl * 16 + xx], u1 = s.du[c0 + 1][sl * 16 + xx];
          ab[0] += u0;
          ab[1] += u1;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
              for (int dx_ = 0; dx_ < 3; ++dx_) {
                const int t = i * 9 + dy_ * 3 + dx_;
                const float gv = s.gx[i][sl + dy_][xx + dx_];
                a1[t][0] = fmaf(u0, gv, a1[t][0]);
                a1[t][1] = fmaf(u1, gv, a1[t][1]);
              }
        }
      }
#endif
    }   // patches

    // ---- combine the 16 pixel-row slices of every channel pair (lanes tid%16) and flush the chunk ------------------
#pragma unroll
    for (int i = 0; i < 27; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float v2 = a2[i][e], v1 = a1[i][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { v2 += __shfl_xor(v2, o, 64); v1 += __shfl_xor(v1, o, 64); }
        a2[i][e] = v2;
        a1[i][e] = v1;
      }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ab[e] += __shfl_xor(ab[e], o, 64);
    if (sl == 0) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = k * CHUNK + 2 * cp + e;
#pragma unroll
        for (int i = 0; i < 27; ++i) {
          dw2[((i / 9) * C + c) * 9 + (i % 9)] = a2[i][e];      // i = o*9 + tap
          dw1[(c * 3 + (i / 9)) * 9 + (i % 9)] = a1[i][e];      // i = in*9 + tap
        }
        db1[c] = ab[e];
      }
    }
  }   // chunks

  __syncthreads();
  if (tid < C) {
    dgw[tid] = s.acc_gn[tid];
    dgb[tid] = s.acc_gn[C + tid];
  }
  if (tid < 3) db2[tid] = s.acc_b2[tid];
}

// grads (+)= sum over the per-block partial rows; thread per element, coalesced across elements
__global__ __launch_bounds__(256) void resblock_param_reduce_kernel(const float* __restrict__ part, int nblk,
                                                                    float* __restrict__ dw1, float* __restrict__ db1,
                                                                    float* __restrict__ dgw, float* __restrict__ dgb,
                                                                    float* __restrict__ dw2, float* __restrict__ db2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= PART_USED) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 3 < nblk; b += 4) {
    s0 += part[(long)b * PART_STRIDE + j];
    s1 += part[(long)(b + 1) * PART_STRIDE + j];
    s2 += part[(long)(b + 2) * PART_STRIDE + j];
    s3 += part[(long)(b + 3) * PART_STRIDE + j];
  }
  for (; b < nblk; ++b) s0 += part[(long)b * PART_STRIDE + j];
  const float v = (s0 + s1) + (s2 + s3);
  float* dst;
  if (j < OFF_B1) dst = dw1 + j;
  else if (j < OFF_GW) dst = db1 + (j - OFF_B1);
  else if (j < OFF_GB) dst = dgw + (j - OFF_GW);
  else if (j < OFF_W2) dst = dgb + (j - OFF_GB);
  else if (j < OFF_B2) dst = dw2 + (j - OFF_W2);
  else dst = db2 + (j - OFF_B2);
  *dst += v;
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

// blocks (= partial rows) the backward uses for P patches; workspace = blocks * neko_patch_resblock_ws_stride floats
int neko_patch_resblock_bwd_blocks_impl(int P) { return P < 512 ? (P < 1 ? 1 : P) : 512; }
int neko_patch_resblock_ws_stride_impl() { return PART_STRIDE; }

int neko_patch_resblock_bwd_impl(const float* x_patches, const float* dy, int P, const float* w1, const float* b1,
                                 const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                                 int mid_channels, int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b,
                                 float* dw2, float* db2, float* workspace, hipStream_t s) {
  (void)b2;
  if (P <= 0) return NEKO_OK;
  if (!x_patches || !dy || !w1 || !b1 || !gn_w || !gn_b || !w2 || !dw1 || !db1 || !dgn_w || !dgn_b || !dw2 || !db2 ||
      !workspace)
    return NEKO_ERR_ARG;
  if (mid_channels != C || num_groups != G) return NEKO_ERR_UNSUPPORTED;
  const int grid = neko_patch_resblock_bwd_blocks_impl(P);
  hipLaunchKernelGGL(resblock_bwd_kernel, dim3(grid), dim3(256), 0, s, x_patches, dy, P, w1, b1, gn_w, gn_b, w2,
                     workspace);
  NEKO_CHECK_LAUNCH();
  hipLaunchKernelGGL(resblock_param_reduce_kernel, dim3((PART_USED + 255) / 256), dim3(256), 0, s, workspace, grid, dw1,
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
