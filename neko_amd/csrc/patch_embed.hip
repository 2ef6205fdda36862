// Image patch embedding: patchify + normalise + ResidualBlock_V2 (forward, and a recomputing backward)
// and the patch position encoding add / scatter.
// Replaces ImageEmbedding.forward (gato/policy/embeddings.py:28-61), ResidualBlock_V2 (:111-131:
// x + conv3x3(3<-C)(GELU(GroupNorm(conv3x3(C<-3)(GELU(x)))))), PatchPosEncoding's lookup/add (:101-110)
// and their autograd.  The 768->d projection (:53) runs on the bf16 GEMM.
//
// One 256-thread block (4 waves) walks 16x16 patches; wave w owns pixels 64w .. 64w+63.  Both 3x3 convolutions and
// both weight gradients are dense contractions over (channel, tap) or over pixels and run on
// v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulation -- the reference runs these convolutions in bf16 under
// autocast); GroupNorm statistics, GELU and the GroupNorm backward stay fp32 on the accumulator
// registers.  On the m-mix batch the image path is ~12k patches per step: the earlier fp32 VALU version of these two
// kernels was 18 % of the step, this one is ~3 %.  Nothing but the normalised patch (768 floats) is kept for
// backward: the backward kernel recomputes conv1/GroupNorm tile by tile and produces all six parameter gradients in
// one pass (per-block partial rows, summed in a fixed order by resblock_param_reduce_kernel).
#include <type_traits>
#include "neko_kernels.h"

namespace {

constexpr int C = 128;        // mid channels (train.py:94 always passes 128)
constexpr int G = 32;         // GroupNorm groups  -> 4 channels per group
constexpr int CPG = C / G;
constexpr int PS = 16;        // patch size
constexpr int HALO = PS + 2;  // 18
constexpr float GN_EPS = 1e-5f;
// layout of one per-block partial-gradient row of the backward kernel
constexpr int OFF_W1 = 0, OFF_B1 = 128 * 27, OFF_GW = OFF_B1 + 128, OFF_GB = OFF_GW + 128, OFF_W2 = OFF_GB + 128,
              OFF_B2 = OFF_W2 + 3 * 128 * 9, PART_USED = OFF_B2 + 3, PART_STRIDE = (PART_USED + 63) / 64 * 64;

// Sum over the 32 lanes that share lane >> 5, by DPP: quad_perm xor 1 / xor 2, row_half_mirror, row_mirror (every lane of a
// 16-lane row then holds its row's total), row_bcast15 into rows 1 and 3.  The total is valid in lanes 16..31 and 48..63
// (callers store from lane 31 / 63).  Five VALU adds with ~8-cycle latency each; the __shfl_xor butterfly this replaces is
// five ds_bpermute round trips through the LDS crossbar (~100+ cycles each), 64 of them per patch in the statistics alone.
__device__ __forceinline__ float half_sum32_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));   // row_bcast15 -> rows 1, 3
  return v;
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

// one pixel's im2col row [32] (bf16): k = ch*9 + dy*3 + dx < 27 from the haloed tile t3[3][18][18], columns 27/28 =
// one_cols, rest 0.  FLIP = false: t3[ch][py+dy][px+dx] (forward neighbourhood); FLIP = true: t3[ch][py-dy+2][px-dx+2]
// (the transposed-convolution neighbourhood of the output gradient).
template <bool FLIP>
__device__ __forceinline__ void write_im2col_row(bf16_t* im, const float* t3, int pix, float one_cols) {
  const float* base = t3 + ((pix >> 4) + (FLIP ? 2 : 0)) * HALO + (pix & 15) + (FLIP ? 2 : 0);
  float v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int off = (k / 9) * HALO * HALO + (FLIP ? -1 : 1) * (((k % 9) / 3) * HALO + (k % 3));
    v[k] = k < 27 ? base[off] : (k < 29 ? one_cols : 0.f);
  }
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
                                                           bf16_t* __restrict__ y16, float* __restrict__ xp,
                                                           float* __restrict__ gn_stats) {
  __shared__ FwdSmem s;
  const int tid = threadIdx.x, py = tid >> 4, px = tid & 15, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  const int nh = H / PS, nw = W / PS, P = n * nh * nw;
  for (int i = tid; i < 3 * HALO * HALO; i += 256) (&s.gx[0][0][0])[i] = 0.f;
  for (int i = tid; i < 27 * HALO * HALO; i += 256) (&s.z[0][0][0])[i] = 0.f;
  stage_params_mfma(s, w1, b1, gw, gb, w2, tid);
  const float bias2[3] = {b2[0], b2[1], b2[2]};
  const float inv_n = 1.0f / (float)(CPG * PS * PS);

  // this thread's pixel of patch q (raw values): requested one patch ahead so the HBM latency hides behind the previous
  // patch's compute instead of sitting in front of the first barrier
  auto load_raw = [&](int q, float (&raw)[3]) {
    const int b = q / (nh * nw), ph = (q / nw) % nh, pw = q % nw;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const long off = (((long)b * 3 + i) * H + ph * PS + py) * W + pw * PS + px;
      raw[i] = U8 ? (float)reinterpret_cast<const unsigned char*>(images)[off] : reinterpret_cast<const float*>(images)[off];
    }
  };
  float nraw[3] = {0.f, 0.f, 0.f};
  if ((int)blockIdx.x < P) load_raw(blockIdx.x, nraw);
  for (int p = blockIdx.x; p < P; p += gridDim.x) {
    float xv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // embeddings.py:40-42: x = (x / 255.0 * 2) - 1 ; x = x / sqrt(patch_size)
      xv[i] = __fsub_rn(__fmul_rn(__fdiv_rn(nraw[i], 255.0f), 2.0f), 1.0f) * 0.25f;
      if (xp) xp[(long)p * 768 + i * 256 + tid] = xv[i];
    }
    __syncthreads();   // previous patch finished reading gx / z
#pragma unroll
    for (int i = 0; i < 3; ++i) s.gx[i][py + 1][px + 1] = gelu_f(xv[i]);
    __syncthreads();
    if (p + (int)gridDim.x < P) load_raw(p + gridDim.x, nraw);
    write_im2col_row<false>(s.im, &s.gx[0][0][0], tid, 1.0f);      // rows 64w .. 64w+63 are written and read by wave w only

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
          v = half_sum32_dpp(v);
          if (l32 == 31) s.red[pass][8 * t + 2 * qq + h][wave] = v;
          if (qq == 3) __builtin_amdgcn_sched_barrier(0);     // bound the live reduction chains (register pressure)
        }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float4 r4 = *reinterpret_cast<const float4*>(&s.red[pass][8 * t + 2 * qq + h][0]);
          const float tot = ((r4.x + r4.y) + (r4.z + r4.w)) * inv_n;
          const float k = pass == 0 ? tot : rsqrtf(tot + GN_EPS);
          // the group statistics go to the backward kernel (64 floats per patch: mean | rstd of group 8t + 2qq + h), which then
          // skips their recomputation: two block barriers and eight cross-lane reductions per channel tile
          if (gn_stats && wave == 0 && l32 == 0) gn_stats[(long)p * 64 + 32 * pass + 8 * t + 2 * qq + h] = k;
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
          for (int j = 0; j < 8; j += 2) {                     // pairs on the packed fp32 pipe (neko_common.h gelu2_f)
            const f32x2_v g2 = gelu2_f(__builtin_elementwise_fma((f32x2_v){acc[t][pt][8 * sidx + j], acc[t][pt][8 * sidx + j + 1]},
                                                                 (f32x2_v){gwv[j], gwv[j + 1]}, (f32x2_v){gbv[j], gbv[j + 1]}));
            v[j] = g2.x;
            v[j + 1] = g2.y;
            if (j == 2) __builtin_amdgcn_sched_barrier(0);     // 4 GELUs in flight at a time (register pressure)
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

// ---- MFMA backward ------------------------------------------------------------------------------------------------
// dy f32 [P,768] is the gradient of the block output (= gradient wrt conv2's output; the identity branch reaches only
// the input image).  Nothing but the normalised patch was kept by the forward: per patch and per 32-channel tile t the
// kernel recomputes conv1 + GroupNorm in registers (same layout as the forward: channel = accumulator row, lane =
// pixel) and then runs four more small contractions on v_mfma_f32_32x32x16_bf16:
//   d_h2^T[c][px] = W2[c][q] * dZ[q][px]      q = (out, dy, dx); dZ = flipped im2col of dy (im3)
//   dW2[q][c]    += dZ[q][px] * h2[px][c]     contraction over this wave's 64 pixels, accumulators live across patches
//   dW1[c][k]    += d_h1[px][c] * im2col(gelu x)[px][k]   (column 27 of the im2col is 1.0 -> db1 for free)
// h2 / d_h1 go through wave-private [pixel][channel] bf16 tiles and come back as ds_read_b64_tr_b16 fragments.  The
// per-channel sums of the GroupNorm backward (sum du, sum du*xhat over 256 pixels) use a 32-value reduce-scatter
// butterfly over the 32 pixel lanes (31 exchanges instead of 160) and one LDS exchange across the 4 waves.
struct BwdSmem {
  float gx[3][HALO][HALO];                            // GELU(x), zero halo
  float dh3[3][HALO][HALO];                           // dy, zero halo
  __attribute__((aligned(16))) bf16_t im1[PS * PS * 32];   // im2col of gx       [pixel][k]
  __attribute__((aligned(16))) bf16_t im3[PS * PS * 32];   // flipped im2col of dy [pixel][q]
  __attribute__((aligned(16))) bf16_t tt[PS * PS * 32];    // [pixel][32 channels of tile t]: first the h2 tile (dW2 operand),
                                                           // then, once those MFMAs have read it, the d_h1 tile (dW1 operand)
  __attribute__((aligned(16))) bf16_t w1[C * 32];     // [c][k], piece ^= (c>>2)&3 (bias hi/lo in columns 27/28)
  __attribute__((aligned(16))) bf16_t w2c[C * 32];    // [c][q], piece ^= (c>>2)&3
  __attribute__((aligned(16))) float red[2][8][4];    // GroupNorm statistics partials [pass][group of tile][wave]
  float stat[64];                                     // GroupNorm statistics of the patch handed over by the forward: mean[32] | rstd[32]
  float cs[4][2][32];                                 // per-wave channel sums [wave][kind][channel of tile]
  float cst[2][32];                                   // their totals
  float gw[C], gb[C];
  float acc_gn[2][C];                                 // dgamma / dbeta accumulated over the block's patches
  float fin[3][4];
};

// tile [row = k][col] with 64-B rows (32 bf16), 16-B piece ^= (row>>2)&3: 32x32x16 MFMA fragment whose 32 rows/cols
// are the tile COLUMNS and whose k = tile rows 16*ks .. 16*ks+15 (+ rowbase)
__device__ __forceinline__ bf16x8_v tr_frag32(const bf16_t* tile, int rowbase, int ks, int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  const int col = 16 * (g & 1) + 4 * (c16 & 3);
  const int krow = rowbase + ks * 16 + 8 * (g >> 1) + (c16 >> 2);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const char* b = reinterpret_cast<const char*>(tile);
  const int off_lo = krow * 64 + ((((col >> 3) ^ ((krow >> 2) & 3)) << 4) | ((col & 7) << 1));
  const int off_hi = (krow + 4) * 64 + ((((col >> 3) ^ (((krow + 4) >> 2) & 3)) << 4) | ((col & 7) << 1));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b + off_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b + off_hi));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_v, r);
}

// v[0..31] summed over the 32 lanes that share lane>>5: afterwards lane l32 holds the total of index l32.
// Recursive template so that every v[] index is a literal (a `half >>= 1` loop is not unrolled by hipcc and turned the
// register array into 32-way select chains: 7,700 v_cmp/v_cndmask pairs).
// value of lane (lane ^ HALF): DPP where the pattern exists inside a 16-lane row (xor 1, 2: quad_perm; xor 8 = rotate by 8;
// xor 4 = rotate by 4 one way or the other), the LDS crossbar (ds_bpermute) only for the cross-row xor 16
template <int HALF>
__device__ __forceinline__ float xor_lane(float x, int l32) {
  const int xi = __float_as_int(x);
  if constexpr (HALF == 1) return __int_as_float(__builtin_amdgcn_mov_dpp(xi, 0xB1, 0xf, 0xf, true));
  else if constexpr (HALF == 2) return __int_as_float(__builtin_amdgcn_mov_dpp(xi, 0x4E, 0xf, 0xf, true));
  else if constexpr (HALF == 8) return __int_as_float(__builtin_amdgcn_mov_dpp(xi, 0x128, 0xf, 0xf, true));     // row_ror:8
  else if constexpr (HALF == 4) {
    const int up = __builtin_amdgcn_mov_dpp(xi, 0x12C, 0xf, 0xf, true);      // row_ror:12: lane i <- lane (i + 4) % 16
    const int dn = __builtin_amdgcn_mov_dpp(xi, 0x124, 0xf, 0xf, true);      // row_ror:4 : lane i <- lane (i - 4) % 16
    return __int_as_float((l32 & 4) ? dn : up);
  } else return __shfl_xor(x, HALF, 64);
}
template <int HALF>
__device__ __forceinline__ void reduce_scatter_step(float (&v)[32], int l32) {
  const bool up = (l32 & HALF) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float keep = up ? v[HALF + i] : v[i];
    const float send = up ? v[i] : v[HALF + i];
    v[i] = keep + xor_lane<HALF>(send, l32);
  }
  if constexpr (HALF > 1) reduce_scatter_step<HALF / 2>(v, l32);
}
__device__ __forceinline__ float reduce_scatter32(float (&v)[32], int l32) {
  reduce_scatter_step<16>(v, l32);
  return v[0];
}

// Occupancy: 77 KB of LDS and <= 256 VGPRs per lane -> TWO blocks (8 waves, 2 per SIMD) per CU.  The first version kept
// separate h2 / d_h1 tiles (93 KB: one block, one wave per SIMD per CU) and every barrier, LDS round trip and shuffle
// chain of the per-patch dependency chain was exposed: 539 us per call at MFMA busy 3.9 % / HBM 0.9 % (r01 counters).
// STATS: the GroupNorm statistics come from the forward pass (gn_stats, see resblock_fwd_kernel) instead of being recomputed: per
// channel tile two block barriers, eight 32-lane reductions and two LDS exchanges fewer (22 -> 17 barriers per patch; round 5).
template <bool STATS>
__global__ __launch_bounds__(256, 2) void resblock_bwd_kernel(const float* __restrict__ xp, const float* __restrict__ dy,
                                                              int P, const float* __restrict__ w1,
                                                              const float* __restrict__ b1, const float* __restrict__ gw,
                                                              const float* __restrict__ gb, const float* __restrict__ w2,
                                                              float* __restrict__ part, const float* __restrict__ gn_stats) {
  // every block writes one partial row [PART_STRIDE] = dw1 | db1 | dgamma | dbeta | dw2 | db2 (plain stores);
  // resblock_param_reduce_kernel sums the rows in a fixed order (atomics onto 54 shared lines serialised in L2).
  float* const dw1 = part + (long)blockIdx.x * PART_STRIDE + OFF_W1;
  float* const db1 = part + (long)blockIdx.x * PART_STRIDE + OFF_B1;
  float* const dgw = part + (long)blockIdx.x * PART_STRIDE + OFF_GW;
  float* const dgb = part + (long)blockIdx.x * PART_STRIDE + OFF_GB;
  float* const dw2 = part + (long)blockIdx.x * PART_STRIDE + OFF_W2;
  float* const db2 = part + (long)blockIdx.x * PART_STRIDE + OFF_B2;
  __shared__ BwdSmem s;
  const int tid = threadIdx.x, py = tid >> 4, px = tid & 15, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l32 = lane & 31;
  for (int i = tid; i < 3 * HALO * HALO; i += 256) { (&s.gx[0][0][0])[i] = 0.f; (&s.dh3[0][0][0])[i] = 0.f; }
  for (int i = tid; i < 2 * C; i += 256) (&s.acc_gn[0][0])[i] = 0.f;
  for (int i = tid; i < C * 32; i += 256) {
    const int c = i >> 5, k = i & 31;
    float v = 0.f;
    if (k < 27) v = w1[c * 27 + k];
    else if (k == 27) v = bf16_to_f32(f32_to_bf16(b1[c]));
    else if (k == 28) v = b1[c] - bf16_to_f32(f32_to_bf16(b1[c]));
    const int dst = c * 32 + ((((k >> 3) ^ ((c >> 2) & 3)) << 3) | (k & 7));
    s.w1[dst] = f32_to_bf16(v);
    s.w2c[dst] = f32_to_bf16(k < 27 ? w2[((k / 9) * C + c) * 9 + (k % 9)] : 0.f);      // conv2.weight [o][c][3][3]
  }
  if (tid < C) { s.gw[tid] = gw[tid]; s.gb[tid] = gb[tid]; }
  const float inv_n = 1.0f / (float)(CPG * PS * PS);

  // Weight gradients: per channel tile t two 32x32 outputs (dW2[q][32t + n], dW1[32t + m][k]), each a contraction over the
  // patch's 256 pixels = 16 k-steps.  Wave w owns ONE of the two outputs (wkind = w & 1: 0 dW2, 1 dW1) and HALF of the
  // k-steps (pixels 128 * (w >> 1) ..): 4 x 16 accumulator registers per wave that live across patches.  (Each wave
  // contracting its own 64 pixels for both outputs needed 128, which with the recompute's own 200 registers forced one
  // wave per SIMD.)  The tiles a wave reads were written by other waves: see the barriers in channel_tile.
  const int wkind = __builtin_amdgcn_readfirstlane(wave & 1), wkhalf = __builtin_amdgcn_readfirstlane(wave >> 1);
  f32x16 dwacc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[t][r] = 0.f;
  float b2acc[3] = {0.f, 0.f, 0.f};

  // next patch's pixels are requested one patch ahead (see the forward kernel)
  float nxv[3] = {0.f, 0.f, 0.f}, ng3[3] = {0.f, 0.f, 0.f}, nstat = 0.f;
  auto load_px = [&](int q) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      nxv[i] = xp[(long)q * 768 + i * 256 + tid];
      ng3[i] = dy[(long)q * 768 + i * 256 + tid];
    }
    if (STATS && tid < 64) nstat = gn_stats[(long)q * 64 + tid];
  };
  if ((int)blockIdx.x < P) load_px(blockIdx.x);
#pragma unroll 1
  for (int p = blockIdx.x; p < P; p += gridDim.x) {
    {
      float xv[3], g3[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        xv[i] = nxv[i];
        g3[i] = ng3[i];
        b2acc[i] += g3[i];
      }
      const float st = nstat;
      __syncthreads();      // previous patch is done with the halo tiles, cs/cst, red and stat
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        s.gx[i][py + 1][px + 1] = gelu_f(xv[i]);
        s.dh3[i][py + 1][px + 1] = g3[i];
      }
      if (STATS && tid < 64) s.stat[tid] = st;
    }
    __syncthreads();
    if (p + (int)gridDim.x < P) load_px(p + gridDim.x);
    write_im2col_row<false>(s.im1, &s.gx[0][0][0], tid, 1.0f);     // rows 64w .. 64w+63: written and read by wave w only
    write_im2col_row<true>(s.im3, &s.dh3[0][0][0], tid, 0.0f);

    // one 32-channel tile; t must be a compile-time constant (dw?acc[t] are register arrays): the body is too large
    // for the unroller's threshold, so it is instantiated four times through a generic lambda instead of a loop
    auto channel_tile = [&](auto TT) __attribute__((always_inline)) {
      constexpr int t = decltype(TT)::value;
      // ---- conv1 recompute (tile t) and d_h2 (tile t) for this wave's 2 pixel tiles -------------------------------
      f32x16 acc[2], dacc[2];
      {
        const int c = 32 * t + l32;
        bf16x8_v a1[2], a2[2];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int o = c * 32 + (((2 * sidx + h) ^ ((c >> 2) & 3)) << 3);
          a1[sidx] = *reinterpret_cast<const bf16x8_v*>(&s.w1[o]);
          a2[sidx] = *reinterpret_cast<const bf16x8_v*>(&s.w2c[o]);
        }
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const int pix = 64 * wave + 32 * pt + l32;
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[pt][r] = 0.f; dacc[pt][r] = 0.f; }
#pragma unroll
          for (int sidx = 0; sidx < 2; ++sidx) {
            acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[sidx], im2col_frag(s.im1, pix, sidx, h), acc[pt], 0, 0, 0);
            dacc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[sidx], im2col_frag(s.im3, pix, sidx, h), dacc[pt], 0, 0, 0);
          }
        }
      }
      // ---- GroupNorm statistics of the tile's 8 groups (qq, h): mean removed, then scaled by rstd, in place ----------
      float rstd[4];
      if constexpr (STATS) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float mu = s.stat[8 * t + 2 * qq + h];
          rstd[qq] = s.stat[32 + 8 * t + 2 * qq + h];
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pt][4 * qq + e] = (acc[pt][4 * qq + e] - mu) * rstd[qq];
        }
        // the statistics barriers used to separate the previous tile's dW1 products (readers of the [pixel][channel] tile) from this
        // tile's h2 stores into it; tile 0 sits behind the two barriers of the patch start
        if (t > 0) __syncthreads();
      } else {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          float v = 0.f;
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = acc[pt][4 * qq + e];
              v = pass == 0 ? v + x : fmaf(x, x, v);
            }
          v = half_sum32_dpp(v);
          if (l32 == 31) s.red[pass][2 * qq + h][wave] = v;
        }
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float4 r4 = *reinterpret_cast<const float4*>(&s.red[pass][2 * qq + h][0]);
          const float tot = ((r4.x + r4.y) + (r4.z + r4.w)) * inv_n;
          const float k = pass == 0 ? tot : rsqrtf(tot + GN_EPS);
          if (pass == 1) rstd[qq] = k;
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[pt][4 * qq + e] = pass == 0 ? acc[pt][4 * qq + e] - k : acc[pt][4 * qq + e] * k;
        }
      }
      }   // !STATS
      // ---- h2 = GELU(u), du = d_h2 * GELU'(u), u = xhat*gamma + beta; per-channel sums over this lane's 2 pixels -------
      // one pixel tile at a time, h2 written to the [pixel][channel] tile straight away (registers 4qq..4qq+3 are channels
      // 8qq + 4h + {0..3}): 16 live h2 values instead of 32, and gamma / beta are re-read from LDS where needed instead
      // of being held across the barriers -- the kernel has to fit 256 registers for two waves per SIMD
      float sums[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) sums[r] = 0.f;
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        float tv[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {                     // channel pairs (c, c + 1) on the packed fp32 pipe
          const int c = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
          const f32x2_v xh = {acc[pt][r], acc[pt][r + 1]};
          const f32x2_v u = __builtin_elementwise_fma(xh, (f32x2_v){s.gw[c], s.gw[c + 1]}, (f32x2_v){s.gb[c], s.gb[c + 1]});
          f32x2_v h2, gp;
          gelu_and_grad2_f(u, h2, gp);
          tv[r] = h2.x;                                                          // h2
          tv[r + 1] = h2.y;
          const f32x2_v du = (f32x2_v){dacc[pt][r], dacc[pt][r + 1]} * gp;
          dacc[pt][r] = du.x;                                                    // d(GN out)
          dacc[pt][r + 1] = du.y;
          sums[r] += du.x;                     // kind 0: sum du      -> dbeta
          sums[r + 1] += du.y;
          sums[16 + r] = fmaf(du.x, xh.x, sums[16 + r]);   // kind 1: sum du*xhat -> dgamma
          sums[16 + r + 1] = fmaf(du.y, xh.y, sums[16 + r + 1]);
          if ((r & 3) == 2) __builtin_amdgcn_sched_barrier(0);     // 4 GELUs in flight at a time (register pressure)
        }
        const int pix = 64 * wave + 32 * pt + l32;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          *reinterpret_cast<uint2*>(&s.tt[pix * 32 + ((qq ^ ((pix >> 2) & 3)) << 3) + 4 * h]) =
              make_uint2(pack_bf16x2(tv[4 * qq], tv[4 * qq + 1]), pack_bf16x2(tv[4 * qq + 2], tv[4 * qq + 3]));
      }
      {
        const float tot = reduce_scatter32(sums, l32);       // lane l32 holds index l32 = kind*16 + r
        const int r = l32 & 15;
        s.cs[wave][l32 >> 4][(r & 3) + 8 * (r >> 2) + 4 * h] = tot;
      }
      __syncthreads();       // h2 tile and channel-sum partials of every wave are visible
      // ---- dW2 += dZ^T . h2 (waves of kind 0), between the two barriers of the channel sums: after the second one the tile
      // is free for d_h1
      if (wkind == 0) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16x8_v a4 = tr_frag32(s.im3, 128 * wkhalf, ks, lane);      // rows = q
          const bf16x8_v b4 = tr_frag32(s.tt, 128 * wkhalf, ks, lane);       // cols = channel of tile t
          dwacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a4, b4, dwacc[t], 0, 0, 0);
        }
      }
      if (tid < 64) {
        const int kind = tid >> 5, cl = tid & 31;
        const float v = (s.cs[0][kind][cl] + s.cs[1][kind][cl]) + (s.cs[2][kind][cl] + s.cs[3][kind][cl]);
        s.cst[kind][cl] = v;
        s.acc_gn[kind == 1 ? 0 : 1][32 * t + cl] += v;       // [0] dgamma, [1] dbeta
      }
      __syncthreads();
      // ---- d_h1 = rstd_g * (du*gamma - A_g/N - xhat*B_g/N) -> the tile --------------------------------------------------
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float gm[4], A = 0.f, Bv = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cl = e + 8 * qq + 4 * h;
          gm[e] = s.gw[32 * t + cl];
          A = fmaf(gm[e], s.cst[0][cl], A);
          Bv = fmaf(gm[e], s.cst[1][cl], Bv);
        }
        A *= inv_n;
        Bv *= inv_n;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const int pix = 64 * wave + 32 * pt + l32;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * qq + e;
            v[e] = rstd[qq] * (fmaf(dacc[pt][r], gm[e], -A) - acc[pt][r] * Bv);
          }
          *reinterpret_cast<uint2*>(&s.tt[pix * 32 + ((qq ^ ((pix >> 2) & 3)) << 3) + 4 * h]) =
              make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
      }
      __syncthreads();       // d_h1 tile of every wave is visible (the next tile's h2 is written two barriers later)
      // ---- dW1 += d_h1^T . im2col (waves of kind 1) ---------------------------------------------------------------------
      if (wkind == 1) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16x8_v a5 = tr_frag32(s.tt, 128 * wkhalf, ks, lane);       // rows = channel of tile t
          const bf16x8_v b5 = tr_frag32(s.im1, 128 * wkhalf, ks, lane);      // cols = k (27: ones -> db1)
          dwacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a5, b5, dwacc[t], 0, 0, 0);
        }
      }
    };   // channel tile
    channel_tile(std::integral_constant<int, 0>{});
    channel_tile(std::integral_constant<int, 1>{});
    channel_tile(std::integral_constant<int, 2>{});
    channel_tile(std::integral_constant<int, 3>{});
  }     // patches

  // ---- block result: the two k-halves of every output meet in LDS (im1 is dead), one channel tile at a time -----------
  __syncthreads();
  float* scr = reinterpret_cast<float*>(s.im1);      // [wave][32 x 32] floats
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
      scr[wave * 1024 + m * 32 + l32] = dwacc[t][r];
    }
    __syncthreads();
    for (int i = tid; i < 2048; i += 256) {
      const int which = i >> 10, e = i & 1023;             // waves (which, which + 2) hold the two halves of output `which`
      const float v = scr[which * 1024 + e] + scr[(which + 2) * 1024 + e];
      const int m = e >> 5, n = e & 31;
      if (which == 0) {                              // dW2[q = m][c = 32t + n]
        if (m < 27) dw2[((m / 9) * C + 32 * t + n) * 9 + (m % 9)] = v;
      } else {                                       // dW1[c = 32t + m][k = n]; k == 27 is the ones column
        if (n < 27) dw1[(32 * t + m) * 27 + n] = v;
        else if (n == 27) db1[32 * t + m] = v;
      }
    }
    __syncthreads();
  }
  if (tid < C) {
    dgw[tid] = s.acc_gn[0][tid];
    dgb[tid] = s.acc_gn[1][tid];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float r = wave_sum(b2acc[i]);
    if (lane == 0) s.fin[i][wave] = r;
  }
  __syncthreads();
  if (tid < 3) db2[tid] = (s.fin[tid][0] + s.fin[tid][1]) + (s.fin[tid][2] + s.fin[tid][3]);
}

// grads (+)= sum over the per-block partial rows, fixed order.  Block = 64 elements x 4 row groups (row r belongs to
// group r & 3), coalesced across elements; the 4 group sums meet in LDS.
__global__ __launch_bounds__(256) void resblock_param_reduce_kernel(const float* __restrict__ part, int nblk,
                                                                    float* __restrict__ dw1, float* __restrict__ db1,
                                                                    float* __restrict__ dgw, float* __restrict__ dgb,
                                                                    float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ float sm[4][64];
  const int e = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f;
  if (j < PART_USED) {
    int b = rg;
    for (; b + 4 < nblk; b += 8) {
      s0 += part[(long)b * PART_STRIDE + j];
      s1 += part[(long)(b + 4) * PART_STRIDE + j];
    }
    if (b < nblk) s0 += part[(long)b * PART_STRIDE + j];
  }
  sm[rg][e] = s0 + s1;
  __syncthreads();
  if (rg != 0 || j >= PART_USED) return;
  const float v = (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
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
                                 hipStream_t s, float* gn_stats) {
  if (n <= 0) return NEKO_OK;
  if (!images || !w1 || !b1 || !gn_w || !gn_b || !w2 || !b2 || !y16) return NEKO_ERR_ARG;
  if (H <= 0 || W <= 0 || (H % PS) || (W % PS)) return NEKO_ERR_ARG;   // "Image dimensions must be divisible by patch size"
  if (mid_channels != C || num_groups != G) return NEKO_ERR_UNSUPPORTED;
  const int P = n * (H / PS) * (W / PS);
  const int grid = P < 2048 ? P : 2048;
  if (images_are_u8)
    hipLaunchKernelGGL((resblock_fwd_kernel<true>), dim3(grid), dim3(256), 0, s, images, n, H, W, w1, b1, gn_w, gn_b,
                       w2, b2, y16, x_patches, gn_stats);
  else
    hipLaunchKernelGGL((resblock_fwd_kernel<false>), dim3(grid), dim3(256), 0, s, images, n, H, W, w1, b1, gn_w, gn_b,
                       w2, b2, y16, x_patches, gn_stats);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// blocks (= partial rows) the backward uses for P patches; workspace = blocks * neko_patch_resblock_ws_stride floats
int neko_patch_resblock_bwd_blocks_impl(int P) { return P < 512 ? (P < 1 ? 1 : P) : 512; }   // two blocks per CU
int neko_patch_resblock_ws_stride_impl() { return PART_STRIDE; }

int neko_patch_resblock_bwd_impl(const float* x_patches, const float* dy, int P, const float* w1, const float* b1,
                                 const float* gn_w, const float* gn_b, const float* w2, const float* b2,
                                 int mid_channels, int num_groups, float* dw1, float* db1, float* dgn_w, float* dgn_b,
                                 float* dw2, float* db2, float* workspace, hipStream_t s, const float* gn_stats) {
  (void)b2;
  if (P <= 0) return NEKO_OK;
  if (!x_patches || !dy || !w1 || !b1 || !gn_w || !gn_b || !w2 || !dw1 || !db1 || !dgn_w || !dgn_b || !dw2 || !db2 ||
      !workspace)
    return NEKO_ERR_ARG;
  if (mid_channels != C || num_groups != G) return NEKO_ERR_UNSUPPORTED;
  const int grid = neko_patch_resblock_bwd_blocks_impl(P);
  if (gn_stats)
    hipLaunchKernelGGL((resblock_bwd_kernel<true>), dim3(grid), dim3(256), 0, s, x_patches, dy, P, w1, b1, gn_w, gn_b, w2,
                       workspace, gn_stats);
  else
    hipLaunchKernelGGL((resblock_bwd_kernel<false>), dim3(grid), dim3(256), 0, s, x_patches, dy, P, w1, b1, gn_w, gn_b, w2,
                       workspace, gn_stats);
  NEKO_CHECK_LAUNCH();
  hipLaunchKernelGGL(resblock_param_reduce_kernel, dim3((PART_USED + 63) / 64), dim3(256), 0, s, workspace, grid, dw1,
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

// ... and from HOST-sorted (position, patch) pairs (round 5: the host draws the patch positions, embeddings.py:63-110, so it sorts them too)
long neko_patch_pos_add_bwd_sorted_ws_bytes_impl(int P, int d) { return P > 0 ? (long)neko_segsum_sorted_ws_bytes_impl(P, d) : 0; }
int neko_patch_pos_add_bwd_sorted_impl(const float* dout, const unsigned* hkeys, const int* hidx, const unsigned* wkeys, const int* widx,
                                       float* d_row_emb, float* d_col_emb, int P, int d, int nrows, void* ws, long ws_bytes, hipStream_t s) {
  if (P <= 0) return NEKO_OK;
  if (!dout || !hkeys || !hidx || !wkeys || !widx || !d_row_emb || !d_col_emb || !ws || nrows <= 0) return NEKO_ERR_ARG;
  if ((unsigned)nrows >= NEKO_SEGSUM_KEY_NONE) return NEKO_ERR_UNSUPPORTED;
  int rc = neko_segsum_rows_sorted_impl(dout, d, hkeys, hidx, P, d, d_row_emb, d, nrows, nullptr, ws, (size_t)ws_bytes, s);
  if (rc != NEKO_OK) return rc;
  return neko_segsum_rows_sorted_impl(dout, d, wkeys, widx, P, d, d_col_emb, d, nrows, nullptr, ws, (size_t)ws_bytes, s);
}

long neko_patch_pos_add_bwd_det_ws_bytes_impl(int P, int d) { return P > 0 ? (long)neko_segsum_ws_bytes_impl(P, d) : 0; }
// the same two gradients without atomics (segsum.hip): every table row is the sum of its patches' rows in patch order
int neko_patch_pos_add_bwd_det_impl(const float* dout, const int* hpos, const int* wpos, float* d_row_emb, float* d_col_emb, int P,
                                    int d, int nrows, void* ws, long ws_bytes, hipStream_t s) {
  if (P <= 0) return NEKO_OK;
  if (!dout || !hpos || !wpos || !d_row_emb || !d_col_emb || !ws || nrows <= 0) return NEKO_ERR_ARG;
  if ((unsigned)nrows >= NEKO_SEGSUM_KEY_NONE) return NEKO_ERR_UNSUPPORTED;     // the sort orders on 20 key bits
  int rc = neko_segsum_rows_impl(dout, d, reinterpret_cast<const unsigned*>(hpos), P, d, d_row_emb, d, nrows, nullptr, ws, (size_t)ws_bytes, s);
  if (rc != NEKO_OK) return rc;
  return neko_segsum_rows_impl(dout, d, reinterpret_cast<const unsigned*>(wpos), P, d, d_col_emb, d, nrows, nullptr, ws, (size_t)ws_bytes, s);
}
