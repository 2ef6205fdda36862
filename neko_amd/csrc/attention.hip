// Fused causal + key-padding-masked self-attention, forward and backward, bf16 MFMA / fp32 softmax.
// Replaces Attention._attn + split_heads/merge_heads + mask prep of the reference
// (gato/transformers/trajectory_gpt2.py:163-188,190-201,222-226,252 and :663-679) and their autograd.
//
// Mask semantics are the reference's, literally:
//     s = (q.k)/sqrt(hd);  s = (key <= query) ? s : -1e4   (REPLACE, finite);  s += (1-mask[key]) * -1e4 (ADD)
// so padded *query* rows still produce the reference's finite outputs (they see every key).
// A query tile with no masked query row only visits key tiles up to its diagonal (and skips a fully
// padded left prefix): for those rows every skipped term is exp(<= -1e4 - m) == 0 in fp32.
//
// Layout: qkv bf16 [B*T, 3*d] as written by the c_attn GEMM (q | k | v, head h at columns h*hd);
// out bf16 [B*T, d] (merge_heads layout); lse fp32 [B,H,T]; no (B,H,T,T) tensor ever exists.
//
// gfx950 mapping: 4 waves per block, "one lane owns one query (or key) column": scores are computed
// transposed, S^T = K.Q^T with v_mfma_f32_32x32x16_bf16, so the 32x32 accumulator leaves each lane with
// 16 keys of ONE query -> the online softmax is in-lane plus a single lane^32 exchange, and the bf16
// probabilities are already the B operand of the P.V MFMA (O^T = V^T.P^T) with the key order
// permuted identically on the V^T side (LDS image [hd][key], 8-byte fragment reads).  K / V^T / Q^T / dO^T
// tiles are staged through padded LDS images; everything is fp32 except MFMA operands.
#include <cstdlib>
#include "neko_kernels.h"

extern int neko_attn_path_mode();

namespace {

constexpr int NT = 256;
#ifndef NEKO_FWD_WAVES
#define NEKO_FWD_WAVES 4   // forward waves per SIMD at hd = 32
#endif
#ifndef NEKO_DKV_WAVES
#define NEKO_DKV_WAVES 3   // dK/dV waves per SIMD at hd = 32: the branch-free inner loop spills at 4 (128 VGPRs) and is 12 % slower
#endif
constexpr int KT = 64;          // keys (or queries, in dK/dV) per inner tile
constexpr int TSTR = KT * 2 + 8;  // transposed image row stride in bytes (136: conflict-free ds_read_b64)
constexpr float MASK_VAL = -10000.0f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
// the softmax runs in the exp2 domain: scores are pre-multiplied by log2(e) (folded into the 1/sqrt(hd) scale),
// v_exp_f32 is 2^x natively, and LSE is converted back to natural log when stored.
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// Workgroups are dealt to the 8 XCDs round-robin by linear id (x fastest).  With T = 1024 the grid is 8 tiles wide,
// so tile index == XCD: one XCD got every heaviest causal tile (16 key tiles) and another every lightest (2) -- the
// kernel ran at the speed of the first.  Rotating the tile index by (head + batch) gives every XCD the same mix.
__device__ __forceinline__ int rotated_tile() { return (int)((blockIdx.x + blockIdx.y + blockIdx.z) % gridDim.x); }

template <int HD> struct Cfg {
  static constexpr int NSTR = HD * 2 + 16;    // natural image row stride (bytes)
  static constexpr int KS = HD / 16;          // MFMA k-steps over the head dim
  static constexpr int IB = HD / 32;          // 32-row blocks over the head dim
  static constexpr int CH = HD / 8;           // 16-B chunks per row
  static constexpr int LPT = (KT * CH) / NT;  // 16-B loads per thread per 64-row tile (>=1)
  static constexpr int NAT_BYTES = KT * NSTR;
  static constexpr int TR_BYTES = HD * TSTR;
};

// ---- staging helpers (64 rows x HD of a [*, ld] bf16 matrix) ------------------------------------
template <int HD>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ base, long ld, int row0, int nrows, int tid,
                                          uint4 (&reg)[Cfg<HD>::LPT]) {
#pragma unroll
  for (int i = 0; i < Cfg<HD>::LPT; ++i) {
    const int c = tid + NT * i;
    const int row = c / Cfg<HD>::CH, ch = c % Cfg<HD>::CH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + row < nrows) v = *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * ld + ch * 8);
    reg[i] = v;
  }
}
template <int HD>
__device__ __forceinline__ void tile_store_nat(char* lds, int tid, const uint4 (&reg)[Cfg<HD>::LPT]) {
#pragma unroll
  for (int i = 0; i < Cfg<HD>::LPT; ++i) {
    const int c = tid + NT * i;
    const int row = c / Cfg<HD>::CH, ch = c % Cfg<HD>::CH;
    *reinterpret_cast<uint4*>(lds + row * Cfg<HD>::NSTR + ch * 16) = reg[i];
  }
}
template <int HD>
__device__ __forceinline__ void tile_store_tr(char* lds, int tid, const uint4 (&reg)[Cfg<HD>::LPT]) {
#pragma unroll
  for (int i = 0; i < Cfg<HD>::LPT; ++i) {
    const int c = tid + NT * i;
    const int row = c / Cfg<HD>::CH, ch = c % Cfg<HD>::CH;
    const uint32_t w[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
#pragma unroll
    for (int e = 0; e < 8; ++e)
      *reinterpret_cast<uint16_t*>(lds + (ch * 8 + e) * TSTR + row * 2) = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
  }
}

// A fragment from a natural image: row (rowbase + lane%32), head-dim slots ks*16 + 8*(lane/32) + 0..7
template <int HD>
__device__ __forceinline__ bf16x8_v frag_nat(const char* lds, int rowbase, int ks, int lane) {
  const uint4 v = *reinterpret_cast<const uint4*>(lds + (rowbase + (lane & 31)) * Cfg<HD>::NSTR +
                                                  (ks * 2 + (lane >> 5)) * 16);
  return __builtin_bit_cast(bf16x8_v, v);
}
// A fragment from a transposed image [hd][64]: row (hdbase + lane%32); contraction slots j=0..7 map to
// column 16*s + 8*(j>>2) + 4*(lane/32) + (j&3)  == the order the 32x32 accumulator leaves in a lane.
__device__ __forceinline__ bf16x8_v frag_tr(const char* lds, int hdbase, int s, int lane) {
  const char* p = lds + (hdbase + (lane & 31)) * TSTR + (16 * s + 4 * (lane >> 5)) * 2;
  const uint2 lo = *reinterpret_cast<const uint2*>(p);
  const uint2 hi = *reinterpret_cast<const uint2*>(p + 16);
  return __builtin_bit_cast(bf16x8_v, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// B fragment for contraction step s (16 columns of the 64-wide tile) from two 32x32 accumulators
__device__ __forceinline__ bf16x8_v frag_from_acc(const f32x16 (&t)[2], int s) {
  const f32x16& a = t[s >> 1];
  const int o = 8 * (s & 1);
  const uint4 r = make_uint4(pack_bf16x2(a[o + 0], a[o + 1]), pack_bf16x2(a[o + 2], a[o + 3]),
                             pack_bf16x2(a[o + 4], a[o + 5]), pack_bf16x2(a[o + 6], a[o + 7]));
  return __builtin_bit_cast(bf16x8_v, r);
}
// own-row B fragments (lane's query/key row, head-dim slots ks*16 + 8*(lane/32)..+7) straight from HBM
template <int HD>
__device__ __forceinline__ void row_frags(const bf16_t* __restrict__ rowptr, bool valid, int lane,
                                          bf16x8_v (&f)[Cfg<HD>::KS]) {
#pragma unroll
  for (int ks = 0; ks < Cfg<HD>::KS; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid) v = *reinterpret_cast<const uint4*>(rowptr + ks * 16 + (lane >> 5) * 8);
    f[ks] = __builtin_bit_cast(bf16x8_v, v);
  }
}
// attention-probability dropout index (2-D so that four consecutive keys of one query always share a hash word, for
// any T): element (row = (b*H + h)*T + q, key) keeps iff byte (key & 3) of drop_word(row * T4 + (key >> 2)) >= thr,
// T4 = ceil(T / 4).
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x, int i) {   // value of lane (lane & ~3) + i, i literal 0..3
  switch (i) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x00, 0xf, 0xf, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x55, 0xf, 0xf, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xAA, 0xf, 0xf, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xFF, 0xf, 0xf, true);
  }
}
// index inside a 32-wide accumulator tile of register r for this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// =====================================================================================================
// forward
// =====================================================================================================
template <int HD, bool DROP>
__device__ __forceinline__ void attn_fwd_tile(const bf16_t* __restrict__ qkv, const float* __restrict__ kbias,
                                                      const int* __restrict__ kstart, bf16_t* __restrict__ out,
                                                      float* __restrict__ lse, int B, int T, int H, float scale,
                                                      uint32_t drop_thr, uint32_t drop_key, float drop_scale, const int tile) {
  using C = Cfg<HD>;
  __shared__ __attribute__((aligned(16))) char smem[C::NAT_BYTES + C::TR_BYTES + KT * 4 + 16];
  char* ldsK = smem;
  char* ldsVt = smem + C::NAT_BYTES;
  float* ldsKb = reinterpret_cast<float*>(smem + C::NAT_BYTES + C::TR_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = tile;
  const int h = blockIdx.y, b = blockIdx.z;
  const int d = H * HD;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + (long)b * T * ld + h * HD;
  const bf16_t* kbase = qbase + d;
  const bf16_t* vbase = qbase + 2 * d;
  const float* kb = kbias + (long)b * T;

  const int q0 = qt * 128;
  const int qw0 = q0 + wave * 32;
  const int q = qw0 + (lane & 31);
  const bool qvalid = q < T;

  bf16x8_v qf[C::KS];
  row_frags<HD>(qbase + (long)q * ld, qvalid, lane, qf);

  // does this query tile hold a masked query row?  (then the whole key range is visited)
  int masked_q = 0;
  if (tid < 128 && q0 + tid < T) masked_q = (kb[q0 + tid] != 0.f);
  const int full = __syncthreads_or(masked_q);
  // only the waves that hold a masked (padded) query row need the keys beyond their diagonal: for every other
  // row those scores are -1e4 and exp(-1e4 - m) == 0 in fp32
  const bool wave_full = __builtin_amdgcn_ballot_w64(qvalid && kb[min(q, T - 1)] != 0.f) != 0;
  const int qmax = min(q0 + 127, T - 1);
  const int kt_end = full ? (T + KT - 1) / KT : qmax / KT + 1;
  const int kt_beg = full ? 0 : (kstart ? kstart[b] / KT : 0);

  f32x16 o[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float scale2 = scale * LOG2E;

  uint4 rk[C::LPT], rv[C::LPT];
  float rkb = 0.f;
  auto prefetch = [&](int kt) {
    const int k0 = kt * KT;
    tile_load<HD>(kbase, ld, k0, T, tid, rk);
    tile_load<HD>(vbase, ld, k0, T, tid, rv);
    if (tid < KT) rkb = (k0 + tid < T) ? kb[k0 + tid] : 0.f;
  };
  if (kt_beg < kt_end) prefetch(kt_beg);

  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const int k0 = kt * KT;
    __syncthreads();
    tile_store_nat<HD>(ldsK, tid, rk);
    tile_store_tr<HD>(ldsVt, tid, rv);
    if (tid < KT) ldsKb[tid] = rkb;
    const int has_pad = __syncthreads_or(tid < KT && rkb != 0.f);
#ifndef NEKO_ATTN_DIAG_NOLOAD
    if (kt + 1 < kt_end) prefetch(kt + 1);
#endif

    // two 32-key sub-tiles, one at a time (only 16 score registers live): S^T = K.Q^T -> masks -> online
    // softmax (lane pair (l, l^32) shares a query) -> O^T += V^T.P^T
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!wave_full && k0 + t * 32 > qw0 + 31) continue;     // wave-uniform: nothing visible, no masked row
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks)
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_nat<HD>(ldsK, t * 32, ks, lane), qf[ks], st, 0, 0, 0);
      // wave-uniform choice of the cheap path: every key of the sub-tile is visible to every query of the wave
      const bool interior = (k0 + t * 32 + 31 <= qw0) && !has_pad && (k0 + KT <= T);
      if (interior) {
        // scale folded into the exp2 argument below: the running max is tracked on scaled values, max commutes with a
        // positive scale
      } else {
        const int lim_causal = q - k0 - t * 32 - 4 * (lane >> 5);        // key <= q  <=>  c(r) <= lim_causal
        const int lim_len = T - 1 - k0 - t * 32 - 4 * (lane >> 5);       // key <  T  <=>  c(r) <= lim_len
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          float v = (c <= lim_causal) ? st[r] * scale2 : MASK_VAL * LOG2E;
          v = fmaf(ldsKb[t * 32 + c + 4 * (lane >> 5)], LOG2E, v);
          st[r] = (c <= lim_len) ? v : -INFINITY;
        }
      }
      float mx = fmaxf(st[0], st[1]);
#pragma unroll
      for (int r = 2; r < 16; ++r) mx = fmaxf(mx, st[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float sc = interior ? scale2 : 1.0f;      // interior scores are still unscaled
      // lazy rescale: the reference maximum only moves when some row's maximum grew by more than 2^8 (wave-uniform
      // decision).  Probabilities are then taken relative to a slightly stale maximum (p <= 256, harmless in fp32
      // accumulators and bf16 operands) and the result is algebraically the same after the final division by l;
      // it removes the per-sub-tile o *= alpha (8 packed multiplies of ~53 VALU instructions) almost always.
      const float cand = mx * sc;
      if (__builtin_amdgcn_ballot_w64(cand > m_run + 8.0f) != 0) {
        const float m_new = fmaxf(m_run, cand);
        const float alpha = exp2_fast(m_run - m_new);   // 2^(-inf) = 0 on the first tile
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < C::IB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        m_run = m_new;
      }
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        st[r] = exp2_fast(fmaf(st[r], sc, -m_run));
        st[r + 1] = exp2_fast(fmaf(st[r + 1], sc, -m_run));
        ps0 += st[r];
        ps1 += st[r + 1];
      }
      l_run += ps0 + ps1;
      if (DROP) {   // attn_dropout on the probabilities (trajectory_gpt2.py:179): the normaliser stays undropped
        const uint32_t g0 = ((uint32_t)(b * H + h) * (uint32_t)T + (uint32_t)q) * (uint32_t)((T + 3) >> 2) +
                            (uint32_t)((k0 + t * 32 + 4 * (lane >> 5)) >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // registers 4j..4j+3 = keys +8j .. +8j+3: one word
          const uint32_t w = drop_word(g0 + 2 * j, drop_key);
#pragma unroll
          for (int e = 0; e < 4; ++e) st[4 * j + e] = drop_byte_keep(w, e, drop_thr) ? st[4 * j + e] : 0.f;   // survivor scale: folded into the final 1/l
        }
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const uint4 pw = make_uint4(pack_bf16x2(st[8 * h2 + 0], st[8 * h2 + 1]), pack_bf16x2(st[8 * h2 + 2], st[8 * h2 + 3]),
                                    pack_bf16x2(st[8 * h2 + 4], st[8 * h2 + 5]), pack_bf16x2(st[8 * h2 + 6], st[8 * h2 + 7]));
        const bf16x8_v pf = __builtin_bit_cast(bf16x8_v, pw);
#pragma unroll
        for (int i = 0; i < C::IB; ++i)
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(ldsVt, i * 32, 2 * t + h2, lane), pf, o[i], 0, 0, 0);
      }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qvalid) {
    const float inv = (DROP ? drop_scale : 1.0f) / l_tot;
    bf16_t* orow = out + ((long)b * T + q) * d + h * HD;
#pragma unroll
    for (int i = 0; i < C::IB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
        pk.y = pack_bf16x2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
      }
    if (lane < 32) lse[((long)b * H + h) * T + q] = m_run * LN2 + __logf(l_tot);
  }
}

template <int HD, bool DROP>
__global__ __launch_bounds__(NT, (HD <= 32 ? NEKO_FWD_WAVES : (HD <= 64 ? 3 : 2))) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ kbias,
                                                      const int* __restrict__ kstart, bf16_t* __restrict__ out,
                                                      float* __restrict__ lse, int B, int T, int H, float scale,
                                                      uint32_t drop_thr, uint32_t drop_key, float drop_scale) {
  // one workgroup = a heavy and a light causal tile (G-1-p and p): every workgroup walks the same number of
  // 64-row tiles.  With one tile per workgroup the whole grid was co-resident and the kernel lasted as long as its
  // heaviest workgroup (16 tile iterations at T = 1024 against an average of 9).
  if (DROP) drop_key += neko_drop_salt();
  const int G = (T + 127) / 128, p = rotated_tile();
  const int first = G - 1 - p, second = p;
  attn_fwd_tile<HD, DROP>(qkv, kbias, kstart, out, lse, B, T, H, scale, drop_thr, drop_key, drop_scale, first);
  if (second != first) {
    __syncthreads();
    attn_fwd_tile<HD, DROP>(qkv, kbias, kstart, out, lse, B, T, H, scale, drop_thr, drop_key, drop_scale, second);
  }
}

// =====================================================================================================
// backward prep: D[b,h,q] = sum_hd dO*O  and per (b, 64-query tile) "holds a masked query row" flags
// =====================================================================================================
__global__ void attn_bwd_prep_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                     const float* __restrict__ kbias, float* __restrict__ D, int* __restrict__ qflags,
                                     int B, int T, int H, int HD, float inv_drop_scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*T*H
  const long total = (long)B * T * H;
  if (idx < total) {
    const int h = idx % H;
    const long row = idx / H;
    const int b = row / T, q = row % T;
    const bf16_t* po = o + row * (long)H * HD + h * HD;
    const bf16_t* pd = dout + row * (long)H * HD + h * HD;
    float acc = 0.f;
    for (int c = 0; c < HD; c += 8) {
      const uint4 a = *reinterpret_cast<const uint4*>(po + c);
      const uint4 g = *reinterpret_cast<const uint4*>(pd + c);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += bf16_to_f32((bf16_t)(aw[e] & 0xffff)) * bf16_to_f32((bf16_t)(gw[e] & 0xffff));
        acc += bf16_to_f32((bf16_t)(aw[e] >> 16)) * bf16_to_f32((bf16_t)(gw[e] >> 16));
      }
    }
    D[((long)b * H + h) * T + q] = acc * inv_drop_scale;      // D / s: the survivor scale s is folded out of dS (see dq / dkv)
  }
  // bit 0 / bit 1: the first / second 32-query half of the 64-query tile holds a masked (padded) query row
  const int nqt = (T + KT - 1) / KT;
  if (idx < (long)B * nqt) {
    const int b = idx / nqt, t = idx % nqt;
    int f = 0;
    for (int i = t * KT; i < min(T, (t + 1) * KT); ++i) f |= (kbias[(long)b * T + i] != 0.f) << ((i - t * KT) >> 5);
    qflags[idx] = f;
  }
}

// =====================================================================================================
// backward dQ: lanes own queries (same geometry as forward)
// =====================================================================================================
template <int HD, bool DROP>
__device__ __forceinline__ void attn_bwd_dq_tile(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                         const float* __restrict__ kbias, const int* __restrict__ kstart,
                                                         const float* __restrict__ lse, const float* __restrict__ Dv,
                                                         bf16_t* __restrict__ dqkv, int B, int T, int H, float scale,
                                                         uint32_t drop_thr, uint32_t drop_key, float drop_scale, const int tile) {
  using C = Cfg<HD>;
  __shared__ __attribute__((aligned(16))) char smem[2 * C::NAT_BYTES + C::TR_BYTES + KT * 4 + 16];
  char* ldsK = smem;
  char* ldsV = smem + C::NAT_BYTES;
  char* ldsKt = smem + 2 * C::NAT_BYTES;
  float* ldsKb = reinterpret_cast<float*>(smem + 2 * C::NAT_BYTES + C::TR_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = tile;
  const int h = blockIdx.y, b = blockIdx.z;
  const int d = H * HD;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + (long)b * T * ld + h * HD;
  const bf16_t* kbase = qbase + d;
  const bf16_t* vbase = qbase + 2 * d;
  const float* kb = kbias + (long)b * T;

  const int q0 = qt * 128, qw0 = q0 + wave * 32;
  const int q = qw0 + (lane & 31);
  const bool qvalid = q < T;

  bf16x8_v qf[C::KS], dof[C::KS];
  row_frags<HD>(qbase + (long)q * ld, qvalid, lane, qf);
  row_frags<HD>(dout + ((long)b * T + q) * d + h * HD, qvalid, lane, dof);
  const float my_lse = (qvalid ? lse[((long)b * H + h) * T + q] : 0.f) * LOG2E;
  const float scale2 = scale * LOG2E;
  const float my_D = qvalid ? Dv[((long)b * H + h) * T + q] : 0.f;

  int masked_q = 0;
  if (tid < 128 && q0 + tid < T) masked_q = (kb[q0 + tid] != 0.f);
  const int full = __syncthreads_or(masked_q);
  // only the waves that hold a masked (padded) query row need the keys beyond their diagonal: for every other
  // row those scores are -1e4 and exp(-1e4 - m) == 0 in fp32
  const bool wave_full = __builtin_amdgcn_ballot_w64(qvalid && kb[min(q, T - 1)] != 0.f) != 0;
  const int qmax = min(q0 + 127, T - 1);
  const int kt_end = full ? (T + KT - 1) / KT : qmax / KT + 1;
  const int kt_beg = full ? 0 : (kstart ? kstart[b] / KT : 0);

  f32x16 dq[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  uint4 rk[C::LPT], rv[C::LPT];
  float rkb = 0.f;
  auto prefetch = [&](int kt) {
    const int k0 = kt * KT;
    tile_load<HD>(kbase, ld, k0, T, tid, rk);
    tile_load<HD>(vbase, ld, k0, T, tid, rv);
    if (tid < KT) rkb = (k0 + tid < T) ? kb[k0 + tid] : 0.f;
  };
  if (kt_beg < kt_end) prefetch(kt_beg);

  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const int k0 = kt * KT;
    __syncthreads();
    tile_store_nat<HD>(ldsK, tid, rk);
    tile_store_tr<HD>(ldsKt, tid, rk);
    tile_store_nat<HD>(ldsV, tid, rv);
    if (tid < KT) ldsKb[tid] = rkb;
    const int has_pad = __syncthreads_or(tid < KT && rkb != 0.f);
    if (kt + 1 < kt_end) prefetch(kt + 1);

    // one 32-key sub-tile at a time (16 score + 16 dP registers live)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!wave_full && k0 + t * 32 > qw0 + 31) continue;     // wave-uniform: nothing visible, no masked row
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_nat<HD>(ldsK, t * 32, ks, lane), qf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_nat<HD>(ldsV, t * 32, ks, lane), dof[ks], dpt, 0, 0, 0);
      }
      // dS^T = P^T o (dP^T - D); zero where the score was REPLACED by the causal constant
      const int lim_causal = q - k0 - t * 32 - 4 * (lane >> 5);
      const int lim_len = T - 1 - k0 - t * 32 - 4 * (lane >> 5);
      const uint32_t g0 = ((uint32_t)(b * H + h) * (uint32_t)T + (uint32_t)q) * (uint32_t)((T + 3) >> 2) +
                          (uint32_t)((k0 + t * 32 + 4 * (lane >> 5)) >> 2);
      // wave-uniform fast path: every key of the sub-tile is visible to every query of the wave and none is padded
      const bool interior = (k0 + t * 32 + 31 <= qw0) && !has_pad && (k0 + KT <= T);
      if (interior) {
        const float nlse = -my_lse;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = DROP ? drop_word(g0 + 2 * j, drop_key) : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * j + e;
            const float pv = exp2_fast(fmaf(st[r], scale2, nlse));
            float dpe = dpt[r];
            if (DROP) dpe = drop_byte_keep(w, e, drop_thr) ? dpe : 0.f;
            st[r] = pv * (dpe - my_D);
          }
        }
      } else
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w = DROP ? drop_word(g0 + 2 * j, drop_key) : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * j + e;
          const int c = (r & 3) + 8 * (r >> 2);
          const bool causal_ok = c <= lim_causal;
          const float sv = fmaf(ldsKb[t * 32 + c + 4 * (lane >> 5)], LOG2E, causal_ok ? st[r] * scale2 : MASK_VAL * LOG2E);
          const float pv = exp2_fast((c <= lim_len) ? sv - my_lse : -INFINITY);      // select, not a branch: 2^-inf = 0
          float dpe = dpt[r];
          if (DROP) dpe = drop_byte_keep(w, e, drop_thr) ? dpe : 0.f;
          st[r] = causal_ok ? pv * (dpe - my_D) : 0.f;
        }
      }
      // dQ^T += K^T . dS^T
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const uint4 w = make_uint4(pack_bf16x2(st[8 * h2 + 0], st[8 * h2 + 1]), pack_bf16x2(st[8 * h2 + 2], st[8 * h2 + 3]),
                                   pack_bf16x2(st[8 * h2 + 4], st[8 * h2 + 5]), pack_bf16x2(st[8 * h2 + 6], st[8 * h2 + 7]));
        const bf16x8_v df = __builtin_bit_cast(bf16x8_v, w);
#pragma unroll
        for (int i = 0; i < C::IB; ++i)
          dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(ldsKt, i * 32, 2 * t + h2, lane), df, dq[i], 0, 0, 0);
      }
    }
  }

  // dS was formed as P o (keep*dP - D/s): the dropout survivor scale s multiplies the result once, here
  const float qs = scale * (DROP ? drop_scale : 1.0f);
  if (qvalid) {
    bf16_t* orow = dqkv + ((long)b * T + q) * ld + h * HD;
#pragma unroll
    for (int i = 0; i < C::IB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dq[i][4 * g + 0] * qs, dq[i][4 * g + 1] * qs);
        pk.y = pack_bf16x2(dq[i][4 * g + 2] * qs, dq[i][4 * g + 3] * qs);
        *reinterpret_cast<uint2*>(orow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
      }
  }
}

template <int HD, bool DROP>
__global__ __launch_bounds__(NT, (HD <= 32 ? 4 : (HD <= 64 ? 2 : 1))) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                         const float* __restrict__ kbias, const int* __restrict__ kstart,
                                                         const float* __restrict__ lse, const float* __restrict__ Dv,
                                                         bf16_t* __restrict__ dqkv, int B, int T, int H, float scale,
                                                         uint32_t drop_thr, uint32_t drop_key, float drop_scale) {
  // one workgroup = a heavy and a light causal tile (G-1-p and p): every workgroup walks the same number of
  // 64-row tiles.  With one tile per workgroup the whole grid was co-resident and the kernel lasted as long as its
  // heaviest workgroup (16 tile iterations at T = 1024 against an average of 9).
  if (DROP) drop_key += neko_drop_salt();
  const int G = (T + 127) / 128, p = rotated_tile();
  const int first = G - 1 - p, second = p;
  attn_bwd_dq_tile<HD, DROP>(qkv, dout, kbias, kstart, lse, Dv, dqkv, B, T, H, scale, drop_thr, drop_key, drop_scale, first);
  if (second != first) {
    __syncthreads();
    attn_bwd_dq_tile<HD, DROP>(qkv, dout, kbias, kstart, lse, Dv, dqkv, B, T, H, scale, drop_thr, drop_key, drop_scale, second);
  }
}

// =====================================================================================================
// backward dK/dV: lanes own keys; loop over 64-query tiles
// =====================================================================================================
template <int HD, bool DROP>
__device__ __forceinline__ void attn_bwd_dkv_tile(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ kbias, const float* __restrict__ lse,
                                                          const float* __restrict__ Dv, const int* __restrict__ qflags,
                                                          bf16_t* __restrict__ dqkv, int B, int T, int H, float scale,
                                                          uint32_t drop_thr, uint32_t drop_key, float drop_scale, const int tile) {
  using C = Cfg<HD>;
  __shared__ __attribute__((aligned(16))) char smem[2 * C::NAT_BYTES + 2 * C::TR_BYTES + 2 * KT * 4];
  char* ldsQ = smem;
  char* ldsdO = smem + C::NAT_BYTES;
  char* ldsQt = smem + 2 * C::NAT_BYTES;
  char* ldsdOt = smem + 2 * C::NAT_BYTES + C::TR_BYTES;
  float* ldsLse = reinterpret_cast<float*>(smem + 2 * C::NAT_BYTES + 2 * C::TR_BYTES);
  float* ldsD = ldsLse + KT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kblk = tile;
  const int h = blockIdx.y, b = blockIdx.z;
  const int d = H * HD;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + (long)b * T * ld + h * HD;
  const bf16_t* kbase = qbase + d;
  const bf16_t* vbase = qbase + 2 * d;
  const bf16_t* dobase = dout + (long)b * T * d + h * HD;
  const float* lse_b = lse + ((long)b * H + h) * T;
  const float* D_b = Dv + ((long)b * H + h) * T;

  const int k0 = kblk * 128, kw0 = k0 + wave * 32;
  const int key = kw0 + (lane & 31);
  const bool kvalid = key < T;
  const float my_kb = (kvalid ? kbias[(long)b * T + key] : 0.f) * LOG2E;
  const float scale2 = scale * LOG2E;

  bf16x8_v kf[C::KS], vf[C::KS];
  row_frags<HD>(kbase + (long)key * ld, kvalid, lane, kf);
  row_frags<HD>(vbase + (long)key * ld, kvalid, lane, vf);

  f32x16 dk[C::IB], dv[C::IB];
#pragma unroll
  for (int i = 0; i < C::IB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  const int nqt = (T + KT - 1) / KT;
  const int qt_causal = k0 / KT;   // first query tile that can see this key block causally

  uint4 rq[C::LPT], rdo[C::LPT];
  float rl = 0.f;
  auto next_tile = [&](int t) {   // next query tile >= t that has to be visited (block-uniform)
    while (t < nqt && t < qt_causal && !qflags[b * nqt + t]) ++t;
    return t;
  };
  auto prefetch = [&](int t) {
    const int q0 = t * KT;
    tile_load<HD>(qbase, ld, q0, T, tid, rq);
    tile_load<HD>(dobase, d, q0, T, tid, rdo);
    if (tid < KT) rl = (q0 + tid < T) ? lse_b[q0 + tid] * LOG2E : 0.f;
    else if (tid < 2 * KT) rl = (q0 + tid - KT < T) ? D_b[q0 + tid - KT] : 0.f;
  };
  int qt = next_tile(0);
  if (qt < nqt) prefetch(qt);

  while (qt < nqt) {
    const int q0 = qt * KT;
    __syncthreads();
    tile_store_nat<HD>(ldsQ, tid, rq);
    tile_store_tr<HD>(ldsQt, tid, rq);
    tile_store_nat<HD>(ldsdO, tid, rdo);
    tile_store_tr<HD>(ldsdOt, tid, rdo);
    if (tid < 2 * KT) ldsLse[tid] = rl;   // ldsD follows ldsLse
    __syncthreads();
    const int qn = next_tile(qt + 1);
    if (qn < nqt) prefetch(qn);

    // S = Q . K^T and dP = dO . V^T : rows = queries, lane column = own key; one 32-query sub-tile at a time
    const int qfl = qflags[b * nqt + qt];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // wave-uniform skip: no query of the sub-tile sees a key of this wave causally and none is a masked row
      if (q0 + t * 32 + 31 < kw0 && !((qfl >> t) & 1)) continue;
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_nat<HD>(ldsQ, t * 32, ks, lane), kf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_nat<HD>(ldsdO, t * 32, ks, lane), vf[ks], dpt, 0, 0, 0);
      }
      const int lim_causal = q0 + t * 32 + 4 * (lane >> 5) - key;    // key <= query  <=>  -c(r) <= lim_causal
      const int lim_len = T - 1 - q0 - t * 32 - 4 * (lane >> 5);     // query < T     <=>   c(r) <= lim_len
      // dropout words: rows of this sub-tile are registers, the 4 lanes of a quad own the 4 keys of one group -> lane
      // (key & 3) = i hashes rows 4j + i and the quad shares the 16 words by DPP (3.25 instead of 11 ops per element)
      uint32_t mine[4] = {0u, 0u, 0u, 0u};
      const int ksh = 8 * (lane & 3);
      if (DROP) {
        const uint32_t T4 = (uint32_t)((T + 3) >> 2);
        const uint32_t gq = ((uint32_t)(b * H + h) * (uint32_t)T + (uint32_t)(q0 + t * 32 + 4 * (lane >> 5) + (lane & 3))) * T4 +
                            (uint32_t)(key >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j] = drop_word(gq + (uint32_t)(8 * j) * T4, drop_key);    // row c = (lane&3) + 8j
      }
      // wave-uniform fast path: every query of the sub-tile sees every key of this wave, no key is padded or invalid
      const bool interior = (q0 + t * 32 >= kw0 + 31) && (q0 + t * 32 + 31 < T) && (kw0 + 31 < T) &&
                            __builtin_amdgcn_ballot_w64(my_kb != 0.f) == 0;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float lse_q = ldsLse[ql], d_q = ldsD[ql];
          const float pv = exp2_fast(fmaf(st[r], scale2, -lse_q));
          float pd = pv, dpe = dpt[r];
          if (DROP) {
            const bool keep = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr;
            pd = keep ? pv : 0.f;
            dpe = keep ? dpe : 0.f;
          }
          st[r] = pd;
          dpt[r] = pv * (dpe - d_q);
        }
      } else
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        const int ql = t * 32 + c + 4 * (lane >> 5);
        const bool causal_ok = (-c) <= lim_causal;
        const float sv = (causal_ok ? st[r] * scale2 : MASK_VAL * LOG2E) + my_kb;
        const float lse_q = ldsLse[ql], d_q = ldsD[ql];      // unconditional: a load inside ?: compiles to a branch
        const float pv = exp2_fast((c <= lim_len && kvalid) ? sv - lse_q : -INFINITY);        // select: 2^-inf = 0
        float pd = pv, dpe = dpt[r];
        if (DROP) {
          const bool keep = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr;   // word of row r
          pd = keep ? pv : 0.f;
          dpe = keep ? dpe : 0.f;
        }
        st[r] = pd;                                              // dropped P (for dV)
        dpt[r] = causal_ok ? pv * (dpe - d_q) : 0.f;             // dS        (for dK)
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const uint4 wp = make_uint4(pack_bf16x2(st[8 * h2 + 0], st[8 * h2 + 1]), pack_bf16x2(st[8 * h2 + 2], st[8 * h2 + 3]),
                                    pack_bf16x2(st[8 * h2 + 4], st[8 * h2 + 5]), pack_bf16x2(st[8 * h2 + 6], st[8 * h2 + 7]));
        const uint4 wd = make_uint4(pack_bf16x2(dpt[8 * h2 + 0], dpt[8 * h2 + 1]), pack_bf16x2(dpt[8 * h2 + 2], dpt[8 * h2 + 3]),
                                    pack_bf16x2(dpt[8 * h2 + 4], dpt[8 * h2 + 5]), pack_bf16x2(dpt[8 * h2 + 6], dpt[8 * h2 + 7]));
        const bf16x8_v pf = __builtin_bit_cast(bf16x8_v, wp);
        const bf16x8_v df = __builtin_bit_cast(bf16x8_v, wd);
#pragma unroll
        for (int i = 0; i < C::IB; ++i) {
          dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(ldsdOt, i * 32, 2 * t + h2, lane), pf, dv[i], 0, 0, 0);
          dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(ldsQt, i * 32, 2 * t + h2, lane), df, dk[i], 0, 0, 0);
        }
      }
    }
    qt = qn;
  }

  // P and dP were masked but not scaled in the loop (ldsD holds D/s): the survivor scale s is applied once, here
  const float vsc = DROP ? drop_scale : 1.0f, ksc = scale * vsc;
  if (kvalid) {
    bf16_t* krow = dqkv + ((long)b * T + key) * ld + d + h * HD;
    bf16_t* vrow = krow + d;
#pragma unroll
    for (int i = 0; i < C::IB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dk[i][4 * g + 0] * ksc, dk[i][4 * g + 1] * ksc);
        pk.y = pack_bf16x2(dk[i][4 * g + 2] * ksc, dk[i][4 * g + 3] * ksc);
        *reinterpret_cast<uint2*>(krow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
        pk.x = pack_bf16x2(dv[i][4 * g + 0] * vsc, dv[i][4 * g + 1] * vsc);
        pk.y = pack_bf16x2(dv[i][4 * g + 2] * vsc, dv[i][4 * g + 3] * vsc);
        *reinterpret_cast<uint2*>(vrow + i * 32 + 8 * g + 4 * (lane >> 5)) = pk;
      }
  }
}

template <int HD, bool DROP>
__global__ __launch_bounds__(NT, (HD <= 32 ? NEKO_DKV_WAVES : (HD <= 64 ? 2 : 1))) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ kbias, const float* __restrict__ lse,
                                                          const float* __restrict__ Dv, const int* __restrict__ qflags,
                                                          bf16_t* __restrict__ dqkv, int B, int T, int H, float scale,
                                                          uint32_t drop_thr, uint32_t drop_key, float drop_scale) {
  // one workgroup = a heavy and a light causal tile (G-1-p and p): every workgroup walks the same number of
  // 64-row tiles.  With one tile per workgroup the whole grid was co-resident and the kernel lasted as long as its
  // heaviest workgroup (16 tile iterations at T = 1024 against an average of 9).
  if (DROP) drop_key += neko_drop_salt();
  const int G = (T + 127) / 128, p = rotated_tile();
  const int first = p, second = G - 1 - p;
  attn_bwd_dkv_tile<HD, DROP>(qkv, dout, kbias, lse, Dv, qflags, dqkv, B, T, H, scale, drop_thr, drop_key, drop_scale, first);
  if (second != first) {
    __syncthreads();
    attn_bwd_dkv_tile<HD, DROP>(qkv, dout, kbias, lse, Dv, qflags, dqkv, B, T, H, scale, drop_thr, drop_key, drop_scale, second);
  }
}

template <int HD>
int fwd_launch(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T, int H,
               int thr, unsigned key, float dscale, hipStream_t s) {
  const float scale = 1.0f / sqrtf((float)HD);
  dim3 grid(((T + 127) / 128 + 1) / 2, H, B);      // tile pairs, see the kernels
  if (thr)
    hipLaunchKernelGGL((attn_fwd_kernel<HD, true>), grid, dim3(NT), 0, s, qkv, kbias, kstart, out, lse, B, T, H, scale,
                       (uint32_t)thr, key, dscale);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<HD, false>), grid, dim3(NT), 0, s, qkv, kbias, kstart, out, lse, B, T, H, scale,
                       0u, key, dscale);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
template <int HD>
int bwd_launch(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* kbias, const int* kstart,
               const float* lse, float* D, int* qflags, bf16_t* dqkv, int B, int T, int H, int thr, unsigned key,
               float dscale, const uint32_t* dmask, hipStream_t s) {
  const float scale = 1.0f / sqrtf((float)HD);
  if (neko_attn_path_mode() != 1 && neko_attn_res_applicable(T, HD))
    return neko_attn_bwd_res_impl(qkv, out, dout, kbias, kstart, lse, D, dqkv, B, T, H, thr, key, dscale, dmask, s);
  if (neko_attn_path_mode() != 1 && neko_attn_stream_applicable(T, HD))      // hd = 64 / 128: DMA-ring kernels (attention_stream.hip)
    return neko_attn_bwd_stream_impl(qkv, out, dout, kbias, kstart, lse, D, dqkv, B, T, H, HD, thr, key, dscale, s);
  const long total = (long)B * T * H;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, dout, kbias, D,
                     qflags, B, T, H, HD, thr ? 1.0f / dscale : 1.0f);
  NEKO_CHECK_LAUNCH();
  dim3 grid(((T + 127) / 128 + 1) / 2, H, B);      // tile pairs, see the kernels
  if (thr) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, true>), grid, dim3(NT), 0, s, qkv, dout, kbias, kstart, lse, D, dqkv, B, T,
                       H, scale, (uint32_t)thr, key, dscale);
    NEKO_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, true>), grid, dim3(NT), 0, s, qkv, dout, kbias, lse, D, qflags, dqkv, B, T,
                       H, scale, (uint32_t)thr, key, dscale);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, false>), grid, dim3(NT), 0, s, qkv, dout, kbias, kstart, lse, D, dqkv, B, T,
                       H, scale, 0u, key, dscale);
    NEKO_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, false>), grid, dim3(NT), 0, s, qkv, dout, kbias, lse, D, qflags, dqkv, B, T,
                       H, scale, 0u, key, dscale);
  }
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

}  // namespace

// schedule selection: 0 = automatic (head-resident kernels of attention_res.hip when they apply; their backward in one pass for
// T > 256, as two kernels below: attention_res.hip), 2 = head-resident with the two-kernel backward at every length (bit-reproducible:
// no order-dependent dQ sums), 3 = head-resident with the one-pass backward at every length (tests, A/B runs), 1 = always the
// streaming kernels of this file.  A tuning / test knob, not part of the numerics: all schedules compute the same sums.
static int g_attn_path = [] {              // NEKO_ATTN_PATH=0..3 sets the initial mode (A/B runs of whole steps)
  const char* e = getenv("NEKO_ATTN_PATH");
  const int v = e ? atoi(e) : 0;
  return v >= 0 && v <= 3 ? v : 0;
}();
int neko_attn_path_mode() { return g_attn_path; }
// Calling thread's request for the bit-reproducible (two-kernel) head-resident backward at every length: thread-local, so a caller that
// wants reproducible gradients never changes the schedule another thread's calls see (the process-wide knob was being
// flipped around every backward call)
static thread_local int t_bwd_reproducible = 0;
int neko_attn_bwd_reproducible_mode() { return t_bwd_reproducible; }
int neko_attn_bwd_reproducible_impl(int on) {
  const int prev = t_bwd_reproducible;
  if (on == 0 || on == 1) t_bwd_reproducible = on;
  return prev;
}
int neko_attn_set_path_impl(int mode) {
  const int prev = g_attn_path;
  if (mode >= 0 && mode <= 3) g_attn_path = mode;
  return prev;
}

// dwords of the dropout keep-mask buffer neko_attn_fwd fills for neko_attn_bwd (0: the schedule in use re-hashes instead)
long neko_attn_mask_dwords_impl(int B, int T, int H, int hd) {
  return g_attn_path != 1 ? neko_attn_res_mask_dwords(B, T, H, hd) : 0;
}

int neko_attn_fwd_impl(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T,
                       int H, int hd, int drop_thr, unsigned drop_key, float drop_scale, uint32_t* dmask, hipStream_t s) {
  if (B <= 0 || T <= 0) return NEKO_OK;
  if (!qkv || !kbias || !out || !lse || H <= 0 || drop_thr < 0 || drop_thr > 255) return NEKO_ERR_ARG;
  if (g_attn_path != 1 && neko_attn_res_applicable(T, hd))
    return neko_attn_fwd_res_impl(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, dmask, s);
  if (g_attn_path != 1 && neko_attn_stream_applicable(T, hd))       // hd = 64 / 128: DMA-ring kernels (attention_stream.hip)
    return neko_attn_fwd_stream_impl(qkv, kbias, kstart, out, lse, B, T, H, hd, drop_thr, drop_key, drop_scale, s);
  switch (hd) {
    case 32: return fwd_launch<32>(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, s);
    case 64: return fwd_launch<64>(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, s);
    case 128: return fwd_launch<128>(qkv, kbias, kstart, out, lse, B, T, H, drop_thr, drop_key, drop_scale, s);
    default: return NEKO_ERR_UNSUPPORTED;
  }
}

// workspace: D fp32 [B*H*T] and qflags int32 [B*ceil(T/64)]
int neko_attn_bwd_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* kbias, const int* kstart,
                       const float* lse, float* D, int* qflags, bf16_t* dqkv, int B, int T, int H, int hd,
                       int drop_thr, unsigned drop_key, float drop_scale, const uint32_t* dmask, hipStream_t s) {
  if (B <= 0 || T <= 0) return NEKO_OK;
  if (!qkv || !out || !dout || !kbias || !lse || !D || !qflags || !dqkv || H <= 0) return NEKO_ERR_ARG;
  switch (hd) {
    case 32: return bwd_launch<32>(qkv, out, dout, kbias, kstart, lse, D, qflags, dqkv, B, T, H, drop_thr, drop_key,
                                   drop_scale, dmask, s);
    case 64: return bwd_launch<64>(qkv, out, dout, kbias, kstart, lse, D, qflags, dqkv, B, T, H, drop_thr, drop_key,
                                   drop_scale, dmask, s);
    case 128: return bwd_launch<128>(qkv, out, dout, kbias, kstart, lse, D, qflags, dqkv, B, T, H, drop_thr, drop_key,
                                   drop_scale, dmask, s);
    default: return NEKO_ERR_UNSUPPORTED;
  }
}

NEKO_DEFINE_SALT_SETTER(attention)
