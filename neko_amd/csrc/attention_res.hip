// Head-resident attention for hd = 32, T <= 1024: one workgroup owns one (batch, head) and keeps the head's whole
// K and V (forward, dQ) or Q and dO (dK/dV) in LDS -- 2 x T x 64 B = 128 KB at T = 1024 of the 160 KB a gfx950 CU has.
// Same arithmetic, masks (reference-literal finite -1e4 semantics, trajectory_gpt2.py:163-188,663-679), dropout
// indexing and output layout as the streaming kernels in attention.hip; what changes is the schedule:
//
//   * the streaming kernels restage a 64-key tile per iteration behind two block barriers plus a block-wide OR
//     (three more barriers); at hd = 32 a wave has only ~250 VALU + 8 MFMA instructions of work per tile, so the
//     barriers and the 4.5x re-read of every K/V tile by the query tiles above it dominated (wave-parked 34 % +
//     issue-stalled 20 % of wave cycles in the first PMC pass).  Here there is ONE barrier per workgroup: after the
//     images are staged every wave runs its own key loop with no further synchronisation.
//   * causal balance: the 32-row blocks of the head are handed out heaviest first from an LDS counter (longest-
//     processing-time-first), so any wave count balances: 16 waves per workgroup (4 per SIMD, <= 128 VGPRs) in the
//     forward, 12 (3 per SIMD, <= 168 VGPRs: the backward bodies spill at 128) in dQ and dK/dV.
//   * the images are natural [row][32] bf16 with 64-byte rows and the 16-byte piece index XOR (row>>2)&3: conflict-
//     free for both the ds_read_b128 row fragments (S = K.Q^T operands) and the ds_read_b64_tr_b16 column fragments
//     (the K^T / V^T / Q^T / dO^T operands), so no transposed copy is built.
//   * workgroup ids are paired so that heads 2j and 2j+1 of one batch row (the two halves of every 128-byte line of
//     the [B*T, 3d] qkv matrix) run on the same XCD.
#include <atomic>
#include "neko_kernels.h"

extern int neko_attn_path_mode();
extern int neko_attn_bwd_reproducible_mode();

namespace {

constexpr float MASK_VAL = -10000.0f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
#ifdef NEKO_ATTN_TRACE
// phase trace of the dK/dV kernel (diagnostic builds, tools/attn_trace.py): per workgroup and wave 4 x 8 bytes =
// s_memrealtime (100 MHz) at kernel entry, behind the staging barrier, after the wave's last work item; sub-tiles done
__device__ unsigned long long* g_neko_attn_trace = nullptr;
extern "C" int neko_attn_diag_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_neko_attn_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define NEKO_ATRACE(slot, val)                                                                                     \
  do {                                                                                                              \
    if (g_neko_attn_trace && (threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                          \
      g_neko_attn_trace[((long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 4 + (slot)] = (val);                         \
  } while (0)
#else
#define NEKO_ATRACE(slot, val) do { } while (0)
#endif
#ifndef NEKO_DKV_WAVES
#define NEKO_DKV_WAVES 12    // waves per dK/dV workgroup; 16 (4 per SIMD, 128 VGPRs: 60 spilled) measured 347 -> 401 us for the backward
#endif
#ifndef NEKO_ATTN_FUSED_ABL
#define NEKO_ATTN_FUSED_ABL 0   // one-pass backward ablations (wrong results): 1 read-add-write without the lock, 2 no dQ accumulation
#endif
#ifndef NEKO_ATTN_ABL
#define NEKO_ATTN_ABL 0      // dK/dV ablations for tools/attn_bench.py (wrong results): 1 no elementwise math in interior
#endif                       // sub-tiles, 2 operand fragments of one fixed query block (the LDS reads leave the loop), 4 no mask loads
#define NEKO_ATTN_Q0(q0) ((NEKO_ATTN_ABL & 2) ? 0 : (q0))
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
// Two fp32 values per issue slot (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, neko_common.h f32x2_v): these kernels are bound
// by the VALU issue rate, and everything around the exponential that is plain arithmetic on an accumulator pair packs.
// bit `bit` of w sign-extended to a dword (all ones / zero) in ONE instruction; written as asm because the compiler turns
// `x & sbfe(w, bit, 1)` into v_and + v_cmp + v_cndmask
template <typename T = void>
__device__ __forceinline__ uint32_t keep_bits(uint32_t w, int bit) {
  uint32_t r;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "n"(bit));
  return r;
}
__device__ __forceinline__ f32x2_v pk2(float a, float b) { return (f32x2_v){a, b}; }
__device__ __forceinline__ f32x2_v exp2_fast2(f32x2_v x) { return (f32x2_v){__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

// image: [rows][32 bf16], 64-byte rows; 16-byte piece p of row r is stored at piece p ^ ((r>>2)&3)
__device__ __forceinline__ int img_off(int row, int piece) { return row * 64 + ((piece ^ ((row >> 2) & 3)) << 4); }

// MFMA operand with the image ROWS as its 32 rows/columns: row (rowbase + lane%32), head-dim slots ks*16 + 8*(lane/32) + 0..7
__device__ __forceinline__ bf16x8_v frag_rows(const char* img, int rowbase, int ks, int lane) {
  const uint4 v = *reinterpret_cast<const uint4*>(img + img_off(rowbase + (lane & 31), ks * 2 + (lane >> 5)));
  return __builtin_bit_cast(bf16x8_v, v);
}
// MFMA operand with the image COLUMNS (head dim) as its 32 rows: contraction slot j = image row
// rowbase + 16*s + 8*(j>>2) + 4*(lane/32) + (j&3) -- the order a 32x32 accumulator leaves in a lane (frag_from_acc).
// rowbase must be a multiple of 32.  Each 16-lane group reads a [4 rows][16 columns] block transposed.
__device__ __forceinline__ bf16x8_v frag_cols(const char* img, int rowbase, int s, int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  const int col = 16 * (g & 1) + 4 * (c16 & 3);
  const int r_lo = 16 * s + 4 * (g >> 1) + (c16 >> 2), r_hi = r_lo + 8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const char* base = img + rowbase * 64 + ((col & 7) << 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + r_lo * 64 + (((col >> 3) ^ ((r_lo >> 2) & 3)) << 4)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + r_hi * 64 + (((col >> 3) ^ ((r_hi >> 2) & 3)) << 4)));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_v, r);
}
// registers 8*s2 .. 8*s2+7 of a 32x32 accumulator as a bf16 operand (s2 = 0, 1)
__device__ __forceinline__ bf16x8_v frag_from_acc(const f32x16& a, int s2) {
  const int o = 8 * s2;
  const uint4 r = make_uint4(pack_bf16x2(a[o + 0], a[o + 1]), pack_bf16x2(a[o + 2], a[o + 3]),
                             pack_bf16x2(a[o + 4], a[o + 5]), pack_bf16x2(a[o + 6], a[o + 7]));
  return __builtin_bit_cast(bf16x8_v, r);
}
// (requesting the rows of a wave's NEXT work item while it works on the current one -- the queue pop moved one item ahead,
// a second register set -- was measured and not kept: forward 131 -> 140 us, backward 354 -> 368 us at B = 32)
// own-row operand straight from HBM (row pointer, head-dim slots ks*16 + 8*(lane/32)..+7)
__device__ __forceinline__ void row_frags(const bf16_t* __restrict__ rowptr, bool valid, int lane, bf16x8_v (&f)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid) v = *reinterpret_cast<const uint4*>(rowptr + ks * 16 + (lane >> 5) * 8);
    f[ks] = __builtin_bit_cast(bf16x8_v, v);
  }
}
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x, int i) {   // value of lane (lane & ~3) + i, i literal 0..3
  switch (i) {
    case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x00, 0xf, 0xf, true);
    case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x55, 0xf, 0xf, true);
    case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xAA, 0xf, 0xf, true);
    default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xFF, 0xf, 0xf, true);
  }
}
// ---- dropout keep decisions computed ONCE (forward) and reused by both backward kernels ------------------------------
// The forward kernel's compare "hash byte >= threshold" for accumulator register r lands in an SGPR pair: a 64-bit lane
// mask whose low word is {keep(query q0 + l, key k0 + c(r))}_l and whose high word is the same for key k0 + c(r) + 4
// (c(r) = (r & 3) + 8 (r >> 2)).  Those 16 pairs = 32 dwords per 32 x 32 sub-tile are written with scalar stores
// (s_store_dwordx4: wave-uniform data never touches a VGPR) to   mask[(b*H + h)][query block][key block][32]   and
//   * dQ (same lane = query geometry) reads them back with scalar loads and applies each as the SGPR-pair operand of ONE
//     v_cndmask_b32 -- instead of 10 VALU per hash word + compare + select per element;
//   * dK/dV (lane = key) loads, per lane, the dword of ITS key (bits = the 32 queries of the block) and tests bit
//     c(r) + 4 (lane / 32) with one v_bfe_i32 per register.
// tools/probe/sstore_probe.hip checks the scalar-store / scalar-load round trip across kernels on gfx950.
typedef uint32_t u32x4_s __attribute__((ext_vector_type(4)));
// loads through the constant address space with a wave-uniform address are always scalar loads (s_load_dwordx*); through a
// plain global pointer hipcc only selects them when it can prove that no store of the kernel may alias
typedef __attribute__((address_space(4))) const unsigned long long const_u64;
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  return reinterpret_cast<const void*>(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                                       (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a));
}
// two lane masks -> 16 bytes at base + OFF (base wave-uniform, in SGPRs)
template <int OFF>
__device__ __forceinline__ void sstore_masks(uint32_t* base, unsigned long long m0, unsigned long long m1) {
  const u32x4_s q = {(uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32)};
  asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(q), "s"(base), "n"(OFF) : "memory");
}
// position of key `kl` (0..31) of a sub-tile inside its 32 dwords: pair r = (kl & 3) + 4 (kl >> 3), half (kl >> 2) & 1
__device__ __forceinline__ int mask_slot_of_key(int kl) { return 2 * ((kl & 3) + 4 * (kl >> 3)) + ((kl >> 2) & 1); }
// x where the lane's bit of the 64-bit lane mask is set, else 0: ONE v_cndmask with the mask as its SGPR-pair operand
__device__ __forceinline__ float keep_lanes(float x, unsigned long long lane_mask) {
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(lane_mask));
  return r;
}
// workgroup id -> (batch*H + head): ids n and n+8 run on the same XCD and get heads 2p and 2p+1
__device__ __forceinline__ int pair_remap(int n, int total) {
  const int full = total & ~15;
  if (n >= full) return n;
  return 2 * ((n >> 4) * 8 + (n & 7)) + ((n >> 3) & 1);
}

// next work item of the workgroup's queue (wave-uniform); the counter lives in LDS and is zeroed before the staging barrier
__device__ __forceinline__ int next_item(int* counter, int lane) {
  int v = 0;
  if (lane == 0) v = atomicAdd(counter, 1);
  return __builtin_amdgcn_readfirstlane(v);
}

// item -> 32-row block for the kernels whose lanes own queries: blocks in `heavy` (they hold a masked query row and
// visit every key) first, then the remaining blocks latest (longest causal range) first.  Wave-uniform scalar work.
__device__ __forceinline__ int block_of_item(int item, uint32_t heavy, int nblk) {
  const int nh = __popc(heavy);
  if (item < nh) {
    uint32_t m = heavy;
    for (int i = 0; i < item; ++i) m &= m - 1;
    return __ffs(m) - 1;
  }
  uint32_t m = ~heavy & (nblk >= 32 ? 0xffffffffu : ((1u << nblk) - 1u));
  for (int i = 0; i < item - nh; ++i) m &= ~(1u << (31 - __clz(m)));
  return 31 - __clz(m);
}
__device__ __forceinline__ bool frag_nonzero(const bf16x8_v& f) {
  const uint4 u = __builtin_bit_cast(uint4, f);
  return (u.x | u.y | u.z | u.w) != 0u;
}

// sum over the 8 bf16 pairs of two 16-byte pieces
__device__ __forceinline__ float dot8_bf16(const uint4& a, const uint4& b) {
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc = fmaf(__uint_as_float(aw[e] << 16), __uint_as_float(bw[e] << 16), acc);
    acc = fmaf(__uint_as_float(aw[e] & 0xffff0000u), __uint_as_float(bw[e] & 0xffff0000u), acc);
  }
  return acc;
}

// stage two [T][32] bf16 matrices (row strides lda / ldb elements) into swizzled images, rows >= T zero
__device__ __forceinline__ void stage_pair(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b, long ldb,
                                           char* imgA, char* imgB, int T, int Tp, int tid, int nthr) {
  // every load of a round is in flight before the first LDS store waits for one: a round is one HBM round trip, and with
  // one workgroup per CU nothing overlaps it -- 6 pieces per thread make T = 1024 a single round at 768 threads
  const int total = Tp * 4;
  constexpr int NB = 6;
  for (int c0 = 0; c0 < total; c0 += nthr * NB) {
    uint4 ra[NB], rb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = c0 + i * nthr + tid, row = c >> 2, p = c & 3;
      ra[i] = make_uint4(0, 0, 0, 0);
      rb[i] = make_uint4(0, 0, 0, 0);
      if (c < total && row < T) {
        ra[i] = *reinterpret_cast<const uint4*>(a + (long)row * lda + p * 8);
        rb[i] = *reinterpret_cast<const uint4*>(b + (long)row * ldb + p * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = c0 + i * nthr + tid, row = c >> 2, p = c & 3;
      if (c < total) {
        *reinterpret_cast<uint4*>(imgA + img_off(row, p)) = ra[i];
        *reinterpret_cast<uint4*>(imgB + img_off(row, p)) = rb[i];
      }
    }
  }
}
// the two halves of pad_mask_of below: the loads (issued with the rest of a prologue's loads) and the wave ballots
__device__ __forceinline__ void pad_mask_load(const float* __restrict__ kb, int T, int nblk, int lane, float (&v)[16]) {
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) v[jj] = kb[min(jj * 64 + lane, T - 1)];        // clamped, unconditional: no branch, no wait
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) v[jj] = (2 * jj < nblk && jj * 64 + lane < T) ? v[jj] : 0.f;
}
__device__ __forceinline__ uint32_t pad_mask_ballot(const float (&v)[16]) {
  uint32_t m = 0;
#pragma unroll
  for (int jj = 0; jj < 16; ++jj) {
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(v[jj] != 0.f);
    if ((uint32_t)bal) m |= 1u << (2 * jj);
    if ((uint32_t)(bal >> 32)) m |= 2u << (2 * jj);
  }
  return m;
}
// bit j: 32-position block j holds a position with a non-zero key bias (a padded key == a masked query row)
__device__ __forceinline__ uint32_t pad_mask_of(const float* __restrict__ kb, int T, int nblk, int lane) {
  // nblk <= 32 here (T <= 1024): the 16 loads are issued together -- as a loop of load / ballot / load they were 16
  // dependent memory round trips at the head of every workgroup, with nothing else resident on the CU to hide them
  float v[16];
  pad_mask_load(kb, T, nblk, lane, v);
  return pad_mask_ballot(v);
}

// Where a (sequence, head) lives.  Uniform batches: sequence b holds rows [b T, (b+1) T).  Packed ("varlen") batches
// (gato_policy.py:408-416 is the left-pad this removes): seq_off[b] .. seq_off[b+1] are the rows of sequence b,
// every sequence with its own length, ONE launch for all of them (length buckets needed one launch per bucket).  Per-(b, h)
// arrays (lse, D) are laid out [sequence][head][position] = row0 * H + h * T + q in both cases; the keep masks of sequence b
// start at mask_off[b] dwords; the dropout hash walks unique row ids with the row stride of the LONGEST sequence.
struct SeqGeom {
  int T;             // length of this sequence
  long row0;         // its first row in the [rows, ...] matrices
  long hrow;         // index of (b, h, position 0) in lse / D and unique row id of the dropout hash
  long mask0;        // dword offset of (b, h) in the keep-mask buffer
  uint32_t T4;       // dropout hash: words per row
};
__device__ __forceinline__ SeqGeom seq_geom(int b, int h, int H, int T_uniform, const int* __restrict__ seq_off,
                                            const long long* __restrict__ mask_off, int T4_varlen) {
  SeqGeom g;
  if (seq_off) {
    g.row0 = seq_off[b];
    g.T = seq_off[b + 1] - (int)g.row0;
    g.T4 = (uint32_t)T4_varlen;
  } else {
    g.row0 = (long)b * T_uniform;
    g.T = T_uniform;
    g.T4 = (uint32_t)((T_uniform + 3) >> 2);
  }
  g.hrow = g.row0 * H + (long)h * g.T;
  const long nblk = (g.T + 31) >> 5;
  g.mask0 = (mask_off ? (long)mask_off[b] : (long)b * H * nblk * nblk * 32) + (long)h * nblk * nblk * 32;
  return g;
}

// =====================================================================================================
// forward
// =====================================================================================================
template <bool DROP, bool MASK>
__global__ __launch_bounds__(1024) void attn_fwd_res_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ kbias,
                                                            const int* __restrict__ kstart, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int B, int T_uniform, int H, float scale,
                                                            uint32_t drop_thr, uint32_t drop_key, float drop_scale,
                                                            uint32_t* __restrict__ dmask, const int* __restrict__ seq_off,
                                                            const long long* __restrict__ mask_off, int T4_varlen) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (DROP) drop_key += neko_drop_salt();
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int hb = pair_remap(blockIdx.x, B * H);
  const int b = hb / H, h = hb % H;
  const SeqGeom sg = seq_geom(b, h, H, T_uniform, seq_off, mask_off, T4_varlen);
  const int T = sg.T;
  const int Tp = (T + 31) & ~31, nblk = Tp >> 5;
  char* imgK = smem;
  char* imgV = smem + Tp * 64;
  float* ldsKb = reinterpret_cast<float*>(smem + Tp * 128);
  int* queue = reinterpret_cast<int*>(ldsKb + Tp);
  if (tid == 0) *queue = 0;
  const int d = H * 32;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + sg.row0 * ld + h * 32;
  const float* kb = kbias + sg.row0;

  stage_pair(qbase + d, ld, qbase + 2 * d, ld, imgK, imgV, T, Tp, tid, nthr);
  for (int i = tid; i < Tp; i += nthr) ldsKb[i] = (i < T) ? kb[i] * LOG2E : 0.f;
  const int kb_first = kstart ? (kstart[b] >> 5) : 0;
  __syncthreads();
  const uint32_t padmask = pad_mask_of(ldsKb, Tp, nblk, lane);     // from the LDS copy (zero beyond T): no second trip to memory

  const float scale2 = scale * LOG2E;
#pragma unroll 1
  for (int item = next_item(queue, lane); item < nblk; item = next_item(queue, lane)) {
    const int qb = block_of_item(item, padmask, nblk);   // heaviest first
    const int q = qb * 32 + (lane & 31);
    const bool qvalid = q < T;
    bf16x8_v qf[2];
    row_frags(qbase + (long)q * ld, qvalid, lane, qf);
    // a wave that holds a masked (padded) query row visits every key: the reference's finite masks let such a row see
    // them; for every other row the keys beyond its diagonal contribute exp(-1e4 - m) == 0 in fp32
    const bool wave_full = (padmask >> qb) & 1;
    const int kb_beg = wave_full ? 0 : min(kb_first, qb);
    const int kb_end = wave_full ? nblk : qb + 1;

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

#pragma unroll 1
    for (int kbk = kb_beg; kbk < kb_end; ++kbk) {
      const int k0 = kbk * 32;
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(imgK, k0, ks, lane), qf[ks], st, 0, 0, 0);
      const bool interior = (kbk < qb) && !((padmask >> kbk) & 1) && (k0 + 32 <= T);
      if (!interior) {
        const int lim_causal = q - k0 - 4 * (lane >> 5);        // key <= q  <=>  c(r) <= lim_causal
        const int lim_len = T - 1 - k0 - 4 * (lane >> 5);       // key <  T  <=>  c(r) <= lim_len
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          float v = (c <= lim_causal) ? st[r] * scale2 : MASK_VAL * LOG2E;
          v += ldsKb[k0 + c + 4 * (lane >> 5)];
          st[r] = (c <= lim_len) ? v : -INFINITY;
        }
      }
      const float sc = interior ? scale2 : 1.0f;      // interior scores are still unscaled
      // Online softmax WITHOUT a row maximum per sub-tile: the exponentials are taken against the running reference
      // m_run first, and only when some lane's partial sum shows that a score climbed more than ~2^8 above it (or m_run
      // is still -inf: first sub-tile of the row, the sum is inf / NaN) the wave computes the row maxima, moves m_run,
      // rescales and redoes the exponentials.  Softmax is shift invariant, so any reference that keeps the terms inside
      // the fp32 range gives the same result; the 16-way max chain, the lane^32 exchange and the compare that used to
      // run for EVERY sub-tile (~15 VALU + one LDS round trip of ~125) now run for the first sub-tile of a row and
      // after rare jumps.  The two halves of a row (lanes l, l^32) must share m_run: the decision is a wave ballot.
      f32x16 pr;
      float ps0 = 0.f, ps1 = 0.f;
      bool renorm = kbk == kb_beg;                     // first sub-tile of the rows: m_run is still -inf
      if (!renorm) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_v pv = exp2_fast2(__builtin_elementwise_fma(pk2(st[r], st[r + 1]), (f32x2_v)(sc), (f32x2_v)(-m_run)));
          pr[r] = pv.x;
          pr[r + 1] = pv.y;
          ps0 += pv.x;
          ps1 += pv.y;
        }
        renorm = __builtin_amdgcn_ballot_w64(!((ps0 + ps1) < 256.0f)) != 0;
      }
      if (renorm) {
        float mx = fmaxf(st[0], st[1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) mx = fmaxf(mx, st[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * sc);
        const float alpha = exp2_fast(m_run - m_new);   // 2^(-inf) = 0 on the first tile
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        m_run = m_new;
        ps0 = 0.f;
        ps1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_v pv = exp2_fast2(__builtin_elementwise_fma(pk2(st[r], st[r + 1]), (f32x2_v)(sc), (f32x2_v)(-m_run)));
          pr[r] = pv.x;
          pr[r + 1] = pv.y;
          ps0 += pv.x;
          ps1 += pv.y;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = pr[r];
      l_run += ps0 + ps1;
      if (DROP) {   // attn_dropout on the probabilities (trajectory_gpt2.py:179): the normaliser stays undropped
        const uint32_t g0 = ((uint32_t)sg.hrow + (uint32_t)q) * sg.T4 + (uint32_t)((k0 + 4 * (lane >> 5)) >> 2);
        unsigned long long km[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // registers 4j..4j+3 = keys +8j .. +8j+3: one word
          const uint32_t w = drop_word(g0 + 2 * j, drop_key);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool keep = drop_byte_keep(w, e, drop_thr);
            if (MASK) km[4 * j + e] = __builtin_amdgcn_ballot_w64(keep);     // the compare's own SGPR pair
            st[4 * j + e] = keep ? st[4 * j + e] : 0.f;
          }
        }
        if (MASK) {     // the sub-tile's 16 lane masks -> mask[(b*H+h)][qb][kbk][32 dwords], 8 scalar stores
          uint32_t* mp = const_cast<uint32_t*>(static_cast<const uint32_t*>(
              uniform_ptr(dmask + sg.mask0 + ((long)qb * nblk + kbk) * 32)));
          sstore_masks<0>(mp, km[0], km[1]);
          sstore_masks<16>(mp, km[2], km[3]);
          sstore_masks<32>(mp, km[4], km[5]);
          sstore_masks<48>(mp, km[6], km[7]);
          sstore_masks<64>(mp, km[8], km[9]);
          sstore_masks<80>(mp, km[10], km[11]);
          sstore_masks<96>(mp, km[12], km[13]);
          sstore_masks<112>(mp, km[14], km[15]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgV, k0, s2, lane), frag_from_acc(st, s2), o, 0, 0, 0);
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qvalid) {
      const float inv = (DROP ? drop_scale : 1.0f) / l_tot;
      bf16_t* orow = out + (sg.row0 + q) * d + h * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(o[4 * g + 0] * inv, o[4 * g + 1] * inv);
        pk.y = pack_bf16x2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + 8 * g + 4 * (lane >> 5)) = pk;
      }
      if (lane < 32) lse[sg.hrow + q] = m_run * LN2 + __logf(l_tot);
    }
  }
  if (DROP && MASK)   // scalar stores sit in the scalar data cache until written back
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// =====================================================================================================
// backward dQ: lanes own queries (same geometry as forward); images K and V
// =====================================================================================================
template <bool DROP, bool MASK>
__global__ __launch_bounds__(768) void attn_dq_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ kbias, const int* __restrict__ kstart,
                                                           const float* __restrict__ lse, const bf16_t* __restrict__ outp,
                                                           float* __restrict__ Dout, bf16_t* __restrict__ dqkv, int B, int T_uniform,
                                                           int H, float scale, uint32_t drop_thr, uint32_t drop_key,
                                                           float drop_scale, const uint32_t* __restrict__ dmask,
                                                           const int* __restrict__ seq_off, const long long* __restrict__ mask_off,
                                                           int T4_varlen) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (DROP) drop_key += neko_drop_salt();
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int hb = pair_remap(blockIdx.x, B * H);
  const int b = hb / H, h = hb % H;
  const SeqGeom sg = seq_geom(b, h, H, T_uniform, seq_off, mask_off, T4_varlen);
  const int T = sg.T;
  const int Tp = (T + 31) & ~31, nblk = Tp >> 5;
  char* imgK = smem;
  char* imgV = smem + Tp * 64;
  float* ldsKb = reinterpret_cast<float*>(smem + Tp * 128);
  int* queue = reinterpret_cast<int*>(ldsKb + Tp);
  if (tid == 0) *queue = 0;
  const int d = H * 32;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + sg.row0 * ld + h * 32;
  const float* kb = kbias + sg.row0;

  stage_pair(qbase + d, ld, qbase + 2 * d, ld, imgK, imgV, T, Tp, tid, nthr);
  for (int i = tid; i < Tp; i += nthr) ldsKb[i] = (i < T) ? kb[i] * LOG2E : 0.f;
  const int kb_first = kstart ? (kstart[b] >> 5) : 0;
  __syncthreads();
  const uint32_t padmask = pad_mask_of(ldsKb, Tp, nblk, lane);     // from the LDS copy (zero beyond T): no second trip to memory

  const float scale2 = scale * LOG2E;
#pragma unroll 1
  for (int item = next_item(queue, lane); item < nblk; item = next_item(queue, lane)) {
    const int qb = block_of_item(item, padmask, nblk);   // heaviest first
    const int q = qb * 32 + (lane & 31);
    const bool qvalid = q < T;
    bf16x8_v qf[2], dof[2];
    row_frags(qbase + (long)q * ld, qvalid, lane, qf);
    row_frags(dout + (sg.row0 + q) * d + h * 32, qvalid, lane, dof);
    const float my_lse = (qvalid ? lse[sg.hrow + q] : 0.f) * LOG2E;
    // D = sum_hd dO.O of the own row (divided by the dropout survivor scale, which is folded out of dS): the lane pair
    // (l, l^32) holds the two halves of the row -- the separate D pass of the streaming kernels is not needed here
    bf16x8_v of[2];
    row_frags(outp + (sg.row0 + q) * d + h * 32, qvalid, lane, of);
    float my_D = dot8_bf16(__builtin_bit_cast(uint4, dof[0]), __builtin_bit_cast(uint4, of[0])) +
                 dot8_bf16(__builtin_bit_cast(uint4, dof[1]), __builtin_bit_cast(uint4, of[1]));
    my_D += __shfl_xor(my_D, 32, 64);
    if (DROP) my_D *= 1.0f / drop_scale;
    if (qvalid && lane < 32) Dout[sg.hrow + q] = my_D;      // the dK/dV kernel (launched after this one) stages it
    // a masked query row whose dO is exactly zero (the training case: no loss reaches a padded position) has dP = D = 0,
    // hence dS = 0 for every key: the keys beyond the diagonal are then needed by no row of the block
    const bool live_masked = __builtin_amdgcn_ballot_w64(qvalid && ldsKb[q] != 0.f && (frag_nonzero(dof[0]) || frag_nonzero(dof[1]))) != 0;
    const bool wave_full = ((padmask >> qb) & 1) && live_masked;
    const int kb_beg = wave_full ? 0 : min(kb_first, qb);
    const int kb_end = wave_full ? nblk : qb + 1;

    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;

    // lane masks of the forward's keep decisions for this query block, one sub-tile (16 x 64 bit, scalar loads) ahead
    const const_u64* mrow = reinterpret_cast<const const_u64*>(reinterpret_cast<uintptr_t>(
        (DROP && MASK) ? uniform_ptr(dmask + sg.mask0 + (long)qb * nblk * 32) : nullptr));
    unsigned long long mcur[16], mnext[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) mcur[r] = (DROP && MASK) ? mrow[kb_beg * 16 + r] : 0ull;

#pragma unroll 1
    for (int kbk = kb_beg; kbk < kb_end; ++kbk) {
      const int k0 = kbk * 32;
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(imgK, k0, ks, lane), qf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(imgV, k0, ks, lane), dof[ks], dpt, 0, 0, 0);
      }
      if (DROP && MASK) {       // behind the operand reads of this sub-tile: the loads have a whole sub-tile to land
        __builtin_amdgcn_sched_barrier(0);
        const int kn = min(kbk + 1, kb_end - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) mnext[r] = mrow[kn * 16 + r];
        __builtin_amdgcn_sched_barrier(0);
      }
      // dS^T = P^T o (keep*dP^T - D/s); zero where the score was REPLACED by the causal constant
      const uint32_t g0 = ((uint32_t)sg.hrow + (uint32_t)q) * sg.T4 + (uint32_t)((k0 + 4 * (lane >> 5)) >> 2);
      const bool interior = (kbk < qb) && !((padmask >> kbk) & 1) && (k0 + 32 <= T);
      if (interior) {
        const float nlse = -my_lse;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = (DROP && !MASK) ? drop_word(g0 + 2 * j, drop_key) : 0u;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int r = 4 * j + e;
            const f32x2_v pv = exp2_fast2(__builtin_elementwise_fma(pk2(st[r], st[r + 1]), (f32x2_v)(scale2), (f32x2_v)(nlse)));
            float dp0 = dpt[r], dp1 = dpt[r + 1];
            if (DROP) {
              dp0 = MASK ? keep_lanes(dp0, mcur[r]) : (drop_byte_keep(w, e, drop_thr) ? dp0 : 0.f);
              dp1 = MASK ? keep_lanes(dp1, mcur[r + 1]) : (drop_byte_keep(w, e + 1, drop_thr) ? dp1 : 0.f);
            }
            const f32x2_v ds = pv * (pk2(dp0, dp1) - (f32x2_v)(my_D));
            st[r] = ds.x;
            st[r + 1] = ds.y;
          }
        }
      } else {
        const int lim_causal = q - k0 - 4 * (lane >> 5);
        const int lim_len = T - 1 - k0 - 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = (DROP && !MASK) ? drop_word(g0 + 2 * j, drop_key) : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * j + e;
            const int c = (r & 3) + 8 * (r >> 2);
            const bool causal_ok = c <= lim_causal;
            const float sv = ldsKb[k0 + c + 4 * (lane >> 5)] + (causal_ok ? st[r] * scale2 : MASK_VAL * LOG2E);
            const float pv = exp2_fast((c <= lim_len) ? sv - my_lse : -INFINITY);      // select, not a branch: 2^-inf = 0
            float dpe = dpt[r];
            if (DROP) dpe = MASK ? keep_lanes(dpe, mcur[r]) : (drop_byte_keep(w, e, drop_thr) ? dpe : 0.f);
            st[r] = causal_ok ? pv * (dpe - my_D) : 0.f;
          }
        }
      }
      if (DROP && MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mcur[r] = mnext[r];
      }
      // dQ^T += K^T . dS^T
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgK, k0, s2, lane), frag_from_acc(st, s2), dq, 0, 0, 0);
    }

    // dS was formed as P o (keep*dP - D/s): the dropout survivor scale s multiplies the result once, here
    const float qs = scale * (DROP ? drop_scale : 1.0f);
    if (qvalid) {
      bf16_t* orow = dqkv + (sg.row0 + q) * ld + h * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dq[4 * g + 0] * qs, dq[4 * g + 1] * qs);
        pk.y = pack_bf16x2(dq[4 * g + 2] * qs, dq[4 * g + 3] * qs);
        *reinterpret_cast<uint2*>(orow + 8 * g + 4 * (lane >> 5)) = pk;
      }
    }
  }
}

// =====================================================================================================
// backward dK/dV: lanes own keys; images Q and dO, per-query lse and D in LDS
// =====================================================================================================
template <bool DROP, bool MASK>
__global__ __launch_bounds__(NEKO_DKV_WAVES * 64) void attn_dkv_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ kbias, const float* __restrict__ lse,
                                                            const float* __restrict__ Din, bf16_t* __restrict__ dqkv, int B,
                                                            int T_uniform, int H, float scale, uint32_t drop_thr,
                                                            uint32_t drop_key, float drop_scale,
                                                            const uint32_t* __restrict__ dmask, const int* __restrict__ seq_off,
                                                            const long long* __restrict__ mask_off, int T4_varlen) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  NEKO_ATRACE(0, __builtin_amdgcn_s_memrealtime());
  if (DROP) drop_key += neko_drop_salt();
  const int hb = pair_remap(blockIdx.x, B * H);
  const int b = hb / H, h = hb % H;
  const SeqGeom sg = seq_geom(b, h, H, T_uniform, seq_off, mask_off, T4_varlen);
  const int T = sg.T;
  const int Tp = (T + 31) & ~31, nblk = Tp >> 5;
  char* imgQ = smem;
  char* imgdO = smem + Tp * 64;
  float* ldsLse = reinterpret_cast<float*>(smem + Tp * 128);
  float* ldsD = ldsLse + Tp;
  int* queue = reinterpret_cast<int*>(ldsD + Tp);
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  if (tid == 0) *queue = 0;
  const int d = H * 32;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + sg.row0 * ld + h * 32;
  const bf16_t* dobase = dout + sg.row0 * d + h * 32;
  const float* kb = kbias + sg.row0;
  const float* lse_b = lse + sg.hrow;
  const float* D_b = Din + sg.hrow;

  // Staging: Q and dO images, -lse and D[q] = sum_hd dO.O / s (written by the dQ kernel, which runs first and needs the sum
  // for its own rows anyway: a third of this prologue's bytes were the O rows read only to form it).  EVERY global load
  // of the prologue -- two 16-byte pieces per (row, piece) slot, lse, D and the key-bias words behind the padding mask --
  // is issued before the first one is waited for.  The workgroup is alone on its CU and all 256 CUs stage at the same
  // time: tools/attn_trace.py measured 16 us from kernel entry to the staging barrier (3 TB/s of 64-byte row pieces).
  float padv[16];
  pad_mask_load(kb, T, nblk, lane, padv);
  {
    constexpr int NB = 6;                                  // 6 * nthr slots >= 4 * Tp pieces: one round
    const int total = Tp * 4;
    for (int c0 = 0; c0 < total; c0 += nthr * NB) {
      uint4 rq[NB], rdo[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int c = c0 + i * nthr + tid, row = c >> 2, pc = c & 3;
        rq[i] = make_uint4(0, 0, 0, 0);
        rdo[i] = make_uint4(0, 0, 0, 0);
        if (c < total && row < T) {
          rq[i] = *reinterpret_cast<const uint4*>(qbase + (long)row * ld + pc * 8);
          rdo[i] = *reinterpret_cast<const uint4*>(dobase + (long)row * d + pc * 8);
        }
      }
      float lv[2] = {0.f, 0.f}, dvv[2] = {0.f, 0.f};
      if (c0 == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = i * nthr + tid;
          if (r < T) { lv[i] = lse_b[r]; dvv[i] = D_b[r]; }
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int c = c0 + i * nthr + tid, row = c >> 2, pc = c & 3;
        if (c < total) {
          *reinterpret_cast<uint4*>(imgQ + img_off(row, pc)) = rq[i];
          *reinterpret_cast<uint4*>(imgdO + img_off(row, pc)) = rdo[i];
        }
      }
      if (c0 == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = i * nthr + tid;
          if (r < Tp) { ldsLse[r] = -lv[i] * LOG2E; ldsD[r] = dvv[i]; }      // -lse: the exponent is fma(s, scale, -lse)
        }
      }
    }
    for (int r = 2 * nthr + tid; r < Tp; r += nthr) {      // (Tp > 2 * nthr only)
      ldsLse[r] = (r < T) ? -lse_b[r] * LOG2E : 0.f;
      ldsD[r] = (r < T) ? D_b[r] : 0.f;
    }
  }
  const uint32_t padmask = pad_mask_ballot(padv);            // bit j: query block j holds a masked (padded) row
  __syncthreads();
  // bit j of qactive: query block j holds a masked row whose dO is not exactly zero.  Only such rows reach keys beyond
  // their diagonal (dO == 0 gives dP = D = 0, so P^T.dO and dS vanish): in training no loss reaches a padded position
  // and the early blocks are visited by the causal range only.
  uint32_t qactive = 0;
  for (uint32_t m = padmask; m; m &= m - 1) {
    const int j = __ffs(m) - 1, row = j * 32 + (lane & 31);
    const uint4 u0 = *reinterpret_cast<const uint4*>(imgdO + img_off(row, (lane >> 5) * 2));
    const uint4 u1 = *reinterpret_cast<const uint4*>(imgdO + img_off(row, (lane >> 5) * 2 + 1));
    const bool nz = ((u0.x | u0.y | u0.z | u0.w | u1.x | u1.y | u1.z | u1.w) != 0u) && row < T && kb[row] != 0.f;
    if (__builtin_amdgcn_ballot_w64(nz) != 0) qactive |= 1u << j;
  }

  const float scale2 = scale * LOG2E;
  const uint32_t T4 = sg.T4;
  NEKO_ATRACE(1, __builtin_amdgcn_s_memrealtime());
  unsigned long long ntiles_traced = 0;
#pragma unroll 1
  for (int kbw = next_item(queue, lane); kbw < nblk; kbw = next_item(queue, lane)) {   // early key blocks are seen by
    const int kw0 = kbw * 32;                                                          // the most queries: heaviest first
    const int key = kw0 + (lane & 31);
    const bool kvalid = key < T;
    const float my_kb = (kvalid ? kb[key] : 0.f) * LOG2E;
    bf16x8_v kf[2], vf[2];
    row_frags(qbase + d + (long)key * ld, kvalid, lane, kf);
    row_frags(qbase + 2 * d + (long)key * ld, kvalid, lane, vf);
    const bool keys_plain = (kw0 + 31 < T) && !((padmask >> kbw) & 1);

    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
    const int ksh = 8 * (lane & 3);

    // query blocks this key block meets: the masked-and-live ones below the diagonal (qactive), then every block from the
    // diagonal on.  next_qb() is wave-uniform scalar work; the dropout mask dword of the NEXT visited block (this lane's
    // key, 32 query bits) is requested one sub-tile ahead.
    auto next_qb = [&](int from) {
      int nq = from;
      while (nq < kbw && !((qactive >> nq) & 1)) ++nq;
      return nq;
    };
    // keep words of the forward pass for this lane's key: mask[(head, query block, this key block)][mask_slot_of_key], one
    // dword (32 query bits) per visited block, requested one block ahead.  Measured and not kept (tools/attn_trace.py:
    // 1.14 us per sub-tile without dropout, 1.58 us with, 1.39 us with the loads ablated): two blocks of lead through two
    // alternating registers, the same through a per-wave LDS-DMA ring (global_load_lds_dword + counted vmcnt), and a
    // key-block-major mask layout that makes this kernel's walk sequential -- 1.58 / 1.63 / 1.74 us: it is not latency.
    const uint32_t* mcol = (DROP && MASK) ? dmask + sg.mask0 + (long)kbw * 32 + mask_slot_of_key(lane & 31) : nullptr;
    auto mask_word = [&](int q) { return (DROP && MASK && !(NEKO_ATTN_ABL & 4)) ? mcol[(long)min(q, nblk - 1) * nblk * 32] : 0xFFFFFFFFu; };
    auto sub_tile = [&](const int qb, const uint32_t wraw) {
      const int q0 = qb * 32;
      const uint32_t wsh = wraw >> (4 * (lane >> 5));           // bit c(r): keep(query q0 + c(r) + 4 (lane / 32), this key)
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(imgQ, NEKO_ATTN_Q0(q0), ks, lane), kf[ks], st, 0, 0, 0);
        dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(imgdO, NEKO_ATTN_Q0(q0), ks, lane), vf[ks], dpt, 0, 0, 0);
      }
      // dropout words: rows of this sub-tile are registers, the 4 lanes of a quad own the 4 keys of one group -> lane
      // (key & 3) = i hashes rows 4j + i and the quad shares the 16 words by DPP
      uint32_t mine[4] = {0u, 0u, 0u, 0u};
      if (DROP && !MASK) {
        const uint32_t gq = ((uint32_t)sg.hrow + (uint32_t)(q0 + 4 * (lane >> 5) + (lane & 3))) * T4 +
                            (uint32_t)(key >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j] = drop_word(gq + (uint32_t)(8 * j) * T4, drop_key);    // row c = (lane&3) + 8j
      }
      const float* lq = ldsLse + q0 + 4 * (lane >> 5);
      const float* dq_ = ldsD + q0 + 4 * (lane >> 5);
      // wave-uniform fast path: every query of the block sees every key of this wave, no key is padded or invalid
      const bool interior = (qb > kbw) && (q0 + 31 < T) && keys_plain;
      if (interior && (NEKO_ATTN_ABL & 1)) {
        // ablation: no elementwise work at all
      } else if (interior) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int c = (r & 3) + 8 * (r >> 2);          // r even: c + 1 is the column of r + 1
          const f32x2_v nlse_q = pk2(lq[c], lq[c + 1]), d_q = pk2(dq_[c], dq_[c + 1]);
          const f32x2_v pv = exp2_fast2(__builtin_elementwise_fma(pk2(st[r], st[r + 1]), (f32x2_v)(scale2), nlse_q));
          f32x2_v ds;
          if (DROP) {
            // dS = P o (M o dP - D) = (M o P) o dP - P o D: ONE masked quantity (the dropped P that dV needs anyway)
            // instead of two, and the rest is a packed multiply and a packed fma
            f32x2_v pd;
            if (MASK) {
              pd.x = __uint_as_float(__float_as_uint(pv.x) & keep_bits(wsh, c));      // all ones where kept
              pd.y = __uint_as_float(__float_as_uint(pv.y) & keep_bits(wsh, c + 1));
            } else {
              pd.x = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr ? pv.x : 0.f;
              pd.y = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], (r + 1) & 3), ksh, 8) >= drop_thr ? pv.y : 0.f;
            }
            ds = __builtin_elementwise_fma(pd, pk2(dpt[r], dpt[r + 1]), -(pv * d_q));
            st[r] = pd.x;
            st[r + 1] = pd.y;
          } else {
            ds = pv * (pk2(dpt[r], dpt[r + 1]) - d_q);
            st[r] = pv.x;
            st[r + 1] = pv.y;
          }
          dpt[r] = ds.x;
          dpt[r + 1] = ds.y;
        }
      } else {
        const int lim_causal = q0 + 4 * (lane >> 5) - key;    // key <= query  <=>  -c(r) <= lim_causal
        const int lim_len = T - 1 - q0 - 4 * (lane >> 5);     // query < T     <=>   c(r) <= lim_len
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const bool causal_ok = (-c) <= lim_causal;
          const float sv = (causal_ok ? st[r] * scale2 : MASK_VAL * LOG2E) + my_kb;
          const float nlse_q = lq[c], d_q = dq_[c];     // unconditional: a load inside ?: compiles to a branch
          const float pv = exp2_fast((c <= lim_len && kvalid) ? sv + nlse_q : -INFINITY);       // select: 2^-inf = 0
          float pd = pv, dpe = dpt[r];
          if (DROP) {
            if (MASK) {
              const uint32_t km = (uint32_t)__builtin_amdgcn_sbfe((int)wsh, c, 1);
              pd = __uint_as_float(__float_as_uint(pv) & km);
              dpe = __uint_as_float(__float_as_uint(dpe) & km);
            } else {
              const bool keep = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr;   // word of row r
              pd = keep ? pv : 0.f;
              dpe = keep ? dpe : 0.f;
            }
          }
          st[r] = pd;                                              // dropped P (for dV)
          dpt[r] = causal_ok ? pv * (dpe - d_q) : 0.f;             // dS        (for dK)
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgdO, NEKO_ATTN_Q0(q0), s2, lane), frag_from_acc(st, s2), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgQ, NEKO_ATTN_Q0(q0), s2, lane), frag_from_acc(dpt, s2), dk, 0, 0, 0);
      }
#ifdef NEKO_ATTN_TRACE
      ++ntiles_traced;
#endif
    };
    int qb = next_qb(0);
    uint32_t wnext = mask_word(qb);
#pragma unroll 1
    while (qb < nblk) {
      const int qb_next = next_qb(qb + 1);
      const uint32_t w = wnext;
      wnext = mask_word(qb_next);
      sub_tile(qb, w);
      qb = qb_next;
    }

    // P and dP were masked but not scaled in the loop (D holds D/s): the survivor scale s is applied once, here
    const float vsc = DROP ? drop_scale : 1.0f, ksc = scale * vsc;
    if (kvalid) {
      bf16_t* krow = dqkv + (sg.row0 + key) * ld + d + h * 32;
      bf16_t* vrow = krow + d;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dk[4 * g + 0] * ksc, dk[4 * g + 1] * ksc);
        pk.y = pack_bf16x2(dk[4 * g + 2] * ksc, dk[4 * g + 3] * ksc);
        *reinterpret_cast<uint2*>(krow + 8 * g + 4 * (lane >> 5)) = pk;
        pk.x = pack_bf16x2(dv[4 * g + 0] * vsc, dv[4 * g + 1] * vsc);
        pk.y = pack_bf16x2(dv[4 * g + 2] * vsc, dv[4 * g + 3] * vsc);
        *reinterpret_cast<uint2*>(vrow + 8 * g + 4 * (lane >> 5)) = pk;
      }
    }
  }
  NEKO_ATRACE(2, __builtin_amdgcn_s_memrealtime());
  NEKO_ATRACE(3, ntiles_traced);
}

// =====================================================================================================
// backward in ONE pass over S / dP (round 4): lanes own keys, dK / dV in registers, dQ through LDS
// =====================================================================================================
// The two kernels above each rebuild S = Q.K^T, the exponentials, the dropout decisions and dP = dO.V^T for every 32 x 32
// sub-tile (6 + 8 MFMAs, 2 x 16 v_exp_f32, twice the elementwise block).  Here every sub-tile is visited once, by the wave that
// owns its KEY block: P^T.dO and dS^T.Q accumulate in registers as in attn_dkv_res_kernel, and the sub-tile's dQ contribution
// dS.K -- a contraction over the keys, which are the LANES of this layout -- goes through a per-wave 2 KB LDS scratch: the packed
// bf16 dS fragments (the very registers the dK MFMA consumes) are stored as [key][query], read back transposed with
// ds_read_b64_tr_b16 as the A operand of two more MFMAs against K^T (the key block's own K rows, transposed once through the same
// scratch), and the 32 x 32 fp32 result is added into an LDS accumulator dQ[query][32] (four 16-B read-add-writes per lane under a
// per-query-block lock).  10 MFMAs and one elementwise block per sub-tile instead of 14 and two.
// LDS: Q and dO images and the fp32 dQ accumulator for 512 queries = 128 KB, so a head of up to 1024 positions runs in two
// phases (queries 0..511, then 512..1023); the eight waves own key blocks w and 15 - w across both phases (their dK / dV partial
// sums stay in registers over the phase boundary: 17 + 32 sub-tiles each, balanced), the key blocks from 16 on are handed out from
// a queue in the second phase, longest first.  Shorter heads (T <= 512) are one phase with every key block from the queue.
// The order in which the waves add to a dQ block depends on their timing: dQ is reproducible only to fp32 rounding (then rounded
// to bf16); the two-kernel form stays available for bit-reproducible runs (neko_attn_set_path(2), NEKO_DETERMINISTIC).
constexpr int FUSED_Q = 512;      // queries resident per phase
constexpr int FUSED_W = 8;        // waves per workgroup (2 per SIMD)

// frag_cols for the dS scratch: rows = keys, and inside a row the 8-byte units (4 queries each) sit where the writer's packed
// accumulator fragments put them: piece 2 s2 + hh of the row holds queries 16 s2 + 4 hh + {0..3} and 16 s2 + 8 + 4 hh + {0..3}
__device__ __forceinline__ bf16x8_v frag_cols_ds(const char* scr, int s, int lane) {
  const int g = lane >> 4, c16 = lane & 15;
  const int v = c16 & 3;                                          // natural unit inside the 16-query half g & 1
  const int piece = 2 * (g & 1) + (v & 1), byte = 8 * (v >> 1);
  const int r_lo = 16 * s + 4 * (g >> 1) + (c16 >> 2), r_hi = r_lo + 8;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(scr + r_lo * 64 + ((piece ^ ((r_lo >> 2) & 3)) << 4) + byte));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(scr + r_hi * 64 + ((piece ^ ((r_hi >> 2) & 3)) << 4) + byte));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_v, r);
}

template <bool DROP, bool MASK>
__global__ __launch_bounds__(FUSED_W * 64) void attn_bwd_fused_res_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ outp, const bf16_t* __restrict__ dout, const float* __restrict__ kbias,
    const int* __restrict__ kstart, const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int B, int T_uniform, int H, float scale, uint32_t drop_thr,
    uint32_t drop_key, float drop_scale, const uint32_t* __restrict__ dmask, const int* __restrict__ seq_off,
    const long long* __restrict__ mask_off, int T4_varlen, int Rmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (DROP) drop_key += neko_drop_salt();
  const int hb = pair_remap(blockIdx.x, B * H);
  const int b = hb / H, h = hb % H;
  const SeqGeom sg = seq_geom(b, h, H, T_uniform, seq_off, mask_off, T4_varlen);
  const int T = sg.T;
  const int Tp = (T + 31) & ~31, nblk = Tp >> 5;
  // LDS carve (Rmax = resident query rows the launch was sized for)
  char* imgQ = smem;
  char* imgdO = imgQ + Rmax * 64;
  float* dQacc = reinterpret_cast<float*>(imgdO + Rmax * 64);
  float* ldsLse = dQacc + Rmax * 32;
  float* ldsD = ldsLse + Rmax;
  char* scratch = reinterpret_cast<char*>(ldsD + Rmax);
  int* queue = reinterpret_cast<int*>(scratch + FUSED_W * 2048);
  int* locks = queue + 4;                                    // one per resident query block (16)
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* scr = scratch + wave * 2048;
  const int d = H * 32;
  const long ld = 3L * d;
  const bf16_t* qbase = qkv + sg.row0 * ld + h * 32;
  const bf16_t* dobase = dout + sg.row0 * d + h * 32;
  const bf16_t* obase = outp + sg.row0 * d + h * 32;
  const float* kb = kbias + sg.row0;
  const float* lse_b = lse + sg.hrow;

  float padv[16];
  pad_mask_load(kb, T, nblk, lane, padv);
  const uint32_t padmask = pad_mask_ballot(padv);            // bit j: block j holds a padded position
  const float scale2 = scale * LOG2E;
  const uint32_t T4 = sg.T4;
  const int nph = Tp > FUSED_Q ? 2 : 1;
  const float inv_ds = DROP ? 1.0f / drop_scale : 1.0f;
  const int kb_first = kstart ? (kstart[b] >> 5) : 0;           // key blocks below it hold padded keys only

  f32x16 sdk0, sdv0, sdk1, sdv1;       // partial dK / dV of the wave's two static key blocks, carried over the phase boundary
#pragma unroll
  for (int r = 0; r < 16; ++r) { sdk0[r] = 0.f; sdv0[r] = 0.f; sdk1[r] = 0.f; sdv1[r] = 0.f; }

#pragma unroll 1
  for (int ph = 0; ph < nph; ++ph) {
    const int qlo = ph * FUSED_Q;
    const int R = min(Tp - qlo, FUSED_Q);                    // resident rows of this phase (multiple of 32)
    const int qb_lo = qlo >> 5, qb_hi = qb_lo + (R >> 5);    // query blocks [qb_lo, qb_hi)
    if (ph) __syncthreads();                                 // every wave is past the previous phase's images and accumulator
    if (tid == 0) *queue = 0;
    if (tid < 16) locks[tid] = 0;
    // ---- staging: Q, dO images (rows qlo ..), D = sum dO.O / s, -lse, zeroed dQ accumulator --------------------------------------
    {
      constexpr int NB = 4;                                  // 4 * 512 slots = 512 rows x 4 pieces: one round
      const int total = R * 4;
      for (int c0 = 0; c0 < total; c0 += nthr * NB) {
        uint4 rq[NB], rdo[NB], ro[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int c = c0 + i * nthr + tid, row = c >> 2, pc = c & 3;
          rq[i] = make_uint4(0, 0, 0, 0);
          rdo[i] = make_uint4(0, 0, 0, 0);
          ro[i] = make_uint4(0, 0, 0, 0);
          if (c < total && qlo + row < T) {
            rq[i] = *reinterpret_cast<const uint4*>(qbase + (long)(qlo + row) * ld + pc * 8);
            rdo[i] = *reinterpret_cast<const uint4*>(dobase + (long)(qlo + row) * d + pc * 8);
            ro[i] = *reinterpret_cast<const uint4*>(obase + (long)(qlo + row) * d + pc * 8);
          }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int c = c0 + i * nthr + tid, row = c >> 2, pc = c & 3;
          float part = dot8_bf16(rdo[i], ro[i]);             // the 4 pieces of a row sit in the 4 lanes of a quad
          part += __shfl_xor(part, 1, 64);
          part += __shfl_xor(part, 2, 64);
          if (c < total) {
            *reinterpret_cast<uint4*>(imgQ + img_off(row, pc)) = rq[i];
            *reinterpret_cast<uint4*>(imgdO + img_off(row, pc)) = rdo[i];
            if (pc == 0) ldsD[row] = part * inv_ds;
          }
        }
      }
      for (int r = tid; r < R; r += nthr) ldsLse[r] = (qlo + r < T) ? -lse_b[qlo + r] * LOG2E : 0.f;
      for (int i = tid; i < R * 8; i += nthr) reinterpret_cast<float4*>(dQacc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // bit j of qactive: query block j (of this phase's window) holds a masked row whose dO is not exactly zero -- only such rows
    // reach keys beyond their diagonal (see attn_dkv_res_kernel)
    uint32_t qactive = 0;
    for (uint32_t m = padmask & (qb_hi >= 32 ? 0xffffffffu : ((1u << qb_hi) - 1u)) & ~((1u << qb_lo) - 1u); m; m &= m - 1) {
      const int j = __ffs(m) - 1, row = (j - qb_lo) * 32 + (lane & 31);
      const uint4 u0 = *reinterpret_cast<const uint4*>(imgdO + img_off(row, (lane >> 5) * 2));
      const uint4 u1 = *reinterpret_cast<const uint4*>(imgdO + img_off(row, (lane >> 5) * 2 + 1));
      const bool nz = ((u0.x | u0.y | u0.z | u0.w | u1.x | u1.y | u1.z | u1.w) != 0u) && qlo + row < T && kb[qlo + row] != 0.f;
      if (__builtin_amdgcn_ballot_w64(nz) != 0) qactive |= 1u << j;
    }

    // ---- this wave's key blocks of the phase ---------------------------------------------------------------------------------
#pragma unroll 1
    for (int u = 0;; ++u) {
      const bool is_static = nph == 2 && u < 2;
      int kbw;
      if (is_static) kbw = u == 0 ? wave : 15 - wave;
      else {
        if (nph == 2 && ph == 0) break;
        kbw = (nph == 2 ? 16 : 0) + next_item(queue, lane);
        if (kbw >= qb_hi) break;                             // key blocks beyond the window's last query block see none of it
      }
      const int kw0 = kbw * 32;
      const int key = kw0 + (lane & 31);
      const bool kvalid = key < T;
      const float my_kb = (kvalid ? kb[key] : 0.f) * LOG2E;
      bf16x8_v kf[2], vf[2];
      row_frags(qbase + d + (long)key * ld, kvalid, lane, kf);
      row_frags(qbase + 2 * d + (long)key * ld, kvalid, lane, vf);
      const bool keys_valid = kw0 + 31 < T;                       // every key of the block exists
    const bool keys_biased = (padmask >> kbw) & 1;              // some key of the block is padded: its bias joins the exponent
      // K^T of the block (head dim on the lanes) through the scratch: the B operand of the dQ products
      *reinterpret_cast<uint4*>(scr + img_off(lane & 31, lane >> 5)) = __builtin_bit_cast(uint4, kf[0]);
      *reinterpret_cast<uint4*>(scr + img_off(lane & 31, 2 + (lane >> 5))) = __builtin_bit_cast(uint4, kf[1]);
      bf16x8_v kT[2];
      kT[0] = frag_cols(scr, 0, 0, lane);
      kT[1] = frag_cols(scr, 0, 1, lane);

      f32x16 dk, dv;
      if (is_static && ph == 1) {
        if (u == 0) { dk = sdk0; dv = sdv0; } else { dk = sdk1; dv = sdv1; }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
      }
      const int ksh = 8 * (lane & 3);
      const bool kpad_all = kbw < kb_first;                  // all-padding key block: only masked-and-live query blocks see it
      auto next_qb = [&](int from) {                         // next visited query block >= from inside the window
        int nq = max(from, qb_lo);
        while (nq < qb_hi && (nq < kbw || kpad_all) && !((qactive >> nq) & 1)) ++nq;
        return nq;
      };
      const uint32_t* mcol = (DROP && MASK) ? dmask + sg.mask0 + (long)kbw * 32 + mask_slot_of_key(lane & 31) : nullptr;
      auto mask_word = [&](int q) { return (DROP && MASK) ? mcol[(long)min(q, nblk - 1) * nblk * 32] : 0xFFFFFFFFu; };
      // this lane's row of a query block inside the accumulator: row (lane & 31), 16-B unit u of the row (4 head-dim columns) at
      // unit u ^ ((row >> 1) & 7) -- the 16 lanes of a ds_read_b128 group then cover 16 different units of the 256-B bank row
      const int dq_swz = ((lane & 31) >> 1) & 7;
      float* dq_lane = dQacc + (lane & 31) * 32;                           // + local query block * 1024
      auto sub_tile = [&](const int qb, const uint32_t wraw) {
        const int q0 = qb * 32, ql0 = q0 - qlo;                 // global / resident first query of the block
        const uint32_t wsh = wraw >> (4 * (lane >> 5));
        f32x16 st, dpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(imgQ, ql0, ks, lane), kf[ks], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(imgdO, ql0, ks, lane), vf[ks], dpt, 0, 0, 0);
        }
        uint32_t mine[4] = {0u, 0u, 0u, 0u};
        if (DROP && !MASK) {
          const uint32_t gq = ((uint32_t)sg.hrow + (uint32_t)(q0 + 4 * (lane >> 5) + (lane & 3))) * T4 + (uint32_t)(key >> 2);
#pragma unroll
          for (int j = 0; j < 4; ++j) mine[j] = drop_word(gq + (uint32_t)(8 * j) * T4, drop_key);
        }
        const float* lq = ldsLse + ql0 + 4 * (lane >> 5);
        const float* dq_ = ldsD + ql0 + 4 * (lane >> 5);
        // a padded key needs no other treatment than its additive bias (the reference ADDS the padding mask, trajectory_gpt2.py:
        // 663-679), so key blocks that hold padded keys stay on the fast path here; in attn_dkv_res_kernel they take the per-element
        // path for every query block, which makes key block 0 of every left-padded sequence ~1.5x as expensive as the others (and
        // with the static key-block ownership of this kernel that wave would hold up the whole phase)
        const bool interior = (qb > kbw) && (q0 + 31 < T) && keys_valid;
        if (interior) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int c = (r & 3) + 8 * (r >> 2);
            f32x2_v nlse_q = pk2(lq[c], lq[c + 1]);
            const f32x2_v d_q = pk2(dq_[c], dq_[c + 1]);
            if (keys_biased) nlse_q += (f32x2_v)(my_kb);          // wave-uniform branch
            const f32x2_v pv = exp2_fast2(__builtin_elementwise_fma(pk2(st[r], st[r + 1]), (f32x2_v)(scale2), nlse_q));
            f32x2_v ds;
            if (DROP) {
              f32x2_v pd;
              if (MASK) {
                pd.x = __uint_as_float(__float_as_uint(pv.x) & keep_bits(wsh, c));
                pd.y = __uint_as_float(__float_as_uint(pv.y) & keep_bits(wsh, c + 1));
              } else {
                pd.x = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr ? pv.x : 0.f;
                pd.y = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], (r + 1) & 3), ksh, 8) >= drop_thr ? pv.y : 0.f;
              }
              ds = __builtin_elementwise_fma(pd, pk2(dpt[r], dpt[r + 1]), -(pv * d_q));
              st[r] = pd.x;
              st[r + 1] = pd.y;
            } else {
              ds = pv * (pk2(dpt[r], dpt[r + 1]) - d_q);
              st[r] = pv.x;
              st[r + 1] = pv.y;
            }
            dpt[r] = ds.x;
            dpt[r + 1] = ds.y;
          }
        } else {
          const int lim_causal = q0 + 4 * (lane >> 5) - key;
          const int lim_len = T - 1 - q0 - 4 * (lane >> 5);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            const bool causal_ok = (-c) <= lim_causal;
            const float sv = (causal_ok ? st[r] * scale2 : MASK_VAL * LOG2E) + my_kb;
            const float nlse_q = lq[c], d_q = dq_[c];
            const float pv = exp2_fast((c <= lim_len && kvalid) ? sv + nlse_q : -INFINITY);
            float pd = pv, dpe = dpt[r];
            if (DROP) {
              if (MASK) {
                const uint32_t km = (uint32_t)__builtin_amdgcn_sbfe((int)wsh, c, 1);
                pd = __uint_as_float(__float_as_uint(pv) & km);
                dpe = __uint_as_float(__float_as_uint(dpe) & km);
              } else {
                const bool keep = __builtin_amdgcn_ubfe(quad_bcast(mine[r >> 2], r & 3), ksh, 8) >= drop_thr;
                pd = keep ? pv : 0.f;
                dpe = keep ? dpe : 0.f;
              }
            }
            st[r] = pd;
            dpt[r] = causal_ok ? pv * (dpe - d_q) : 0.f;
          }
        }
        const bf16x8_v dsf0 = frag_from_acc(dpt, 0), dsf1 = frag_from_acc(dpt, 1);
        // dS as [key][query] into the scratch (one 16-B piece per fragment), ahead of the dV / dK products that hide the round trip
        *reinterpret_cast<uint4*>(scr + img_off(lane & 31, lane >> 5)) = __builtin_bit_cast(uint4, dsf0);
        *reinterpret_cast<uint4*>(scr + img_off(lane & 31, 2 + (lane >> 5))) = __builtin_bit_cast(uint4, dsf1);
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgdO, ql0, 0, lane), frag_from_acc(st, 0), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgQ, ql0, 0, lane), dsf0, dk, 0, 0, 0);
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgdO, ql0, 1, lane), frag_from_acc(st, 1), dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols(imgQ, ql0, 1, lane), dsf1, dk, 0, 0, 0);
        // dQ^T[hd][query] += K^T[hd][key] . dS^T[key][query]: lanes = queries, registers = head-dim columns c(r) + 4 (lane / 32),
        // i.e. four 16-B units of the lane's accumulator row.  The accumulator block of a query block is shared by the waves
        // (one per key block): read-add-write under the block's lock, taken by lane 0 with a compare-and-swap that is issued
        // BEFORE the products (its round trip hides under them); LDS float atomics are not an option -- ds_add_f32 retires about
        // one lane per three cycles (16 of them per sub-tile: 4.4 ms per call instead of 0.65, profiles/r04_attn_fused.txt).
        int* lock = locks + (ql0 >> 5);
        int busy = 0;
#if NEKO_ATTN_FUSED_ABL == 0
        {
          int expect = 0;
          if (lane == 0) busy = __hip_atomic_compare_exchange_strong(lock, &expect, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0 : 1;
        }
#endif
        f32x16 dqt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[r] = 0.f;
        dqt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT[0], frag_cols_ds(scr, 0, lane), dqt, 0, 0, 0);
        dqt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT[1], frag_cols_ds(scr, 1, lane), dqt, 0, 0, 0);
#if NEKO_ATTN_FUSED_ABL == 0
        busy = __builtin_amdgcn_readfirstlane(busy);
        while (busy) {                                           // another wave is adding to this query block: a few hundred cycles
          __builtin_amdgcn_s_sleep(2);
          int expect = 0;
          if (lane == 0) busy = __hip_atomic_compare_exchange_strong(lock, &expect, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0 : 1;
          busy = __builtin_amdgcn_readfirstlane(busy);
        }
#endif
        float* drow = dq_lane + ql0 * 32;
#if NEKO_ATTN_FUSED_ABL != 2
        float4 acc4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc4[j] = *reinterpret_cast<const float4*>(drow + (((2 * j + (lane >> 5)) ^ dq_swz) << 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc4[j].x += dqt[4 * j]; acc4[j].y += dqt[4 * j + 1]; acc4[j].z += dqt[4 * j + 2]; acc4[j].w += dqt[4 * j + 3];
          *reinterpret_cast<float4*>(drow + (((2 * j + (lane >> 5)) ^ dq_swz) << 2)) = acc4[j];
        }
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(dqt[r]));
#endif
#if NEKO_ATTN_FUSED_ABL == 0
        if (lane == 0) __hip_atomic_store(lock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
      };
      int qb = next_qb(0);
      uint32_t wnext = mask_word(qb);
#pragma unroll 1
      while (qb < qb_hi) {
        const int qb_next = next_qb(qb + 1);
        const uint32_t w = wnext;
        wnext = mask_word(qb_next);
        sub_tile(qb, w);
        qb = qb_next;
      }

      if (is_static && ph == 0) {
        if (u == 0) { sdk0 = dk; sdv0 = dv; } else { sdk1 = dk; sdv1 = dv; }
      } else {
        const float vsc = DROP ? drop_scale : 1.0f, ksc = scale * vsc;
        if (kvalid) {
          bf16_t* krow = dqkv + (sg.row0 + key) * ld + d + h * 32;
          bf16_t* vrow = krow + d;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack_bf16x2(dk[4 * g + 0] * ksc, dk[4 * g + 1] * ksc);
            pk.y = pack_bf16x2(dk[4 * g + 2] * ksc, dk[4 * g + 3] * ksc);
            *reinterpret_cast<uint2*>(krow + 8 * g + 4 * (lane >> 5)) = pk;
            pk.x = pack_bf16x2(dv[4 * g + 0] * vsc, dv[4 * g + 1] * vsc);
            pk.y = pack_bf16x2(dv[4 * g + 2] * vsc, dv[4 * g + 3] * vsc);
            *reinterpret_cast<uint2*>(vrow + 8 * g + 4 * (lane >> 5)) = pk;
          }
        }
      }
    }
    __syncthreads();                                         // every wave's locked read-add-write of the dQ accumulator has completed
    // ---- dQ rows of the phase: fp32 accumulator -> bf16, scaled once ---------------------------------------------------------
    {
      const float qs = scale * (DROP ? drop_scale : 1.0f);
      for (int c = tid; c < R * 4; c += nthr) {
        const int row = c >> 2, pc = c & 3;
        if (qlo + row < T) {
          const int swz = ((row & 31) >> 1) & 7;
          const float4 a = *reinterpret_cast<const float4*>(dQacc + row * 32 + (((2 * pc) ^ swz) << 2));
          const float4 b4 = *reinterpret_cast<const float4*>(dQacc + row * 32 + (((2 * pc + 1) ^ swz) << 2));
          const uint4 pk = make_uint4(pack_bf16x2(a.x * qs, a.y * qs), pack_bf16x2(a.z * qs, a.w * qs),
                                      pack_bf16x2(b4.x * qs, b4.y * qs), pack_bf16x2(b4.z * qs, b4.w * qs));
          *reinterpret_cast<uint4*>(dqkv + (sg.row0 + qlo + row) * ld + h * 32 + pc * 8) = pk;
        }
      }
    }
  }
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
                 hipSuccess
             ? NEKO_OK
             : NEKO_ERR_LAUNCH;
}
// The > 64 KiB dynamic-LDS limit is a per-DEVICE attribute of a kernel (a process-wide `static const int once` left every
// device but the first without it): one bit per device ordinal, set after fn() -- the allow_lds calls of one kernel family -- succeeded
// on the current device.  Setting it twice from two threads is harmless.
typedef std::atomic<unsigned long long> LdsLimitDone;
template <class Fn>
int lds_limit_once_per_device(LdsLimitDone& done, Fn&& fn) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NEKO_ERR_LAUNCH;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return NEKO_OK;
  const int rc = fn();
  if (rc != NEKO_OK) return rc;
  done.fetch_or(bit, std::memory_order_release);
  return NEKO_OK;
}

}  // namespace

bool neko_attn_res_applicable(int T, int hd) { return hd == 32 && T >= 1 && T <= 1024; }

// dwords of the dropout keep-mask buffer the head-resident kernels exchange (0 when they do not apply)
long neko_attn_res_mask_dwords(int B, int T, int H, int hd) {
  if (!neko_attn_res_applicable(T, hd)) return 0;
  const long nblk = (T + 31) / 32;
  return (long)B * H * nblk * nblk * 32;
}

// seq_off / mask_off null: the uniform (B, T) batch; otherwise nseq = B packed sequences (see SeqGeom) whose longest has T rows
int neko_attn_fwd_res_impl(const bf16_t* qkv, const float* kbias, const int* kstart, bf16_t* out, float* lse, int B, int T,
                           int H, int drop_thr, unsigned drop_key, float drop_scale, uint32_t* dmask, hipStream_t s,
                           const int* seq_off, const long long* mask_off) {
  const int Tp = (T + 31) & ~31, nblk = Tp / 32, nw = min(16, (nblk + 1) / 2);
  const size_t lds = (size_t)Tp * 128 + (size_t)Tp * 4 + 16;
  const float scale = 1.0f / sqrtf(32.0f);
  const int T4 = (T + 3) >> 2;
  static LdsLimitDone done_fwd;
  const int once = lds_limit_once_per_device(done_fwd, [] {
    return allow_lds(attn_fwd_res_kernel<true, true>, 160 * 1024) | allow_lds(attn_fwd_res_kernel<true, false>, 160 * 1024) |
           allow_lds(attn_fwd_res_kernel<false, false>, 160 * 1024);
  });
  if (once != NEKO_OK) return once;
  if (drop_thr && dmask)
    hipLaunchKernelGGL((attn_fwd_res_kernel<true, true>), dim3(B * H), dim3(64 * nw), lds, s, qkv, kbias, kstart, out, lse, B, T,
                       H, scale, (uint32_t)drop_thr, drop_key, drop_scale, dmask, seq_off, mask_off, T4);
  else if (drop_thr)
    hipLaunchKernelGGL((attn_fwd_res_kernel<true, false>), dim3(B * H), dim3(64 * nw), lds, s, qkv, kbias, kstart, out, lse, B, T,
                       H, scale, (uint32_t)drop_thr, drop_key, drop_scale, nullptr, seq_off, mask_off, T4);
  else
    hipLaunchKernelGGL((attn_fwd_res_kernel<false, false>), dim3(B * H), dim3(64 * nw), lds, s, qkv, kbias, kstart, out, lse, B, T,
                       H, scale, 0u, drop_key, drop_scale, nullptr, seq_off, mask_off, T4);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// D = sum dO.O / s is formed by the dQ kernel for its rows and left in the D workspace for the dK/dV kernel (qflags untouched).
// dmask: the keep masks the forward call of the same (qkv, drop_key) wrote, or null (the kernels re-hash the decisions).
int neko_attn_bwd_res_impl(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* kbias, const int* kstart,
                           const float* lse, float* D, bf16_t* dqkv, int B, int T, int H, int drop_thr, unsigned drop_key,
                           float drop_scale, const uint32_t* dmask, hipStream_t s, const int* seq_off,
                           const long long* mask_off) {
  if (!D) return NEKO_ERR_ARG;          // f32 [rows * H]: the dQ kernel leaves sum_hd dO.O / s there for the dK/dV kernel
  // One pass over S / dP (attn_bwd_fused_res_kernel) or the two kernels below.  Automatic: one pass above 256 positions.  A short head is
  // one phase of at most 8 key blocks handed to 8 waves from the queue -- triangle sizes 1..8, half the waves idle at the end -- and
  // the two kernels are ahead: 42.7 vs 73.6 us at T = 128, 96 vs 118 at 256, 224 vs 206 at 384, 334 vs 285 at 512 (B = 64, 24 heads,
  // dropout 0.1); README-size steps (T = 240): c2 5.90 -> 5.75 ms, c3 5.95 -> 5.85; T = 494 (c4) stays with one pass (11.29 vs 11.56).
  // Round 5: above 512 positions (two staging phases, eight waves, 64 registers of carried partial sums) the two kernels are ahead inside
  // the step -- m-mix 64 x 1024: 36.41 -> 36.10 ms over three alternating rounds on one box, profiles/r05_attn_path_ab.txt (round 4
  // measured them level) -- while the single-phase lengths keep the one-pass kernel (c4, T = 494: 11.1 vs 11.15-11.75 ms).  The metric's
  // sequence length therefore runs the bit-reproducible form again.
  // Round 6: the two kernels are the automatic choice at EVERY length -- the one-pass kernel adds dQ up in the order its waves arrive, so a
  // configs[3] run (T = 494) was not run-to-run reproducible by default, for 0.05 ms of a 10.4 ms step
  // (profiles/r06_c4_attn_path_ab.txt: 10.35 vs 10.40 ms over three alternating rounds).  neko_attn_set_path(3) still selects it.
  const int pm = neko_attn_bwd_reproducible_mode() ? 2 : neko_attn_path_mode();       // (a thread's reproducibility request outranks the knob)
  if (pm == 3) {
    const int Tp = (T + 31) & ~31, Rmax = min(Tp, FUSED_Q);
    const size_t lds = (size_t)Rmax * (128 + 128 + 8) + FUSED_W * 2048 + 16 + 64;
    const float scale = 1.0f / sqrtf(32.0f);
    const int T4 = (T + 3) >> 2;
    static LdsLimitDone done_fused;
    const int once_f = lds_limit_once_per_device(done_fused, [] {
      return allow_lds(attn_bwd_fused_res_kernel<true, true>, 160 * 1024) | allow_lds(attn_bwd_fused_res_kernel<true, false>, 160 * 1024) |
             allow_lds(attn_bwd_fused_res_kernel<false, false>, 160 * 1024);
    });
    if (once_f != NEKO_OK) return once_f;
#define NEKO_BWD_FUSED(DROPV, MASKV, THR, MP)                                                                                       \
    hipLaunchKernelGGL((attn_bwd_fused_res_kernel<DROPV, MASKV>), dim3(B * H), dim3(FUSED_W * 64), lds, s, qkv, out, dout, kbias, kstart, lse, \
                       dqkv, B, T, H, scale, (uint32_t)(THR), drop_key, drop_scale, MP, seq_off, mask_off, T4, Rmax)
    if (drop_thr && dmask) NEKO_BWD_FUSED(true, true, drop_thr, dmask);
    else if (drop_thr) NEKO_BWD_FUSED(true, false, drop_thr, nullptr);
    else NEKO_BWD_FUSED(false, false, 0, nullptr);
#undef NEKO_BWD_FUSED
    NEKO_CHECK_LAUNCH();
    return NEKO_OK;
  }
  const int Tp = (T + 31) & ~31, nblk = Tp / 32, nw = min(12, (nblk + 1) / 2), nw_kv = min(NEKO_DKV_WAVES, (nblk + 1) / 2);
  const size_t lds_q = (size_t)Tp * 128 + (size_t)Tp * 4 + 16, lds_kv = (size_t)Tp * 128 + (size_t)Tp * 8 + 16;
  const float scale = 1.0f / sqrtf(32.0f);
  const int T4 = (T + 3) >> 2;
  static LdsLimitDone done_bwd;
  const int once = lds_limit_once_per_device(done_bwd, [] {
    return allow_lds(attn_dq_res_kernel<true, true>, 160 * 1024) | allow_lds(attn_dq_res_kernel<true, false>, 160 * 1024) |
           allow_lds(attn_dq_res_kernel<false, false>, 160 * 1024) | allow_lds(attn_dkv_res_kernel<true, true>, 160 * 1024) |
           allow_lds(attn_dkv_res_kernel<true, false>, 160 * 1024) | allow_lds(attn_dkv_res_kernel<false, false>, 160 * 1024);
  });
  if (once != NEKO_OK) return once;
#define NEKO_BWD_RES(DROPV, MASKV, THR, MP)                                                                                       \
  do {                                                                                                                            \
    hipLaunchKernelGGL((attn_dq_res_kernel<DROPV, MASKV>), dim3(B * H), dim3(64 * nw), lds_q, s, qkv, dout, kbias, kstart, lse,  \
                       out, D, dqkv, B, T, H, scale, (uint32_t)(THR), drop_key, drop_scale, MP, seq_off, mask_off, T4);          \
    NEKO_CHECK_LAUNCH();                                                                                                          \
    hipLaunchKernelGGL((attn_dkv_res_kernel<DROPV, MASKV>), dim3(B * H), dim3(64 * nw_kv), lds_kv, s, qkv, dout, kbias, lse, D, \
                       dqkv, B, T, H, scale, (uint32_t)(THR), drop_key, drop_scale, MP, seq_off, mask_off, T4);                  \
  } while (0)
  if (drop_thr && dmask) NEKO_BWD_RES(true, true, drop_thr, dmask);
  else if (drop_thr) NEKO_BWD_RES(true, false, drop_thr, nullptr);
  else NEKO_BWD_RES(false, false, 0, nullptr);
#undef NEKO_BWD_RES
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

NEKO_DEFINE_SALT_SETTER(attention_res)
