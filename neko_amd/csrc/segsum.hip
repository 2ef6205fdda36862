// Deterministic scatter-add of rows:  out[key[i], :] += src[i, :]  for i = 0 .. n-1, summed IN INDEX ORDER for every destination row.
// Replaces the fp32 atomics of the embedding-table gradients (autograd of the lookups in GatoPolicy.tokenize_input_dicts,
// gato/policy/gato_policy.py:195-432: embed_token / pos_embed_observation / separator_token; PatchPosEncoding, gato/policy/embeddings.py:
// 101-110): the atomic form is order-dependent (1e-7 of an entry, enough to send two identical AdamW runs onto different trajectories)
// and spends its time on contended table rows (the separator row takes thousands of adds per column, a position row ~64).
//
//   1. (key, index) pairs are sorted by key with a stable radix sort (rocPRIM: the one library call of the path);
//   2. one wave per chunk of 32 sorted entries walks its entries in order, 8 source rows in flight: runs that lie inside the chunk are
//      added to their destination row at once (this wave is the row's only writer), the first and the last run of the chunk -- which may
//      continue in the neighbouring chunks -- are left as partial rows;
//   3. one wave per chain head adds the partial rows of a run that spans chunks in chunk order and writes the destination row.
// Every sum has a fixed order: the result is bit-identical from run to run.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "neko_kernels.h"

namespace {

constexpr int SEG_CH = 32;                      // sorted entries per chunk
constexpr unsigned KEY_NONE = NEKO_SEGSUM_KEY_NONE;     // entry without a destination (sorted to the end, skipped); keys use 20 bits
constexpr int KEY_BITS = 20;

__global__ void iota_kernel(int* idx, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = i;
}

struct ChunkMeta { unsigned first_key, last_key; int nruns, pad; };

// destination row of key k: out + k * ld, or the extra row for k == nrows (the separator token's own parameter); null for a key
// outside the table (k > nrows, or k == nrows without an extra row): such entries are dropped, never written anywhere
__device__ __forceinline__ float* dest_row(float* out, long ld, int nrows, float* extra, unsigned k) {
  if (k < (unsigned)nrows) return out + (long)k * ld;
  return k == (unsigned)nrows ? extra : nullptr;
}

__global__ __launch_bounds__(64) void segsum_chunk_kernel(const float* __restrict__ src, long ld_src, const unsigned* __restrict__ keys,
                                                          const int* __restrict__ idx, int n, int d, float* __restrict__ out, long ld_out,
                                                          int nrows, float* __restrict__ extra, ChunkMeta* __restrict__ meta,
                                                          float* __restrict__ part) {
  const int chunk = blockIdx.x, lane = threadIdx.x;
  const int e0 = chunk * SEG_CH;
  const int cnt_all = min(SEG_CH, n - e0);
  // this lane's entry (lanes >= 32 idle here); the walk below reads entries by readlane
  const unsigned kv = lane < cnt_all ? keys[e0 + lane] : KEY_NONE;
  const int iv = lane < cnt_all ? idx[e0 + lane] : 0;
  const int cnt = __popcll(__builtin_amdgcn_ballot_w64(kv != KEY_NONE));          // valid entries are a prefix (NONE sorts last)
  float* pfirst = part + ((long)chunk * 2 + 0) * d;
  float* plast = part + ((long)chunk * 2 + 1) * d;
  int nruns = 0;
  unsigned first_key = KEY_NONE, last_key = KEY_NONE;
  for (int c0 = 0; c0 < d; c0 += 256) {               // 64 lanes x float4 per pass over the chunk
    const int c = c0 + 4 * lane;
    const bool ok = c < d;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned cur = KEY_NONE;
    int run = -1;
    for (int j0 = 0; j0 < cnt; j0 += 8) {
      float4 v[8];
      unsigned kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                   // eight source rows requested before the first is added
        const int j = min(j0 + u, cnt - 1);
        kk[u] = (unsigned)__shfl((int)kv, j, 64);
        const int row = __shfl(iv, j, 64);
        v[u] = ok ? *reinterpret_cast<const float4*>(src + (long)row * ld_src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u >= cnt) break;
        if (kk[u] != cur) {                            // wave-uniform: a run ends
          if (run == 0) { if (ok) *reinterpret_cast<float4*>(pfirst + c) = acc; }
          else if (run > 0 && ok) {                    // a run inside the chunk: this wave is its row's only writer
            float* drow = dest_row(out, ld_out, nrows, extra, cur);
            if (drow) {
              float4* dst = reinterpret_cast<float4*>(drow + c);
              float4 o = *dst;
              o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
              *dst = o;
            }
          }
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
          cur = kk[u];
          ++run;
        }
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      }
    }
    if (cnt > 0 && ok) *reinterpret_cast<float4*>((run == 0 ? pfirst : plast) + c) = acc;     // the chunk's last run (may continue)
    if (c0 == 0) {
      nruns = run + 1;
      first_key = (unsigned)__shfl((int)kv, 0, 64);
      last_key = cur;
    }
  }
  if (lane == 0) meta[chunk] = ChunkMeta{cnt > 0 ? first_key : KEY_NONE, cnt > 0 ? last_key : KEY_NONE, cnt > 0 ? nruns : 0, 0};
}

// blockIdx.y = 0: the chunk's first run, 1: its last run (when the chunk holds more than one).  A run whose first entry lies in this
// chunk is a chain head: it adds the partial rows of the following chunks for as long as they continue it, in chunk order.
__global__ __launch_bounds__(64) void segsum_chain_kernel(const ChunkMeta* __restrict__ meta, const float* __restrict__ part, int nchunks,
                                                          int d, float* __restrict__ out, long ld_out, int nrows,
                                                          float* __restrict__ extra) {
  const int chunk = blockIdx.x, which = blockIdx.y, lane = threadIdx.x;
  const ChunkMeta m = meta[chunk];
  if (m.nruns == 0) return;
  unsigned key;
  if (which == 0) {
    if (chunk > 0 && meta[chunk - 1].last_key == m.first_key) return;        // continues a run that started earlier
    key = m.first_key;
  } else {
    if (m.nruns < 2) return;                                                 // the only run is handled as `first`
    key = m.last_key;
  }
  const bool open_end = which == 1 || m.nruns == 1;                          // the run reaches the end of its chunk
  int stop = chunk + 1;                                                      // chunks [chunk + 1, stop) continue it
  if (open_end)
    while (stop < nchunks && meta[stop].nruns > 0 && meta[stop].first_key == key) {
      ++stop;
      if (meta[stop - 1].nruns > 1) break;
    }
  float* dst_row = dest_row(out, ld_out, nrows, extra, key);
  if (!dst_row) return;                                                      // key outside the table
  for (int c = 4 * lane; c < d; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(part + ((long)chunk * 2 + (which == 1 ? 1 : 0)) * d + c);
    float4 acc = a;
    for (int j = chunk + 1; j < stop; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(part + ((long)j * 2) * d + c);
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    float4* dst = reinterpret_cast<float4*>(dst_row + c);
    float4 o = *dst;
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    *dst = o;
  }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }
size_t sort_temp_bytes(int n) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr,
                            (unsigned)n, 0, KEY_BITS, (hipStream_t)0);
  return bytes;
}

}  // namespace

// bytes of workspace one call on n entries of d columns needs
size_t neko_segsum_ws_bytes_impl(int n, int d) {
  if (n <= 0) return 0;
  const size_t nch = (size_t)(n + SEG_CH - 1) / SEG_CH;
  return align256((size_t)n * 4) * 3 + align256(nch * sizeof(ChunkMeta)) + align256(nch * 2 * (size_t)d * 4) + align256(sort_temp_bytes(n));
}

// The same sums for entries that arrive ALREADY SORTED by key (round 5: the host builds the packing descriptors and draws the patch
// positions, so it can sort those keys itself -- numpy's stable sort of 65536 16-bit keys is under a millisecond -- and the device-side
// radix sort with its half-dozen launches drops out): keys_sorted ascending with KEY_NONE entries last, idx_sorted the source rows,
// ties in source order.  Bit-identical to neko_segsum_rows_impl on the unsorted pairs.
size_t neko_segsum_sorted_ws_bytes_impl(int n, int d) {
  if (n <= 0) return 0;
  const size_t nch = (size_t)(n + SEG_CH - 1) / SEG_CH;
  return align256(nch * sizeof(ChunkMeta)) + align256(nch * 2 * (size_t)d * 4);
}
int neko_segsum_rows_sorted_impl(const float* src, long ld_src, const unsigned* keys_sorted, const int* idx_sorted, int n, int d, float* out,
                                 long ld_out, int nrows, float* extra, void* ws, size_t ws_bytes, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!src || !keys_sorted || !idx_sorted || !out || !ws || (d & 3) || (ld_src & 3) || (ld_out & 3)) return NEKO_ERR_ARG;
  if (ws_bytes < neko_segsum_sorted_ws_bytes_impl(n, d)) return NEKO_ERR_ARG;
  const int nch = (n + SEG_CH - 1) / SEG_CH;
  char* w = static_cast<char*>(ws);
  ChunkMeta* meta = reinterpret_cast<ChunkMeta*>(w);                w += align256((size_t)nch * sizeof(ChunkMeta));
  float* part = reinterpret_cast<float*>(w);
  hipLaunchKernelGGL(segsum_chunk_kernel, dim3(nch), dim3(64), 0, s, src, ld_src, keys_sorted, idx_sorted, n, d, out, ld_out, nrows, extra,
                     meta, part);
  NEKO_CHECK_LAUNCH();
  hipLaunchKernelGGL(segsum_chain_kernel, dim3(nch, 2), dim3(64), 0, s, meta, part, nch, d, out, ld_out, nrows, extra);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}

// out[keys[i], :] += src[i, :] in index order; keys[i] == nrows goes to `extra` (may be null if no such key), KEY_NONE entries are skipped.
// d % 4 == 0, rows 16-B aligned.  `keys` is not modified.
int neko_segsum_rows_impl(const float* src, long ld_src, const unsigned* keys, int n, int d, float* out, long ld_out, int nrows,
                          float* extra, void* ws, size_t ws_bytes, hipStream_t s) {
  if (n <= 0) return NEKO_OK;
  if (!src || !keys || !out || !ws || (d & 3) || (ld_src & 3) || (ld_out & 3)) return NEKO_ERR_ARG;
  if (ws_bytes < neko_segsum_ws_bytes_impl(n, d)) return NEKO_ERR_ARG;
  const int nch = (n + SEG_CH - 1) / SEG_CH;
  char* w = static_cast<char*>(ws);
  unsigned* keys_sorted = reinterpret_cast<unsigned*>(w);           w += align256((size_t)n * 4);
  int* idx_in = reinterpret_cast<int*>(w);                          w += align256((size_t)n * 4);
  int* idx_sorted = reinterpret_cast<int*>(w);                      w += align256((size_t)n * 4);
  ChunkMeta* meta = reinterpret_cast<ChunkMeta*>(w);                w += align256((size_t)nch * sizeof(ChunkMeta));
  float* part = reinterpret_cast<float*>(w);                        w += align256((size_t)nch * 2 * d * 4);
  size_t temp_bytes = sort_temp_bytes(n);
  hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, idx_in, n);
  NEKO_CHECK_LAUNCH();
  if (rocprim::radix_sort_pairs(w, temp_bytes, keys, keys_sorted, idx_in, idx_sorted, (unsigned)n, 0, KEY_BITS, s) != hipSuccess)
    return NEKO_ERR_LAUNCH;
  hipLaunchKernelGGL(segsum_chunk_kernel, dim3(nch), dim3(64), 0, s, src, ld_src, keys_sorted, idx_sorted, n, d, out, ld_out, nrows, extra,
                     meta, part);
  NEKO_CHECK_LAUNCH();
  hipLaunchKernelGGL(segsum_chain_kernel, dim3(nch, 2), dim3(64), 0, s, meta, part, nch, d, out, ld_out, nrows, extra);
  NEKO_CHECK_LAUNCH();
  return NEKO_OK;
}
