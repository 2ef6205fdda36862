// Round 5 probe: does the per-CU lock of gemm_glds.hip (HW_REG_HW_ID / HW_REG_XCC_ID -> lock index) give one lock per CU?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/cu_lock_probe.hip -o /tmp/cu_lock_probe && /tmp/cu_lock_probe
// 2048 blocks of 256 threads with 72 KB of LDS (two per CU): each takes its CU's lock, holds it ~5 us, releases it.  The host counts the
// distinct lock indices, the blocks per index and the overlapping hold intervals inside an index (must be 0) and across indices.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ unsigned g_lock[4096 * 16];

__global__ __launch_bounds__(256) void probe(unsigned long long* rec, int use_lock, int hold_10ns) {
  extern __shared__ char smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
  const unsigned key = (((hw >> 8) & 0xffu) | ((xcc & 0xfu) << 8)) & 4095u;
  unsigned* lock = &g_lock[key * 16];
  unsigned long long t_arrive = __builtin_amdgcn_s_memrealtime();
  if (use_lock) {
    if (threadIdx.x == 0) while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(16);
    __syncthreads();
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)hold_10ns) __builtin_amdgcn_s_sleep(8);
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  if (use_lock && threadIdx.x == 0) atomicExch(lock, 0u);
  if (threadIdx.x == 0) {
    rec[blockIdx.x * 4 + 0] = key | ((unsigned long long)hw << 32);
    rec[blockIdx.x * 4 + 1] = t0;
    rec[blockIdx.x * 4 + 2] = t1;
    rec[blockIdx.x * 4 + 3] = t_arrive;
  }
  if (threadIdx.x == 9999) smem[0] = 1;
}

int main() {
  const int nblk = 2048;
  unsigned long long* d;
  CK(hipMalloc(&d, nblk * 4 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
  for (int use_lock = 0; use_lock < 2; ++use_lock) {
    CK(hipMemset(d, 0, nblk * 4 * 8));
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 72 * 1024, 0, d, use_lock, 500);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nblk * 4);
    CK(hipMemcpy(h.data(), d, nblk * 4 * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> by;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nblk; ++b) {
      by[(unsigned)(h[b * 4] & 0xffffffffu)].push_back({h[b * 4 + 1], h[b * 4 + 2]});
      tmin = std::min(tmin, h[b * 4 + 3]);
      tmax = std::max(tmax, h[b * 4 + 2]);
    }
    int overlaps = 0, maxper = 0, minper = 1 << 30;
    for (auto& kv : by) {
      auto& v = kv.second;
      std::sort(v.begin(), v.end());
      for (size_t i = 1; i < v.size(); ++i) if (v[i].first < v[i - 1].second) ++overlaps;
      maxper = std::max(maxper, (int)v.size());
      minper = std::min(minper, (int)v.size());
    }
    printf("lock %d: %zu distinct lock indices, blocks per index %d..%d, overlapping holds inside an index %d, wall %.1f us\n", use_lock,
           by.size(), minper, maxper, overlaps, (tmax - tmin) / 100.0);
    if (use_lock == 0) {
      printf("  example HW_ID words:");
      for (int b = 0; b < 8; ++b) printf(" %08llx(key %llu)", h[b * 4] >> 32, h[b * 4] & 0xffffffffu);
      printf("\n");
    }
  }
  return 0;
}
