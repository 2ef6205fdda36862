// Probe: are scalar stores (s_store_dwordx4 + s_dcache_wb) usable on gfx950 for wave-uniform data (lane masks)?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/sstore_probe.hip -o /tmp/sstore_probe && /tmp/sstore_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void producer(uint32_t* out, int iter) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  // four lane masks that depend on (wave, iter)
  const unsigned long long m0 = __builtin_amdgcn_ballot_w64(((lane * 7 + wave + iter) & 3) == 0);
  const unsigned long long m1 = __builtin_amdgcn_ballot_w64(((lane * 5 + wave + iter) & 7) < 3);
  u32x4 q = {(uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32)};
  const uintptr_t a = reinterpret_cast<uintptr_t>(out + (long)wave * 4);
  uint32_t* dst = reinterpret_cast<uint32_t*>(((uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                                              (uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)a));
  asm volatile("s_store_dwordx4 %0, %1, 0x0" ::"s"(q), "s"(dst) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ void consumer(const uint32_t* __restrict__ in, uint32_t* out2) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(in) + (long)__builtin_amdgcn_readfirstlane(wave) * 2;
  const unsigned long long m0 = p[0], m1 = p[1];      // uniform address: s_load
  float x = 1.0f, y;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(y) : "v"(x), "s"(m0));
  float z;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(z) : "v"(x), "s"(m1));
  out2[(long)wave * 64 + lane] = (y != 0.f ? 1u : 0u) | (z != 0.f ? 2u : 0u);
}
int main() {
  const int blocks = 4096, threads = 256, waves = blocks * threads / 64;
  uint32_t *d, *d2;
  hipMalloc(&d, waves * 16);
  hipMalloc(&d2, (size_t)waves * 64 * 4);
  std::vector<uint32_t> h(waves * 4), h2((size_t)waves * 64);
  long bad = 0, bad2 = 0;
  for (int iter = 0; iter < 20; ++iter) {
    hipLaunchKernelGGL(producer, dim3(blocks), dim3(threads), 0, 0, d, iter);
    hipLaunchKernelGGL(consumer, dim3(blocks), dim3(threads), 0, 0, d, d2);
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), d2, (size_t)waves * 64 * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < waves; ++w) {
      unsigned long long m0 = 0, m1 = 0;
      for (int l = 0; l < 64; ++l) {
        if (((l * 7 + w + iter) & 3) == 0) m0 |= 1ull << l;
        if (((l * 5 + w + iter) & 7) < 3) m1 |= 1ull << l;
      }
      if (h[w * 4] != (uint32_t)m0 || h[w * 4 + 1] != (uint32_t)(m0 >> 32) || h[w * 4 + 2] != (uint32_t)m1 || h[w * 4 + 3] != (uint32_t)(m1 >> 32)) ++bad;
      for (int l = 0; l < 64; ++l) {
        const uint32_t want = ((m0 >> l) & 1) | (((m1 >> l) & 1) << 1);
        if (h2[(size_t)w * 64 + l] != want) ++bad2;
      }
    }
  }
  printf("scalar-store probe: %d waves x 20 iterations: %ld wrong stored quads, %ld wrong consumer lanes\n", waves, bad, bad2);
  return (bad || bad2) ? 1 : 0;
}
