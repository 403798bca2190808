// Returning global atomics on hot counters: agent scope (memory side) vs workgroup scope on XCC-private counters
// (executed in the issuing XCC's L2).  Checks that no update is lost and prints the time per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_scope_probe.hip -o /tmp/atomic_scope_probe && /tmp/atomic_scope_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }

template <int SCOPE_WG>
__global__ __launch_bounds__(256) void probe(uint32_t* counters, uint32_t* sink, uint32_t* xcc_seen, int rounds, int n_counters) {
  const uint32_t x = SCOPE_WG ? xcc_id() : 0u;
  if (threadIdx.x == 0) atomicOr(&xcc_seen[blockIdx.x & 1023], 1u << xcc_id());
  uint32_t acc = 0;
  for (int r = 0; r < rounds; ++r) {
    if ((int)threadIdx.x < n_counters) {
      uint32_t* c = counters + x * 4096 + (threadIdx.x + r * 131) % n_counters + (r % 16) * n_counters;
      if (SCOPE_WG) acc += __hip_atomic_fetch_add(c, 1u + (acc & 1u) * 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else acc += __hip_atomic_fetch_add(c, 1u + (acc & 1u) * 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}

int main() {
  const int n_wg = 4096, rounds = 16, n_counters = 128;
  uint32_t *counters, *sink, *seen;
  hipMalloc(&counters, 16 * 4096 * 4); hipMalloc(&sink, 4); hipMalloc(&seen, 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipMemset(counters, 0, 16 * 4096 * 4); hipMemset(seen, 0, 4096);
      hipEventRecord(a);
      if (mode) hipLaunchKernelGGL(probe<1>, dim3(n_wg), dim3(256), 0, 0, counters, sink, seen, rounds, n_counters);
      else hipLaunchKernelGGL(probe<0>, dim3(n_wg), dim3(256), 0, 0, counters, sink, seen, rounds, n_counters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    std::vector<uint32_t> h(16 * 4096), s(1024);
    hipMemcpy(h.data(), counters, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), seen, 4096, hipMemcpyDeviceToHost);
    unsigned long long total = 0; for (auto v : h) total += v;
    uint32_t mask = 0; int mixed = 0; for (auto v : s) { mask |= v; if (v & (v - 1)) ++mixed; }
    printf("%s scope: %.3f ms per launch (%d WGs x %d rounds x %d returning atomics), sum %llu expected %llu, xcc mask 0x%x, blockIdx%%1024 classes seen on >1 xcc: %d\n",
           mode ? "workgroup (XCC-private counters)" : "agent", best, n_wg, rounds, n_counters, total,
           (unsigned long long)n_wg * rounds * n_counters, mask, mixed);
  }
  return 0;
}
