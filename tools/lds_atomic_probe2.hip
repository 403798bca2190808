// LDS integer atomic throughput on gfx950: returnless ds_add_u32 vs ds_add_u64 vs returning CAS, as a function of the
// number of ACTIVE lanes and of same-address conflicts (what the hash-grid merge table issues).
// hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe2.hip -o /tmp/p2 && /tmp/p2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void probe(float* out, unsigned long long* cyc, int active, int conflict, int iters) {
  __shared__ unsigned long long b64[4096];
  __shared__ unsigned int b32[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) { b64[i] = 0; b32[i] = 0xFFFFFFFFu; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool on = lane < active;
  int addr = (lane / conflict) * 7 + wave * 97;
  unsigned long long t0 = clock64();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int a = (addr + it * 67) & 4095;
    if (on) {
      if (MODE == 0) atomicAdd(&b32[a], 3u);                        // ds_add_u32 (returnless)
      if (MODE == 1) atomicAdd(&b64[a], 0x100000003ull);            // ds_add_u64 (returnless)
      if (MODE == 2) acc += (float)atomicCAS(&b32[a], 0xFFFFFFFFu, (unsigned)a);  // ds_cmpst_rtn_b32
      if (MODE == 3) acc += (float)atomicAdd(&b32[a], 1u);          // ds_add_rtn_u32
    }
  }
  __syncthreads();
  unsigned long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)b64[threadIdx.x] + (float)b32[threadIdx.x];
}
int main() {
  const int blocks = 256, iters = 2000, waves = 16;  // 16 waves per CU, like four aggregation workgroups
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
  const char* names[4] = {"ds_add_u32", "ds_add_u64", "ds_cmpst_rtn_b32", "ds_add_rtn_u32"};
  for (int mode = 0; mode < 4; ++mode) for (int active : {64, 16, 4}) for (int conflict : {1, 4, 64}) {
    if (conflict > active && conflict != 64) continue;
    auto launch = [&](auto k) { hipLaunchKernelGGL(k, blocks, waves * 64, 0, 0, out, cyc, active, conflict, iters); };
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) launch(probe<0>); if (mode == 1) launch(probe<1>); if (mode == 2) launch(probe<2>); if (mode == 3) launch(probe<3>);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("%-18s active=%2d conflict=%2d : %.1f CU-cycles per wave-instruction\n", names[mode], active, conflict, avg / iters / waves);
  }
  return 0;
}
