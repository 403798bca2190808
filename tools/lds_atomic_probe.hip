// LDS atomic throughput probe (gfx950): cycles per wave-instruction for ds_add_f32 / ds_add_rtn_u32 / ds_cmpst_rtn
// at different conflict degrees.  One workgroup of W waves per CU-sized grid; reports cycles/instr/wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void probe(float* out, unsigned long long* cyc, int conflict, int iters) {
  __shared__ float buf[16384];
  __shared__ unsigned int ibuf[4096];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ibuf[i] = 0xFFFFFFFFu;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // `conflict` lanes share one address; groups of `conflict` consecutive lanes
  int addr = (lane / conflict) + wave * 64;
  unsigned long long t0 = clock64();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    int a = (addr + it * 67) & 4095;
    if (MODE == 0) atomicAdd(&buf[a], 1.0f);
    if (MODE == 1) acc += (float)atomicAdd(&ibuf[a], 1u);
    if (MODE == 2) acc += (float)atomicCAS(&ibuf[a], 0xFFFFFFFFu, (unsigned)a);
    if (MODE == 3) buf[a] = acc + it;           // plain ds_write for reference
    if (MODE == 4) acc += buf[a];               // plain ds_read
  }
  __syncthreads();
  unsigned long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + buf[threadIdx.x];
}
int main() {
  const int blocks = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
  const char* names[5] = {"ds_add_f32", "ds_add_rtn_u32", "ds_cmpst_rtn_b32", "ds_write_b32", "ds_read_b32"};
  for (int waves : {1, 4, 12}) for (int mode = 0; mode < 5; ++mode) for (int conflict : {1, 2, 8, 64}) {
    auto launch = [&](auto k) { hipLaunchKernelGGL(k, blocks, waves * 64, 0, 0, out, cyc, conflict, iters); };
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) launch(probe<0>); if (mode == 1) launch(probe<1>); if (mode == 2) launch(probe<2>);
      if (mode == 3) launch(probe<3>); if (mode == 4) launch(probe<4>);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    printf("waves/WG=%2d %-18s conflict=%2d : %.1f cycles per wave-instr (per-WG wall %.1f cyc/iter)\n", waves, names[mode], conflict,
           avg / iters / waves, avg / iters);
  }
  return 0;
}
