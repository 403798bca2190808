// Sustained fp32 MFMA rate of the chip: every SIMD issues independent v_mfma_f32_16x16x4_f32 chains.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
void run(int wgs_per_cu, const char* label) {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  const int grid = 256 * wgs_per_cu, iters = 4000;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double flops = (double)grid * 4 /*waves*/ * iters * 8.0 * CHAINS * 2048.0;
  printf("%s: chains %d, %d WG/CU: %.3f ms, %.1f TFLOP/s\n", label, CHAINS, wgs_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(1, "1 wave/SIMD");
  run<2>(1, "1 wave/SIMD");
  run<4>(1, "1 wave/SIMD");
  run<4>(2, "2 waves/SIMD");
  run<4>(4, "4 waves/SIMD");
  return 0;
}
