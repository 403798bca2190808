// Does VALU work overlap with MFMA work on one SIMD?  Each wave runs R rounds of [NM back-to-back MFMAs][NV VALU fmas];
// compare 1 and 2 waves per SIMD and an interleaved order (1 MFMA : NV/NM VALU ops).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, bool INTERLEAVE>
__global__ __launch_bounds__(256) void k(float* out, int rounds) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int r = 0; r < rounds; ++r) {
    if (!INTERLEAVE) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i & 7] = fmaf(v[i & 7], 1.0001f, 0.5f);
    } else {
      constexpr int PER = NM > 0 ? NV / (NM > 0 ? NM : 1) : 0;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PER; ++i) v[(m * PER + i) & 7] = fmaf(v[(m * PER + i) & 7], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NV, bool IL>
void run(int wgs_per_cu, const char* label) {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  const int grid = 256 * wgs_per_cu, rounds = 2000;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<NM, NV, IL>), dim3(grid), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((k<NM, NV, IL>), dim3(grid), dim3(256), 0, 0, out, rounds);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double cyc = ms * 1e-3 * 2.4e9 / rounds;  // cycles per round (at 2.4 GHz nominal)
  printf("%-34s NM=%3d NV=%4d %d wave/SIMD: %8.0f cycles/round  (MFMA alone = %d, VALU alone ~ %d)\n", label, NM, NV, wgs_per_cu,
         cyc, NM * 32 * wgs_per_cu, NV * 4 * wgs_per_cu);
  hipFree(out);
}
int main() {
  run<64, 0, false>(1, "MFMA only");
  run<0, 256, false>(1, "VALU only");
  run<64, 256, false>(1, "blocks, same wave");
  run<64, 256, true>(1, "interleaved 1:4, same wave");
  run<64, 256, false>(2, "blocks, two waves");
  run<64, 256, true>(2, "interleaved 1:4, two waves");
  run<64, 512, false>(2, "blocks, two waves");
  run<64, 512, true>(1, "interleaved 1:8, same wave");
  return 0;
}
