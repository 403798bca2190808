// Verifies the f32 MFMA operand/accumulator lane maps on the actual gfx950 before the fused MLP relies on them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k16(const float* A, const float* B, float* D) {  // A 16x4 row-major, B 4x16 row-major
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k32(const float* A, const float* B, float* D) {  // A 32x2, B 2x32
  int l = threadIdx.x;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  auto run = [](int M, int K, bool big) {
    std::vector<float> A(M * K), B(K * M), D(M * M), R(M * M, 0.f);
    for (int i = 0; i < M * K; ++i) { A[i] = (float)((i * 7 + 3) % 13) - 6; B[i] = (float)((i * 5 + 1) % 11) - 5; }
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) for (int k = 0; k < K; ++k) R[i * M + j] += A[i * K + k] * B[k * M + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    if (big) hipLaunchKernelGGL(k32, 1, 64, 0, 0, dA, dB, dD); else hipLaunchKernelGGL(k16, 1, 64, 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < M * M; ++i) bad += D[i] != R[i];
    printf("mfma %dx%dx%d layout: %s (%d mismatches)\n", M, M, K, bad ? "MISMATCH" : "ok", bad);
  };
  run(16, 4, false); run(32, 2, true);
  return 0;
}
