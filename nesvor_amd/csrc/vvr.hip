// Similarity sums for volume-to-volume rigid registration (gfx950).
//
// Replaces, for one optimisation step of the reference's `VVR` (nesvor/svort/registration.py:143-247), the K
// evaluations of  warp (rigid transform of the target's point list + F.grid_sample of the source, :233-247)  +
// loss reduction (:166-170)  that its finite-difference gradient performs one after the other (K = 1 + 2 x 6 poses:
// :150-163): here every point is read once and sampled under all K poses, and the launch returns the five moment
// sums per pose from which NCC or MSE follow.
//
// Sampling = F.grid_sample(mode="bilinear", padding_mode="zeros", align_corners=True): normalised coordinate g in
// [-1, 1] -> index (g + 1) / 2 * (size - 1); corners outside the volume contribute 0.
// Gather-bound (8 reads per point and pose, the source volume is L2-resident); sums leave the chip as
// one fp64 atomic per workgroup and moment.
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/nesvor_hip.h"

namespace {

__device__ __forceinline__ float fetch(const float* __restrict__ v, int x, int y, int z, int W, int H, int D) {
  return (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D) ? v[((size_t)z * H + y) * W + x] : 0.f;
}

template <int KT>
__global__ __launch_bounds__(256) void vvr_similarity_kernel(const float* __restrict__ src, int D, int H, int W,
                                                             const float* __restrict__ pts, const float* __restrict__ tgt,
                                                             const float* __restrict__ mats, float ux, float uy, float uz,
                                                             int64_t M, double* __restrict__ sums /* (KT,3) */,
                                                             double* __restrict__ tsums /* (2) or null */) {
  __shared__ float smat[KT * 12];
  __shared__ float red[4][KT * 3 + 2];
  for (int e = threadIdx.x; e < KT * 12; e += blockDim.x) smat[e] = mats[e];
  __syncthreads();
  float aI[KT], aII[KT], aIJ[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) aI[k] = aII[k] = aIJ[k] = 0.f;
  float aJ = 0.f, aJJ = 0.f;
  const float hx = 0.5f * (W - 1), hy = 0.5f * (H - 1), hz = 0.5f * (D - 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const float t = tgt[i];
    aJ += t; aJJ = fmaf(t, t, aJJ);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const float* m = smat + 12 * k;  // rows of [R | t], translation applied first: R (p + t)
      const float qx = px + m[3], qy = py + m[7], qz = pz + m[11];
      const float gx = (m[0] * qx + m[1] * qy + m[2] * qz) * ux, gy = (m[4] * qx + m[5] * qy + m[6] * qz) * uy,
                  gz = (m[8] * qx + m[9] * qy + m[10] * qz) * uz;
      const float fx = (gx + 1.f) * hx, fy = (gy + 1.f) * hy, fz = (gz + 1.f) * hz;
      const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
      const float wx = fx - x0f, wy = fy - y0f, wz = fz - z0f;
      // (int) of a huge or NaN coordinate is clamped by the range test in fetch()
      const int x0 = (int)fmaxf(fminf(x0f, 1e9f), -1e9f), y0 = (int)fmaxf(fminf(y0f, 1e9f), -1e9f), z0 = (int)fmaxf(fminf(z0f, 1e9f), -1e9f);
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float w = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy) * ((c & 4) ? wz : 1.f - wz);
        v = fmaf(w, fetch(src, x0 + (c & 1), y0 + ((c >> 1) & 1), z0 + (c >> 2), W, H, D), v);
      }
      aI[k] += v; aII[k] = fmaf(v, v, aII[k]); aIJ[k] = fmaf(v, t, aIJ[k]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const float s0 = wave_sum_dpp(aI[k]), s1 = wave_sum_dpp(aII[k]), s2 = wave_sum_dpp(aIJ[k]);
    if (lane == 0) { red[wave][3 * k] = s0; red[wave][3 * k + 1] = s1; red[wave][3 * k + 2] = s2; }
  }
  {
    const float s0 = wave_sum_dpp(aJ), s1 = wave_sum_dpp(aJJ);
    if (lane == 0) { red[wave][3 * KT] = s0; red[wave][3 * KT + 1] = s1; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < KT * 3 + 2; e += blockDim.x) {
    const double s = ((double)red[0][e] + (double)red[1][e]) + ((double)red[2][e] + (double)red[3][e]);
    if (e < KT * 3) atomicAdd(sums + e, s);
    else if (tsums != nullptr) atomicAdd(tsums + (e - KT * 3), s);
  }
}

}  // namespace

extern "C" int nesvor_vvr_similarity(const float* source, int D, int H, int W, const float* points, const float* target,
                                     const float* mats, const float* to_unit_xyz, int64_t M, int K, double* sums,
                                     double* target_sums, void* stream) {
  if (M <= 0 || K <= 0) return 0;
  if (D < 1 || H < 1 || W < 1) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 3 * (size_t)K, st);
  if (e != hipSuccess) return (int)e;
  if (target_sums != nullptr) {
    e = hipMemsetAsync(target_sums, 0, sizeof(double) * 2, st);
    if (e != hipSuccess) return (int)e;
  }
  int64_t blocks = (M + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const float ux = to_unit_xyz[0], uy = to_unit_xyz[1], uz = to_unit_xyz[2];
  int k = 0;
  bool first = true;
  while (k < K) {  // 13 poses per pass (one finite-difference gradient), single poses otherwise
    double* ts = first ? target_sums : nullptr;
    if (K - k >= 13) {
      hipLaunchKernelGGL((vvr_similarity_kernel<13>), dim3((unsigned)blocks), dim3(256), 0, st, source, D, H, W, points, target,
                         mats + 12 * (size_t)k, ux, uy, uz, M, sums + 3 * (size_t)k, ts);
      k += 13;
    } else {
      hipLaunchKernelGGL((vvr_similarity_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, source, D, H, W, points, target,
                         mats + 12 * (size_t)k, ux, uy, uz, M, sums + 3 * (size_t)k, ts);
      k += 1;
    }
    first = false;
  }
  return (int)hipGetLastError();
}
