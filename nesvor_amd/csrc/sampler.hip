// PSF sampling + rigid transform of slice pixels for gfx950 (forward and backward).
//
// Replaces the head of NeSVoR.forward (nesvor/nesvor/models.py:267-278) and
// mat_transform_points (nesvor/transform/transform.py:259-271, trans_first = True):
//     x[b,s] = R_k ( xyz[b] + noise[b,s] * sigma_k + t_k ),   k = slice_idx[b]
//     u[b,s] = (x[b,s] - bb0) / (bb1 - bb0)                    (INR.forward, models.py:143)
// In the reference this is ~8 elementwise launches plus a batched 3x3 matmul over N = B*S points
// that rocBLAS executes in ~7 ms (and two more in backward).  Here: one streaming launch each way.
// One wave per pixel: lanes stride over the S samples (coalesced 12-byte triples), the pose of the
// pixel's slice lives in SGPR-uniform registers.
// Backward: d mat_k = sum_s [ dx (p + t)^T | R^T dx ] reduced across the wave (DPP) and written per
// pixel (B,3,4); the caller index_adds pixels into slices (n is a few hundred).
#include <hip/hip_runtime.h>
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void psf_transform_fwd(const float* __restrict__ mat, const int64_t* __restrict__ slice_idx,
                                                         const float* __restrict__ xyz, const float* __restrict__ sigma,
                                                         const float* __restrict__ noise, const float* __restrict__ bb,
                                                         float* __restrict__ x, float* __restrict__ u, int B, int S) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t k = slice_idx[b];
  const float* m = mat + k * 12;
  const float r00 = m[0], r01 = m[1], r02 = m[2], t0 = m[3];
  const float r10 = m[4], r11 = m[5], r12 = m[6], t1 = m[7];
  const float r20 = m[8], r21 = m[9], r22 = m[10], t2 = m[11];
  const float px = xyz[3 * b], py = xyz[3 * b + 1], pz = xyz[3 * b + 2];
  const float sx = sigma[3 * k], sy = sigma[3 * k + 1], sz = sigma[3 * k + 2];
  const float b0x = bb[0], b0y = bb[1], b0z = bb[2];
  const float ex = bb[3] - b0x, ey = bb[4] - b0y, ez = bb[5] - b0z;
  for (int s = lane; s < S; s += 64) {
    const size_t o = ((size_t)b * S + s) * 3;
    // same association as the reference: (xyz + noise * sigma) + T, then row . vector left to right
    const float qx = (px + noise[o] * sx) + t0, qy = (py + noise[o + 1] * sy) + t1, qz = (pz + noise[o + 2] * sz) + t2;
    const float X = r00 * qx + r01 * qy + r02 * qz;
    const float Y = r10 * qx + r11 * qy + r12 * qz;
    const float Z = r20 * qx + r21 * qy + r22 * qz;
    x[o] = X; x[o + 1] = Y; x[o + 2] = Z;
    if (u != nullptr) { u[o] = (X - b0x) / ex; u[o + 1] = (Y - b0y) / ey; u[o + 2] = (Z - b0z) / ez; }
  }
}

__global__ __launch_bounds__(256) void psf_transform_bwd(const float* __restrict__ mat, const int64_t* __restrict__ slice_idx,
                                                         const float* __restrict__ xyz, const float* __restrict__ sigma,
                                                         const float* __restrict__ noise, const float* __restrict__ bb,
                                                         const float* __restrict__ dx, const float* __restrict__ du,
                                                         float* __restrict__ dmat, int B, int S) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t k = slice_idx[b];
  const float* m = mat + k * 12;
  const float r00 = m[0], r01 = m[1], r02 = m[2], t0 = m[3];
  const float r10 = m[4], r11 = m[5], r12 = m[6], t1 = m[7];
  const float r20 = m[8], r21 = m[9], r22 = m[10], t2 = m[11];
  const float px = xyz[3 * b], py = xyz[3 * b + 1], pz = xyz[3 * b + 2];
  const float sx = sigma[3 * k], sy = sigma[3 * k + 1], sz = sigma[3 * k + 2];
  const float ex = bb[3] - bb[0], ey = bb[4] - bb[1], ez = bb[5] - bb[2];
  float g[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = 0.f;
  for (int s = lane; s < S; s += 64) {
    const size_t o = ((size_t)b * S + s) * 3;
    float gx = dx != nullptr ? dx[o] : 0.f, gy = dx != nullptr ? dx[o + 1] : 0.f, gz = dx != nullptr ? dx[o + 2] : 0.f;
    if (du != nullptr) { gx += du[o] / ex; gy += du[o + 1] / ey; gz += du[o + 2] / ez; }
    const float qx = (px + noise[o] * sx) + t0, qy = (py + noise[o + 1] * sy) + t1, qz = (pz + noise[o + 2] * sz) + t2;
    g[0] += gx * qx; g[1] += gx * qy; g[2] += gx * qz;
    g[4] += gy * qx; g[5] += gy * qy; g[6] += gy * qz;
    g[8] += gz * qx; g[9] += gz * qy; g[10] += gz * qz;
    g[3] += r00 * gx + r10 * gy + r20 * gz;   // d t = R^T g
    g[7] += r01 * gx + r11 * gy + r21 * gz;
    g[11] += r02 * gx + r12 * gy + r22 * gz;
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = wave_sum_dpp(g[i]);
  if (lane < 12) {
    float v = g[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) v = lane == i ? g[i] : v;
    dmat[(size_t)b * 12 + lane] = v;
  }
}

}  // namespace

extern "C" int nesvor_psf_transform_forward(const float* mat, const int64_t* slice_idx, const float* xyz,
                                            const float* sigma, const float* noise, const float* bb, float* x,
                                            float* u, int B, int S, void* stream) {
  if (B <= 0 || S <= 0) return 0;
  hipLaunchKernelGGL(psf_transform_fwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     noise, bb, x, u, B, S);
  return (int)hipGetLastError();
}

extern "C" int nesvor_psf_transform_backward(const float* mat, const int64_t* slice_idx, const float* xyz,
                                             const float* sigma, const float* noise, const float* bb, const float* dx,
                                             const float* du, float* dmat, int B, int S, void* stream) {
  if (B <= 0 || S <= 0) return 0;
  hipLaunchKernelGGL(psf_transform_bwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     noise, bb, dx, du, dmat, B, S);
  return (int)hipGetLastError();
}
