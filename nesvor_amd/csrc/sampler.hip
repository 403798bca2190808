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

// PSF noise without a noise tensor: a counter-based generator (Philox4x32-10, the construction torch's device generator
// uses too) keyed by a 64-bit seed; counter = (sample index, stream offset).  One call yields four 32-bit words -> two
// Box-Muller pairs -> the sample's three N(0,1) draws (the fourth is dropped).  The forward and the backward of a step
// evaluate the same function of (seed, offset, sample), so nothing is stored; `nesvor_psf_noise` materialises the draws
// (tests, debugging).  The reference draws torch.randn (models.py:270): any N(0,1) stream is the same model.
struct NoiseSource {
  const float* noise;      // explicit draws (B, S, 3), or null:
  uint64_t seed, offset;   // counter-based draws
};
__device__ __forceinline__ void philox_normal3(uint64_t seed, uint64_t offset, uint64_t index, float (&n)[3]) {
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  // 24-bit uniforms: u in (0, 1] for the logarithm, v in [0, 1) for the angle
  const float u1 = (float)((c0 >> 8) + 1u) * 5.9604644775390625e-8f, v1 = (float)(c1 >> 8) * 5.9604644775390625e-8f;
  const float u2 = (float)((c2 >> 8) + 1u) * 5.9604644775390625e-8f, v2 = (float)(c3 >> 8) * 5.9604644775390625e-8f;
  const float r1 = sqrtf(-2.f * logf(u1)), r2 = sqrtf(-2.f * logf(u2));
  float s1, co1, co2;
  sincospif(2.f * v1, &s1, &co1);
  co2 = cospif(2.f * v2);
  n[0] = r1 * co1; n[1] = r1 * s1; n[2] = r2 * co2;
}
__device__ __forceinline__ void draw3(const NoiseSource& src, size_t sample, float (&n)[3]) {
  if (src.noise != nullptr) { n[0] = src.noise[3 * sample]; n[1] = src.noise[3 * sample + 1]; n[2] = src.noise[3 * sample + 2]; }
  else philox_normal3(src.seed, src.offset, sample, n);
}

__global__ __launch_bounds__(256) void psf_noise_kernel(NoiseSource src, float* __restrict__ out, int64_t n_samples) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_samples) return;
  float n[3];
  philox_normal3(src.seed, src.offset, (uint64_t)i, n);
  out[3 * i] = n[0]; out[3 * i + 1] = n[1]; out[3 * i + 2] = n[2];
}

__global__ __launch_bounds__(256) void psf_transform_fwd(const float* __restrict__ mat, const int64_t* __restrict__ slice_idx,
                                                         const float* __restrict__ xyz, const float* __restrict__ sigma,
                                                         const NoiseSource noise, const float* __restrict__ bb,
                                                         float* __restrict__ x, float* __restrict__ u, int B, int S,
                                                         const float* __restrict__ emb = nullptr, float* __restrict__ se = nullptr, int ks = 0) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t k = slice_idx[b];
  // optional: the pixel's row of the slice embedding (the per-pixel input of sigma_net / b_net), gathered by the wave that
  // looks the slice up anyway - one launch less per training iteration
  if (se != nullptr)
    for (int c = lane; c < ks; c += 64) se[(size_t)b * ks + c] = emb[(size_t)k * ks + c];
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
    float xi[3];
    draw3(noise, (size_t)b * S + s, xi);
    const float qx = (px + xi[0] * sx) + t0, qy = (py + xi[1] * sy) + t1, qz = (pz + xi[2] * sz) + t2;
    const float X = r00 * qx + r01 * qy + r02 * qz;
    const float Y = r10 * qx + r11 * qy + r12 * qz;
    const float Z = r20 * qx + r21 * qy + r22 * qz;
    if (x != nullptr) { x[o] = X; x[o + 1] = Y; x[o + 2] = Z; }
    if (u != nullptr) { u[o] = (X - b0x) / ex; u[o + 1] = (Y - b0y) / ey; u[o + 2] = (Z - b0z) / ez; }
  }
}

__global__ __launch_bounds__(256) void psf_transform_bwd(const float* __restrict__ mat, const int64_t* __restrict__ slice_idx,
                                                         const float* __restrict__ xyz, const float* __restrict__ sigma,
                                                         const NoiseSource noise, const float* __restrict__ bb,
                                                         const float* __restrict__ dx, const float* __restrict__ du,
                                                         float* __restrict__ dmat, int B, int S,
                                                         float* __restrict__ dmat_slice = nullptr) {  // optional (n,3,4): the pixel's gradient ADDED to its slice's
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
    float xi[3];
    draw3(noise, (size_t)b * S + s, xi);
    const float qx = (px + xi[0] * sx) + t0, qy = (py + xi[1] * sy) + t1, qz = (pz + xi[2] * sz) + t2;
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
    if (dmat != nullptr) dmat[(size_t)b * 12 + lane] = v;
    if (dmat_slice != nullptr) atomicAdd(dmat_slice + k * 12 + lane, v);
  }
}

}  // namespace

extern "C" int nesvor_psf_transform_forward(const float* mat, const int64_t* slice_idx, const float* xyz,
                                            const float* sigma, const float* noise, const float* bb, float* x,
                                            float* u, int B, int S, void* stream) {
  if (B <= 0 || S <= 0) return 0;
  if (noise == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(psf_transform_fwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     NoiseSource{noise, 0, 0}, bb, x, u, B, S);
  return (int)hipGetLastError();
}

extern "C" int nesvor_psf_transform_backward(const float* mat, const int64_t* slice_idx, const float* xyz,
                                             const float* sigma, const float* noise, const float* bb, const float* dx,
                                             const float* du, float* dmat, int B, int S, void* stream) {
  if (B <= 0 || S <= 0) return 0;
  if (noise == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(psf_transform_bwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     NoiseSource{noise, 0, 0}, bb, dx, du, dmat, B, S);
  return (int)hipGetLastError();
}

// the same two operators with the PSF noise drawn inside the kernels (see philox_normal3)
extern "C" int nesvor_psf_transform_forward_rng(const float* mat, const int64_t* slice_idx, const float* xyz,
                                                const float* sigma, uint64_t seed, uint64_t offset, const float* bb,
                                                float* x, float* u, int B, int S, void* stream) {
  if (B <= 0 || S <= 0) return 0;
  hipLaunchKernelGGL(psf_transform_fwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     NoiseSource{nullptr, seed, offset}, bb, x, u, B, S);
  return (int)hipGetLastError();
}

// nesvor_psf_transform_forward_rng that also gathers se[b, :] = embedding[slice_idx[b], :] (B, ks)
extern "C" int nesvor_psf_transform_forward_rng_gather(const float* mat, const int64_t* slice_idx, const float* xyz,
                                                       const float* sigma, uint64_t seed, uint64_t offset, const float* bb,
                                                       float* x, float* u, int B, int S, const float* embedding, float* se, int ks,
                                                       void* stream) {
  if (B <= 0 || S <= 0) return 0;
  if (ks > 0 && (embedding == nullptr || se == nullptr)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(psf_transform_fwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     NoiseSource{nullptr, seed, offset}, bb, x, u, B, S, embedding, ks > 0 ? se : nullptr, ks);
  return (int)hipGetLastError();
}

extern "C" int nesvor_psf_transform_backward_rng(const float* mat, const int64_t* slice_idx, const float* xyz,
                                                 const float* sigma, uint64_t seed, uint64_t offset, const float* bb,
                                                 const float* dx, const float* du, float* dmat, int B, int S, void* stream) {
  if (B <= 0 || S <= 0) return 0;
  hipLaunchKernelGGL(psf_transform_bwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     NoiseSource{nullptr, seed, offset}, bb, dx, du, dmat, B, S);
  return (int)hipGetLastError();
}

// ... that ALSO (or only: dpix may be NULL) adds every pixel's gradient to its slice's row of dmat_slice (n,3,4) - the index_add of
// the reference's autograd (transform.py:274-280 through slice_idx), without a second launch over dpix
extern "C" int nesvor_psf_transform_backward_rng_slices(const float* mat, const int64_t* slice_idx, const float* xyz, const float* sigma,
                                                        uint64_t seed, uint64_t offset, const float* bb, const float* dx, const float* du,
                                                        float* dpix, float* dmat_slice, int B, int S, void* stream) {
  if (B <= 0 || S <= 0) return 0;
  hipLaunchKernelGGL(psf_transform_bwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, mat, slice_idx, xyz, sigma,
                     NoiseSource{nullptr, seed, offset}, bb, dx, du, dpix, B, S, dmat_slice);
  return (int)hipGetLastError();
}

extern "C" int nesvor_psf_noise(uint64_t seed, uint64_t offset, float* out, int64_t n_samples, void* stream) {
  if (n_samples <= 0) return 0;
  hipLaunchKernelGGL(psf_noise_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     NoiseSource{nullptr, seed, offset}, out, n_samples);
  return (int)hipGetLastError();
}
