// Slice acquisition forward operator A for gfx950: volume -> PSF-blurred slices.
//
// Behavioural spec: `slice_acquisition_forward_cuda_kernel`
// (nesvor/slice_acquisition/slice_acq_cuda_kernel.cu:17-171).  Gather-only.
// One thread per slice pixel.  The PSF is compacted once per workgroup into an
// LDS list of its non-zero taps (offset + weight, original order preserved so
// the fp32 accumulation order matches the reference), which removes the
// zero-tap branches from the inner loop; the per-slice rotation is read through
// the scalar path.  Volume reads are 8 scattered dwords per tap and stay in
// L2/Infinity Cache for phantom-sized volumes (128^3 fp32 = 8 MB).
#include <hip/hip_runtime.h>
#include "common.h"

namespace {

constexpr int kMaxTaps = 1024;

struct Tap { float x, y, z, w; };

template <bool INTERP_PSF>
__global__ __launch_bounds__(256) void slice_acq_fwd(
    const float* __restrict__ transforms, const float* __restrict__ vol, const uint8_t* __restrict__ vol_mask,
    const uint8_t* __restrict__ slices_mask, const float* __restrict__ psf, float* __restrict__ slices,
    float* __restrict__ slices_weight, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
    float res_slice) {
  __shared__ Tap taps[kMaxTaps];
  __shared__ int n_taps;
  if (threadIdx.x == 0) {
    int cnt = 0, ip = 0;
    for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
      for (int iy = -h_p / 2; iy < (h_p + 1) / 2; ++iy)
        for (int ix = -w_p / 2; ix < (w_p + 1) / 2; ++ix, ++ip) {
          float pv = psf[ip];
          if (pv != 0.f && cnt < kMaxTaps) taps[cnt++] = Tap{(float)ix, (float)iy, (float)iz, pv};
        }
    n_taps = cnt;
  }
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * h * w) return;
  if (slices_mask != nullptr && !slices_mask[idx]) return;
  const int ix = idx % w, iy = (idx / w) % h, in = idx / ((int64_t)h * w);
  const float* t = transforms + (size_t)in * 12;
  const float r11 = t[0], r12 = t[1], r13 = t[2], r21 = t[4], r22 = t[5], r23 = t[6], r31 = t[8], r32 = t[9], r33 = t[10];
  // the reference evaluates this affine map in double and rounds once (.cu:46-47)
  const float px = (float)((ix - (w - 1) / 2.) * (double)res_slice + (double)t[3]);
  const float py = (float)((iy - (h - 1) / 2.) * (double)res_slice + (double)t[7]);
  const float pz = t[11];
  const float xc = r11 * px + r12 * py + r13 * pz + (W - 1) / 2.f;
  const float yc = r21 * px + r22 * py + r23 * pz + (H - 1) / 2.f;
  const float zc = r31 * px + r32 * py + r33 * pz + (D - 1) / 2.f;
  const int Sy = W, Sz = H * W;
  float val = 0.f, wsum = 0.f;
  const int nt = n_taps;
  for (int k = 0; k < nt; ++k) {
    const Tap tp = taps[k];
    const float x = xc + r11 * tp.x + r12 * tp.y + r13 * tp.z;
    const float y = yc + r21 * tp.x + r22 * tp.y + r23 * tp.z;
    const float z = zc + r31 * tp.x + r32 * tp.y + r33 * tp.z;
    if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
    if (INTERP_PSF) {
      // nearest voxel, PSF re-interpolated at the voxel's offset from the centre
      const int xr = (int)floorf(x + 0.5f), yr = (int)floorf(y + 0.5f), zr = (int)floorf(z + 0.5f);
      const int iv = zr * Sz + yr * Sy + xr;
      if (vol_mask != nullptr && !vol_mask[iv]) continue;
      const float dx = xr - xc, dy = yr - yc, dz = zr - zc;
      const float xp = r11 * dx + r21 * dy + r31 * dz + (w_p - 1) / 2.f;
      const float yp = r12 * dx + r22 * dy + r32 * dz + (h_p - 1) / 2.f;
      const float zp = r13 * dx + r23 * dy + r33 * dz + (d_p - 1) / 2.f;
      if (xp < 0 || yp < 0 || zp < 0 || xp >= w_p - 1 || yp >= h_p - 1 || zp >= d_p - 1) continue;
      const int xf = (int)floorf(xp), yf = (int)floorf(yp), zf = (int)floorf(zp);
      const float wx = xp - xf, wy = yp - yf, wz = zp - zf;
      const float* p0 = psf + (zf * h_p + yf) * w_p + xf;
      const int py_ = w_p, pz_ = w_p * h_p;
      float pw = 0.f;
      pw += (1 - wx) * (1 - wy) * (1 - wz) * p0[0];
      pw += wx * (1 - wy) * (1 - wz) * p0[1];
      pw += (1 - wx) * wy * (1 - wz) * p0[py_];
      pw += (1 - wx) * (1 - wy) * wz * p0[pz_];
      pw += wx * wy * (1 - wz) * p0[1 + py_];
      pw += wx * (1 - wy) * wz * p0[1 + pz_];
      pw += (1 - wx) * wy * wz * p0[py_ + pz_];
      pw += wx * wy * wz * p0[1 + py_ + pz_];
      val += pw * vol[iv];
      wsum += pw;
    } else {
      const int xf = (int)floorf(x), yf = (int)floorf(y), zf = (int)floorf(z);
      const float wx = x - xf, wy = y - yf, wz = z - zf;
      const int iv = zf * Sz + yf * Sy + xf;
      // corner order as the reference accumulates: 000,100,010,001,110,101,011,111
      const int off[8] = {0, 1, Sy, Sz, 1 + Sy, 1 + Sz, Sy + Sz, 1 + Sy + Sz};
      const float cw[8] = {(1 - wx) * (1 - wy) * (1 - wz), wx * (1 - wy) * (1 - wz), (1 - wx) * wy * (1 - wz),
                           (1 - wx) * (1 - wy) * wz,       wx * wy * (1 - wz),       wx * (1 - wy) * wz,
                           (1 - wx) * wy * wz,             wx * wy * wz};
      float v8[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v8[c] = vol[iv + off[c]];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (vol_mask == nullptr || vol_mask[iv + off[c]]) {
          const float pw = cw[c] * tp.w;
          val += pw * v8[c];
          wsum += pw;
        }
      }
    }
  }
  if (wsum > 0) {
    slices[idx] = val / wsum;
    if (slices_weight != nullptr) slices_weight[idx] = wsum;
  }
}

}  // namespace

extern "C" int nesvor_slice_acq_forward(const float* transforms, const float* vol, const uint8_t* vol_mask,
                                        const uint8_t* slices_mask, const float* psf, float* slices,
                                        float* slices_weight, int D, int H, int W, int d_p, int h_p, int w_p, int n,
                                        int h, int w, float res_slice, int interp_psf, void* stream) {
  const int64_t total = (int64_t)n * h * w;
  if (total <= 0) return 0;
  if ((int64_t)d_p * h_p * w_p > kMaxTaps) return (int)hipErrorInvalidValue;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (interp_psf)
    hipLaunchKernelGGL(slice_acq_fwd<true>, grid, block, 0, (hipStream_t)stream, transforms, vol, vol_mask, slices_mask,
                       psf, slices, slices_weight, D, H, W, d_p, h_p, w_p, n, h, w, res_slice);
  else
    hipLaunchKernelGGL(slice_acq_fwd<false>, grid, block, 0, (hipStream_t)stream, transforms, vol, vol_mask, slices_mask,
                       psf, slices, slices_weight, D, H, W, d_p, h_p, w_p, n, h, w, res_slice);
  return (int)hipGetLastError();
}
