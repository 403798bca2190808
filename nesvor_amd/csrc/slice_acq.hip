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

// LDS_TAPS: the PSF's non-zero taps fit the LDS list (every PSF get_PSF builds for clinical geometries: (9,5,5) = 153
// taps for 1.5 x 1.5 x 3 mm slices on a 1 mm grid); larger PSFs (thick slices on a fine grid, e.g. 6 mm on 0.5 mm = 1813
// taps) walk the dense PSF array in global memory instead - same arithmetic, same order.
template <bool INTERP_PSF, bool LDS_TAPS>
__global__ __launch_bounds__(256) void slice_acq_fwd(
    const float* __restrict__ transforms, const float* __restrict__ vol, const uint8_t* __restrict__ vol_mask,
    const uint8_t* __restrict__ slices_mask, const float* __restrict__ psf, float* __restrict__ slices,
    float* __restrict__ slices_weight, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
    float res_slice) {
  __shared__ Tap taps[LDS_TAPS ? kMaxTaps : 1];
  __shared__ int n_taps;
  if constexpr (LDS_TAPS) {
    if (threadIdx.x == 0) {
      int cnt = 0, ip = 0;
      for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
        for (int iy = -h_p / 2; iy < (h_p + 1) / 2; ++iy)
          for (int ix = -w_p / 2; ix < (w_p + 1) / 2; ++ix, ++ip) {
            float pv = psf[ip];
            if (pv != 0.f) taps[cnt++] = Tap{(float)ix, (float)iy, (float)iz, pv};  // the host checked d_p h_p w_p <= kMaxTaps
          }
      n_taps = cnt;
    }
    __syncthreads();
  }
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
  const int nt = LDS_TAPS ? n_taps : d_p * h_p * w_p;
  const int tx0 = -w_p / 2, ty0 = -h_p / 2, tz0 = -d_p / 2;
  for (int k = 0; k < nt; ++k) {
    Tap tp;
    if constexpr (LDS_TAPS) {
      tp = taps[k];
    } else {
      const float pv = psf[k];
      if (pv == 0.f) continue;
      const int kz = k / (h_p * w_p), kr = k - kz * (h_p * w_p), ky = kr / w_p;
      tp = Tap{(float)(tx0 + kr - ky * w_p), (float)(ty0 + ky), (float)(tz0 + kz), pv};
    }
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

// ------------------------------------------------------------------ adjoint A^T and backward of A
// The reference scatters every (pixel, PSF tap) into 8 voxels with atomicAdd
// (slice_acq_cuda_kernel.cu:173-470 backward, :472-670 adjoint).  On MI355X global atomics execute
// memory-side (~16 G/s): one adjoint of 3 stacks of 77 x 151^2 pixels would be 13 G atomics.  Both
// operators are therefore evaluated as a GATHER OVER VOXELS: a voxel collects, slice by slice, the
// samples p = R (q_pixel + tap) + c0 that have it as one of their 8 trilinear corners
// (-1 <= p - v < 1 per axis).  In the slice frame that is the small box |q + tap - R^T (v - c0)| < sqrt 3,
// so per slice only ~6 x 6 pixels x <= 4^3 taps are visited.  No atomics, each voxel written once.
//   pass 1 (per pixel): PSF weight = sum of the taps inside the volume; coef = value / weight under the
//                       operator's activity rule (adjoint: weight >= 0.5; backward: grad != 0, weight != 0)
//   pass 2 (per voxel): vol[v] = sum coef * psf[tap] * trilinear(p - v)   [+ the same sum with 1/weight]
//   backward only     : grad_transforms by one workgroup per slice (block reduction, no atomics)
constexpr float kSqrt3 = 1.7320509f;

struct PixelGeom { float qx, qy, qz, xc, yc, zc; };

__device__ __forceinline__ PixelGeom pixel_geom(const float* t, int ix, int iy, int h, int w, float res_slice, int D, int H, int W) {
  PixelGeom g;
  g.qx = (float)((ix - (w - 1) / 2.) * (double)res_slice + (double)t[3]);
  g.qy = (float)((iy - (h - 1) / 2.) * (double)res_slice + (double)t[7]);
  g.qz = t[11];
  g.xc = t[0] * g.qx + t[1] * g.qy + t[2] * g.qz + (W - 1) / 2.f;
  g.yc = t[4] * g.qx + t[5] * g.qy + t[6] * g.qz + (H - 1) / 2.f;
  g.zc = t[8] * g.qx + t[9] * g.qy + t[10] * g.qz + (D - 1) / 2.f;
  return g;
}

// mode 0: adjoint (value = slices, active iff weight >= 0.5); mode 1: backward (value = grad_slices,
// active iff value != 0 and weight != 0).  coef[idx] = value / weight, cw[idx] = 1 / weight (0 if inactive).
__global__ __launch_bounds__(256) void slice_acq_pixel_coef(const float* __restrict__ transforms, const float* __restrict__ psf,
                                                            const float* __restrict__ value, const uint8_t* __restrict__ slices_mask,
                                                            float* __restrict__ coef, float* __restrict__ cw, int D, int H, int W,
                                                            int d_p, int h_p, int w_p, int n, int h, int w, float res_slice, int mode) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * h * w) return;
  float c = 0.f, iw = 0.f;
  const float val = value[idx];
  const bool on = (slices_mask == nullptr || slices_mask[idx]) && (mode == 0 || val != 0.f);
  if (on) {
    const int ix = idx % w, iy = (idx / w) % h, in = idx / ((int64_t)h * w);
    const float* t = transforms + (size_t)in * 12;
    const PixelGeom g = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    float weight = 0.f;
    int ip = 0;
    for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
      for (int iyp = -h_p / 2; iyp < (h_p + 1) / 2; ++iyp)
        for (int ixp = -w_p / 2; ixp < (w_p + 1) / 2; ++ixp, ++ip) {
          const float pv = psf[ip];
          if (pv == 0.f) continue;
          const float x = g.xc + t[0] * ixp + t[1] * iyp + t[2] * iz;
          const float y = g.yc + t[4] * ixp + t[5] * iyp + t[6] * iz;
          const float z = g.zc + t[8] * ixp + t[9] * iyp + t[10] * iz;
          if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
          weight += pv;
        }
    const bool active = mode == 0 ? weight >= 0.5f : weight != 0.f;
    if (active) { c = val / weight; iw = 1.f / weight; }
  }
  coef[idx] = c;
  if (cw != nullptr) cw[idx] = iw;
}

__global__ __launch_bounds__(256) void slice_acq_adjoint_gather(const float* __restrict__ transforms, const float* __restrict__ psf,
                                                                const float* __restrict__ coef, const float* __restrict__ cw,
                                                                const uint8_t* __restrict__ vol_mask, float* __restrict__ vol,
                                                                float* __restrict__ vol_weight, int D, int H, int W, int d_p,
                                                                int h_p, int w_p, int n, int h, int w, float res_slice,
                                                                int equalize, int accumulate) {
  const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (iv >= (int64_t)D * H * W) return;
  const int vx = iv % W, vy = (iv / W) % H, vz = iv / ((int64_t)H * W);
  float acc = 0.f, accw = 0.f;
  if (vol_mask == nullptr || vol_mask[iv]) {
    const float d0 = vx - (W - 1) / 2.f, d1 = vy - (H - 1) / 2.f, d2 = vz - (D - 1) / 2.f;
    const int tz0 = -d_p / 2, tz1 = (d_p + 1) / 2 - 1, ty0 = -h_p / 2, ty1 = (h_p + 1) / 2 - 1, tx0 = -w_p / 2, tx1 = (w_p + 1) / 2 - 1;
    for (int k = 0; k < n; ++k) {
      const float* t = transforms + (size_t)k * 12;
      // voxel in the slice frame: r = R^T (v - c0)
      const float rz = t[2] * d0 + t[6] * d1 + t[10] * d2;
      const float ez = rz - t[11];
      const int izlo = max(tz0, (int)ceilf(ez - kSqrt3)), izhi = min(tz1, (int)floorf(ez + kSqrt3));
      if (izlo > izhi) continue;
      const float rx = t[0] * d0 + t[4] * d1 + t[8] * d2, ry = t[1] * d0 + t[5] * d1 + t[9] * d2;
      const int ixlo = max(0, (int)ceilf((rx - t[3] - kSqrt3 - tx1) / res_slice + (w - 1) / 2.f));
      const int ixhi = min(w - 1, (int)floorf((rx - t[3] + kSqrt3 - tx0) / res_slice + (w - 1) / 2.f));
      const int iylo = max(0, (int)ceilf((ry - t[7] - kSqrt3 - ty1) / res_slice + (h - 1) / 2.f));
      const int iyhi = min(h - 1, (int)floorf((ry - t[7] + kSqrt3 - ty0) / res_slice + (h - 1) / 2.f));
      for (int iy = iylo; iy <= iyhi; ++iy)
        for (int ix = ixlo; ix <= ixhi; ++ix) {
          const size_t pidx = ((size_t)k * h + iy) * w + ix;
          const float cf = coef[pidx];
          const float cwv = cw != nullptr ? cw[pidx] : 0.f;
          if (cf == 0.f && cwv == 0.f) continue;
          const PixelGeom g = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
          const int jxlo = max(tx0, (int)ceilf(rx - g.qx - kSqrt3)), jxhi = min(tx1, (int)floorf(rx - g.qx + kSqrt3));
          const int jylo = max(ty0, (int)ceilf(ry - g.qy - kSqrt3)), jyhi = min(ty1, (int)floorf(ry - g.qy + kSqrt3));
          for (int iz = izlo; iz <= izhi; ++iz)
            for (int jy = jylo; jy <= jyhi; ++jy)
              for (int jx = jxlo; jx <= jxhi; ++jx) {
                const float pv = psf[((iz - tz0) * h_p + (jy - ty0)) * w_p + (jx - tx0)];
                if (pv == 0.f) continue;
                const float x = g.xc + t[0] * jx + t[1] * jy + t[2] * iz;
                const float y = g.yc + t[4] * jx + t[5] * jy + t[6] * iz;
                const float z = g.zc + t[8] * jx + t[9] * jy + t[10] * iz;
                if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
                const float ex = x - vx, ey = y - vy, ezz = z - vz;
                if (ex < -1.f || ex >= 1.f || ey < -1.f || ey >= 1.f || ezz < -1.f || ezz >= 1.f) continue;
                // the voxel must be floor(p) or floor(p) + 1 on every axis, with the scatter's weights
                const float wx = ex >= 0.f ? 1.f - ex : 1.f + ex, wy = ey >= 0.f ? 1.f - ey : 1.f + ey, wz = ezz >= 0.f ? 1.f - ezz : 1.f + ezz;
                const float wgt = wx * wy * wz * pv;
                acc += wgt * cf;
                accw += wgt * cwv;
              }
        }
    }
  }
  if (equalize && accw > 0.f) acc /= accw;
  if (accumulate) vol[iv] += acc; else vol[iv] = acc;
  if (vol_weight != nullptr) vol_weight[iv] = accw;
}

// d L / d transforms of the forward operator: one workgroup per slice, block reduction of 12 sums.
__global__ __launch_bounds__(256) void slice_acq_bwd_transforms(const float* __restrict__ transforms, const float* __restrict__ vol,
                                                                const uint8_t* __restrict__ vol_mask, const float* __restrict__ psf,
                                                                const float* __restrict__ coef, float* __restrict__ grad_transforms,
                                                                int D, int H, int W, int d_p, int h_p, int w_p, int h, int w,
                                                                float res_slice) {
  __shared__ float red[4][12];
  const int k = blockIdx.x;
  const float* t = transforms + (size_t)k * 12;
  const int Sy = W, Sz = H * W;
  float g[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = 0.f;
  for (int pix = threadIdx.x; pix < h * w; pix += blockDim.x) {
    const float gs = coef[(size_t)k * h * w + pix];
    if (gs == 0.f) continue;
    const int ix = pix % w, iy = pix / w;
    const PixelGeom pg = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    int ip = 0;
    for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
      for (int iyp = -h_p / 2; iyp < (h_p + 1) / 2; ++iyp)
        for (int ixp = -w_p / 2; ixp < (w_p + 1) / 2; ++ixp, ++ip) {
          const float pv = psf[ip];
          if (pv == 0.f) continue;
          const float x = pg.xc + t[0] * ixp + t[1] * iyp + t[2] * iz;
          const float y = pg.yc + t[4] * ixp + t[5] * iyp + t[6] * iz;
          const float z = pg.zc + t[8] * ixp + t[9] * iyp + t[10] * iz;
          if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
          const int xf = (int)floorf(x), yf = (int)floorf(y), zf = (int)floorf(z);
          const float wx = x - xf, wy = y - yf, wz = z - zf;
          const int i0 = zf * Sz + yf * Sy + xf;
          const float pgs = pv * gs;
          float dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
            const int ic = i0 + cx + cy * Sy + cz * Sz;
            if (vol_mask != nullptr && !vol_mask[ic]) continue;
            const float val = pgs * vol[ic];
            const float ax = cx ? wx : 1.f - wx, ay = cy ? wy : 1.f - wy, az = cz ? wz : 1.f - wz;
            dx += (cx ? val : -val) * ay * az;
            dy += (cy ? val : -val) * ax * az;
            dz += (cz ? val : -val) * ax * ay;
          }
          const float ox = pg.qx + ixp, oy = pg.qy + iyp, oz = pg.qz + iz;
          g[0] += dx * ox; g[1] += dx * oy; g[2] += dx * oz;
          g[4] += dy * ox; g[5] += dy * oy; g[6] += dy * oz;
          g[8] += dz * ox; g[9] += dz * oy; g[10] += dz * oz;
          g[3] += dx * t[0] + dy * t[4] + dz * t[8];
          g[7] += dx * t[1] + dy * t[5] + dz * t[9];
          g[11] += dx * t[2] + dy * t[6] + dz * t[10];
        }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float s = wave_sum_dpp(g[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12) grad_transforms[(size_t)k * 12 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// in-place "equalise a gradient" step of adjoint_backward (slice_acq_cuda_kernel.cu:672-693 with is_grad = true)
__global__ void slice_acq_equalize_grad(float* __restrict__ grad_vol, const float* __restrict__ vol_weight, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float wgt = vol_weight[i];
  if (wgt > 0.f) grad_vol[i] /= (wgt < 1e-3f ? 1e-3f : wgt);
}

// Backward of A^T (slice_acq_cuda_kernel.cu:695-950, linear mode): a gather per slice pixel,
//   grad_slices[p] = sum_taps psf * trilinear(grad_vol) / sum_taps psf            (taps inside the volume)
//   grad_transforms[k] += d/dT of the same expression with the corner values weighted by
//                         (slices[p] - vol[corner]) (equalised adjoint) or slices[p].
// One workgroup per slice; the 12 pose sums are block-reduced (the reference adds them with atomics).
__global__ __launch_bounds__(256) void slice_acq_adjoint_bwd(const float* __restrict__ transforms, const float* __restrict__ grad_vol,
                                                             const float* __restrict__ psf, const float* __restrict__ slices,
                                                             const uint8_t* __restrict__ slices_mask, const float* __restrict__ vol,
                                                             const uint8_t* __restrict__ vol_mask, float* __restrict__ grad_slices,
                                                             float* __restrict__ grad_transforms, int D, int H, int W, int d_p,
                                                             int h_p, int w_p, int h, int w, float res_slice) {
  __shared__ float red[4][12];
  const int k = blockIdx.x;
  const float* t = transforms + (size_t)k * 12;
  const int Sy = W, Sz = H * W;
  float g[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = 0.f;
  for (int pix = threadIdx.x; pix < h * w; pix += blockDim.x) {
    const size_t idx = (size_t)k * h * w + pix;
    if (slices_mask != nullptr && !slices_mask[idx]) continue;
    const int ix = pix % w, iy = pix / w;
    const PixelGeom pg = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    const float sv = slices[idx];
    float val = 0.f, weight = 0.f;
    float gp[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gp[i] = 0.f;
    int ip = 0;
    for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
      for (int iyp = -h_p / 2; iyp < (h_p + 1) / 2; ++iyp)
        for (int ixp = -w_p / 2; ixp < (w_p + 1) / 2; ++ixp, ++ip) {
          const float pv = psf[ip];
          if (pv == 0.f) continue;
          const float x = pg.xc + t[0] * ixp + t[1] * iyp + t[2] * iz;
          const float y = pg.yc + t[4] * ixp + t[5] * iyp + t[6] * iz;
          const float z = pg.zc + t[8] * ixp + t[9] * iyp + t[10] * iz;
          if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
          const int xf = (int)floorf(x), yf = (int)floorf(y), zf = (int)floorf(z);
          const float wx = x - xf, wy = y - yf, wz = z - zf;
          const int i0 = zf * Sz + yf * Sy + xf;
          float v_ = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
            const int ic = i0 + cx + cy * Sy + cz * Sz;
            if (vol_mask != nullptr && !vol_mask[ic]) continue;
            const float gv = grad_vol[ic];
            const float ax = cx ? wx : 1.f - wx, ay = cy ? wy : 1.f - wy, az = cz ? wz : 1.f - wz;
            v_ += ax * ay * az * gv;
            const float s = (vol == nullptr ? sv : sv - vol[ic]) * gv;
            dx += (cx ? s : -s) * ay * az;
            dy += (cy ? s : -s) * ax * az;
            dz += (cz ? s : -s) * ax * ay;
          }
          val += pv * v_;
          weight += pv;
          dx *= pv; dy *= pv; dz *= pv;
          const float ox = pg.qx + ixp, oy = pg.qy + iyp, oz = pg.qz + iz;
          gp[0] += dx * ox; gp[1] += dx * oy; gp[2] += dx * oz;
          gp[4] += dy * ox; gp[5] += dy * oy; gp[6] += dy * oz;
          gp[8] += dz * ox; gp[9] += dz * oy; gp[10] += dz * oz;
          gp[3] += dx * t[0] + dy * t[4] + dz * t[8];
          gp[7] += dx * t[1] + dy * t[5] + dz * t[9];
          gp[11] += dx * t[2] + dy * t[6] + dz * t[10];
        }
    if (weight > 0.f) {
      if (grad_slices != nullptr) grad_slices[idx] = val / weight;
#pragma unroll
      for (int i = 0; i < 12; ++i) g[i] += gp[i] / weight;
    }
  }
  if (grad_transforms == nullptr) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float s = wave_sum_dpp(g[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12) grad_transforms[(size_t)k * 12 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int nesvor_slice_acq_adjoint_backward(const float* transforms, float* grad_vol, const float* vol_weight,
                                                 const uint8_t* vol_mask, const float* psf, const float* slices,
                                                 const uint8_t* slices_mask, const float* vol, float* grad_slices,
                                                 float* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p,
                                                 int n, int h, int w, float res_slice, int equalize, void* stream) {
  if ((int64_t)n * h * w <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (equalize) {
    if (vol_weight == nullptr || vol == nullptr) return (int)hipErrorInvalidValue;
    const int64_t nv = (int64_t)D * H * W;
    hipLaunchKernelGGL(slice_acq_equalize_grad, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, grad_vol, vol_weight, nv);
  }
  hipLaunchKernelGGL(slice_acq_adjoint_bwd, dim3((unsigned)n), dim3(256), 0, st, transforms, (const float*)grad_vol, psf, slices,
                     slices_mask, equalize ? vol : (const float*)nullptr, vol_mask, grad_slices, grad_transforms, D, H, W, d_p, h_p,
                     w_p, h, w, res_slice);
  return (int)hipGetLastError();
}

extern "C" int nesvor_slice_acq_forward(const float* transforms, const float* vol, const uint8_t* vol_mask,
                                        const uint8_t* slices_mask, const float* psf, float* slices,
                                        float* slices_weight, int D, int H, int W, int d_p, int h_p, int w_p, int n,
                                        int h, int w, float res_slice, int interp_psf, void* stream) {
  const int64_t total = (int64_t)n * h * w;
  if (total <= 0) return 0;
  const bool lds = (int64_t)d_p * h_p * w_p <= kMaxTaps;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define NESVOR_LAUNCH_FWD(I, L)                                                                                           \
  hipLaunchKernelGGL((slice_acq_fwd<I, L>), grid, block, 0, (hipStream_t)stream, transforms, vol, vol_mask, slices_mask, psf, \
                     slices, slices_weight, D, H, W, d_p, h_p, w_p, n, h, w, res_slice)
  if (interp_psf) { if (lds) NESVOR_LAUNCH_FWD(true, true); else NESVOR_LAUNCH_FWD(true, false); }
  else { if (lds) NESVOR_LAUNCH_FWD(false, true); else NESVOR_LAUNCH_FWD(false, false); }
#undef NESVOR_LAUNCH_FWD
  return (int)hipGetLastError();
}

extern "C" int nesvor_slice_acq_adjoint_forward(const float* transforms, const float* psf, const float* slices,
                                                const uint8_t* slices_mask, const uint8_t* vol_mask, float* vol,
                                                float* vol_weight, float* scratch, int D, int H, int W, int d_p, int h_p,
                                                int w_p, int n, int h, int w, float res_slice, int equalize, void* stream) {
  const int64_t np = (int64_t)n * h * w, nv = (int64_t)D * H * W;
  if (nv <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  float* coef = scratch;
  float* cw = scratch + np;
  if (np > 0)
    hipLaunchKernelGGL(slice_acq_pixel_coef, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, transforms, psf, slices,
                       slices_mask, coef, cw, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, 0);
  hipLaunchKernelGGL(slice_acq_adjoint_gather, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, transforms, psf, coef,
                     (equalize || vol_weight != nullptr) ? cw : (const float*)nullptr, vol_mask, vol, vol_weight, D, H, W, d_p, h_p,
                     w_p, n, h, w, res_slice, equalize, 0);
  return (int)hipGetLastError();
}

extern "C" int nesvor_slice_acq_backward(const float* transforms, const float* vol, const uint8_t* vol_mask,
                                         const float* psf, const float* grad_slices, const uint8_t* slices_mask,
                                         float* grad_vol, float* grad_transforms, float* scratch, int D, int H, int W,
                                         int d_p, int h_p, int w_p, int n, int h, int w, float res_slice, void* stream) {
  const int64_t np = (int64_t)n * h * w, nv = (int64_t)D * H * W;
  if (np <= 0 || nv <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  float* coef = scratch;
  hipLaunchKernelGGL(slice_acq_pixel_coef, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, transforms, psf, grad_slices,
                     slices_mask, coef, (float*)nullptr, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, 1);
  if (grad_vol != nullptr)
    hipLaunchKernelGGL(slice_acq_adjoint_gather, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, transforms, psf, coef,
                       (const float*)nullptr, vol_mask, grad_vol, (float*)nullptr, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, 0, 0);
  if (grad_transforms != nullptr)
    hipLaunchKernelGGL(slice_acq_bwd_transforms, dim3((unsigned)n), dim3(256), 0, st, transforms, vol, vol_mask, psf, coef,
                       grad_transforms, D, H, W, d_p, h_p, w_p, h, w, res_slice);
  return (int)hipGetLastError();
}
