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
// Full-wave sum for float or double (the 12 pose sums per slice: once per workgroup, off the inner loops)
template <typename T> __device__ __forceinline__ T wave_sum_any(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}


constexpr int kMaxTaps = 1024;

template <typename T> struct Tap { T x, y, z, w; };

// LDS_TAPS: the PSF's non-zero taps fit the LDS list (every PSF get_PSF builds for clinical geometries: (9,5,5) = 153
// taps for 1.5 x 1.5 x 3 mm slices on a 1 mm grid); larger PSFs (thick slices on a fine grid, e.g. 6 mm on 0.5 mm = 1813
// taps) walk the dense PSF array in global memory instead - same arithmetic, same order.
template <typename T, bool INTERP_PSF, bool LDS_TAPS>
__global__ __launch_bounds__(256) void slice_acq_fwd(
    const T* __restrict__ transforms, const T* __restrict__ vol, const uint8_t* __restrict__ vol_mask,
    const uint8_t* __restrict__ slices_mask, const T* __restrict__ psf, T* __restrict__ slices,
    T* __restrict__ slices_weight, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
    T res_slice) {
  __shared__ Tap<T> taps[LDS_TAPS ? kMaxTaps : 1];
  __shared__ int n_taps;
  if constexpr (LDS_TAPS) {
    if (threadIdx.x == 0) {
      int cnt = 0, ip = 0;
      for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
        for (int iy = -h_p / 2; iy < (h_p + 1) / 2; ++iy)
          for (int ix = -w_p / 2; ix < (w_p + 1) / 2; ++ix, ++ip) {
            T pv = psf[ip];
            if (pv != (T)0.0) taps[cnt++] = Tap<T>{(T)ix, (T)iy, (T)iz, pv};  // the host checked d_p h_p w_p <= kMaxTaps
          }
      n_taps = cnt;
    }
    __syncthreads();
  }
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * h * w) return;
  if (slices_mask != nullptr && !slices_mask[idx]) return;
  const int ix = idx % w, iy = (idx / w) % h, in = idx / ((int64_t)h * w);
  const T* t = transforms + (size_t)in * 12;
  const T r11 = t[0], r12 = t[1], r13 = t[2], r21 = t[4], r22 = t[5], r23 = t[6], r31 = t[8], r32 = t[9], r33 = t[10];
  // the reference evaluates this affine map in double and rounds once (.cu:46-47)
  const T px = (T)((ix - (w - 1) / 2.) * (double)res_slice + (double)t[3]);
  const T py = (T)((iy - (h - 1) / 2.) * (double)res_slice + (double)t[7]);
  const T pz = t[11];
  const T xc = r11 * px + r12 * py + r13 * pz + (W - 1) / (T)2.0;
  const T yc = r21 * px + r22 * py + r23 * pz + (H - 1) / (T)2.0;
  const T zc = r31 * px + r32 * py + r33 * pz + (D - 1) / (T)2.0;
  const int Sy = W, Sz = H * W;
  T val = (T)0.0, wsum = (T)0.0;
  const int nt = LDS_TAPS ? n_taps : d_p * h_p * w_p;
  const int tx0 = -w_p / 2, ty0 = -h_p / 2, tz0 = -d_p / 2;
  for (int k = 0; k < nt; ++k) {
    Tap<T> tp;
    if constexpr (LDS_TAPS) {
      tp = taps[k];
    } else {
      const T pv = psf[k];
      if (pv == (T)0.0) continue;
      const int kz = k / (h_p * w_p), kr = k - kz * (h_p * w_p), ky = kr / w_p;
      tp = Tap<T>{(T)(tx0 + kr - ky * w_p), (T)(ty0 + ky), (T)(tz0 + kz), pv};
    }
    const T x = xc + r11 * tp.x + r12 * tp.y + r13 * tp.z;
    const T y = yc + r21 * tp.x + r22 * tp.y + r23 * tp.z;
    const T z = zc + r31 * tp.x + r32 * tp.y + r33 * tp.z;
    if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
    if (INTERP_PSF) {
      // nearest voxel, PSF re-interpolated at the voxel's offset from the centre
      const int xr = (int)floor(x + (T)0.5), yr = (int)floor(y + (T)0.5), zr = (int)floor(z + (T)0.5);
      const int iv = zr * Sz + yr * Sy + xr;
      if (vol_mask != nullptr && !vol_mask[iv]) continue;
      const T dx = xr - xc, dy = yr - yc, dz = zr - zc;
      const T xp = r11 * dx + r21 * dy + r31 * dz + (w_p - 1) / (T)2.0;
      const T yp = r12 * dx + r22 * dy + r32 * dz + (h_p - 1) / (T)2.0;
      const T zp = r13 * dx + r23 * dy + r33 * dz + (d_p - 1) / (T)2.0;
      if (xp < 0 || yp < 0 || zp < 0 || xp >= w_p - 1 || yp >= h_p - 1 || zp >= d_p - 1) continue;
      const int xf = (int)floor(xp), yf = (int)floor(yp), zf = (int)floor(zp);
      const T wx = xp - xf, wy = yp - yf, wz = zp - zf;
      const T* p0 = psf + (zf * h_p + yf) * w_p + xf;
      const int py_ = w_p, pz_ = w_p * h_p;
      T pw = (T)0.0;
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
      const int xf = (int)floor(x), yf = (int)floor(y), zf = (int)floor(z);
      const T wx = x - xf, wy = y - yf, wz = z - zf;
      const int iv = zf * Sz + yf * Sy + xf;
      // corner order as the reference accumulates: 000,100,010,001,110,101,011,111
      const int off[8] = {0, 1, Sy, Sz, 1 + Sy, 1 + Sz, Sy + Sz, 1 + Sy + Sz};
      const T cw[8] = {(1 - wx) * (1 - wy) * (1 - wz), wx * (1 - wy) * (1 - wz), (1 - wx) * wy * (1 - wz),
                           (1 - wx) * (1 - wy) * wz,       wx * wy * (1 - wz),       wx * (1 - wy) * wz,
                           (1 - wx) * wy * wz,             wx * wy * wz};
      T v8[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v8[c] = vol[iv + off[c]];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (vol_mask == nullptr || vol_mask[iv + off[c]]) {
          const T pw = cw[c] * tp.w;
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
template <typename T> __device__ __forceinline__ constexpr T sqrt3() { return (T)1.7320508075688772; }

template <typename T> struct PixelGeom { T qx, qy, qz, xc, yc, zc; };

template <typename T>
__device__ __forceinline__ PixelGeom<T> pixel_geom(const T* t, int ix, int iy, int h, int w, T res_slice, int D, int H, int W) {
  PixelGeom<T> g;
  g.qx = (T)((ix - (w - 1) / 2.) * (double)res_slice + (double)t[3]);
  g.qy = (T)((iy - (h - 1) / 2.) * (double)res_slice + (double)t[7]);
  g.qz = t[11];
  g.xc = t[0] * g.qx + t[1] * g.qy + t[2] * g.qz + (W - 1) / (T)2.0;
  g.yc = t[4] * g.qx + t[5] * g.qy + t[6] * g.qz + (H - 1) / (T)2.0;
  g.zc = t[8] * g.qx + t[9] * g.qy + t[10] * g.qz + (D - 1) / (T)2.0;
  return g;
}

// mode 0: adjoint (value = slices, active iff weight >= 0.5); mode 1: backward (value = grad_slices,
// active iff value != 0 and weight != 0).  coef[idx] = value / weight, cw[idx] = 1 / weight (0 if inactive).
template <typename T>
__global__ __launch_bounds__(256) void slice_acq_pixel_coef(const T* __restrict__ transforms, const T* __restrict__ psf,
                                                            const T* __restrict__ value, const uint8_t* __restrict__ slices_mask,
                                                            T* __restrict__ coef, T* __restrict__ cw, int D, int H, int W,
                                                            int d_p, int h_p, int w_p, int n, int h, int w, T res_slice, int mode) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * h * w) return;
  T c = (T)0.0, iw = (T)0.0;
  const T val = value[idx];
  const bool on = (slices_mask == nullptr || slices_mask[idx]) && (mode == 0 || val != (T)0.0);
  if (on) {
    const int ix = idx % w, iy = (idx / w) % h, in = idx / ((int64_t)h * w);
    const T* t = transforms + (size_t)in * 12;
    const PixelGeom<T> g = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    T weight = (T)0.0;
    int ip = 0;
    for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
      for (int iyp = -h_p / 2; iyp < (h_p + 1) / 2; ++iyp)
        for (int ixp = -w_p / 2; ixp < (w_p + 1) / 2; ++ixp, ++ip) {
          const T pv = psf[ip];
          if (pv == (T)0.0) continue;
          const T x = g.xc + t[0] * ixp + t[1] * iyp + t[2] * iz;
          const T y = g.yc + t[4] * ixp + t[5] * iyp + t[6] * iz;
          const T z = g.zc + t[8] * ixp + t[9] * iyp + t[10] * iz;
          if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
          weight += pv;
        }
    const bool active = mode == 0 ? weight >= (T)0.5 : weight != (T)0.0;
    if (active) { c = val / weight; iw = (T)1.0 / weight; }
  }
  coef[idx] = c;
  if (cw != nullptr) cw[idx] = iw;
}

template <typename T>
__global__ __launch_bounds__(256) void slice_acq_adjoint_gather(const T* __restrict__ transforms, const T* __restrict__ psf,
                                                                const T* __restrict__ coef, const T* __restrict__ cw,
                                                                const uint8_t* __restrict__ vol_mask, T* __restrict__ vol,
                                                                T* __restrict__ vol_weight, int D, int H, int W, int d_p,
                                                                int h_p, int w_p, int n, int h, int w, T res_slice,
                                                                int equalize, int accumulate) {
  const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (iv >= (int64_t)D * H * W) return;
  const int vx = iv % W, vy = (iv / W) % H, vz = iv / ((int64_t)H * W);
  T acc = (T)0.0, accw = (T)0.0;
  if (vol_mask == nullptr || vol_mask[iv]) {
    const T d0 = vx - (W - 1) / (T)2.0, d1 = vy - (H - 1) / (T)2.0, d2 = vz - (D - 1) / (T)2.0;
    const int tz0 = -d_p / 2, tz1 = (d_p + 1) / 2 - 1, ty0 = -h_p / 2, ty1 = (h_p + 1) / 2 - 1, tx0 = -w_p / 2, tx1 = (w_p + 1) / 2 - 1;
    for (int k = 0; k < n; ++k) {
      const T* t = transforms + (size_t)k * 12;
      // voxel in the slice frame: r = R^T (v - c0)
      const T rz = t[2] * d0 + t[6] * d1 + t[10] * d2;
      const T ez = rz - t[11];
      const int izlo = max(tz0, (int)ceil(ez - sqrt3<T>())), izhi = min(tz1, (int)floor(ez + sqrt3<T>()));
      if (izlo > izhi) continue;
      const T rx = t[0] * d0 + t[4] * d1 + t[8] * d2, ry = t[1] * d0 + t[5] * d1 + t[9] * d2;
      const int ixlo = max(0, (int)ceil((rx - t[3] - sqrt3<T>() - tx1) / res_slice + (w - 1) / (T)2.0));
      const int ixhi = min(w - 1, (int)floor((rx - t[3] + sqrt3<T>() - tx0) / res_slice + (w - 1) / (T)2.0));
      const int iylo = max(0, (int)ceil((ry - t[7] - sqrt3<T>() - ty1) / res_slice + (h - 1) / (T)2.0));
      const int iyhi = min(h - 1, (int)floor((ry - t[7] + sqrt3<T>() - ty0) / res_slice + (h - 1) / (T)2.0));
      for (int iy = iylo; iy <= iyhi; ++iy)
        for (int ix = ixlo; ix <= ixhi; ++ix) {
          const size_t pidx = ((size_t)k * h + iy) * w + ix;
          const T cf = coef[pidx];
          const T cwv = cw != nullptr ? cw[pidx] : (T)0.0;
          if (cf == (T)0.0 && cwv == (T)0.0) continue;
          const PixelGeom<T> g = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
          const int jxlo = max(tx0, (int)ceil(rx - g.qx - sqrt3<T>())), jxhi = min(tx1, (int)floor(rx - g.qx + sqrt3<T>()));
          const int jylo = max(ty0, (int)ceil(ry - g.qy - sqrt3<T>())), jyhi = min(ty1, (int)floor(ry - g.qy + sqrt3<T>()));
          for (int iz = izlo; iz <= izhi; ++iz)
            for (int jy = jylo; jy <= jyhi; ++jy)
              for (int jx = jxlo; jx <= jxhi; ++jx) {
                const T pv = psf[((iz - tz0) * h_p + (jy - ty0)) * w_p + (jx - tx0)];
                if (pv == (T)0.0) continue;
                const T x = g.xc + t[0] * jx + t[1] * jy + t[2] * iz;
                const T y = g.yc + t[4] * jx + t[5] * jy + t[6] * iz;
                const T z = g.zc + t[8] * jx + t[9] * jy + t[10] * iz;
                if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
                const T ex = x - vx, ey = y - vy, ezz = z - vz;
                if (ex < -(T)1.0 || ex >= (T)1.0 || ey < -(T)1.0 || ey >= (T)1.0 || ezz < -(T)1.0 || ezz >= (T)1.0) continue;
                // the voxel must be floor(p) or floor(p) + 1 on every axis, with the scatter's weights
                const T wx = ex >= (T)0.0 ? (T)1.0 - ex : (T)1.0 + ex, wy = ey >= (T)0.0 ? (T)1.0 - ey : (T)1.0 + ey, wz = ezz >= (T)0.0 ? (T)1.0 - ezz : (T)1.0 + ezz;
                const T wgt = wx * wy * wz * pv;
                acc += wgt * cf;
                accw += wgt * cwv;
              }
        }
    }
  }
  if (equalize && accw > (T)0.0) acc /= accw;
  if (accumulate) vol[iv] += acc; else vol[iv] = acc;
  if (vol_weight != nullptr) vol_weight[iv] = accw;
}

// d L / d transforms of the forward operator: one workgroup per slice, block reduction of 12 sums.
template <typename T>
__global__ __launch_bounds__(256) void slice_acq_bwd_transforms(const T* __restrict__ transforms, const T* __restrict__ vol,
                                                                const uint8_t* __restrict__ vol_mask, const T* __restrict__ psf,
                                                                const T* __restrict__ coef, T* __restrict__ grad_transforms,
                                                                int D, int H, int W, int d_p, int h_p, int w_p, int h, int w,
                                                                T res_slice) {
  __shared__ T red[4][12];
  const int k = blockIdx.x;
  const T* t = transforms + (size_t)k * 12;
  const int Sy = W, Sz = H * W;
  T g[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = (T)0.0;
  for (int pix = threadIdx.x; pix < h * w; pix += blockDim.x) {
    const T gs = coef[(size_t)k * h * w + pix];
    if (gs == (T)0.0) continue;
    const int ix = pix % w, iy = pix / w;
    const PixelGeom<T> pg = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    int ip = 0;
    for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
      for (int iyp = -h_p / 2; iyp < (h_p + 1) / 2; ++iyp)
        for (int ixp = -w_p / 2; ixp < (w_p + 1) / 2; ++ixp, ++ip) {
          const T pv = psf[ip];
          if (pv == (T)0.0) continue;
          const T x = pg.xc + t[0] * ixp + t[1] * iyp + t[2] * iz;
          const T y = pg.yc + t[4] * ixp + t[5] * iyp + t[6] * iz;
          const T z = pg.zc + t[8] * ixp + t[9] * iyp + t[10] * iz;
          if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
          const int xf = (int)floor(x), yf = (int)floor(y), zf = (int)floor(z);
          const T wx = x - xf, wy = y - yf, wz = z - zf;
          const int i0 = zf * Sz + yf * Sy + xf;
          const T pgs = pv * gs;
          T dx = (T)0.0, dy = (T)0.0, dz = (T)0.0;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
            const int ic = i0 + cx + cy * Sy + cz * Sz;
            if (vol_mask != nullptr && !vol_mask[ic]) continue;
            const T val = pgs * vol[ic];
            const T ax = cx ? wx : (T)1.0 - wx, ay = cy ? wy : (T)1.0 - wy, az = cz ? wz : (T)1.0 - wz;
            dx += (cx ? val : -val) * ay * az;
            dy += (cy ? val : -val) * ax * az;
            dz += (cz ? val : -val) * ax * ay;
          }
          const T ox = pg.qx + ixp, oy = pg.qy + iyp, oz = pg.qz + iz;
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
    const T s = wave_sum_any(g[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12) grad_transforms[(size_t)k * 12 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// in-place "equalise a gradient" step of adjoint_backward (slice_acq_cuda_kernel.cu:672-693 with is_grad = true)
template <typename T>
__global__ void slice_acq_equalize_grad(T* __restrict__ grad_vol, const T* __restrict__ vol_weight, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T wgt = vol_weight[i];
  if (wgt > (T)0.0) grad_vol[i] /= (wgt < (T)1e-3 ? (T)1e-3 : wgt);
}

// Backward of A^T (slice_acq_cuda_kernel.cu:695-950, linear mode): a gather per slice pixel,
//   grad_slices[p] = sum_taps psf * trilinear(grad_vol) / sum_taps psf            (taps inside the volume)
//   grad_transforms[k] += d/dT of the same expression with the corner values weighted by
//                         (slices[p] - vol[corner]) (equalised adjoint) or slices[p].
// One workgroup per slice; the 12 pose sums are block-reduced (the reference adds them with atomics).
template <typename T>
__global__ __launch_bounds__(256) void slice_acq_adjoint_bwd(const T* __restrict__ transforms, const T* __restrict__ grad_vol,
                                                             const T* __restrict__ psf, const T* __restrict__ slices,
                                                             const uint8_t* __restrict__ slices_mask, const T* __restrict__ vol,
                                                             const uint8_t* __restrict__ vol_mask, T* __restrict__ grad_slices,
                                                             T* __restrict__ grad_transforms, int D, int H, int W, int d_p,
                                                             int h_p, int w_p, int h, int w, T res_slice) {
  __shared__ T red[4][12];
  const int k = blockIdx.x;
  const T* t = transforms + (size_t)k * 12;
  const int Sy = W, Sz = H * W;
  T g[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = (T)0.0;
  for (int pix = threadIdx.x; pix < h * w; pix += blockDim.x) {
    const size_t idx = (size_t)k * h * w + pix;
    if (slices_mask != nullptr && !slices_mask[idx]) continue;
    const int ix = pix % w, iy = pix / w;
    const PixelGeom<T> pg = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    const T sv = slices[idx];
    T val = (T)0.0, weight = (T)0.0;
    T gp[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gp[i] = (T)0.0;
    int ip = 0;
    for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
      for (int iyp = -h_p / 2; iyp < (h_p + 1) / 2; ++iyp)
        for (int ixp = -w_p / 2; ixp < (w_p + 1) / 2; ++ixp, ++ip) {
          const T pv = psf[ip];
          if (pv == (T)0.0) continue;
          const T x = pg.xc + t[0] * ixp + t[1] * iyp + t[2] * iz;
          const T y = pg.yc + t[4] * ixp + t[5] * iyp + t[6] * iz;
          const T z = pg.zc + t[8] * ixp + t[9] * iyp + t[10] * iz;
          if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
          const int xf = (int)floor(x), yf = (int)floor(y), zf = (int)floor(z);
          const T wx = x - xf, wy = y - yf, wz = z - zf;
          const int i0 = zf * Sz + yf * Sy + xf;
          T v_ = (T)0.0, dx = (T)0.0, dy = (T)0.0, dz = (T)0.0;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
            const int ic = i0 + cx + cy * Sy + cz * Sz;
            if (vol_mask != nullptr && !vol_mask[ic]) continue;
            const T gv = grad_vol[ic];
            const T ax = cx ? wx : (T)1.0 - wx, ay = cy ? wy : (T)1.0 - wy, az = cz ? wz : (T)1.0 - wz;
            v_ += ax * ay * az * gv;
            const T s = (vol == nullptr ? sv : sv - vol[ic]) * gv;
            dx += (cx ? s : -s) * ay * az;
            dy += (cy ? s : -s) * ax * az;
            dz += (cz ? s : -s) * ax * ay;
          }
          val += pv * v_;
          weight += pv;
          dx *= pv; dy *= pv; dz *= pv;
          const T ox = pg.qx + ixp, oy = pg.qy + iyp, oz = pg.qz + iz;
          gp[0] += dx * ox; gp[1] += dx * oy; gp[2] += dx * oz;
          gp[4] += dy * ox; gp[5] += dy * oy; gp[6] += dy * oz;
          gp[8] += dz * ox; gp[9] += dz * oy; gp[10] += dz * oz;
          gp[3] += dx * t[0] + dy * t[4] + dz * t[8];
          gp[7] += dx * t[1] + dy * t[5] + dz * t[9];
          gp[11] += dx * t[2] + dy * t[6] + dz * t[10];
        }
    if (weight > (T)0.0) {
      if (grad_slices != nullptr) grad_slices[idx] = val / weight;
#pragma unroll
      for (int i = 0; i < 12; ++i) g[i] += gp[i] / weight;
    }
  }
  if (grad_transforms == nullptr) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const T s = wave_sum_any(g[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12) grad_transforms[(size_t)k * 12 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}


template <typename T>
int slice_acq_adjoint_backward_impl(const T* transforms, T* grad_vol, const T* vol_weight,
                                                 const uint8_t* vol_mask, const T* psf, const T* slices,
                                                 const uint8_t* slices_mask, const T* vol, T* grad_slices,
                                                 T* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p,
                                                 int n, int h, int w, T res_slice, int equalize, void* stream) {
  if ((int64_t)n * h * w <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (equalize) {
    if (vol_weight == nullptr || vol == nullptr) return (int)hipErrorInvalidValue;
    const int64_t nv = (int64_t)D * H * W;
    hipLaunchKernelGGL(slice_acq_equalize_grad<T>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, grad_vol, vol_weight, nv);
  }
  hipLaunchKernelGGL(slice_acq_adjoint_bwd<T>, dim3((unsigned)n), dim3(256), 0, st, transforms, (const T*)grad_vol, psf, slices,
                     slices_mask, equalize ? vol : (const T*)nullptr, vol_mask, grad_slices, grad_transforms, D, H, W, d_p, h_p,
                     w_p, h, w, res_slice);
  return (int)hipGetLastError();
}

template <typename T>
int slice_acq_forward_impl(const T* transforms, const T* vol, const uint8_t* vol_mask,
                                        const uint8_t* slices_mask, const T* psf, T* slices,
                                        T* slices_weight, int D, int H, int W, int d_p, int h_p, int w_p, int n,
                                        int h, int w, T res_slice, int interp_psf, void* stream) {
  const int64_t total = (int64_t)n * h * w;
  if (total <= 0) return 0;
  const bool lds = (int64_t)d_p * h_p * w_p <= kMaxTaps;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define NESVOR_LAUNCH_FWD(I, L)                                                                                           \
  hipLaunchKernelGGL((slice_acq_fwd<T, I, L>), grid, block, 0, (hipStream_t)stream, transforms, vol, vol_mask, slices_mask, psf, \
                     slices, slices_weight, D, H, W, d_p, h_p, w_p, n, h, w, res_slice)
  if (interp_psf) { if (lds) NESVOR_LAUNCH_FWD(true, true); else NESVOR_LAUNCH_FWD(true, false); }
  else { if (lds) NESVOR_LAUNCH_FWD(false, true); else NESVOR_LAUNCH_FWD(false, false); }
#undef NESVOR_LAUNCH_FWD
  return (int)hipGetLastError();
}

template <typename T>
int slice_acq_adjoint_forward_impl(const T* transforms, const T* psf, const T* slices,
                                                const uint8_t* slices_mask, const uint8_t* vol_mask, T* vol,
                                                T* vol_weight, T* scratch, int D, int H, int W, int d_p, int h_p,
                                                int w_p, int n, int h, int w, T res_slice, int equalize, void* stream) {
  const int64_t np = (int64_t)n * h * w, nv = (int64_t)D * H * W;
  if (nv <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  T* coef = scratch;
  T* cw = scratch + np;
  if (np > 0)
    hipLaunchKernelGGL(slice_acq_pixel_coef<T>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, transforms, psf, slices,
                       slices_mask, coef, cw, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, 0);
  hipLaunchKernelGGL(slice_acq_adjoint_gather<T>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, transforms, psf, coef,
                     (equalize || vol_weight != nullptr) ? cw : (const T*)nullptr, vol_mask, vol, vol_weight, D, H, W, d_p, h_p,
                     w_p, n, h, w, res_slice, equalize, 0);
  return (int)hipGetLastError();
}

template <typename T>
int slice_acq_backward_impl(const T* transforms, const T* vol, const uint8_t* vol_mask,
                                         const T* psf, const T* grad_slices, const uint8_t* slices_mask,
                                         T* grad_vol, T* grad_transforms, T* scratch, int D, int H, int W,
                                         int d_p, int h_p, int w_p, int n, int h, int w, T res_slice, void* stream) {
  const int64_t np = (int64_t)n * h * w, nv = (int64_t)D * H * W;
  if (np <= 0 || nv <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  T* coef = scratch;
  hipLaunchKernelGGL(slice_acq_pixel_coef<T>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, transforms, psf, grad_slices,
                     slices_mask, coef, (T*)nullptr, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, 1);
  if (grad_vol != nullptr)
    hipLaunchKernelGGL(slice_acq_adjoint_gather<T>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, transforms, psf, coef,
                       (const T*)nullptr, vol_mask, grad_vol, (T*)nullptr, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, 0, 0);
  if (grad_transforms != nullptr)
    hipLaunchKernelGGL(slice_acq_bwd_transforms<T>, dim3((unsigned)n), dim3(256), 0, st, transforms, vol, vol_mask, psf, coef,
                       grad_transforms, D, H, W, d_p, h_p, w_p, h, w, res_slice);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------ interp_psf mode of A^T and of both backwards
// `interp_psf = true` (nearest voxel + the PSF re-interpolated at the voxel's offset from the pixel centre) exists in all
// four reference kernels (slice_acq_cuda_kernel.cu:71-109 forward, :229-257 / :279-370 backward, :526-572 / :582-606
// adjoint, :754-836 adjoint backward), and no caller of the reference ever switches it on.  It is built here for
// completeness of the module, one thread per slice pixel exactly as the reference formulates it; the scatters are
// global atomics (memory-side on MI355X, ~16 G/s: fine for a mode nothing iterates on - the linear mode above is the
// one that was re-formulated as a gather).
template <typename T> struct InterpTap { int iv, xr, yr, zr; T pw, gx, gy, gz; };

// One PSF tap at (x, y, z) already known to lie inside the volume.  false: the rounded voxel's offset falls outside the
// PSF support.  GRAD: also d pw / d (x_psf, y_psf, z_psf), the corner sums of the reference (.cu:311-352).
template <typename T, bool GRAD>
__device__ __forceinline__ bool interp_tap(const T* __restrict__ t, const PixelGeom<T>& g, const T* __restrict__ psf, int d_p,
                                           int h_p, int w_p, T x, T y, T z, int Sy, int Sz, InterpTap<T>& o) {
  o.xr = (int)floor(x + (T)0.5); o.yr = (int)floor(y + (T)0.5); o.zr = (int)floor(z + (T)0.5);
  o.iv = o.zr * Sz + o.yr * Sy + o.xr;
  const T dx = o.xr - g.xc, dy = o.yr - g.yc, dz = o.zr - g.zc;
  const T xp = t[0] * dx + t[4] * dy + t[8] * dz + (w_p - 1) / (T)2.0;
  const T yp = t[1] * dx + t[5] * dy + t[9] * dz + (h_p - 1) / (T)2.0;
  const T zp = t[2] * dx + t[6] * dy + t[10] * dz + (d_p - 1) / (T)2.0;
  if (xp < 0 || yp < 0 || zp < 0 || xp >= w_p - 1 || yp >= h_p - 1 || zp >= d_p - 1) return false;
  const int xf = (int)floor(xp), yf = (int)floor(yp), zf = (int)floor(zp);
  const T wx = xp - xf, wy = yp - yf, wz = zp - zf;
  const T* p0 = psf + (zf * h_p + yf) * w_p + xf;
  const int py = w_p, pz = w_p * h_p;
  const T c000 = p0[0], c100 = p0[1], c010 = p0[py], c001 = p0[pz], c110 = p0[1 + py], c101 = p0[1 + pz], c011 = p0[py + pz],
          c111 = p0[1 + py + pz];
  T pw = (T)0.0;  // corner order of the reference: 000, 100, 010, 001, 110, 101, 011, 111
  pw += (1 - wx) * (1 - wy) * (1 - wz) * c000;
  pw += wx * (1 - wy) * (1 - wz) * c100;
  pw += (1 - wx) * wy * (1 - wz) * c010;
  pw += (1 - wx) * (1 - wy) * wz * c001;
  pw += wx * wy * (1 - wz) * c110;
  pw += wx * (1 - wy) * wz * c101;
  pw += (1 - wx) * wy * wz * c011;
  pw += wx * wy * wz * c111;
  o.pw = pw;
  if constexpr (GRAD) {
    T gx = (T)0.0, gy = (T)0.0, gz = (T)0.0;
    gx -= (1 - wy) * (1 - wz) * c000; gy -= (1 - wx) * (1 - wz) * c000; gz -= (1 - wx) * (1 - wy) * c000;
    gx += (1 - wy) * (1 - wz) * c100; gy -= wx * (1 - wz) * c100;       gz -= wx * (1 - wy) * c100;
    gx -= wy * (1 - wz) * c010;       gy += (1 - wx) * (1 - wz) * c010; gz -= (1 - wx) * wy * c010;
    gx -= (1 - wy) * wz * c001;       gy -= (1 - wx) * wz * c001;       gz += (1 - wx) * (1 - wy) * c001;
    gx += wy * (1 - wz) * c110;       gy += wx * (1 - wz) * c110;       gz -= wx * wy * c110;
    gx += (1 - wy) * wz * c101;       gy -= wx * wz * c101;             gz += wx * (1 - wy) * c101;
    gx -= wy * wz * c011;             gy += (1 - wx) * wz * c011;       gz += (1 - wx) * wy * c011;
    gx += wy * wz * c111;             gy += wx * wz * c111;             gz += wx * wy * c111;
    o.gx = gx; o.gy = gy; o.gz = gz;
  }
  return true;
}

// the taps of a pixel: body(tap) for every non-zero PSF tap whose position lies inside the volume and whose rounded voxel
// lies inside the PSF support
template <typename T, bool GRAD, typename Body>
__device__ __forceinline__ void for_interp_taps(const T* __restrict__ t, const PixelGeom<T>& g, const T* __restrict__ psf, int d_p,
                                                int h_p, int w_p, int D, int H, int W, Body body) {
  const int Sy = W, Sz = H * W;
  int ip = 0;
  for (int iz = -d_p / 2; iz < (d_p + 1) / 2; ++iz)
    for (int iyp = -h_p / 2; iyp < (h_p + 1) / 2; ++iyp)
      for (int ixp = -w_p / 2; ixp < (w_p + 1) / 2; ++ixp, ++ip) {
        if (psf[ip] == (T)0.0) continue;
        const T x = g.xc + t[0] * ixp + t[1] * iyp + t[2] * iz;
        const T y = g.yc + t[4] * ixp + t[5] * iyp + t[6] * iz;
        const T z = g.zc + t[8] * ixp + t[9] * iyp + t[10] * iz;
        if (x < 0 || y < 0 || z < 0 || x >= W - 1 || y >= H - 1 || z >= D - 1) continue;
        InterpTap<T> tap;
        if (!interp_tap<T, GRAD>(t, g, psf, d_p, h_p, w_p, x, y, z, Sy, Sz, tap)) continue;
        body(tap);
      }
}

// the 12 pose sums of a pixel -> grad_transforms of its slice (wave reduction, one atomic per wave and entry)
template <typename T>
__device__ __forceinline__ void add_pose_grad(T* __restrict__ grad_transforms, int in, const T (&gt)[12], bool uniform_slice) {
  if (uniform_slice) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const T s = wave_sum_any(gt[k]);
      if ((threadIdx.x & 63) == 0 && s != (T)0.0) atomicAdd(grad_transforms + (size_t)in * 12 + k, s);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (gt[k] != (T)0.0) atomicAdd(grad_transforms + (size_t)in * 12 + k, gt[k]);
  }
}

// backward of A (.cu:173-470, interp branch): grad_vol[voxel] += pw gs,  grad_transforms from d pw / d pose
template <typename T>
__global__ __launch_bounds__(256) void slice_acq_bwd_interp(const T* __restrict__ transforms, const T* __restrict__ vol,
                                                            const uint8_t* __restrict__ vol_mask, const T* __restrict__ psf,
                                                            const T* __restrict__ grad_slices, const uint8_t* __restrict__ slices_mask,
                                                            T* __restrict__ grad_vol, T* __restrict__ grad_transforms, int D, int H,
                                                            int W, int d_p, int h_p, int w_p, int n, int h, int w, T res_slice) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n * h * w;
  const int64_t cidx = idx < total ? idx : total - 1;
  const int ix = cidx % w, iy = (cidx / w) % h, in = cidx / ((int64_t)h * w);
  const T* t = transforms + (size_t)in * 12;
  T gt[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) gt[k] = (T)0.0;
  T gs = idx < total && (slices_mask == nullptr || slices_mask[idx]) ? grad_slices[idx] : (T)0.0;
  if (gs != (T)0.0) {
    const PixelGeom<T> g = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    T weight = (T)0.0;  // pass 1: no vol_mask here (.cu:229-257)
    for_interp_taps<T, false>(t, g, psf, d_p, h_p, w_p, D, H, W, [&](const InterpTap<T>& tap) { weight += tap.pw; });
    if (weight != (T)0.0) {
      gs /= weight;
      const T cx = (W - 1) / (T)2.0, cy = (H - 1) / (T)2.0, cz = (D - 1) / (T)2.0;
      for_interp_taps<T, true>(t, g, psf, d_p, h_p, w_p, D, H, W, [&](const InterpTap<T>& tap) {
        if (vol_mask != nullptr && !vol_mask[tap.iv]) return;
        if (grad_vol != nullptr) atomicAdd(grad_vol + tap.iv, tap.pw * gs);
        if (grad_transforms != nullptr) {
          const T tmp = gs * vol[tap.iv];
          const T dx = tap.gx * tmp, dy = tap.gy * tmp, dz = tap.gz * tmp;
          gt[0] += dx * (tap.xr - cx); gt[1] += dy * (tap.xr - cx); gt[2] += dz * (tap.xr - cx); gt[3] -= dx;
          gt[4] += dx * (tap.yr - cy); gt[5] += dy * (tap.yr - cy); gt[6] += dz * (tap.yr - cy); gt[7] -= dy;
          gt[8] += dx * (tap.zr - cz); gt[9] += dy * (tap.zr - cz); gt[10] += dz * (tap.zr - cz); gt[11] -= dz;
        }
      });
    }
  }
  if (grad_transforms != nullptr) {
    // a wave lies in one slice iff its first and last pixel do (then one atomic per wave instead of 64)
    const int64_t first = idx - (threadIdx.x & 63), last = first + 63;
    const bool uniform = last < total && first / ((int64_t)h * w) == last / ((int64_t)h * w);
    add_pose_grad(grad_transforms, in, gt, uniform);
  }
}

// A^T (.cu:472-670, interp branch): vol[voxel] += pw / weight s, vol_weight[voxel] += pw / weight for weight >= 0.5
template <typename T>
__global__ __launch_bounds__(256) void slice_acq_adjoint_interp(const T* __restrict__ transforms, const T* __restrict__ psf,
                                                                const T* __restrict__ slices, const uint8_t* __restrict__ slices_mask,
                                                                const uint8_t* __restrict__ vol_mask, T* __restrict__ vol,
                                                                T* __restrict__ vol_weight, int D, int H, int W, int d_p, int h_p,
                                                                int w_p, int n, int h, int w, T res_slice) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * h * w) return;
  if (slices_mask != nullptr && !slices_mask[idx]) return;
  const T s = slices[idx];
  const int ix = idx % w, iy = (idx / w) % h, in = idx / ((int64_t)h * w);
  const T* t = transforms + (size_t)in * 12;
  const PixelGeom<T> g = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
  T weight = (T)0.0;
  for_interp_taps<T, false>(t, g, psf, d_p, h_p, w_p, D, H, W, [&](const InterpTap<T>& tap) { weight += tap.pw; });
  if (weight < (T)0.5) return;  // border
  for_interp_taps<T, false>(t, g, psf, d_p, h_p, w_p, D, H, W, [&](const InterpTap<T>& tap) {
    if (vol_mask != nullptr && !vol_mask[tap.iv]) return;
    const T pn = tap.pw / weight;
    atomicAdd(vol + tap.iv, pn * s);
    if (vol_weight != nullptr) atomicAdd(vol_weight + tap.iv, pn);
  });
}

template <typename T>
__global__ void slice_acq_equalize_vol(T* __restrict__ vol, const T* __restrict__ vol_weight, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && vol_weight[i] > (T)0.0) vol[i] /= vol_weight[i];
}

// backward of A^T (.cu:695-950, interp branch): grad_slices = sum pw g[voxel] / sum pw,  pose gradient from d pw / d pose
template <typename T>
__global__ __launch_bounds__(256) void slice_acq_adjoint_bwd_interp(const T* __restrict__ transforms, const T* __restrict__ grad_vol,
                                                                    const T* __restrict__ psf, const T* __restrict__ slices,
                                                                    const uint8_t* __restrict__ slices_mask, const T* __restrict__ vol,
                                                                    const uint8_t* __restrict__ vol_mask, T* __restrict__ grad_slices,
                                                                    T* __restrict__ grad_transforms, int D, int H, int W, int d_p,
                                                                    int h_p, int w_p, int n, int h, int w, T res_slice) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n * h * w;
  const int64_t cidx = idx < total ? idx : total - 1;
  const int ix = cidx % w, iy = (cidx / w) % h, in = cidx / ((int64_t)h * w);
  const T* t = transforms + (size_t)in * 12;
  T gt[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) gt[k] = (T)0.0;
  if (idx < total && (slices_mask == nullptr || slices_mask[idx])) {
    const PixelGeom<T> g = pixel_geom(t, ix, iy, h, w, res_slice, D, H, W);
    const T sv = slices[idx];
    const T cx = (W - 1) / (T)2.0, cy = (H - 1) / (T)2.0, cz = (D - 1) / (T)2.0;
    T val = (T)0.0, weight = (T)0.0;
    T acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = (T)0.0;
    // the reference tests the voxel mask before the PSF support; both only skip the tap, the order does not matter
    for_interp_taps<T, true>(t, g, psf, d_p, h_p, w_p, D, H, W, [&](const InterpTap<T>& tap) {
      if (vol_mask != nullptr && !vol_mask[tap.iv]) return;
      const T gv = grad_vol[tap.iv];
      if (grad_transforms != nullptr) {
        const T s = (vol == nullptr ? sv : sv - vol[tap.iv]) * gv;
        const T dx = tap.gx * s, dy = tap.gy * s, dz = tap.gz * s;
        acc[0] += dx * (tap.xr - cx); acc[1] += dy * (tap.xr - cx); acc[2] += dz * (tap.xr - cx); acc[3] -= dx;
        acc[4] += dx * (tap.yr - cy); acc[5] += dy * (tap.yr - cy); acc[6] += dz * (tap.yr - cy); acc[7] -= dy;
        acc[8] += dx * (tap.zr - cz); acc[9] += dy * (tap.zr - cz); acc[10] += dz * (tap.zr - cz); acc[11] -= dz;
      }
      val += tap.pw * gv;
      weight += tap.pw;
    });
    if (weight > (T)0.0) {
      if (grad_slices != nullptr) grad_slices[idx] = val / weight;
#pragma unroll
      for (int k = 0; k < 12; ++k) gt[k] = acc[k] / weight;
    }
  }
  if (grad_transforms != nullptr) {
    const int64_t first = idx - (threadIdx.x & 63), last = first + 63;
    const bool uniform = last < total && first / ((int64_t)h * w) == last / ((int64_t)h * w);
    add_pose_grad(grad_transforms, in, gt, uniform);
  }
}

// host side of the interp_psf mode.  Outputs are accumulated into: the callers pass zero-filled vol / vol_weight /
// grad_vol / grad_transforms / grad_slices (as the reference's host functions allocate them, .cu:1000-1120)
template <typename T>
int slice_acq_backward_interp_impl(const T* transforms, const T* vol, const uint8_t* vol_mask, const T* psf, const T* grad_slices,
                                   const uint8_t* slices_mask, T* grad_vol, T* grad_transforms, int D, int H, int W, int d_p,
                                   int h_p, int w_p, int n, int h, int w, T res_slice, void* stream) {
  const int64_t np = (int64_t)n * h * w;
  if (np <= 0 || (int64_t)D * H * W <= 0) return 0;
  if (d_p < 2 || h_p < 2 || w_p < 2) return 0;  // no interior cell in the PSF: every tap is skipped
  hipLaunchKernelGGL(slice_acq_bwd_interp<T>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)stream, transforms, vol,
                     vol_mask, psf, grad_slices, slices_mask, grad_vol, grad_transforms, D, H, W, d_p, h_p, w_p, n, h, w, res_slice);
  return (int)hipGetLastError();
}
template <typename T>
int slice_acq_adjoint_forward_interp_impl(const T* transforms, const T* psf, const T* slices, const uint8_t* slices_mask,
                                          const uint8_t* vol_mask, T* vol, T* vol_weight, int D, int H, int W, int d_p, int h_p,
                                          int w_p, int n, int h, int w, T res_slice, int equalize, void* stream) {
  const int64_t np = (int64_t)n * h * w, nv = (int64_t)D * H * W;
  if (np <= 0 || nv <= 0) return 0;
  if (equalize && vol_weight == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (d_p >= 2 && h_p >= 2 && w_p >= 2)
    hipLaunchKernelGGL(slice_acq_adjoint_interp<T>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, transforms, psf, slices,
                       slices_mask, vol_mask, vol, vol_weight, D, H, W, d_p, h_p, w_p, n, h, w, res_slice);
  if (equalize)
    hipLaunchKernelGGL(slice_acq_equalize_vol<T>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, vol, (const T*)vol_weight, nv);
  return (int)hipGetLastError();
}
template <typename T>
int slice_acq_adjoint_backward_interp_impl(const T* transforms, T* grad_vol, const T* vol_weight, const uint8_t* vol_mask,
                                           const T* psf, const T* slices, const uint8_t* slices_mask, const T* vol, T* grad_slices,
                                           T* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p, int n, int h, int w,
                                           T res_slice, int equalize, void* stream) {
  const int64_t np = (int64_t)n * h * w;
  if (np <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (equalize) {
    if (vol_weight == nullptr || vol == nullptr) return (int)hipErrorInvalidValue;
    const int64_t nv = (int64_t)D * H * W;
    hipLaunchKernelGGL(slice_acq_equalize_grad<T>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, grad_vol, vol_weight, nv);
  }
  if (d_p < 2 || h_p < 2 || w_p < 2) return (int)hipGetLastError();
  hipLaunchKernelGGL(slice_acq_adjoint_bwd_interp<T>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, transforms,
                     (const T*)grad_vol, psf, slices, slices_mask, equalize ? vol : (const T*)nullptr, vol_mask, grad_slices,
                     grad_transforms, D, H, W, d_p, h_p, w_p, n, h, w, res_slice);
  return (int)hipGetLastError();
}

}  // namespace

// ---- C entry points: float (the path every caller of the reference uses) and double (AT_DISPATCH_FLOATING_TYPES,
//      slice_acq_cuda_kernel.cu:970-1114)
#define NESVOR_SLICE_ACQ_ENTRY(SUFFIX, T)                                                                                      \
  extern "C" int nesvor_slice_acq_forward##SUFFIX(const T* transforms, const T* vol, const uint8_t* vol_mask,                  \
                                                  const uint8_t* slices_mask, const T* psf, T* slices, T* slices_weight, int D, \
                                                  int H, int W, int d_p, int h_p, int w_p, int n, int h, int w, T res_slice,    \
                                                  int interp_psf, void* stream) {                                               \
    return slice_acq_forward_impl<T>(transforms, vol, vol_mask, slices_mask, psf, slices, slices_weight, D, H, W, d_p, h_p,    \
                                     w_p, n, h, w, res_slice, interp_psf, stream);                                              \
  }                                                                                                                            \
  extern "C" int nesvor_slice_acq_adjoint_forward##SUFFIX(const T* transforms, const T* psf, const T* slices,                  \
                                                          const uint8_t* slices_mask, const uint8_t* vol_mask, T* vol,          \
                                                          T* vol_weight, T* scratch, int D, int H, int W, int d_p, int h_p,     \
                                                          int w_p, int n, int h, int w, T res_slice, int equalize,              \
                                                          void* stream) {                                                       \
    return slice_acq_adjoint_forward_impl<T>(transforms, psf, slices, slices_mask, vol_mask, vol, vol_weight, scratch, D, H, W, \
                                             d_p, h_p, w_p, n, h, w, res_slice, equalize, stream);                              \
  }                                                                                                                            \
  extern "C" int nesvor_slice_acq_backward##SUFFIX(const T* transforms, const T* vol, const uint8_t* vol_mask, const T* psf,   \
                                                   const T* grad_slices, const uint8_t* slices_mask, T* grad_vol,               \
                                                   T* grad_transforms, T* scratch, int D, int H, int W, int d_p, int h_p,       \
                                                   int w_p, int n, int h, int w, T res_slice, void* stream) {                   \
    return slice_acq_backward_impl<T>(transforms, vol, vol_mask, psf, grad_slices, slices_mask, grad_vol, grad_transforms,      \
                                      scratch, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, stream);                             \
  }                                                                                                                            \
  extern "C" int nesvor_slice_acq_adjoint_backward##SUFFIX(const T* transforms, T* grad_vol, const T* vol_weight,              \
                                                           const uint8_t* vol_mask, const T* psf, const T* slices,              \
                                                           const uint8_t* slices_mask, const T* vol, T* grad_slices,            \
                                                           T* grad_transforms, int D, int H, int W, int d_p, int h_p, int w_p,  \
                                                           int n, int h, int w, T res_slice, int equalize, void* stream) {      \
    return slice_acq_adjoint_backward_impl<T>(transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol,        \
                                              grad_slices, grad_transforms, D, H, W, d_p, h_p, w_p, n, h, w, res_slice,         \
                                              equalize, stream);                                                                \
  }                                                                                                                            \
  /* interp_psf = true: same arguments (no scratch), outputs zero-filled by the caller and accumulated into */                 \
  extern "C" int nesvor_slice_acq_adjoint_forward_interp##SUFFIX(const T* transforms, const T* psf, const T* slices,           \
                                                                 const uint8_t* slices_mask, const uint8_t* vol_mask, T* vol,   \
                                                                 T* vol_weight, int D, int H, int W, int d_p, int h_p, int w_p, \
                                                                 int n, int h, int w, T res_slice, int equalize, void* stream) {\
    return slice_acq_adjoint_forward_interp_impl<T>(transforms, psf, slices, slices_mask, vol_mask, vol, vol_weight, D, H, W,   \
                                                    d_p, h_p, w_p, n, h, w, res_slice, equalize, stream);                       \
  }                                                                                                                            \
  extern "C" int nesvor_slice_acq_backward_interp##SUFFIX(const T* transforms, const T* vol, const uint8_t* vol_mask,          \
                                                          const T* psf, const T* grad_slices, const uint8_t* slices_mask,       \
                                                          T* grad_vol, T* grad_transforms, int D, int H, int W, int d_p,        \
                                                          int h_p, int w_p, int n, int h, int w, T res_slice, void* stream) {   \
    return slice_acq_backward_interp_impl<T>(transforms, vol, vol_mask, psf, grad_slices, slices_mask, grad_vol,                \
                                             grad_transforms, D, H, W, d_p, h_p, w_p, n, h, w, res_slice, stream);              \
  }                                                                                                                            \
  extern "C" int nesvor_slice_acq_adjoint_backward_interp##SUFFIX(const T* transforms, T* grad_vol, const T* vol_weight,       \
                                                                  const uint8_t* vol_mask, const T* psf, const T* slices,       \
                                                                  const uint8_t* slices_mask, const T* vol, T* grad_slices,     \
                                                                  T* grad_transforms, int D, int H, int W, int d_p, int h_p,    \
                                                                  int w_p, int n, int h, int w, T res_slice, int equalize,      \
                                                                  void* stream) {                                               \
    return slice_acq_adjoint_backward_interp_impl<T>(transforms, grad_vol, vol_weight, vol_mask, psf, slices, slices_mask, vol, \
                                                     grad_slices, grad_transforms, D, H, W, d_p, h_p, w_p, n, h, w, res_slice,  \
                                                     equalize, stream);                                                         \
  }
NESVOR_SLICE_ACQ_ENTRY(, float)
NESVOR_SLICE_ACQ_ENTRY(_f64, double)
#undef NESVOR_SLICE_ACQ_ENTRY
