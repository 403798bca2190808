// Multi-resolution hash-grid encoding for gfx950: forward, parameter-gradient
// scatter and input gradient.  Algorithm statement: oracle/hashgrid.py (the
// published Instant-NGP / tiny-cuda-nn HashGrid; the reference reaches it via
// tcnn.Encoding at nesvor/nesvor/models.py:25).
//
// Mapping to the machine
// ----------------------
// * grid = (ceil(N/256), L): blockIdx.y is the level, so everything level
//   dependent (scale, resolution, table slice) is wave-uniform and lives in
//   SGPRs, and the dispatcher walks the levels one after another: at any time
//   the chip works on one or two levels whose table slice (<= 4 MiB at
//   T = 2^19, F = 2) is what the per-XCD L2s hold.
// * one thread per (sample, level); the 8 corner fetches are 8 independent
//   F*4-byte loads in flight per lane; samples of one pixel are contiguous
//   (layout (B,S,.)), so a 64-lane wave covers one PSF cloud: coarse levels
//   collapse to a handful of distinct cache lines per load instruction.
// * encoded features are produced/consumed FEATURE-MAJOR (L*F, N) on the fused
//   path so that each lane writes/reads consecutive addresses (coalesced
//   dwordx1/x2 streams); the row-major (N, L*F) layout tinycudann hands to
//   PyTorch is supported for the drop-in module.
// * backward: hardware fp32 atomics (global_atomic_add_f32, built with
//   -munsafe-fp-atomics).  A PSF cloud hits the same cell with most lanes of a
//   wave at the coarse levels, so lanes that share the wave leader's cell are
//   first summed across the wave and committed by one lane
//   (16 atomics instead of 16 x group size); lanes in sparsely shared cells
//   fall through to plain per-lane atomics.
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/nesvor_hip.h"

namespace {

constexpr uint32_t kPrimeY = 2654435761u;
constexpr uint32_t kPrimeZ = 805459861u;

struct LevelParams {
  float scale;
  uint32_t res, size, offset, hashed;
};

__device__ __forceinline__ LevelParams load_level(const nesvor_grid_t& g, int level) {
  LevelParams p;
  p.scale = g.scale[level];
  p.res = g.res[level];
  p.size = g.size[level];
  p.offset = g.offset[level];
  p.hashed = g.hashed[level];
  return p;
}

__device__ __forceinline__ uint32_t corner_index(const LevelParams& p, uint32_t x, uint32_t y, uint32_t z) {
  if (p.hashed) {
    const uint32_t h = x ^ (y * kPrimeY) ^ (z * kPrimeZ);
    return ((p.size & (p.size - 1)) == 0) ? (h & (p.size - 1)) : (h % p.size);
  }
  uint32_t idx = x + y * p.res + z * p.res * p.res;
  if (idx >= p.size) idx %= p.size;  // only on the u == 1 face / out-of-range inputs
  return idx;
}

struct CellPos {
  uint32_t gx, gy, gz;
  float wx, wy, wz;
};

__device__ __forceinline__ CellPos locate(const LevelParams& p, float ux, float uy, float uz) {
  CellPos c;
  const float px = fmaf(p.scale, ux, 0.5f), py = fmaf(p.scale, uy, 0.5f), pz = fmaf(p.scale, uz, 0.5f);
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  c.gx = (uint32_t)(int)fx; c.gy = (uint32_t)(int)fy; c.gz = (uint32_t)(int)fz;
  c.wx = px - fx; c.wy = py - fy; c.wz = pz - fz;
  return c;
}

template <int F>
__device__ __forceinline__ void load_feat(const float* __restrict__ p, float (&v)[F]) {
  if constexpr (F == 1) {
    v[0] = p[0];
  } else if constexpr (F == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
#pragma unroll
    for (int k = 0; k < F; k += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + k);
      v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
    }
  }
}

// ------------------------------------------------------------------ forward
template <int F, int LAYOUT>
__global__ __launch_bounds__(256) void hashgrid_fwd(const nesvor_grid_t g, const float* __restrict__ u,
                                                    const float* __restrict__ table, float* __restrict__ pe,
                                                    int64_t N) {
  const int level = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const LevelParams p = load_level(g, level);
  const float ux = u[3 * i], uy = u[3 * i + 1], uz = u[3 * i + 2];
  const CellPos c = locate(p, ux, uy, uz);
  const float* tab = table + (size_t)p.offset * F;
  float v[8][F];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t idx = corner_index(p, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + (k >> 2));
    load_feat<F>(tab + (size_t)idx * F, v[k]);
  }
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = ((k & 1) ? c.wx : 1.f - c.wx) * (((k >> 1) & 1) ? c.wy : 1.f - c.wy) * ((k >> 2) ? c.wz : 1.f - c.wz);
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = fmaf(w, v[k][f], acc[f]);
  }
  const int E = g.n_levels * F;
  if constexpr (LAYOUT == NESVOR_LAYOUT_ROW_MAJOR) {
    float* o = pe + (size_t)i * E + level * F;
#pragma unroll
    for (int f = 0; f < F; ++f) o[f] = acc[f];
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) pe[(size_t)(level * F + f) * N + i] = acc[f];
  }
}

// ----------------------------------------------------------------- backward
// grad_table += scatter(w * dy);  optionally grad_u[i] += d/du (per level, atomics)
template <int F, int LAYOUT, bool INPUT_GRAD, bool MERGE>
__global__ __launch_bounds__(256) void hashgrid_bwd(const nesvor_grid_t g, const float* __restrict__ u,
                                                    const float* __restrict__ table, const float* __restrict__ dpe,
                                                    float* __restrict__ grad_table, float* __restrict__ grad_u,
                                                    int64_t N) {
  const int level = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < N;
  const LevelParams p = load_level(g, level);
  const int64_t ii = valid ? i : N - 1;
  const float ux = u[3 * ii], uy = u[3 * ii + 1], uz = u[3 * ii + 2];
  const CellPos c = locate(p, ux, uy, uz);
  const int E = g.n_levels * F;
  float dy[F];
  if constexpr (LAYOUT == NESVOR_LAYOUT_ROW_MAJOR) {
    const float* o = dpe + (size_t)ii * E + level * F;
#pragma unroll
    for (int f = 0; f < F; ++f) dy[f] = valid ? o[f] : 0.f;
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) dy[f] = valid ? dpe[(size_t)(level * F + f) * N + ii] : 0.f;
  }
  uint32_t idx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) idx[k] = corner_index(p, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + (k >> 2));

  if constexpr (INPUT_GRAD) {
    // d y_f / d u_d = scale * sum_corners sign_d * w_other * table[corner][f]
    const float* tab = table + (size_t)p.offset * F;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    float v[8][F];
#pragma unroll
    for (int k = 0; k < 8; ++k) load_feat<F>(tab + (size_t)idx[k] * F, v[k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float fd = 0.f;
#pragma unroll
      for (int f = 0; f < F; ++f) fd = fmaf(v[k][f], dy[f], fd);
      const float wxk = (k & 1) ? c.wx : 1.f - c.wx, wyk = ((k >> 1) & 1) ? c.wy : 1.f - c.wy, wzk = (k >> 2) ? c.wz : 1.f - c.wz;
      gx += ((k & 1) ? fd : -fd) * wyk * wzk;
      gy += (((k >> 1) & 1) ? fd : -fd) * wxk * wzk;
      gz += ((k >> 2) ? fd : -fd) * wxk * wyk;
    }
    if (valid) {
      atomicAdd(grad_u + 3 * i + 0, p.scale * gx);
      atomicAdd(grad_u + 3 * i + 1, p.scale * gy);
      atomicAdd(grad_u + 3 * i + 2, p.scale * gz);
    }
  }

  float* gt = grad_table + (size_t)p.offset * F;
  float val[8][F];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = ((k & 1) ? c.wx : 1.f - c.wx) * (((k >> 1) & 1) ? c.wy : 1.f - c.wy) * ((k >> 2) ? c.wz : 1.f - c.wz);
#pragma unroll
    for (int f = 0; f < F; ++f) val[k][f] = w * dy[f];
  }

  bool pending = valid;
  if constexpr (MERGE) {
    // Cell-keyed wave pre-reduction: while a large share of the remaining lanes
    // sits in the leader's cell, sum that group across the wave and let the
    // leader commit it.
    const int lane = threadIdx.x & 63;
    for (int round = 0; round < 4; ++round) {
      const unsigned long long rem = __ballot(pending);
      if (rem == 0) break;
      const int leader = __ffsll((long long)rem) - 1;
      const uint32_t lx = __shfl(c.gx, leader, 64), ly = __shfl(c.gy, leader, 64), lz = __shfl(c.gz, leader, 64);
      const bool mine = pending && c.gx == lx && c.gy == ly && c.gz == lz;
      const unsigned long long grp = __ballot(mine);
      if (__popcll(grp) < 8) break;
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float s = wave_sum(mine ? val[k][f] : 0.f);
          if (lane == leader) atomicAdd(gt + (size_t)idx[k] * F + f, s);
        }
      pending = pending && !mine;
    }
  }
  if (pending) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int f = 0; f < F; ++f) atomicAdd(gt + (size_t)idx[k] * F + f, val[k][f]);
  }
}

template <int F, int LAYOUT>
int launch_fwd(const nesvor_grid_t* g, const float* u, const float* table, float* pe, int64_t N, hipStream_t st) {
  dim3 grid((unsigned)((N + 255) / 256), g->n_levels), block(256);
  hipLaunchKernelGGL((hashgrid_fwd<F, LAYOUT>), grid, block, 0, st, *g, u, table, pe, N);
  return (int)hipGetLastError();
}

template <int F, int LAYOUT>
int launch_bwd(const nesvor_grid_t* g, const float* u, const float* table, const float* dpe, float* gt, float* gu,
               int64_t N, hipStream_t st) {
  dim3 grid((unsigned)((N + 255) / 256), g->n_levels), block(256);
  if (gu != nullptr) {
    hipError_t e = hipMemsetAsync(gu, 0, sizeof(float) * 3 * N, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((hashgrid_bwd<F, LAYOUT, true, true>), grid, block, 0, st, *g, u, table, dpe, gt, gu, N);
  } else {
    hipLaunchKernelGGL((hashgrid_bwd<F, LAYOUT, false, true>), grid, block, 0, st, *g, u, table, dpe, gt, gu, N);
  }
  return (int)hipGetLastError();
}

}  // namespace

#define DISPATCH_F_LAYOUT(FN, ...)                                                              \
  do {                                                                                          \
    const int F_ = grid->n_features;                                                            \
    if (layout == NESVOR_LAYOUT_ROW_MAJOR) {                                                    \
      if (F_ == 1) return FN<1, NESVOR_LAYOUT_ROW_MAJOR>(__VA_ARGS__);                          \
      if (F_ == 2) return FN<2, NESVOR_LAYOUT_ROW_MAJOR>(__VA_ARGS__);                          \
      if (F_ == 4) return FN<4, NESVOR_LAYOUT_ROW_MAJOR>(__VA_ARGS__);                          \
      if (F_ == 8) return FN<8, NESVOR_LAYOUT_ROW_MAJOR>(__VA_ARGS__);                          \
    } else if (layout == NESVOR_LAYOUT_FEATURE_MAJOR) {                                         \
      if (F_ == 1) return FN<1, NESVOR_LAYOUT_FEATURE_MAJOR>(__VA_ARGS__);                      \
      if (F_ == 2) return FN<2, NESVOR_LAYOUT_FEATURE_MAJOR>(__VA_ARGS__);                      \
      if (F_ == 4) return FN<4, NESVOR_LAYOUT_FEATURE_MAJOR>(__VA_ARGS__);                      \
      if (F_ == 8) return FN<8, NESVOR_LAYOUT_FEATURE_MAJOR>(__VA_ARGS__);                      \
    }                                                                                           \
    return (int)hipErrorInvalidValue;                                                           \
  } while (0)

extern "C" int nesvor_hashgrid_forward(const nesvor_grid_t* grid, const float* u, const float* table, float* pe,
                                       int64_t N, int layout, void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  DISPATCH_F_LAYOUT(launch_fwd, grid, u, table, pe, N, (hipStream_t)stream);
}

extern "C" int nesvor_hashgrid_backward(const nesvor_grid_t* grid, const float* u, const float* table,
                                        const float* dpe, float* grad_table, float* grad_u, int64_t N, int layout,
                                        void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  DISPATCH_F_LAYOUT(launch_bwd, grid, u, table, dpe, grad_table, grad_u, N, (hipStream_t)stream);
}
