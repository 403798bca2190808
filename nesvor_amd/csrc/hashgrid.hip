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
// * one thread per (sample, level); samples of one pixel are contiguous
//   (layout (B,S,.)), so a 256-thread workgroup covers one PSF cloud.  Where
//   the lattice box of its cells at the level has <= 1024 vertices, the box is
//   copied into LDS once and the 8 corner fetches are LDS reads; otherwise
//   they are 8 independent F*4-byte global loads in flight per lane.
// * encoded features are produced/consumed FEATURE-MAJOR (L*F, N) on the fused
//   path so that each lane writes/reads consecutive addresses (coalesced
//   dwordx1/x2 streams); the row-major (N, L*F) layout tinycudann hands to
//   PyTorch is supported for the drop-in module.
// * backward: global fp32 atomics are executed memory-side on MI355X (they
//   drop the line from the XCD L2; measured ~16 G atomics/s chip-wide), so a
//   tcnn-style "one atomicAdd per corner" scatter costs 8-16 ms at N = 2^20.
//   The backward is therefore an owner-computes scatter in two launches:
//     (1) hashgrid_bwd_aggregate: one workgroup per 256 consecutive samples
//         (= one PSF cloud), sorted once by Morton code; per level a segmented
//         wave scan on the VALU sums runs of lanes in the same cell; the run
//         tails go through a workgroup-wide merge table in LDS (64-bit fixed
//         point; slots addressed by position in the cloud's lattice box where it
//         fits, by the low bits of the vertex's lattice coordinates - a window
//         over the box - with double hashing on a miss otherwise), and the merged (entry,
//         grad) records are binned by table chunk and appended to that chunk's
//         queue in HBM (one LDS counter op per record, one returning atomic per
//         non-empty (workgroup, chunk) pair, on a counter set private to the
//         XCC the workgroup runs on);
//     (2) hashgrid_bwd_owner: one workgroup per table chunk accumulates its
//         queue into LDS (integer-CAS adds: ds_add_f32 retires only ~1 lane
//         per 3 cycles on gfx950) and adds the chunk to grad_table with
//         plain coalesced read-modify-writes - it is the only writer.
//   Records that do not fit a queue fall back to global atomics, so the
//   result is exact for any input distribution.  A batch whose consecutive
//   points are NOT spatially clustered (NESVOR_LAYOUT_UNCLUSTERED) is first put
//   into the order of a coarse lattice's cells (sort_place / sort_scan /
//   sort_compact below; feature-major d pe re-ordered into rows by
//   gather_dy_rows_kernel) and then takes the same two launches.  The legacy all-atomics
//   kernel is kept as hashgrid_bwd (used for tiny N and as a cross-check).
#include <hip/hip_runtime.h>
#include <mutex>
#include <type_traits>
#include <unordered_map>
#include "common.h"
#include "adam.h"
#include "../../include/nesvor_hip.h"

// Timing ablations (tools/ablate_hashgrid.py): -DNESVOR_ABLATE=<bits> compiles pieces of the aggregation pass out.
// Results are wrong with any bit set; the default build has none.
#ifndef NESVOR_ABLATE
#define NESVOR_ABLATE 0
#endif
#define NESVOR_ABL(bit) ((NESVOR_ABLATE & (bit)) != 0)
#ifndef NESVOR_SORT_SPAN
#define NESVOR_SORT_SPAN 256  // samples sorted together by Morton code: the whole workgroup (64 = per wave and 128: same time within 1 %)
#endif

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
  // a dense level has res^3 <= size < 2^32, i.e. res^2 < 2^22: the full-rate 24-bit multiplies are exact for every
  // in-range vertex (v_mul_lo_u32 issues at a quarter of the rate); out-of-range inputs get a different - but in every
  // kernel the same - garbage index before the modulo
  uint32_t idx = x + __umul24(y, p.res) + __umul24(z, p.res * p.res);
  if (idx >= p.size) idx %= p.size;  // only on the u == 1 face / out-of-range inputs
  return idx;
}

// Inclusive prefix sum over the first 32 lanes of a wave (lane = level; NESVOR_MAX_LEVELS = 32): DPP row_shr inside the two
// 16-lane rows, then row 0's total onto row 1.
__device__ __forceinline__ uint32_t scan32_u32(uint32_t v, int lane) {
#define NESVOR_SCAN_STEP_(SHR)                                                                        \
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 | (SHR), 0xf, 0xf, true);
  NESVOR_SCAN_STEP_(1) NESVOR_SCAN_STEP_(2) NESVOR_SCAN_STEP_(4) NESVOR_SCAN_STEP_(8)
#undef NESVOR_SCAN_STEP_
  const uint32_t row0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
  return v + ((lane >= 16 && lane < 32) ? row0 : 0u);
}

// Round schedule of the per-cloud kernels.  Lane l of the calling wave (all 64 lanes active) holds level l's box volume `vol` (0:
// the box does not fit the LDS table) and chunk count `nch`.  Box levels = the run of levels from `first` with vol != 0; they are
// cut greedily into rounds of consecutive levels whose boxes fit the table together (<= max_slots), whose chunks fit the bucket
// counters (<= max_bkts) and which are at most max_group long.  Returns level l's numbers in lane l: first slot / bucket inside
// its round, the level behind its round and, on a round's first level, the round's totals.  One iteration per ROUND over prefix
// sums (rounds 1-5 walked level by level with run-time v_readlane and single-lane LDS stores: 6 % of a workgroup's life in the
// aggregation pass, tools/hg_timeline.py).
struct RoundSchedule {
  uint32_t slot_off, bkt_off, grp_end, rnd_slots, rnd_bkts;
  int box_end;
};
__device__ __forceinline__ RoundSchedule round_schedule(uint32_t vol, uint32_t nch, int first, int last, int lane, uint32_t max_slots,
                                                        uint32_t max_bkts, int max_group) {
  const unsigned long long no_box = __ballot(vol == 0u) | ~((1ull << last) - 1ull);
  const int e = __builtin_ctzll(no_box & ~((1ull << first) - 1ull));  // first level at or behind `first` that is no box level
  const bool in = lane >= first && lane < e;
  const uint32_t v_in = in ? vol : 0u, n_in = in ? nch : 0u;
  const uint32_t C = scan32_u32(v_in, lane), N = scan32_u32(n_in, lane);
  RoundSchedule r;
  r.slot_off = 0u; r.bkt_off = 0u; r.grp_end = 0u; r.rnd_slots = 0u; r.rnd_bkts = 0u; r.box_end = e;
  int g0 = first;
  while (g0 < e) {  // (uniform: a handful of rounds)
    const uint32_t cb = (uint32_t)__builtin_amdgcn_readlane((int)(C - v_in), g0), nb = (uint32_t)__builtin_amdgcn_readlane((int)(N - n_in), g0);
    const bool ok = lane >= g0 && lane < e && C - cb <= max_slots && N - nb <= max_bkts && lane - g0 < max_group;
    int bnd = __builtin_ctzll((__ballot(!ok) & ~((1ull << g0) - 1ull)) | (1ull << 63));
    bnd = bnd > g0 ? bnd : g0 + 1;  // (a level always fits on its own: vol <= max_slots by construction)
    const bool mine = lane >= g0 && lane < bnd;
    r.slot_off = mine ? (C - v_in) - cb : r.slot_off;
    r.bkt_off = mine ? (N - n_in) - nb : r.bkt_off;
    r.grp_end = mine ? (uint32_t)bnd : r.grp_end;
    const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)C, bnd - 1) - cb, ns = (uint32_t)__builtin_amdgcn_readlane((int)N, bnd - 1) - nb;
    r.rnd_slots = lane == g0 ? cs : r.rnd_slots;
    r.rnd_bkts = lane == g0 ? ns : r.rnd_bkts;
    g0 = bnd;
  }
  return r;
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

// The two corners of a lattice cell that differ in x only, (x, y, z) and (x + 1, y, z), as ONE request where their table
// entries are neighbours (round 6).  Every gather is one L2 transaction, and on input that shares nothing between the lanes of a
// wave - uniform points: 134 M gathers per 2^20 points - the forward runs at the L2s' transaction rate (~0.3 T/s chip-wide:
// 0.40 ms).  On a dense level the pair is adjacent by construction (idx, idx + 1); on a hashed level x ^ (y P1) ^ (z P2) maps an
// even x and x + 1 to entries that differ in bit 0.  F <= 2 (a pair is at most 16 bytes): one load of the pair at the lower
// index - only 4 F-byte aligned, which global loads take - plus, for the lanes whose pair is NOT adjacent (odd x on a hashed
// level, the wrap at the u = 1 face), the aligned pair around i0 and a second request for i1.  Transactions per point and
// level: 4-4.5 (dense) / 6 (hashed) instead of 8.  The values are the same table entries: results do not change.
template <int F>
__device__ __forceinline__ void load_feat_pair(const float* __restrict__ tab, uint32_t i0, uint32_t i1, float (&v0)[F], float (&v1)[F]) {
  if constexpr (F <= 2) {
    const bool adj = (i1 == i0 + 1u) || (i0 == i1 + 1u);
    const uint32_t p = adj ? min(i0, i1) : (i0 & ~1u);   // (level sizes are multiples of 8: the aligned pair around i0 exists)
    float t[2 * F];
    if constexpr (F == 1) {
      struct __attribute__((packed, aligned(4))) P2 { float a, b; };
      const P2 w = *reinterpret_cast<const P2*>(tab + (size_t)p);
      t[0] = w.a; t[1] = w.b;
    } else {
      struct __attribute__((packed, aligned(8))) P4 { float a, b, c, d; };
      const P4 w = *reinterpret_cast<const P4*>(tab + (size_t)p * 2);
      t[0] = w.a; t[1] = w.b; t[2] = w.c; t[3] = w.d;
    }
    const bool second0 = i0 != p;
#pragma unroll
    for (int f = 0; f < F; ++f) { v0[f] = second0 ? t[F + f] : t[f]; v1[f] = second0 ? t[f] : t[F + f]; }
    if (!adj) load_feat<F>(tab + (size_t)i1 * F, v1);
  } else {
    load_feat<F>(tab + (size_t)i0 * F, v0);
    load_feat<F>(tab + (size_t)i1 * F, v1);
  }
}

// ------------------------------------------------------------------ forward
// A workgroup = 256 consecutive samples = (for S = 256) one PSF cloud.  Where the lattice box spanned by its cells at
// this level has at most kFwdSlots vertices (the coarse and middle levels), the box is copied into LDS once - every
// vertex fetched once per workgroup instead of once per touching sample - and the 8 corner reads are LDS reads; the
// fine levels gather from global memory as before.  Cells are biased by +1 so that the unsigned min / max also order
// the cell -1 of points just outside the unit cube.
template <int F, int LAYOUT>
__global__ __launch_bounds__(256) void hashgrid_fwd(const nesvor_grid_t g, const float* __restrict__ u,
                                                    const float* __restrict__ table, float* __restrict__ pe,
                                                    int64_t N, int box_cache, float* __restrict__ pe_absmax) {
  constexpr int kFwdSlots = F <= 2 ? 1024 : (F == 4 ? 512 : 256);  // 8 KB of LDS (2048 slots: no faster)
  __shared__ uint32_t wbox[4][6];
  __shared__ __attribute__((aligned(16))) float cache[kFwdSlots * F];
  const int level = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  const bool valid = i < N;
  const int64_t ii = valid ? i : N - 1;
  const LevelParams p = load_level(g, level);
  const float ux = u[3 * ii], uy = u[3 * ii + 1], uz = u[3 * ii + 2];
  const CellPos c = locate(p, ux, uy, uz);
  const float* tab = table + (size_t)p.offset * F;
  float v[8][F];
  bool cached = false;
  if (box_cache) {  // kernel argument: uniform
    const uint32_t bx = c.gx + 1u, by = c.gy + 1u, bz = c.gz + 1u;
    const uint32_t lo0 = wave_min_u32_dpp(bx), lo1 = wave_min_u32_dpp(by), lo2 = wave_min_u32_dpp(bz);
    const uint32_t hi0 = wave_max_u32_dpp(bx), hi1 = wave_max_u32_dpp(by), hi2 = wave_max_u32_dpp(bz);
    if (lane == 0) {
      uint32_t* w = wbox[tid >> 6];
      w[0] = lo0; w[1] = lo1; w[2] = lo2; w[3] = hi0; w[4] = hi1; w[5] = hi2;
    }
    __syncthreads();
    uint32_t lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = min(min(wbox[0][d], wbox[1][d]), min(wbox[2][d], wbox[3][d]));
      hi[d] = max(max(wbox[0][3 + d], wbox[1][3 + d]), max(wbox[2][3 + d], wbox[3][3 + d]));
      lo[d] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo[d]);
      hi[d] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi[d]);
    }
    const uint32_t ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const uint32_t nx = ex + 2u, ny = ey + 2u, nz = ez + 2u, nxy = nx * ny;
    cached = ex < (uint32_t)kFwdSlots && ey < (uint32_t)kFwdSlots && ez < (uint32_t)kFwdSlots &&
             (uint64_t)nxy * nz <= (uint64_t)kFwdSlots;
    if (cached) {  // workgroup-uniform
      const uint32_t vol = nxy * nz;
      const float inv_nxy = 1.f / (float)nxy, inv_nx = 1.f / (float)nx;
      for (uint32_t s = tid; s < vol; s += 256) {
        const uint32_t z = (uint32_t)(((float)s + 0.5f) * inv_nxy);
        const uint32_t r = s - __umul24(z, nxy);
        const uint32_t y = (uint32_t)(((float)r + 0.5f) * inv_nx);
        const uint32_t idx = corner_index(p, lo[0] - 1u + (r - __umul24(y, nx)), lo[1] - 1u + y, lo[2] - 1u + z);
        float t[F];
        load_feat<F>(tab + (size_t)idx * F, t);
#pragma unroll
        for (int f = 0; f < F; ++f) cache[s * F + f] = t[f];
      }
      __syncthreads();
      const uint32_t s0 = __umul24(__umul24(bz - lo[2], ny) + (by - lo[1]), nx) + (bx - lo[0]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t s = s0 + (k & 1) + ((k >> 1) & 1) * nx + (k >> 2) * nxy;
#pragma unroll
        for (int f = 0; f < F; ++f) v[k][f] = cache[s * F + f];
      }
    }
  }
  if (!cached) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {  // x-pairs: one request where the two entries are neighbours (load_feat_pair)
      const uint32_t i0 = corner_index(p, c.gx, c.gy + ((k >> 1) & 1), c.gz + (k >> 2));
      const uint32_t i1 = corner_index(p, c.gx + 1u, c.gy + ((k >> 1) & 1), c.gz + (k >> 2));
      load_feat_pair<F>(tab, i0, i1, v[k], v[k + 1]);
    }
  }
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  if (valid) {
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
  if (pe_absmax != nullptr) {  // (all lanes arrive here: the wave-wide maximum)
    float m = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) m = fmaxf(m, fabsf(acc[f]));
    publish_absmax_wg(pe_absmax, m);
  }
}

// ------------------------------------------------------- forward, all levels of a PSF cloud in one workgroup
// One workgroup = 256 consecutive samples (for S = 256: one PSF cloud), ALL levels.  The per-(cloud, level) blocks of
// hashgrid_fwd are bound by two serialised memory latencies each (coordinates, then the box copy); here the coordinates
// are read once, the lattice boxes of all levels follow from one bounding box, consecutive levels whose boxes fit the
// LDS copy (512 slots) together share a round (levels 0-6, 7-8, 9-10, 11 for the bench's clouds), and the next round's table
// entries are in flight while the current round interpolates (two copies, one barrier per round).  Levels whose box
// does not fit (the finest one or two of a cloud; almost all for unclustered points) gather from global memory.
// A/B switches of the per-cloud forward (round 4): packed fp32 blend, scalar-base row stores (11 % fewer VALU instructions per
// wave; the launch takes 0.075 ms with or without them - it is latency-bound at full occupancy, profiles/r04_pmc_sq_hashgrid_fwd_cloud_summary.txt -
// and gathers issued as scalar-base inline asm with an explicit vmcnt(0) cost 0.003 ms: the wait also drains the row stores)
#ifndef NESVOR_FWD_PK
#define NESVOR_FWD_PK 1
#endif
#ifndef NESVOR_FWD_SSTORE
#define NESVOR_FWD_SSTORE 1
#endif
// Unclustered input (round 6, NESVOR_LAYOUT_UNCLUSTERED on the forward): `perm` = the points in the order of a coarse lattice's
// cells (sort_points below: the same order the unclustered backward uses); workgroup w takes points perm[256 w ..], so that
// its lattice boxes are small again and the box rounds reach the middle levels.  Row-major output goes straight to the
// points' own rows (a row is 4 L F contiguous bytes, written level by level by one thread); feature-major output would be a
// 4-byte store per (point, feature) into rows of N floats - 33 M scattered sectors at N = 2^20 - so it is written as ROWS in
// the sorted order into `rows` (whole lines) and scatter_pe_rows_kernel turns those into columns (the inverse of
// gather_dy_rows_kernel).
template <int F, int LAYOUT>
__global__ __launch_bounds__(256) void hashgrid_fwd_cloud(const nesvor_grid_t g, const float* __restrict__ u,
                                                          const float* __restrict__ table, float* __restrict__ pe, int64_t N,
                                                          float* __restrict__ pe_absmax,  // optional: raised to max |pe| (the density network's input bound)
                                                          const uint32_t* __restrict__ perm = nullptr, float* __restrict__ rows = nullptr,
                                                          int level_begin = 0, int level_stop = NESVOR_MAX_LEVELS) {  // levels [level_begin, min(level_stop, L)): see nesvor_hashgrid_forward_levels
#ifndef NESVOR_FWD_CLOUD_SLOTS
#define NESVOR_FWD_CLOUD_SLOTS 512
#endif
  constexpr int kSlots = F <= 2 ? NESVOR_FWD_CLOUD_SLOTS : (F == 4 ? 512 : 256);  // slots per copy (two copies).  Measured at F = 2 (N = 2^20 cloud points): 256/512: 0.075 ms, 1024: 0.079, 2048: 0.101, 4096: 0.193 - the finer levels gather from L2 as fast as a bigger copy serves them, and the copy costs occupancy
  constexpr int kMaxGroup = 8;
  constexpr int NRB = kSlots / 256;
  __shared__ __attribute__((aligned(16))) float tcache[2][kSlots * F];
  __shared__ float ubox[4][6];
  __shared__ uint32_t lbox[NESVOR_MAX_LEVELS + 1][8];   // as in hashgrid_bwd_aggregate
  __shared__ uint32_t lpar[NESVOR_MAX_LEVELS + 1][4];   // res, size, offset, hashed
  __shared__ uint32_t slot_off[NESVOR_MAX_LEVELS + 1], grp_end[NESVOR_MAX_LEVELS + 1], rnd_slots[NESVOR_MAX_LEVELS + 1];
  __shared__ int32_t box_end_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t slot_i = (int64_t)blockIdx.x * 256 + tid;   // position in the processing order
  const bool valid = slot_i < N;
  const int64_t si = valid ? slot_i : N - 1;
  const int64_t i = perm != nullptr ? (int64_t)perm[si] : slot_i;  // the point (output index); == slot_i without an order
  const int64_t ii = perm != nullptr ? i : si;
  const int L = g.n_levels, E = L * F;
  const int lb = level_begin, le = min(level_stop, L);  // (uniform kernel arguments)
  const float ux = u[3 * ii], uy = u[3 * ii + 1], uz = u[3 * ii + 2];
  {
    float lo[3] = {ux, uy, uz}, hi[3] = {ux, uy, uz};
#pragma unroll
    for (int d = 0; d < 3; ++d) { lo[d] = -wave_max_f32_dpp(-lo[d]); hi[d] = wave_max_f32_dpp(hi[d]); }
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { ubox[tid >> 6][d] = lo[d]; ubox[tid >> 6][3 + d] = hi[d]; }
    }
  }
  __syncthreads();
  if (tid < 64) {
    float ulo[3], uhi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ulo[d] = fminf(fminf(ubox[0][d], ubox[1][d]), fminf(ubox[2][d], ubox[3][d]));
      uhi[d] = fmaxf(fmaxf(ubox[0][3 + d], ubox[1][3 + d]), fmaxf(ubox[2][3 + d], ubox[3][3 + d]));
    }
    uint32_t b[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (tid < L) {
      // (per-lane index into the kernel arguments = vector loads from the kernarg segment: measured FASTER here than the uniform
      //  loop of scalar loads + selects that the aggregation pass uses - forward 71.8 -> 75.0 us with it)
      const LevelParams p = load_level(g, tid);
      const CellPos blo = locate(p, ulo[0], ulo[1], ulo[2]), bhi = locate(p, uhi[0], uhi[1], uhi[2]);
      const uint32_t ex = bhi.gx - blo.gx, ey = bhi.gy - blo.gy, ez = bhi.gz - blo.gz;
      const bool fits = ex < (uint32_t)kSlots && ey < (uint32_t)kSlots && ez < (uint32_t)kSlots &&
                        (uint64_t)(ex + 2u) * (ey + 2u) * (ez + 2u) <= (uint64_t)kSlots;
      const uint32_t nx = ex + 2u, nxy = nx * (ey + 2u), vol = fits ? nxy * (ez + 2u) : 0u;
      b[0] = blo.gx; b[1] = blo.gy; b[2] = blo.gz; b[3] = ex; b[4] = ey; b[5] = ez; b[6] = vol;
      b[7] = fits ? vol - 2u - nx - nxy : 0u;
      lpar[tid][0] = p.res; lpar[tid][1] = p.size; lpar[tid][2] = p.offset; lpar[tid][3] = p.hashed;
    }
    if (tid <= L) {
#pragma unroll
      for (int q = 0; q < 8; ++q) lbox[tid][q] = b[q];
    }
    const RoundSchedule rs = round_schedule(b[6], 0u, lb, le, tid, (uint32_t)kSlots, 0xFFFFFFFFu, kMaxGroup);
    if (tid == 0) box_end_s = rs.box_end;
    if (tid < NESVOR_MAX_LEVELS) { slot_off[tid] = rs.slot_off; grp_end[tid] = rs.grp_end; rnd_slots[tid] = rs.rnd_slots; }
  }
  __syncthreads();
  const int box_end = __builtin_amdgcn_readfirstlane(box_end_s);
  auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  // feature-major rows: scalar row base + the lane's 32-bit byte offset (4 i < 2^32) - no vector address arithmetic per store
  const bool off32 = NESVOR_FWD_SSTORE && N < ((int64_t)1 << 30);
  const uint32_t i4 = (uint32_t)i * 4u;
  float pe_mx = 0.f;
  auto store_pe = [&](int level, const float (&acc)[F]) __attribute__((always_inline)) {
    if (!valid) return;
#pragma unroll
    for (int f = 0; f < F; ++f) pe_mx = fmaxf(pe_mx, fabsf(acc[f]));
    if (rows != nullptr) {  // (kernel argument: uniform) feature-major output of an ordered batch: rows in the sorted order
      float* o = rows + (size_t)slot_i * E + level * F;
#pragma unroll
      for (int f = 0; f < F; ++f) o[f] = acc[f];
      return;
    }
    if constexpr (LAYOUT == NESVOR_LAYOUT_ROW_MAJOR) {
      float* o = pe + (size_t)i * E + level * F;
#pragma unroll
      for (int f = 0; f < F; ++f) o[f] = acc[f];
    } else {
      if (off32) {
#pragma unroll
        for (int f = 0; f < F; ++f) gstore_b32_sbase(pe + (size_t)(level * F + f) * N, i4, acc[f]);
      } else {
#pragma unroll
        for (int f = 0; f < F; ++f) pe[(size_t)(level * F + f) * N + i] = acc[f];
      }
    }
  };
  auto blend = [&](const CellPos& c, const float (&v)[8][F], float (&acc)[F]) __attribute__((always_inline)) {
    if constexpr (F == 2 && NESVOR_FWD_PK) {
      // packed fp32 (v_pk_mul_f32 / v_pk_fma_f32: two lanes of fp32 per instruction): the x-pair of every weight and the two
      // features of every corner go through one instruction each - the same products and fused multiply-adds in the same
      // order as the scalar form below (bit-identical), 18 instead of 31 VALU instructions per level
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 wx2 = {1.f - c.wx, c.wx};
      const float ay[2] = {1.f - c.wy, c.wy}, az[2] = {1.f - c.wz, c.wz};
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int zy = 0; zy < 4; ++zy) {  // corners k = 2 zy, 2 zy + 1: weights (ax ay) az
        const f32x2 w2 = (wx2 * ay[zy & 1]) * az[zy >> 1];
        a = __builtin_elementwise_fma(f32x2{w2.x, w2.x}, f32x2{v[2 * zy][0], v[2 * zy][1]}, a);
        a = __builtin_elementwise_fma(f32x2{w2.y, w2.y}, f32x2{v[2 * zy + 1][0], v[2 * zy + 1][1]}, a);
      }
      acc[0] = a.x; acc[1] = a.y;
      return;
    }
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = ((k & 1) ? c.wx : 1.f - c.wx) * (((k >> 1) & 1) ? c.wy : 1.f - c.wy) * ((k >> 2) ? c.wz : 1.f - c.wz);
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = fmaf(w, v[k][f], acc[f]);
    }
  };
  // table entries of the slots of round [ra, rb) this thread owns (slots t, t + 256, ...)
  auto fetch_round = [&](int ra, int rb, float (&feat)[NRB][F]) __attribute__((always_inline)) {
    const uint32_t total = sgpr(rnd_slots[ra]);
#pragma unroll
    for (int j = 0; j < NRB; ++j) {
      const uint32_t slot = (uint32_t)j * 256u + (uint32_t)tid;
#pragma unroll
      for (int f = 0; f < F; ++f) feat[j][f] = 0.f;
      if (slot < total) {
        int lv = ra;
        for (int l = ra + 1; l < rb; ++l) lv = slot >= slot_off[l] ? l : lv;
        const uint32_t local = slot - slot_off[lv];
        const uint32_t nx = lbox[lv][3] + 2u, nxy = nx * (lbox[lv][4] + 2u);
        const uint32_t z = (uint32_t)(((float)local + 0.5f) * (1.f / (float)nxy));
        const uint32_t r = local - __umul24(z, nxy);
        const uint32_t y = (uint32_t)(((float)r + 0.5f) * (1.f / (float)nx));
        LevelParams pl;
        pl.scale = 0.f; pl.res = lpar[lv][0]; pl.size = lpar[lv][1]; pl.offset = lpar[lv][2]; pl.hashed = lpar[lv][3];
        const uint32_t key = corner_index(pl, lbox[lv][0] + (r - __umul24(y, nx)), lbox[lv][1] + y, lbox[lv][2] + z);
        load_feat<F>(table + ((size_t)pl.offset + key) * F, feat[j]);
      }
    }
  };
  auto stash_round = [&](int ra, int buf, const float (&feat)[NRB][F]) __attribute__((always_inline)) {
    const uint32_t total = sgpr(rnd_slots[ra]);
#pragma unroll
    for (int j = 0; j < NRB; ++j) {
      const uint32_t slot = (uint32_t)j * 256u + (uint32_t)tid;
      if (slot < total) {
#pragma unroll
        for (int f = 0; f < F; ++f) tcache[buf][slot * F + f] = feat[j][f];
      }
    }
  };
  if (box_end > lb) {
    int ra = lb, rb = (int)sgpr(grp_end[lb]), buf = 0;
    {
      float feat[NRB][F];
      fetch_round(ra, rb, feat);
      stash_round(ra, 0, feat);
    }
    __syncthreads();
    for (;;) {
      const int na = rb, nb = na < box_end ? (int)sgpr(grp_end[na]) : na;
      const bool more = na < box_end;
      float nfeat[NRB][F];
      if (more) fetch_round(na, nb, nfeat);  // in flight while this round interpolates
#pragma unroll 1
      for (int lv = ra; lv < rb; ++lv) {
        const LevelParams p = load_level(g, lv);
        const CellPos c = locate(p, ux, uy, uz);
        const uint32_t x0 = sgpr(lbox[lv][0]), y0 = sgpr(lbox[lv][1]), z0 = sgpr(lbox[lv][2]);
        const uint32_t ny = sgpr(lbox[lv][4]) + 2u, nx = sgpr(lbox[lv][3]) + 2u, nxy = nx * ny;
        const uint32_t s0 = min(__umul24(__umul24(c.gz - z0, ny) + (c.gy - y0), nx) + (c.gx - x0), sgpr(lbox[lv][7])) + sgpr(slot_off[lv]);
        float v[8][F], acc[F];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t sl = s0 + (k & 1) + ((k >> 1) & 1) * nx + (k >> 2) * nxy;
#pragma unroll
          for (int f = 0; f < F; ++f) v[k][f] = tcache[buf][sl * F + f];
        }
        blend(c, v, acc);
        store_pe(lv, acc);
      }
      if (!more) break;
      stash_round(na, buf ^ 1, nfeat);  // the other copy: last read before the previous barrier
      __syncthreads();
      buf ^= 1; ra = na; rb = nb;
    }
  }
  // levels whose box does not fit the copy: two levels at a time, 16 independent gathers per lane in flight
#pragma unroll 1
  for (int lv = box_end; lv < le; lv += 2) {
    const bool two = lv + 1 < le;
    const LevelParams p0 = load_level(g, lv), p1 = load_level(g, two ? lv + 1 : lv);
    const CellPos c0 = locate(p0, ux, uy, uz), c1 = locate(p1, ux, uy, uz);
    const float* tab0 = table + (size_t)p0.offset * F;
    const float* tab1 = table + (size_t)p1.offset * F;
    float v0[8][F], v1[8][F], acc[F];
    {
    // (single requests here: the lanes of a cloud share cache lines, the memory pipeline merges them; the paired form of the
    //  per-level kernel measured SLOWER in this kernel - raster lattice 0.062 -> 0.079 ms, gpurun_out/r06d)
#pragma unroll
    for (int k = 0; k < 8; ++k) load_feat<F>(tab0 + (size_t)corner_index(p0, c0.gx + (k & 1), c0.gy + ((k >> 1) & 1), c0.gz + (k >> 2)) * F, v0[k]);
    if (two) {
#pragma unroll
      for (int k = 0; k < 8; ++k) load_feat<F>(tab1 + (size_t)corner_index(p1, c1.gx + (k & 1), c1.gy + ((k >> 1) & 1), c1.gz + (k >> 2)) * F, v1[k]);
    }
    }
    blend(c0, v0, acc);
    store_pe(lv, acc);
    if (two) {
      blend(c1, v1, acc);
      store_pe(lv + 1, acc);
    }
  }
  if (pe_absmax != nullptr) publish_absmax_wg(pe_absmax, pe_mx);
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

// ------------------------------------------- backward, owner-computes version
constexpr int kMaxChunks = 256;               // table chunks (queues) per level
constexpr int kSubQueues = 8;                 // every queue is split by the XCC the producing workgroup runs on (see below); plan.n_sub <= 8 are used
constexpr uint32_t kTailStride = 4096;        // counters per sub-queue set (>= kMaxChunks * NESVOR_MAX_LEVELS)
constexpr uint32_t kOverflowBase = kTailStride - NESVOR_MAX_LEVELS;  // last words of counter set 0: records per level that found their queue full
constexpr int kOwnerLdsFloats = 8192;         // 32 KiB accumulator per owner workgroup: 4096-entry chunks (measured: 16 / 64 / 128 KiB are slower -
                                              // fewer, hotter queue counters on one side, more reservations per workgroup on the other)

struct BwdPlan {
  uint32_t shift[NESVOR_MAX_LEVELS];           // per level: chunk = 2^shift entries (at most kOwnerLdsFloats / F; the finest levels take half)
  uint32_t n_buckets;
  uint32_t n_chunks[NESVOR_MAX_LEVELS];
  uint32_t bucket_base[NESVOR_MAX_LEVELS];     // first global bucket id of the level
  uint32_t cap[NESVOR_MAX_LEVELS];             // capacity (records) of each of the kSubQueues sub-queues of a bucket of the level
  uint32_t slice[NESVOR_MAX_LEVELS];           // records per owner workgroup of the level
  uint64_t rec_off[NESVOR_MAX_LEVELS];         // first record of the level's queues
  uint32_t n_sub;                              // sub-queues per bucket = XCCs of the device partition (1, 2, 4 or 8)
  uint32_t box_slots;                          // 0: the merge table is always hashed (A/B switch NESVOR_HASHGRID_BOX=0)
  int32_t level_begin, level_end;              // this launch handles levels [level_begin, level_end) (all by default)
  int32_t accumulate_u;                        // input gradient: add to grad_u instead of overwriting (later launches of a split backward)
};
// (further per-launch inputs of the aggregation pass travel as separate kernel arguments: growing this struct moved the
//  compiler's per-level reads of it - plan.shift[level], plan.cap[level] ... - from scalar loads to per-lane global loads,
//  which cost the uniform-points case a factor of three)

// Bitonic sort of one 32-bit key per thread over SPAN consecutive threads of a 256-thread workgroup (ascending).  The 26
// exchanges at distance 1..8 are DPP moves, the 7 at distance 16 / 32 lane swaps (common.h::lane_xor_u32); only the 3
// exchanges across waves go through LDS.  (Rounds 1-3 used __shfl_xor = ds_bpermute for all 33 in-wave exchanges: one
// sort cost 0.03 ms of the pass, profiles/r01_ablate_hashgrid.log.)
template <int K, int J>
__device__ __forceinline__ void bitonic_merge(uint32_t& sv, int tid, uint32_t* sortbuf) {
  uint32_t other;
  if constexpr (J < 64) {
    other = lane_xor_u32<J>(sv, tid & 63);
  } else {
    __syncthreads();
    sortbuf[tid] = sv;
    __syncthreads();
    other = sortbuf[tid ^ J];
  }
  const bool keep_min = ((tid & K) == 0) == ((tid & J) == 0);
  sv = keep_min ? min(sv, other) : max(sv, other);
  if constexpr (J > 1) bitonic_merge<K, J / 2>(sv, tid, sortbuf);
}
template <int K>
__device__ __forceinline__ void bitonic_sort(uint32_t& sv, int tid, uint32_t* sortbuf) {
  if constexpr (K > 2) bitonic_sort<K / 2>(sv, tid, sortbuf);
  bitonic_merge<K, K / 2>(sv, tid, sortbuf);
}

__device__ __forceinline__ uint32_t spread3(uint32_t x) {  // 8 bits -> every third bit
  x &= 0xffu;
  x = (x ^ (x << 8)) & 0x0300f00fu;
  x = (x ^ (x << 4)) & 0x030c30c3u;
  x = (x ^ (x << 2)) & 0x09249249u;
  return x;
}

// Phase 1 of the backward.  One workgroup = 256 consecutive samples (one PSF cloud).
//  * once per workgroup: bitonic-sort the samples by the Morton code of their finest-level cell, so
//    that at every level lanes falling into the same cell sit (mostly) next to each other;
//  * per level: each lane forms its 8 corner contributions and a segmented wave scan on the VALU
//    (DPP row_shr inside 16-lane rows, v_readlane carries across rows) sums runs of equal cells;
//    only the last lane of a run keeps records.  That removes the bulk of the duplication of a PSF
//    cloud (all of it at the coarse levels) without any LDS traffic;
//  * the surviving (entry, grad) records are binned by table chunk: one LDS counter increment per
//    record, one returning global atomic per non-empty (workgroup, chunk) to reserve queue space
//    (its ~2 us latency is hidden behind the next level's register-only work), then the records go
//    straight from registers to the chunk queues.  Residual duplicates (same vertex reached from
//    neighbouring cells / other waves) are summed by the chunk's owner in phase 2.
// LDS: 2 x 64 counters, so occupancy is set by registers only.
// float (|x| < 2^62, an integer after scaling by a power of two) <-> two's-complement 64-bit fixed point without the
// compiler's generic f32<->i64 expansions (see to_fixed).
__device__ __forceinline__ unsigned long long to_fixed(float x) {
  // hi = round(x / 2^32), lo = x - hi 2^32 in [-2^31, 2^31]: exact in fp32 for every x (a floor-based split is not: for a
  // small negative x it forms 2^32 + x, which needs 32 bits - harmless while the scale follows the workgroup's own maximum,
  // 2^-15 relative once a caller-supplied bound sits 2^30 above a workgroup's gradients)
  const float t = rintf(x * 0x1p-32f);
  const int32_t lo = __float2int_rn(fmaf(-t, 0x1p32f, x));  // (to nearest: a truncation bias would add up over the adds of a slot)
  return ((unsigned long long)(uint32_t)((int32_t)t + (lo >> 31)) << 32) | (unsigned long long)(uint32_t)lo;
}
// NESVOR_HG_F64FIX=1 (build flag, off): the conversion through the fp64 pipe.  x s + 1.5 2^52 (one v_cvt_f64_f32, one v_fma_f64
// rounding to nearest) has the exponent of 2^52 for every |x s| < 2^50, so its bit pattern is M + q with M = 0x4338000000000000
// and q = rint(x s) as a two's-complement integer: 2 instructions per value where to_fixed takes 7 and the scaling an eighth.
// The 64-bit LDS adds sum N M + sum q; the low 51 bits of M are zero, so the low 51 bits of a slot are sum q mod 2^51 - read
// back sign-extended from bit 50 (from_fixed51), exact while |sum q| < 2^50.  A slot that received anything is never zero
// (N M = 0 mod 2^64 needs N = 0 mod 2^13; a workgroup has 2^11 contributions).  Measured (gpurun_out/r05p, two alternating
// builds): 6 % fewer VALU instructions, aggregation pass 0.341 -> 0.333 ms isolated, 0.299 -> 0.293 ms in the step - and 11 bits
// less room under a caller-supplied bound (resolution bound 2^-41 instead of 2^-52: a cloud whose gradients lie 2^-30 below the
// bound keeps 2^-11 relative accuracy, test_hashgrid_backward_with_producer_bound[1000.0] fails).  Not worth that: off.
#ifndef NESVOR_HG_F64FIX
#define NESVOR_HG_F64FIX 0
#endif
__device__ __forceinline__ unsigned long long to_fixed51(float x, double scale) {
  return (unsigned long long)__double_as_longlong(fma((double)x, scale, 0x1.8p52));
}
__device__ __forceinline__ float from_fixed51(unsigned long long q) {
  const uint32_t lo = (uint32_t)q;
  const int32_t hi = ((int32_t)(uint32_t)(q >> 32) << 13) >> 13;  // bits 32..50, sign-extended (v_bfe_i32)
  return fmaf((float)(hi + (int32_t)(lo >> 31)), 0x1p32f, (float)(int32_t)lo);
}
__device__ __forceinline__ float from_fixed(unsigned long long q) {
  // q = hi 2^32 + lo with lo taken as SIGNED (hi absorbs its sign bit): a small negative sum is then read as 0 2^32 - |q|,
  // not as -1 2^32 + (2^32 - |q|), whose low word needs 32 bits
  const uint32_t lo = (uint32_t)q;
  return fmaf((float)((int32_t)(q >> 32) + (int32_t)(lo >> 31)), 0x1p32f, (float)(int32_t)lo);
}


// NESVOR_FIXED32 (build macro, F == 2 only): a merge-table slot holds both features as two 32-bit fixed-point fields of one
// 64-bit word (one ds_add_u64 per corner, 3 VALU instructions per value) instead of one 64-bit fixed-point word per
// feature.  Resolution: 2^-22 of 256 max|dy| of the workgroup and level (the 64-bit form resolves 2^-52).  Off by default.
#ifndef NESVOR_FIXED32
#define NESVOR_FIXED32 0
#endif

// BOUND: max |dy| comes from the caller (`dy_bound`) and the pass over dy that determines it is compiled out.  A template
// parameter, not a run-time branch: with the branch in place the compiler restructured the pass it guards, and the kernel
// WITHOUT a bound ran 3x slower on uniform points (4.9 vs 1.6 ms) and 4 % slower on PSF clouds (tools/hg_variants.py).
#ifndef NESVOR_HG_SLOTS
#define NESVOR_HG_SLOTS 1024      // merge-table slots at F <= 2.  A/B (gpurun_out/s2j2): 2048 slots at two workgroups per CU (66 KB of LDS) let levels
                                  // 12-13 take the box path, and the pass goes from 0.346 to 0.505 ms: it lives on its four waves per SIMD
#endif
#ifndef NESVOR_HG_MINBLOCKS
#define NESVOR_HG_MINBLOCKS 4
#endif
// NESVOR_HG_SPATIAL (default): the hashed merge table of the fine levels (boxes that do not fit the table) is addressed by the
// low bits of the vertex's lattice coordinates - slot = x mod 2^a | (y mod 2^b) << a | (z mod 2^c) << (a + b), a + b + c =
// log2(slots), the bits dealt to the axes by the extent of the workgroup's box - instead of a multiplicative hash of the entry
// index.  Two vertices of a cloud then share a first slot only if they lie a whole window apart: replaying 200 PSF clouds
// (tools/sim_spatial_hash.py) 0.6 / 11 / 59 vertices per cloud miss their first slot at levels 13 / 14 / 15 against 41 / 89 / 176
// with the multiplicative hash (313 / 462 / 672 vertices in 1024 slots), and every miss is a serial LDS round trip per probe.
// Misses walk on by double hashing as before.  0: multiplicative hash, table size following the previous level's count.
#ifndef NESVOR_HG_SPATIAL
#define NESVOR_HG_SPATIAL 1
#endif
// -DNESVOR_HG_TIMELINE=1 (tools/hg_timeline.py): thread 0 of the first 64 workgroups of the aggregation pass writes (site, time)
// marks - s_memtime, the 100 MHz constant clock - at its barriers; nesvor_debug_hg_timeline() copies them out.  Where a
// workgroup's ~65 us go, phase by phase, without a profiler in the way.
#ifndef NESVOR_HG_TIMELINE
#define NESVOR_HG_TIMELINE 0
#endif
#if NESVOR_HG_TIMELINE
constexpr int kTlWgs = 64, kTlMarks = 96;
__device__ unsigned long long g_hg_timeline[kTlWgs][kTlMarks];
#define HG_TICK(site)                                                                                              \
  do {                                                                                                             \
    if (blockIdx.x < (unsigned)kTlWgs && threadIdx.x == 0 && tl_n < kTlMarks)                                      \
      g_hg_timeline[blockIdx.x][tl_n++] = ((unsigned long long)(site) << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull); \
  } while (0)
#else
#define HG_TICK(site) do { } while (0)
#endif
template <int F, int LAYOUT, bool INPUT_GRAD, bool MERGE, bool BOUND = false>
__global__ __launch_bounds__(256, F <= 2 ? NESVOR_HG_MINBLOCKS : (F == 4 ? 3 : 2)) void hashgrid_bwd_aggregate(const nesvor_grid_t g, const BwdPlan plan,
                                                              const float* __restrict__ u,
                                                              const float* __restrict__ table,
                                                              const float* __restrict__ dpe,
                                                              float* __restrict__ grad_table,
                                                              float* __restrict__ grad_u, uint32_t* __restrict__ tails,
                                                              uint32_t* __restrict__ tails_next,
                                                              uint32_t* __restrict__ records, int64_t N,
                                                              const float* __restrict__ dy_bound,  // device scalar >= max |dy| over the batch, or null: the kernel reads all dy itself first
                                                              uint8_t* __restrict__ order,         // one byte per sample: the Morton-sorted order of every workgroup's samples
                                                              int order_mode,                      // 1: sort and write `order` (first launch of a backward), 2: read it (later launches of a split backward)
                                                              const uint32_t* __restrict__ perm,   // null, or the points in lattice-cell order (unclustered input: sort_points below): workgroup w takes points perm[256 w ..]
                                                              int dy_by_slot) {                    // 1: dpe is already in that order (row r = point perm[r]): gather_dy_rows_kernel
  // the queue tails of the NEXT backward (the other of the workspace's two tail regions) are zero-filled here, so that
  // no launch of its own is needed for it
  for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < (uint32_t)kSubQueues * kTailStride; t += gridDim.x * 256u) tails_next[t] = 0u;
  constexpr bool kPack = (F == 2) && (NESVOR_FIXED32 != 0);
  constexpr bool kPerLevel = kPack && (NESVOR_FIXED32 >= 2);  // packed fields scaled by the workgroup's max |dy| of EACH level
  constexpr int kWords = kPack ? 1 : F;  // 64-bit words per slot
  __shared__ uint32_t bcount[kMaxChunks];
  // per bucket of the current round: box rounds hold uint4 (reserved position, first record of the sub-queue, capacity,
  // level); the single-level rounds hold uint2 (position, first record) double-buffered by level parity
  __shared__ __attribute__((aligned(16))) uint32_t bb_raw[4 * kMaxChunks];
  uint4* const bbase4 = reinterpret_cast<uint4*>(bb_raw);
  uint2(*const bbase)[kMaxChunks] = reinterpret_cast<uint2(*)[kMaxChunks]>(bb_raw);
  __shared__ uint32_t sortbuf[256];
  // workgroup-wide merge table: slots addressed by position in the lattice box (box rounds; tkeys then holds
  // level << 27 | entry index of every slot) or by open addressing keyed by the level-local entry index
  constexpr int kSlots = F <= 2 ? NESVOR_HG_SLOTS : (F == 4 ? 512 : 256);
  constexpr uint32_t kEmpty = 0xFFFFFFFFu;
  constexpr uint32_t kKeyMask = (1u << 27) - 1u;
  constexpr int kMaxGroup = 8;  // levels per box round
  __shared__ __attribute__((aligned(16))) uint32_t tkeys[kSlots];
  // slot values are 64-bit fixed point: integer LDS atomics are returnless and resolve same-address lanes in
  // hardware (ds_add_f32 retires ~3 cycles per lane on gfx950, a compare-and-swap loop pays a round trip per
  // retry), and the sum no longer depends on the order of the adds
  __shared__ __attribute__((aligned(16))) unsigned long long tvals[kSlots * kWords];
  // box rounds: the table entries of the round's lattice boxes ([slot][F]); the input gradient reads its 8 corners here
  __shared__ __attribute__((aligned(16))) float tcache[INPUT_GRAD && MERGE ? kSlots * F : 1];
  __shared__ float gmax[4];           // per wave max |dy| over all levels of this launch
  __shared__ uint32_t lmax_bits[NESVOR_MAX_LEVELS + 1];  // kPerLevel: max |dy| of the workgroup's samples per level (float bits)
  __shared__ float lscale[NESVOR_MAX_LEVELS + 1][2];     // kPerLevel: fixed-point scale of a level and its inverse
  __shared__ uint32_t merge_stat[2];  // records inserted / drained at the current level (hashed table)
  __shared__ uint32_t slots_log2;     // slots of the hashed table used at the current level: 256 .. kSlots, ~4x the
                                      // previous level's distinct vertices (they grow ~1.3-1.5x per level), so that
                                      // the drain only walks what can be occupied
  __shared__ uint32_t merge_off;      // set once merging stops paying: finer levels skip the table
  __shared__ uint32_t prev_cnt;       // vertices that received something at the previous level (0: unknown)
  __shared__ float ubox[4][6];        // per wave min / max of the samples' coordinates
  // per level: first cell (x,y,z), cells spanned - 1 (x,y,z), vertices of the lattice box (0: the box does not fit the
  // table), largest slot a sample's first corner may take
  __shared__ uint32_t lbox[NESVOR_MAX_LEVELS + 1][8];
  __shared__ uint32_t lwin[NESVOR_MAX_LEVELS + 1];  // NESVOR_HG_SPATIAL: address bits of x | y << 8 | z << 16 in the hashed merge table
  // box rounds (groups of consecutive levels whose boxes share the table): per level the first slot of its box, the
  // first (round-local) bucket of its chunks, and the end of its round
  __shared__ uint32_t slot_off[NESVOR_MAX_LEVELS + 1], bkt_off[NESVOR_MAX_LEVELS + 1], grp_end[NESVOR_MAX_LEVELS + 1];
  __shared__ uint32_t rnd_slots[NESVOR_MAX_LEVELS + 1], rnd_bkts[NESVOR_MAX_LEVELS + 1];  // totals of a round, at its first level
  // per level, for code that indexes levels per LANE (kernel arguments can only be indexed uniformly without a trip
  // through scratch memory): res, size, offset, hashed, queue capacity, first bucket, first record, chunks
  __shared__ uint32_t lpar[NESVOR_MAX_LEVELS + 1][8];
  __shared__ int32_t box_end_s;       // levels [level_begin, box_end) address the table by box slot
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t base = (int64_t)blockIdx.x * 256;
  const int E = g.n_levels * F;
#if NESVOR_HG_TIMELINE
  int tl_n = 0;
  if (blockIdx.x < (unsigned)kTlWgs && tid == 0) for (int k = 0; k < kTlMarks; ++k) g_hg_timeline[blockIdx.x][k] = 0ull;
#endif
  HG_TICK(0);  // entry
  const int level_end = plan.level_end;
  // Queue tails are hot counters (every workgroup reserves space in ~100 of them per level).  The L2s of the eight
  // XCCs are kept coherent by hardware, so a counter shared by all workgroups migrates between L2s on every
  // reservation; a counter set per XCC stays in its own L2 (measured, tools/atomic_scope_probe.hip: 2.1x the
  // throughput).  HW_REG_XCC_ID names the XCC this workgroup really runs on - no assumption about the dispatch order.
  const uint32_t sub = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (plan.n_sub - 1u);
  if (tid < kMaxChunks) bcount[tid] = 0;
  for (int t = tid; t < kSlots; t += 256) tkeys[t] = kEmpty;
  for (int t = tid; t < kSlots * kWords; t += 256) tvals[t] = 0ull;
  if (tid < 2) merge_stat[tid] = 0;
  if (tid == 0) { merge_off = MERGE ? 0u : 1u; slots_log2 = __builtin_ctz(kSlots); prev_cnt = 0u; }
  if constexpr (kPerLevel) {
    if (tid <= NESVOR_MAX_LEVELS) lmax_bits[tid] = 0u;
  }

  // ---- sort the workgroup's samples by Morton code of the finest-level cell
  uint32_t sv;
  {
    const int64_t s0_ = base + tid;
    uint32_t code = 0x00ffffffu;
    if (s0_ < N) {
      const int64_t i0 = perm != nullptr ? (int64_t)perm[s0_] : s0_;
      const LevelParams pf = load_level(g, g.n_levels - 1);
      const CellPos c = locate(pf, u[3 * i0], u[3 * i0 + 1], u[3 * i0 + 2]);
      code = spread3(c.gx) | (spread3(c.gy) << 1) | (spread3(c.gz) << 2);
    }
    sv = (code << 8) | (uint32_t)tid;
    // a later launch of a split backward (data parallel: coarse levels after the fine ones) re-uses the order the first
    // launch found - same samples, same key
    if (order_mode == 2) sv = (uint32_t)order[base + tid];
    else
#pragma unroll 1
    for (int rep = 0; rep < (NESVOR_ABL(1) ? 2 : 1); ++rep) {
      if (rep) sv ^= 0x80000000u;  // (ablation: something to sort the second time)
      bitonic_sort<NESVOR_SORT_SPAN>(sv, tid, sortbuf);
      if (rep) sv ^= 0x80000000u;
    }
  }
  HG_TICK(1);  // sorted
  if (order_mode == 1) order[base + tid] = (uint8_t)(sv & 255u);
  const int64_t slot_i = base + (sv & 255u);  // the sample this lane owns from now on: position in the (cell-ordered) batch ...
  const bool valid = slot_i < N;
  const int64_t i = (valid && perm != nullptr) ? (int64_t)perm[slot_i] : slot_i;  // ... and its index in u / dpe / grad_u
  const int64_t ii = valid ? i : N - 1;
  const int64_t id = dy_by_slot ? (valid ? slot_i : N - 1) : ii;  // row / column of dpe
  const float ux = u[3 * ii], uy = u[3 * ii + 1], uz = u[3 * ii + 2];
  float gux = 0.f, guy = 0.f, guz = 0.f;

  auto load_dy = [&](int level, float (&dy)[F]) __attribute__((always_inline)) {
    if constexpr (LAYOUT == NESVOR_LAYOUT_ROW_MAJOR) {
      const float* o = dpe + (size_t)id * E + level * F;
#pragma unroll
      for (int f = 0; f < F; ++f) dy[f] = valid ? o[f] : 0.f;
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) dy[f] = valid ? dpe[(size_t)(level * F + f) * N + id] : 0.f;
    }
  };

  if constexpr (MERGE) {
    // bounding box of the workgroup's samples: locate() is monotone in u, so the lattice box of every level follows
    // from these six numbers
    float lo[3] = {ux, uy, uz}, hi[3] = {ux, uy, uz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = -wave_max_f32_dpp(-lo[d]);
      hi[d] = wave_max_f32_dpp(hi[d]);
    }
    // ONE fixed-point scale for the whole launch of this workgroup: max |dy| over its samples and all levels (eight
    // levels' loads in flight at a time).  The adds of one slot sum to at most 256 max|dy| (corner weights of a sample
    // sum to 1), which is mapped below 2^61 (2^30 for the packed 32-bit fields); 64-bit words then resolve
    // max|dy| 2^-52, far below an fp32 ulp of any gradient that matters next to the largest one.
    // With a caller-supplied bound (the producer of dy knows its largest magnitude: nesvor_mlp_backward_bounded) that pass
    // over dy - 134 MB at N = 2^20, read a second time level by level below - is skipped: 64-bit words leave room for a
    // bound that is orders of magnitude above this workgroup's own maximum.
    float m = 0.f;
    if constexpr (kPerLevel) __syncthreads();  // lmax_bits zero-filled (no barrier before this point when the sort is skipped)
    if constexpr (BOUND) {
      m = *dy_bound;
      if constexpr (kPerLevel) {  // (a global bound: every level gets the same scale)
        if (tid < NESVOR_MAX_LEVELS) lmax_bits[tid] = __float_as_uint(m);
      }
    } else
    for (int l0 = plan.level_begin; l0 < level_end; l0 += 8) {
      float d[8][F];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (l0 + q < level_end) load_dy(l0 + q, d[q]);
        else {
#pragma unroll
          for (int f = 0; f < F; ++f) d[q][f] = 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float ml = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) ml = fmaxf(ml, fabsf(d[q][f]));
        m = fmaxf(m, ml);
        if constexpr (kPerLevel) {
          ml = wave_max_f32_dpp(ml);
          if (lane == 0 && l0 + q < level_end) atomicMax(&lmax_bits[l0 + q], __float_as_uint(ml));
        }
      }
    }
    m = wave_max_f32_dpp(m);
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { ubox[tid >> 6][d] = lo[d]; ubox[tid >> 6][3 + d] = hi[d]; }
      gmax[tid >> 6] = m;
    }
  }
  __syncthreads();
  HG_TICK(2);  // bounding box / max |dy| published
  float ulo[3] = {0.f, 0.f, 0.f}, uhi[3] = {0.f, 0.f, 0.f};
  float fscale = 1.f, finv = 1.f;
  if constexpr (MERGE) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      ulo[d] = fminf(fminf(ubox[0][d], ubox[1][d]), fminf(ubox[2][d], ubox[3][d]));
      uhi[d] = fmaxf(fmaxf(ubox[0][3 + d], ubox[1][3 + d]), fmaxf(ubox[2][3 + d], ubox[3][3 + d]));
      ulo[d] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ulo[d])));
      uhi[d] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, uhi[d])));
    }
    {
      float mx = 256.f * fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
      mx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mx)));
      // (slot sums stay below 2^30 in the packed fields, 2^49 in the 51-bit words, 2^61 in the 64-bit ones)
      int sexp = (kPack ? 29 : (NESVOR_HG_F64FIX ? 48 : 60)) - ((int)((__float_as_uint(mx) >> 23) & 0xFFu) - 127);
      sexp = sexp > 100 ? 100 : (sexp < -100 ? -100 : sexp);
      fscale = __uint_as_float((uint32_t)(sexp + 127) << 23);
      finv = __uint_as_float((uint32_t)(127 - sexp) << 23);
    }
    // the lattice boxes of all levels at once (wave 0, lane l: level l) instead of two locate() per level in every
    // thread, and the round schedule from them (uniform loop over the levels, v_readlane picks a level's numbers)
    HG_TICK(6);  // box of the samples, fixed-point scale (all threads)
    if (tid < 64) {
      uint32_t b[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      uint32_t nch = 0u;
      // (lane = level: per-lane reads of the kernel arguments are vector loads from the kernarg segment - measured faster here than a
      //  uniform loop of scalar loads with selects: 26 against 93 timeline units)
      const LevelParams p = load_level(g, tid < g.n_levels ? tid : 0);
      const int tl_ = tid < g.n_levels ? tid : 0;
      const uint32_t pl_nch = plan.n_chunks[tl_], pl_cap = plan.cap[tl_], pl_base = plan.bucket_base[tl_], pl_rec = (uint32_t)plan.rec_off[tl_],
                     pl_shift = plan.shift[tl_];
      if (tid < g.n_levels) {
        const CellPos blo = locate(p, ulo[0], ulo[1], ulo[2]), bhi = locate(p, uhi[0], uhi[1], uhi[2]);
        const uint32_t ex = bhi.gx - blo.gx, ey = bhi.gy - blo.gy, ez = bhi.gz - blo.gz;  // cells spanned - 1 (wrap if out of range)
        const bool fits = plan.box_slots != 0u && ex < (uint32_t)kSlots && ey < (uint32_t)kSlots && ez < (uint32_t)kSlots &&
                          (uint64_t)(ex + 2u) * (ey + 2u) * (ez + 2u) <= (uint64_t)kSlots;
        const uint32_t nx = ex + 2u, nxy = nx * (ey + 2u), vol = fits ? nxy * (ez + 2u) : 0u;
        b[0] = blo.gx; b[1] = blo.gy; b[2] = blo.gz; b[3] = ex; b[4] = ey; b[5] = ez; b[6] = vol;
        b[7] = fits ? vol - 2u - nx - nxy : 0u;  // = slot (inside the box) of the first corner of the box's last cell
        {
          // address bits of the spatially addressed table: one at a time to the axis whose box extent per slot is the largest
          const uint32_t vx = min(ex, 1023u) + 2u, vy = min(ey, 1023u) + 2u, vz = min(ez, 1023u) + 2u;
          uint32_t ba = 0, bb = 0, bc = 0;
          for (int n = 0; n < __builtin_ctz(kSlots); ++n) {
            const uint32_t rx = (vx << 12) >> ba, ry = (vy << 12) >> bb, rz = (vz << 12) >> bc;
            if (rz >= rx && rz >= ry) ++bc;
            else if (ry >= rx) ++bb;
            else ++ba;
          }
          lwin[tid] = ba | (bb << 8) | (bc << 16);
        }
        nch = pl_nch;
        lpar[tid][0] = p.res; lpar[tid][1] = p.size; lpar[tid][2] = p.offset; lpar[tid][3] = p.hashed;
        lpar[tid][4] = pl_cap; lpar[tid][5] = pl_base; lpar[tid][6] = pl_rec;
        lpar[tid][7] = pl_shift;
        if constexpr (kPerLevel) {
          // a slot sums at most 256 values of at most max |dy| each: mapped below 2^30
          const float mxl = 256.f * __uint_as_float(lmax_bits[tid]);
          int se = 29 - ((int)((__float_as_uint(mxl) >> 23) & 0xFFu) - 127);
          se = se > 100 ? 100 : (se < -100 ? -100 : se);
          lscale[tid][0] = __uint_as_float((uint32_t)(se + 127) << 23);
          lscale[tid][1] = __uint_as_float((uint32_t)(127 - se) << 23);
        }
      }
      HG_TICK(7);  // per-level boxes, window bits, lpar (lane = level)
      if (tid <= g.n_levels) {
#pragma unroll
        for (int q = 0; q < 8; ++q) lbox[tid][q] = b[q];  // row n_levels: all zero (a level past the end is "not a box")
      }
      // box levels = the longest prefix of levels whose box fits the table; consecutive box levels share a round
      // while their boxes fit the table together and their chunks the bucket counters
      const RoundSchedule rs = round_schedule(b[6], nch, plan.level_begin, level_end, tid, (uint32_t)kSlots, (uint32_t)kMaxChunks, kMaxGroup);
      if (tid == 0) box_end_s = rs.box_end;
      if (tid < NESVOR_MAX_LEVELS) {
        slot_off[tid] = rs.slot_off; bkt_off[tid] = rs.bkt_off; grp_end[tid] = rs.grp_end;
        rnd_slots[tid] = rs.rnd_slots; rnd_bkts[tid] = rs.rnd_bkts;
      }
      HG_TICK(8);  // round schedule (wave 0)
    }
    __syncthreads();
    HG_TICK(3);  // lattice boxes and round schedule
  }
  const int box_end = MERGE ? __builtin_amdgcn_readfirstlane(box_end_s) : plan.level_begin;
  auto sgpr = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };  // wave-uniform values

  // Everything of a level that needs no shared state beyond the box cache: corner slots / indices, run-summed corner
  // values, tail flag.  BOX: the level's vertices are addressed by their slot in the round's table (s0 = slot of the
  // first corner); otherwise by table entry index (idx).
  // (always_inline: with the inline-asm scan the inliner otherwise leaves this a real function - closure, kernel
  // arguments and the value arrays then live in scratch memory: 6x slower)
  auto prepare = [&](auto box_c, int level, const float (&dy)[F], uint32_t (&idx)[8], uint32_t& s0, uint32_t& nx, uint32_t& nxy,
                     float (&val)[8][F], bool& tail) __attribute__((always_inline)) {
    constexpr bool BOX = decltype(box_c)::value;
    // (the level is wave-uniform, but it lives in loops whose exit tests read LDS, so the compiler keeps it in a VGPR - and a
    //  kernel argument indexed by a VGPR is a VECTOR load from the kernarg segment: five of them at the head of every level's
    //  dependency chain.  Through readfirstlane the same reads are scalar loads: worth 1 % of the pass, round 5)
    level = __builtin_amdgcn_readfirstlane(level);
    const LevelParams p = load_level(g, level);
    const CellPos c = locate(p, ux, uy, uz);
    if constexpr (BOX) {
      const uint32_t x0 = sgpr(lbox[level][0]), y0 = sgpr(lbox[level][1]), z0 = sgpr(lbox[level][2]);
      const uint32_t ny = sgpr(lbox[level][4]) + 2u;
      nx = sgpr(lbox[level][3]) + 2u; nxy = nx * ny;
      // (24-bit multiplies: box coordinates are below 2^10; the clamp only matters for NaN coordinates, which fall
      // outside every box: it keeps all eight corners inside the level's part of the table)
      s0 = min(__umul24(__umul24(c.gz - z0, ny) + (c.gy - y0), nx) + (c.gx - x0), sgpr(lbox[level][7])) + sgpr(slot_off[level]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) idx[k] = corner_index(p, c.gx + (k & 1), c.gy + ((k >> 1) & 1), c.gz + (k >> 2));
      s0 = (c.gx & 1023u) | ((c.gy & 1023u) << 10) | ((c.gz & 1023u) << 20);  // (what the spatially addressed merge table needs of the cell)
    }
    if constexpr (INPUT_GRAD) {
      float v[8][F];
      if constexpr (BOX) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t s = s0 + (k & 1) + ((k >> 1) & 1) * nx + (k >> 2) * nxy;
#pragma unroll
          for (int f = 0; f < F; ++f) v[k][f] = tcache[s * F + f];
        }
      } else {
        const float* tab = table + (size_t)p.offset * F;
#pragma unroll
        for (int k = 0; k < 8; ++k) load_feat<F>(tab + (size_t)idx[k] * F, v[k]);
      }
      // d/du of the trilinear blend of q_k = <table[corner k], dy>: differences along one axis, blended along the others
      float q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        q[k] = v[k][0] * dy[0];
#pragma unroll
        for (int f = 1; f < F; ++f) q[k] = fmaf(v[k][f], dy[f], q[k]);
      }
      float dxl[4], lx[4];  // along x at the four (y, z) edges: difference and blend
#pragma unroll
      for (int e = 0; e < 4; ++e) { dxl[e] = q[2 * e + 1] - q[2 * e]; lx[e] = fmaf(c.wx, dxl[e], q[2 * e]); }
      const float sx0 = fmaf(c.wy, dxl[1] - dxl[0], dxl[0]), sx1 = fmaf(c.wy, dxl[3] - dxl[2], dxl[2]);
      const float sx = fmaf(c.wz, sx1 - sx0, sx0);
      const float dy0 = lx[1] - lx[0], dy1 = lx[3] - lx[2];  // along y at z = 0, 1
      const float sy = fmaf(c.wz, dy1 - dy0, dy0);
      const float sz = fmaf(c.wy, dy1, lx[2]) - fmaf(c.wy, dy0, lx[0]);
      gux = fmaf(p.scale, sx, gux); guy = fmaf(p.scale, sy, guy); guz = fmaf(p.scale, sz, guz);
    }
    {
      const float ax[2] = {1.f - c.wx, c.wx}, ay[2] = {1.f - c.wy, c.wy}, az[2] = {1.f - c.wz, c.wz};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float w = ax[k & 1] * ay[(k >> 1) & 1] * az[k >> 2];
#pragma unroll
        for (int f = 0; f < F; ++f) val[k][f] = w * dy[f];
      }
    }
    // Segmented inclusive scan (a run = consecutive lanes of a 16-lane row in the same cell), on the VALU:
    // Hillis-Steele with DPP row_shr (out-of-row sources read 0).  `flag` = a run head lies between the row
    // start and this lane.  Runs are cut at the row boundaries: what a longer run would have merged is merged
    // by the workgroup's table, and the neighbour compares stay on the VALU as row_shr:1 / row_shl:1 DPP moves.
    const int rl = lane & 15;
    const uint32_t px_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.gx, 0x111, 0xf, 0xf, true);
    const uint32_t py_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.gy, 0x111, 0xf, 0xf, true);
    const uint32_t pz_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.gz, 0x111, 0xf, 0xf, true);
    const bool vprev = __builtin_amdgcn_update_dpp(0, (int)valid, 0x111, 0xf, 0xf, true) != 0;
    const bool head = rl == 0 || !valid || !vprev || c.gx != px_ || c.gy != py_ || c.gz != pz_;
    const bool next_head = __builtin_amdgcn_update_dpp(1, (int)head, 0x101, 0xf, 0xf, false) != 0;  // row_shl:1
    tail = valid && (rl == 15 || next_head);
    int flag = head ? 1 : 0;
    // one instruction per value and step: v_fmac_f32 with a DPP source (val += shifted(val) * m); the compiler emits
    // v_mov_dpp + v_fma for the same expression.  s_nop: a DPP read needs two wait states after the VALU write of its
    // source, which the hazard recogniser does not see through inline asm.
#define NESVOR_SCAN_STEP(SHR, CTRL)                                                                        \
    {                                                                                                      \
      const float m = flag ? 0.f : 1.f;                                                                    \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) _Pragma("unroll") for (int f = 0; f < F; ++f)          \
        asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1"  \
                     : "+v"(val[k][f]) : "v"(m));                                                          \
      flag |= __builtin_amdgcn_update_dpp(0, flag, CTRL, 0xf, 0xf, true);                                  \
    }
    if constexpr (!NESVOR_ABL(2)) {
      // all values are written before this point and two wait states pass before the first DPP read
#define NESVOR_FENCE8(f) asm volatile("s_nop 1" : "+v"(val[0][f]), "+v"(val[1][f]), "+v"(val[2][f]), "+v"(val[3][f]), "+v"(val[4][f]), "+v"(val[5][f]), "+v"(val[6][f]), "+v"(val[7][f]))
      NESVOR_FENCE8(0);
      if constexpr (F >= 2) NESVOR_FENCE8(1);
      if constexpr (F >= 4) { NESVOR_FENCE8(2); NESVOR_FENCE8(3); }
      if constexpr (F >= 8) { NESVOR_FENCE8(4); NESVOR_FENCE8(5); NESVOR_FENCE8(6); NESVOR_FENCE8(7); }
#undef NESVOR_FENCE8
      // a step only does something for lanes that have not yet seen the head of their run: at the fine levels runs
      // are one to three lanes long and the wide steps are skipped (wave-uniform test)
      NESVOR_SCAN_STEP(1, 0x111)
      if (__ballot(flag == 0)) {
        NESVOR_SCAN_STEP(2, 0x112)
        if (__ballot(flag == 0)) {
          NESVOR_SCAN_STEP(4, 0x114)
          if (__ballot(flag == 0)) NESVOR_SCAN_STEP(8, 0x118)
        }
      }
    }
#undef NESVOR_SCAN_STEP
  };

  // one corner's F values -> the slot's fixed-point words
  auto scale_of = [&](uint32_t lv) __attribute__((always_inline)) { return kPerLevel ? lscale[lv][0] : fscale; };
  auto inv_of = [&](uint32_t lv) __attribute__((always_inline)) { return kPerLevel ? lscale[lv][1] : finv; };
  auto slot_add = [&](uint32_t slot, const float (&v)[F], float fscale) __attribute__((always_inline)) {
    if constexpr (kPack) {
      const int32_t q0 = __float2int_rn(v[0] * fscale), q1 = __float2int_rn(v[1] * fscale);
      // ((int64)q1 << 32) + (int64)q0: the low field's sign borrows from the high field; undone when the slot is read
      atomicAdd(&tvals[slot], ((unsigned long long)(uint32_t)(q1 + (q0 >> 31)) << 32) | (unsigned long long)(uint32_t)q0);
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if constexpr (NESVOR_HG_F64FIX != 0) atomicAdd(&tvals[f * kSlots + slot], to_fixed51(v[f], (double)fscale));
        else atomicAdd(&tvals[f * kSlots + slot], to_fixed(v[f] * fscale));
      }
    }
  };
  // read + clear one slot; false: nothing was added (or everything cancelled exactly)
  auto slot_take = [&](uint32_t slot, float (&v)[F], float finv) __attribute__((always_inline)) -> bool {
    bool any = false;
    if constexpr (kPack) {
      const unsigned long long w = tvals[slot];
      any = w != 0ull;
      if (any) {
        const int32_t lo = (int32_t)(uint32_t)w;
        const int32_t hi = (int32_t)(uint32_t)(w >> 32) - (lo >> 31);
        v[0] = (float)lo * finv; v[1] = (float)hi * finv;
        tvals[slot] = 0ull;
      }
    } else {
      unsigned long long w[F];
#pragma unroll
      for (int f = 0; f < F; ++f) { w[f] = tvals[f * kSlots + slot]; any = any || w[f] != 0ull; }
      if (any) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
          v[f] = (NESVOR_HG_F64FIX != 0 ? from_fixed51(w[f]) : from_fixed(w[f])) * finv;
          tvals[f * kSlots + slot] = 0ull;
        }
      }
    }
    return any;
  };

  float dy_a[F], dy_b[F];
  int level = plan.level_begin;
  load_dy(level, dy_a);
  if (level + 1 < level_end) load_dy(level + 1, dy_b);

  // ======================================================================================================== (A)
  // Box rounds.  A round = consecutive levels whose lattice boxes fit the table together (for a PSF cloud: levels
  // 0-5, 6-8, 9-10, then one level per round): ONE insertion / drain / queue reservation / record write sequence and
  // two barriers for all of them.  The table work of a level costs the same latency chain whether its box has 27
  // or 700 vertices, so sharing it is what shortens the pass.  Pipeline (one barrier between the lines):
  //     W(r-1) write the previous round's records | D(r) drain the table into records, count them per bucket,
  //                                                 fill the keys and the table copy of round r+1
  //     R(r) reserve queue space (returning global atomics, ~2 us, hidden behind:)
  //                                               | P(r+1) per level of round r+1: prepare + insert the run tails
  constexpr int NRB = kSlots / 256;  // slots per thread: thread t owns slots t, t + 256, ... (fill and drain)
  if constexpr (MERGE) {
    // keys (tkeys) and table entries (tcache) of the slots of round [ra, rb): a slot is read and written by its owner
    // thread only, so only the table copy needs the barrier that follows.  The loads are returned in `feat` so that
    // they can fly while the caller does other work; store_feat() puts them into tcache.
    auto fill_keys = [&](int ra, int rb, float (&feat)[NRB][F]) __attribute__((always_inline)) {
      const uint32_t total = sgpr(rnd_slots[ra]);
#pragma unroll
      for (int j = 0; j < NRB; ++j) {
        const uint32_t slot = (uint32_t)j * 256u + (uint32_t)tid;
#pragma unroll
        for (int f = 0; f < F; ++f) feat[j][f] = 0.f;
        if (slot < total) {
          int lv = ra;
          for (int l = ra + 1; l < rb; ++l) lv = slot >= slot_off[l] ? l : lv;
          const uint32_t local = slot - slot_off[lv];
          const uint32_t nx = lbox[lv][3] + 2u, nxy = nx * (lbox[lv][4] + 2u);
          // local slot -> lattice point (local < 1024: the float quotients are exact after truncation)
          const uint32_t z = (uint32_t)(((float)local + 0.5f) * (1.f / (float)nxy));
          const uint32_t r = local - __umul24(z, nxy);
          const uint32_t y = (uint32_t)(((float)r + 0.5f) * (1.f / (float)nx));
          LevelParams pl;
          pl.scale = 0.f; pl.res = lpar[lv][0]; pl.size = lpar[lv][1]; pl.offset = lpar[lv][2]; pl.hashed = lpar[lv][3];
          const uint32_t key = corner_index(pl, lbox[lv][0] + (r - __umul24(y, nx)), lbox[lv][1] + y, lbox[lv][2] + z);
          tkeys[slot] = ((uint32_t)lv << 27) | key;
          if constexpr (INPUT_GRAD) load_feat<F>(table + ((size_t)pl.offset + key) * F, feat[j]);
        }
      }
    };
    auto store_feat = [&](int ra, const float (&feat)[NRB][F]) __attribute__((always_inline)) {
      if constexpr (INPUT_GRAD) {
        const uint32_t total = sgpr(rnd_slots[ra]);
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          const uint32_t slot = (uint32_t)j * 256u + (uint32_t)tid;
          if (slot < total) {
#pragma unroll
            for (int f = 0; f < F; ++f) tcache[slot * F + f] = feat[j][f];
          }
        }
      }
    };
    // P: prepare + insert every level of round [ra, rb)
    auto insert_round = [&](int ra, int rb) __attribute__((always_inline)) {
#pragma unroll 1
      for (int lv = ra; lv < rb; ++lv) {
        uint32_t idx_unused[8], s0, nx, nxy;
        float val[8][F];
        bool tail;
        prepare(std::true_type{}, lv, dy_a, idx_unused, s0, nx, nxy, val, tail);
#pragma unroll
        for (int f = 0; f < F; ++f) dy_a[f] = dy_b[f];
        if (lv + 2 < level_end) load_dy(lv + 2, dy_b);
        const float fs_lv = scale_of((uint32_t)lv);
        // (Lane-transposed inserts for waves with few run tails - four tails park their values in an LDS strip, 64 lanes convert and
        //  add one value each - were measured in round 4 and bought nothing: a coarse level's time is its latency chain, not the
        //  130 conversion instructions.  DESIGN_LOG.md / profiles/r04_hashgrid_ab_transposed_inserts.log.)
        if (!NESVOR_ABL(4) && tail) {
#pragma unroll
          for (int k = 0; k < 8; ++k) slot_add(s0 + (k & 1) + ((k >> 1) & 1) * nx + (k >> 2) * nxy, val[k], fs_lv);
        }
      }
    };

    if (level < box_end) {
      int ra = level, rb = (int)sgpr(grp_end[level]);
      {
        float feat[NRB][F];
        fill_keys(ra, rb, feat);
        store_feat(ra, feat);
      }
      if constexpr (INPUT_GRAD) __syncthreads();
      HG_TICK(4);  // first round's table copy in place
      insert_round(ra, rb);
      HG_TICK(5);  // a round's inserts issued (0x10 | first level below)
      uint32_t rkey[NRB], rmeta[NRB], rank[NRB], rmask = 0;  // records of the round being finished: entry index,
      float rval[NRB][F];                                    // level << 8 | round-local bucket, rank inside the bucket
      bool have_prev = false;
      for (;;) {
        __syncthreads();  // -- insertions of round [ra, rb) complete; bbase4 of the previous round published
        HG_TICK(0x20 | ra);  // inserts of the round starting at level ra complete
        // W(r-1)
        if (have_prev && !NESVOR_ABL(8)) {
#pragma unroll
          for (int j = 0; j < NRB; ++j) {
            if (!(rmask & (1u << j))) continue;
            const uint4 bb = bbase4[rmeta[j] & 0xFFu];
            const uint32_t pos = bb.x + rank[j];
            if (pos < bb.z) {
              uint32_t* r = records + ((size_t)bb.y + pos) * (1 + F);
              r[0] = rkey[j];
#pragma unroll
              for (int f = 0; f < F; ++f) r[1 + f] = __float_as_uint(rval[j][f]);
            } else {  // queue full: exact fallback (counted: the host grows the level's queues when it sees it)
              atomicAdd(&tails[kOverflowBase + (rmeta[j] >> 8)], 1u);
#pragma unroll
              for (int f = 0; f < F; ++f) atomicAdd(grad_table + ((size_t)lpar[rmeta[j] >> 8][2] + rkey[j]) * F + f, rval[j][f]);
            }
          }
        }
        // D(r): this thread's slots -> records; the next round's keys replace the current ones and its table entries are
        // fetched meanwhile
        const int na = rb, nb_ = na < box_end ? (int)sgpr(grp_end[na]) : na;
        const bool next_box = na < box_end;
        const uint32_t n_slots = sgpr(rnd_slots[ra]);
        uint32_t lk[NRB];
#pragma unroll
        for (int j = 0; j < NRB; ++j) lk[j] = tkeys[j * 256 + tid];
        float nfeat[NRB][F];
        if (next_box) {
          fill_keys(na, nb_, nfeat);
        } else {
#pragma unroll
          for (int j = 0; j < NRB; ++j) tkeys[j * 256 + tid] = kEmpty;  // the hashed table of the next level starts empty
        }
        rmask = 0;
        uint32_t n_last = 0;  // vertices of the round's last level that received something
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          rank[j] = 0; rkey[j] = 0; rmeta[j] = 0;
#pragma unroll
          for (int f = 0; f < F; ++f) rval[j][f] = 0.f;
          const uint32_t slot = (uint32_t)j * 256u + (uint32_t)tid;
          if (slot < n_slots && slot_take(slot, rval[j], inv_of(lk[j] >> 27))) {
            const uint32_t lv = lk[j] >> 27;
            rkey[j] = lk[j] & kKeyMask;
            const uint32_t bucket = bkt_off[lv] + (rkey[j] >> lpar[lv][7]);
            rmeta[j] = (lv << 8) | bucket;
            rmask |= 1u << j;
            rank[j] = atomicAdd(&bcount[bucket], 1u);
            n_last += lv + 1u == (uint32_t)rb ? 1u : 0u;
          }
        }
        if (!next_box) {
          // the last box level's count decides whether the first hashed level merges at all (see the test behind the loop)
          const uint32_t wave_last = (uint32_t)wave_sum_u32(n_last);
          if (lane == 0 && wave_last) atomicAdd(&merge_stat[1], wave_last);
        }
        if (next_box) store_feat(na, nfeat);  // every read of the current round's copy happened before the barrier above
        have_prev = true;
        HG_TICK(0x40 | ra);  // previous records written, round drained
        __syncthreads();  // -- bucket counts of round [ra, rb) complete, table drained, next round's copy in place
        HG_TICK(0x60 | ra);  // ... and everyone has
        // R(r): one returning (memory-side, ~2 us) atomic per non-empty bucket of the round
        const uint32_t nbk = sgpr(rnd_bkts[ra]);
        uint4 mine = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)tid < nbk) {
          int lv = ra;
          for (int l = ra + 1; l < rb; ++l) lv = (uint32_t)tid >= bkt_off[l] ? l : lv;
          const uint32_t chunk = (uint32_t)tid - bkt_off[lv], cap = lpar[lv][4];
          const uint32_t cnt = bcount[tid];
          bcount[tid] = 0;
          uint32_t pos0 = 0;
          if (cnt && !NESVOR_ABL(16)) pos0 = atomicAdd(&tails[sub * kTailStride + lpar[lv][5] + chunk], cnt);
          mine = make_uint4(pos0, lpar[lv][6] + (chunk * plan.n_sub + sub) * cap, cap, (uint32_t)lv);
        }
        // ... hidden behind P(r+1)
        level = rb;
        if (next_box) insert_round(na, nb_);
        HG_TICK(0x80 | ra);  // reservation issued, next round's inserts issued
        if ((uint32_t)tid < nbk) bbase4[tid] = mine;
        HG_TICK(0xA0 | ra);  // reservation returned
        if (!next_box) break;
        ra = na; rb = nb_;
      }
      __syncthreads();
      // The hashed table of the levels behind the boxes must not get crowded: vertices grow by up to 2x per level (1.45x on PSF
      // clouds), a table more than ~3/4 full walks tens of probes per insert.  More than half the slots at the last box level
      // (cell-ordered uniform points: ~600 of the 780 vertices of a level-11 box): the finer levels write direct records.
      // (Rounds 1-4 only looked at a hashed level after it had been inserted.)
      if (tid == 0) {
        if (NESVOR_HG_SPATIAL && merge_stat[1] * 2u > (uint32_t)kSlots) merge_off = 1u;
        prev_cnt = merge_stat[1];
        merge_stat[1] = 0u;
      }
      // W(last box round)
      if (!NESVOR_ABL(8)) {
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          if (!(rmask & (1u << j))) continue;
          const uint4 bb = bbase4[rmeta[j] & 0xFFu];
          const uint32_t pos = bb.x + rank[j];
          if (pos < bb.z) {
            uint32_t* r = records + ((size_t)bb.y + pos) * (1 + F);
            r[0] = rkey[j];
#pragma unroll
            for (int f = 0; f < F; ++f) r[1 + f] = __float_as_uint(rval[j][f]);
          } else {
            atomicAdd(&tails[kOverflowBase + (rmeta[j] >> 8)], 1u);
#pragma unroll
            for (int f = 0; f < F; ++f) atomicAdd(grad_table + ((size_t)lpar[rmeta[j] >> 8][2] + rkey[j]) * F + f, rval[j][f]);
          }
        }
      }
      __syncthreads();  // bb_raw is re-used (as bbase) by the single-level rounds below
    }
  }

  // ================================================================================================= (B), (C)
  // Single-level rounds: hashed merge table while it pays (merge_off is set once, workgroup-uniformly, before a
  // barrier), then direct records.  Two loops in sequence rather than a branch inside one loop, so that the compiler
  // cannot hoist the common second half of the two paths above the branch (which made everything of the next level live
  // during the insertion).
  if (level < level_end) {
    uint32_t idx[8], idx_n[8];
    float val[8][F], val_n[8][F];
    uint32_t su = 0, nxu = 1, nxyu = 1;  // (box-only outputs of prepare)
    bool tail, tail_n = false;
    prepare(std::false_type{}, level, dy_a, idx, su, nxu, nxyu, val, tail);
    bool merge = MERGE;
    // Second half of a level, specialised on the number NR of records a thread can hold (table slots per thread in
    // merge mode, the 8 corners otherwise) so that the merge path does not carry 8 record registers sets through the
    // next level's prepare(): reserve queue space, prepare the next level, write the records.
    // in_place: the current level's idx / val are dead (merge path: the records were drained from the table), so the
    // next level is prepared straight into them - no copy at the end of the level
    auto finish_level = [&](auto& rkey, auto& rank, auto& rval, uint32_t rmask, auto in_place) __attribute__((always_inline)) {
      constexpr int NR = sizeof(rkey) / sizeof(rkey[0]);
      HG_TICK(0xE0 | level);  // single-level round: drained / ranked
      __syncthreads();
      const int lu = __builtin_amdgcn_readfirstlane(level);  // (scalar index into the kernel arguments: see prepare())
      // reserve queue space: one returning (memory-side, ~2 us) atomic per non-empty chunk ...
      const uint32_t nb = plan.n_chunks[lu];
      const uint32_t cap = plan.cap[lu];
      uint32_t my_base = 0;
      if (tid < nb) {
        const uint32_t cnt = bcount[tid];
        if (cnt && !NESVOR_ABL(16)) my_base = atomicAdd(&tails[sub * kTailStride + plan.bucket_base[lu] + tid], cnt);
        bcount[tid] = 0;
      }
      if (merge && tid == 255) {
        // merging stops paying once fewer than a quarter of the records collapse, and must stop before the table gets
        // crowded
        const uint32_t drained = merge_stat[1];
        if (drained * 4u > merge_stat[0] * 3u || drained * 10u > (uint32_t)kSlots * 7u) merge_off = 1u;
        // (the next level at this level's growth: keep it below ~4/5 of the slots)
        if (NESVOR_HG_SPATIAL && prev_cnt != 0u && (uint64_t)drained * drained * 5u > (uint64_t)prev_cnt * (uint32_t)kSlots * 4u) merge_off = 1u;
        prev_cnt = drained;
        uint32_t lg = 8;
        while ((1u << lg) < 4u * drained && (1u << lg) < (uint32_t)kSlots) ++lg;
        slots_log2 = lg;
        merge_stat[0] = 0; merge_stat[1] = 0;
      }
      // ... and hide its latency behind the next level's register-only work
      if (level + 1 < level_end) {
#pragma unroll
        for (int f = 0; f < F; ++f) dy_a[f] = dy_b[f];
        if (level + 2 < level_end) load_dy(level + 2, dy_b);
        if constexpr (decltype(in_place)::value) prepare(std::false_type{}, level + 1, dy_a, idx, su, nxu, nxyu, val, tail);
        else prepare(std::false_type{}, level + 1, dy_a, idx_n, su, nxu, nxyu, val_n, tail_n);
      }
      if (tid < nb) bbase[level & 1][tid] = make_uint2(my_base, (tid * plan.n_sub + sub) * cap);
      HG_TICK(0xF0 | level);  // reservation returned, next level prepared
      __syncthreads();
      // records of the level: (entry, grad...) = (1 + F) words each; a level's queues stay below 2^32 bytes (make_plan)
      char* const level_rec = reinterpret_cast<char*>(records) + plan.rec_off[lu] * (uint64_t)(4 * (1 + F));
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if (NESVOR_ABL(8) || !(rmask & (1u << k))) continue;
        const uint2 bb = bbase[level & 1][rkey[k] >> plan.shift[lu]];
        const uint32_t pos = bb.x + rank[k];
        if (pos < cap) {
          uint32_t* r = reinterpret_cast<uint32_t*>(level_rec + (bb.y + pos) * (uint32_t)(4 * (1 + F)));
          r[0] = rkey[k];
#pragma unroll
          for (int f = 0; f < F; ++f) r[1 + f] = __float_as_uint(rval[k][f]);
        } else {  // queue full: exact fallback
          atomicAdd(&tails[kOverflowBase + level], 1u);
#pragma unroll
          for (int f = 0; f < F; ++f) atomicAdd(grad_table + ((size_t)g.offset[lu] + rkey[k]) * F + f, rval[k][f]);
        }
      }
    };
    auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        idx[k] = idx_n[k];
#pragma unroll
        for (int f = 0; f < F; ++f) val[k][f] = val_n[k][f];
      }
      tail = tail_n;
    };
    if constexpr (MERGE) {
      for (; level < level_end && merge_off == 0u; ++level) {
        constexpr int NR = kSlots / 256;
        uint32_t rkey[NR], rank[NR], rmask = 0;
        float rval[NR][F];
        const uint32_t slog = NESVOR_HG_SPATIAL ? (uint32_t)__builtin_ctz(kSlots) : slots_log2, smask = (1u << slog) - 1u;
        if (!NESVOR_ABL(4) && tail) {
          uint32_t h[8];
          uint32_t pending = 0;
          // claim / find the 8 slots: first probes issued together, collisions walked one by one
          if constexpr (NESVOR_HG_SPATIAL != 0) {
            const uint32_t wb = sgpr(lwin[level]);
            const uint32_t ba = wb & 255u, bb = (wb >> 8) & 255u, bc = wb >> 16;
            const uint32_t mx = (1u << ba) - 1u, my = (1u << bb) - 1u, mz = (1u << bc) - 1u;
            const uint32_t x0 = su & 1023u, y0 = (su >> 10) & 1023u, z0 = su >> 20;
            const uint32_t hx[2] = {x0 & mx, (x0 + 1u) & mx};
            const uint32_t hy[2] = {(y0 & my) << ba, ((y0 + 1u) & my) << ba};
            const uint32_t hz[2] = {(z0 & mz) << (ba + bb), ((z0 + 1u) & mz) << (ba + bb)};
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] = hx[k & 1] | hy[(k >> 1) & 1] | hz[k >> 2];
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] = (idx[k] * 2654435761u) >> (32 - slog);
          }
          uint32_t prev[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) prev[k] = atomicCAS(&tkeys[h[k]], kEmpty, idx[k]);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (prev[k] != kEmpty && prev[k] != idx[k]) pending |= 1u << k;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (pending & (1u << k)) {
              bool done = false;
              // double hashing (key-dependent odd step): at level 15 the table is two thirds full and linear probing
              // clusters (level 15: 0.067 -> 0.050 ms; probing all unresolved corners of a lane per round was slower)
              const uint32_t step = ((idx[k] * 0x9E3779B1u) >> 20) | 1u;
#pragma unroll 1
              for (int probe = 0; probe < 64 && !done; ++probe) {
                h[k] = (h[k] + step) & smask;
                const uint32_t pv = atomicCAS(&tkeys[h[k]], kEmpty, idx[k]);
                done = pv == kEmpty || pv == idx[k];
              }
              if (done) pending &= ~(1u << k);
            }
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (pending & (1u << k)) {  // table crowded: exact fallback
#pragma unroll
              for (int f = 0; f < F; ++f) atomicAdd(grad_table + ((size_t)g.offset[__builtin_amdgcn_readfirstlane(level)] + idx[k]) * F + f, val[k][f]);
            } else {
              slot_add(h[k], val[k], scale_of((uint32_t)level));
            }
          }
        }
        const uint32_t n_tail = __builtin_popcountll(__ballot(tail));
        if (lane == 0 && n_tail) atomicAdd(&merge_stat[0], 8u * n_tail);
        HG_TICK(0xC0 | level);  // hashed level: slots claimed, adds issued
        __syncthreads();
        HG_TICK(0xD0 | level);  // ... by everyone
        // drain: slot -> register record, slot cleared for the next level
        uint32_t mine = 0;
        const uint32_t n_slots = 1u << slog;  // slots in use at this level: thread t drains t, t + 256, ...
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          rkey[j] = 0; rank[j] = 0;
#pragma unroll
          for (int f = 0; f < F; ++f) rval[j][f] = 0.f;
          const uint32_t slot = (uint32_t)j * 256u + (uint32_t)tid;
          if (slot >= n_slots) continue;
          const uint32_t key = tkeys[slot];
          if (key != kEmpty) {
            tkeys[slot] = kEmpty;
            rmask |= 1u << j; rkey[j] = key;
            slot_take(slot, rval[j], inv_of((uint32_t)level));
            rank[j] = atomicAdd(&bcount[key >> plan.shift[__builtin_amdgcn_readfirstlane(level)]], 1u);
            ++mine;
          }
        }
        const uint32_t wave_mine = (uint32_t)wave_sum_u32(mine);
        if (lane == 0 && wave_mine) atomicAdd(&merge_stat[1], wave_mine);
        finish_level(rkey, rank, rval, rmask, std::true_type{});
      }
    }
    merge = false;
    for (; level < level_end; ++level) {
      // rank of every record inside its chunk's span (LDS integer atomics: ~6 cycles / wave-instruction)
      uint32_t rank[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) rank[k] = tail ? atomicAdd(&bcount[idx[k] >> plan.shift[__builtin_amdgcn_readfirstlane(level)]], 1u) : 0u;
      finish_level(idx, rank, val, tail ? 0xFFu : 0u, std::false_type{});
      advance();
    }
  }
  HG_TICK(0x0F);  // last records written
  if constexpr (INPUT_GRAD) {
    if (valid) {
      if (plan.accumulate_u) { gux += grad_u[3 * i]; guy += grad_u[3 * i + 1]; guz += grad_u[3 * i + 2]; }
      grad_u[3 * i] = gux; grad_u[3 * i + 1] = guy; grad_u[3 * i + 2] = guz;
    }
  }
}

constexpr uint32_t kOwnerSlice = 1u << 18;  // records per owner workgroup: a PSF-cloud batch keeps every queue in ONE slice (sole writer, no atomics)

// grid.x = sum over buckets of ceil(cap / kOwnerSlice) slices; a slice past the queue tail exits at once.
constexpr int kOwnerThreads = 512;   // 8 waves, four workgroups per CU (32 KiB of LDS each): measured 0.058 ms vs 0.077 (1024 threads) / 0.073 (256)

// ADAM: the owner pass also applies the optimiser to the table.  The workgroup that completes a chunk's gradient updates the
// chunk's parameters and moments while the gradient is still in its LDS, so the table gradient is never written to (and read
// back from, and zero-filled in) HBM; `grad_table` is read - it may hold a gradient from an earlier backward, or the
// contributions of records that found their queue full - and what was non-zero in it is zero-filled.  The result equals this
// pass without ADAM followed by nesvor_adamw_step(table, grad_table, ..., zero_grad = 1) (bit for bit where a chunk's
// records fit one slice - every chunk of a PSF-cloud batch at the fine levels; in the order of the slices' atomic adds
// otherwise, as without ADAM).  Chunks whose records span several slices: every slice adds its sums to grad_table and takes a
// ticket (`done`, one counter per chunk, zero between launches); the slice that draws the last ticket does the update.
struct OwnerAdam {
  float* param;     // the table (entry 0 of level 0), updated in place
  float* exp_avg;   // first / second moments, same indexing
  float* exp_avg_sq;
  uint32_t* done;   // tickets, one per bucket
  AdamArgs a;
};

template <int F, bool COALESCED, bool ADAM = false>
__global__ __launch_bounds__(kOwnerThreads) void hashgrid_bwd_owner(const nesvor_grid_t g, const BwdPlan plan,
                                                          const uint32_t* __restrict__ tails,
                                                          const uint32_t* __restrict__ records,
                                                          float* __restrict__ grad_table, const OwnerAdam adam,
                                                          uint32_t id_stride,  // workgroup id -> (id * id_stride) mod grid: coprime to the grid size (1: identity)
                                                          uint32_t wg_base) {  // flat id of the launch's first workgroup (a launch for a level range starts at its first level's)
  __shared__ __attribute__((aligned(16))) float acc[kOwnerLdsFloats];
  __shared__ uint32_t ticket_s;
  const int tid = threadIdx.x;
  // decode (level, chunk, slice) from the flat workgroup id
  // (-DNESVOR_OWNER_FINE_FIRST=1: the workgroups of the finest level - the longest queues, 21 K records per chunk at level 15 of a
  //  PSF-cloud batch - get the lowest ids and start first.  Measured in the step, two alternating rounds: owner launch 95-97 us
  //  against 84-85, 1031-1033 against 1045-1050 it/s: the coarse levels' short workgroups fill the gaps better when they lead.
  //  -DNESVOR_OWNER_PF=2 / 4 (more of a chunk's AdamW operands requested before the record phase): 86-87 / 90-91 us against 87.)
#ifndef NESVOR_OWNER_FINE_FIRST
#define NESVOR_OWNER_FINE_FIRST 0
#endif
  uint32_t wg = (uint32_t)(((uint64_t)blockIdx.x * id_stride) % gridDim.x) + wg_base;
  int level = NESVOR_OWNER_FINE_FIRST ? g.n_levels - 1 : 0;
  uint32_t spl = 0;
  for (;; level += NESVOR_OWNER_FINE_FIRST ? -1 : 1) {
    spl = (plan.cap[level] * plan.n_sub + plan.slice[level] - 1) / plan.slice[level];
    const uint32_t cnt = plan.n_chunks[level] * spl;
    if (wg < cnt || (NESVOR_OWNER_FINE_FIRST ? level == 0 : level + 1 >= g.n_levels)) break;
    wg -= cnt;
  }
  if (level < plan.level_begin || level >= plan.level_end) return;  // split backward: another launch owns this level
  const uint32_t chunk = wg / spl, slice = wg % spl;
  const uint32_t gb = plan.bucket_base[level] + chunk;
  // the bucket's records = the concatenation of its sub-queues; record r of that virtual queue sits at
  // r + sum over the sub-queues that end at or before r of their unused tail
  // Latencies taken out of the workgroup's serial chain (round 4; NESVOR_OWNER_EARLY=0 restores the old order): the queue tails
  // are requested first and the LDS accumulator is zero-filled while they fly (it used to wait behind the early-exit test on
  // them); with ADAM the first float4 of the chunk's parameters / moments / old gradient per thread - which depend on nothing
  // the record phase produces - is requested here as well and lands under the record phase.
#ifndef NESVOR_OWNER_EARLY
#define NESVOR_OWNER_EARLY 1
#endif
  uint32_t traw[kSubQueues];
#pragma unroll
  for (int x = 0; x < kSubQueues; ++x) traw[x] = (uint32_t)x < plan.n_sub ? tails[x * kTailStride + gb] : 0u;
  const uint32_t e0 = chunk << plan.shift[level];
  const uint32_t ne = min((uint32_t)(1u << plan.shift[level]), g.size[level] - e0);
  // NESVOR_OWNER_PF float4 per thread and array (parameters, both moments, old gradient) are requested here, before the record
  // phase: 1 = the first of a full chunk's four sweeps (round 4), 4 = the whole chunk (64 more VGPRs: two workgroups per CU)
#ifndef NESVOR_OWNER_PF
#define NESVOR_OWNER_PF 1
#endif
  constexpr int kPF = NESVOR_OWNER_PF;
  float4 pf_p[kPF], pf_m[kPF], pf_v[kPF], pf_o[kPF];
#pragma unroll
  for (int j = 0; j < kPF; ++j) pf_p[j] = pf_m[j] = pf_v[j] = pf_o[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  bool prefetched = false;
  if constexpr (ADAM && NESVOR_OWNER_EARLY) {
    const uint32_t nf = ne * F;
    if (nf % 4u == 0u) {
      const size_t first = ((size_t)g.offset[level] + e0) * F;
      prefetched = true;
#pragma unroll
      for (int j = 0; j < kPF; ++j) {
        const uint32_t t = (uint32_t)tid + (uint32_t)j * kOwnerThreads;
        if (t < nf / 4u) {
          pf_p[j] = reinterpret_cast<const float4*>(adam.param + first)[t];
          pf_m[j] = reinterpret_cast<const float4*>(adam.exp_avg + first)[t];
          pf_v[j] = reinterpret_cast<const float4*>(adam.exp_avg_sq + first)[t];
          pf_o[j] = reinterpret_cast<const float4*>(grad_table + first)[t];
        }
      }
    }
  }
  if (NESVOR_OWNER_EARLY) {
    for (int t = tid; t < kOwnerLdsFloats / 4; t += kOwnerThreads) reinterpret_cast<float4*>(acc)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  uint32_t n = 0;
  uint32_t pre[kSubQueues], gap[kSubQueues];  // pre[x]: first virtual index of sub-queue x; gap[x]: hole before it
#pragma unroll
  for (int x = 0; x < kSubQueues; ++x) {
    const uint32_t nx = min(traw[x], plan.cap[level]);
    pre[x] = n;
    gap[x] = x ? plan.cap[level] - (n - pre[x - 1]) : 0u;
    n += nx;
  }
  auto slot_of = [&](uint32_t r) __attribute__((always_inline)) {
    uint32_t o = r;
#pragma unroll
    for (int x = 1; x < kSubQueues; ++x) o += r >= pre[x] ? gap[x] : 0u;
    return o;
  };
  const uint32_t r0 = slice * plan.slice[level];
  // slices that hold records (ADAM: at least one, which also updates a chunk that received nothing - the moments decay)
  const uint32_t n_part = ADAM ? max(1u, (n + plan.slice[level] - 1u) / plan.slice[level]) : 0u;
  if (ADAM ? slice >= n_part : r0 >= n) return;
  const uint32_t r1 = max(r0, min(n, r0 + plan.slice[level]));
  const bool sole_writer = n <= plan.slice[level];
  if (!NESVOR_OWNER_EARLY) {
    for (int t = tid; t < kOwnerLdsFloats / 4; t += kOwnerThreads) reinterpret_cast<float4*>(acc)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const uint32_t mask = (1u << plan.shift[level]) - 1u;
  const uint32_t* rec = records + (plan.rec_off[level] + (uint64_t)chunk * plan.n_sub * plan.cap[level]) * (1 + F);
  auto add_record = [&](uint32_t key, const float (&v)[F]) __attribute__((always_inline)) {
    const uint32_t local = key & mask;
    if constexpr (F == 2) {  // one 64-bit compare-and-swap adds both features
      unsigned long long* a = reinterpret_cast<unsigned long long*>(&acc[local * 2]);
      unsigned long long old = *reinterpret_cast<volatile unsigned long long*>(a), assumed;
      do {
        assumed = old;
        const float n0 = __uint_as_float((uint32_t)assumed) + v[0], n1 = __uint_as_float((uint32_t)(assumed >> 32)) + v[1];
        old = atomicCAS(a, assumed, ((unsigned long long)__float_as_uint(n1) << 32) | __float_as_uint(n0));
      } while (old != assumed);
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) lds_add_f32_cas(&acc[local * F + f], v[f]);
    }
  };
  // Records of one pixel sit next to each other in the queue and repeat the same few table entries
  // (shared cell corners), so neighbouring lanes must NOT take neighbouring records (measured: the
  // coalesced walk is 2x slower from compare-and-swap retries): each thread walks its own contiguous
  // sub-range, which spreads the lanes over the whole slice.
  // (sub-ranges are multiples of 32 records = 3 x 128 B so that a cache line is fetched by one thread only)
#ifndef NESVOR_OWNER_UNROLL
#define NESVOR_OWNER_UNROLL 8
#endif
  constexpr int kUnroll = NESVOR_OWNER_UNROLL;  // records in flight per thread
  if constexpr (COALESCED) {
    // merged queues: a workgroup of the aggregation pass emits every vertex once per level, so neighbouring
    // records no longer repeat a table entry and neighbouring lanes can take neighbouring records
#if defined(NESVOR_OWNER_PRED) && NESVOR_OWNER_PRED
    // batches of kUnroll predicated loads per thread: a bucket of a few thousand records (the common case) is one or two
    // batches - one or two memory latencies - instead of a batch loop plus a one-record-at-a-time tail loop
    for (uint32_t r = r0 + tid; r < r1; r += kUnroll * kOwnerThreads) {
      uint32_t key[kUnroll];
      float v[kUnroll][F];
#pragma unroll
      for (int j = 0; j < kUnroll; ++j) {
        const uint32_t rr = r + j * kOwnerThreads;
        key[j] = 0xFFFFFFFFu;
        if (rr < r1) {
          const uint32_t* q = rec + (size_t)slot_of(rr) * (1 + F);
          key[j] = q[0];
#pragma unroll
          for (int f = 0; f < F; ++f) v[j][f] = __uint_as_float(q[1 + f]);
        }
      }
#pragma unroll
      for (int j = 0; j < kUnroll; ++j)
        if (key[j] != 0xFFFFFFFFu) add_record(key[j], v[j]);
    }
#else
    uint32_t r = r0 + tid;
    for (; r + (kUnroll - 1) * kOwnerThreads < r1; r += kUnroll * kOwnerThreads) {
      uint32_t key[kUnroll];
      float v[kUnroll][F];
#pragma unroll
      for (int j = 0; j < kUnroll; ++j) {
        const uint32_t* q = rec + (size_t)slot_of(r + j * kOwnerThreads) * (1 + F);
        key[j] = q[0];
#pragma unroll
        for (int f = 0; f < F; ++f) v[j][f] = __uint_as_float(q[1 + f]);
      }
#pragma unroll
      for (int j = 0; j < kUnroll; ++j) add_record(key[j], v[j]);
    }
    for (; r < r1; r += kOwnerThreads) {
      const uint32_t* q = rec + (size_t)slot_of(r) * (1 + F);
      float v[F];
#pragma unroll
      for (int f = 0; f < F; ++f) v[f] = __uint_as_float(q[1 + f]);
      add_record(q[0], v);
    }
#endif
  } else {
  const uint32_t per = (((r1 - r0 + kOwnerThreads - 1) / kOwnerThreads) + 31u) & ~31u;
  uint32_t r = r0 + tid * per;
  const uint32_t rend = min(r1, r + per);
  for (; r + kUnroll <= rend; r += kUnroll) {
    uint32_t key[kUnroll];
    float v[kUnroll][F];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const uint32_t* q = rec + (size_t)slot_of(r + j) * (1 + F);
      key[j] = q[0];
#pragma unroll
      for (int f = 0; f < F; ++f) v[j][f] = __uint_as_float(q[1 + f]);
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) add_record(key[j], v[j]);
  }
  for (; r < rend; ++r) {
    const uint32_t* q = rec + (size_t)slot_of(r) * (1 + F);
    float v[F];
#pragma unroll
    for (int f = 0; f < F; ++f) v[f] = __uint_as_float(q[1 + f]);
    add_record(q[0], v);
  }
  }
  __syncthreads();
  float* out = grad_table + ((size_t)g.offset[level] + e0) * F;
  if constexpr (ADAM) {
    const size_t first = ((size_t)g.offset[level] + e0) * F;
    float* P = adam.param + first;
    float* M = adam.exp_avg + first;
    float* V = adam.exp_avg_sq + first;
    const uint32_t nf = ne * F;
    if (!sole_writer) {
      for (uint32_t t = tid; t < nf; t += kOwnerThreads) {
        const float a = acc[t];
        if (a != 0.f) atomicAdd(out + t, a);
      }
      __threadfence();
      __syncthreads();
      if (tid == 0) ticket_s = atomicAdd(&adam.done[gb], 1u);
      __syncthreads();
      if (ticket_s != n_part - 1u) return;
      if (tid == 0) adam.done[gb] = 0u;  // ready for the next launch
      __threadfence();
      for (uint32_t t = tid; t < nf; t += kOwnerThreads) {
        const float gsum = __hip_atomic_load(out + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float p = P[t], m = M[t], v = V[t];
        adam1(p, gsum, m, v, adam.a);
        P[t] = p; M[t] = m; V[t] = v;
        if (gsum != 0.f) out[t] = 0.f;
      }
      return;
    }
    if (nf % 4u == 0u) {  // (chunks and level offsets are multiples of 8 entries: aligned)
      const float4* a4 = reinterpret_cast<const float4*>(acc);
      float4* o4 = reinterpret_cast<float4*>(out);
      auto sweep = [&](uint32_t t, float4 p, float4 m, float4 v, const float4 o) __attribute__((always_inline)) {
        const float4 a = a4[t];
        adam1(p.x, a.x + o.x, m.x, v.x, adam.a); adam1(p.y, a.y + o.y, m.y, v.y, adam.a);
        adam1(p.z, a.z + o.z, m.z, v.z, adam.a); adam1(p.w, a.w + o.w, m.w, v.w, adam.a);
        reinterpret_cast<float4*>(P)[t] = p; reinterpret_cast<float4*>(M)[t] = m; reinterpret_cast<float4*>(V)[t] = v;
        if (o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) o4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      };
      uint32_t t = tid;
      if (prefetched) {  // (these float4 arrived during the record phase)
#pragma unroll
        for (int j = 0; j < kPF; ++j, t += kOwnerThreads)
          if (t < nf / 4u) sweep(t, pf_p[j], pf_m[j], pf_v[j], pf_o[j]);
      }
      for (; t < nf / 4u; t += kOwnerThreads)
        sweep(t, reinterpret_cast<float4*>(P)[t], reinterpret_cast<float4*>(M)[t], reinterpret_cast<float4*>(V)[t], o4[t]);
    } else {
      for (uint32_t t = tid; t < nf; t += kOwnerThreads) {
        const float o = out[t];
        float p = P[t], m = M[t], v = V[t];
        adam1(p, acc[t] + o, m, v, adam.a);
        P[t] = p; M[t] = m; V[t] = v;
        if (o != 0.f) out[t] = 0.f;
      }
    }
    return;
  }
#ifndef NESVOR_OWNER_F4
#define NESVOR_OWNER_F4 1
#endif
  if (NESVOR_OWNER_F4 && sole_writer && (ne * F) % 4u == 0u) {
    // only writer of the chunk: plain read-modify-write, four floats at a time where any of them is non-zero (chunks and
    // level offsets are multiples of 8 entries, so the float4 accesses are aligned)
    const float4* a4 = reinterpret_cast<const float4*>(acc);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (uint32_t t = tid; t < ne * F / 4u; t += kOwnerThreads) {
      const float4 a = a4[t];
      if (a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f) {
        float4 o = o4[t];
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        o4[t] = o;
      }
    }
  } else {
    for (uint32_t t = tid; t < ne * F; t += kOwnerThreads) {
      const float a = acc[t];
      if (a != 0.f) {
        if (sole_writer) out[t] += a;
        else atomicAdd(out + t, a);  // long queue shared by several slices (rare: coarse level, un-clustered input)
      }
    }
  }
}

inline uint32_t owner_grid(const nesvor_grid_t* g, const BwdPlan& plan, int l0 = 0, int l1 = NESVOR_MAX_LEVELS) {  // workgroups of levels [l0, l1)
  uint32_t n = 0;
  for (int l = l0; l < g->n_levels && l < l1; ++l) n += plan.n_chunks[l] * ((plan.cap[l] * plan.n_sub + plan.slice[l] - 1) / plan.slice[l]);
  return n;
}

// host: chunking / queue plan.  Returns false if the grid does not fit the plan's limits.
inline bool make_plan(const nesvor_grid_t* g, int64_t N, BwdPlan* plan, uint64_t* n_records, const float* queue_scale = nullptr) {
  const int F = g->n_features;
  uint32_t shift0 = 0;
  while ((1u << (shift0 + 1)) * (uint32_t)F <= (uint32_t)kOwnerLdsFloats) ++shift0;
  // NESVOR_HASHGRID_FINE_LEVELS (tuning hook, default 0): that many of the finest levels take half-size chunks (twice the
  // owner workgroups, half as long queues).  Measured on PSF clouds at N = 2^20: owner pass 0.050-0.054 ms for 0..4,
  // aggregation pass +1-3 % - the longest queues are not what sets the owner pass's duration.
  static const int fine_levels = []() { const char* e = getenv("NESVOR_HASHGRID_FINE_LEVELS"); return e ? atoi(e) : 0; }();
  // XCCs of the current device (partition): 32 CUs each on gfx950
  static const uint32_t n_xcc = []() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 8u;
    uint32_t n = 1;
    while (n < (uint32_t)kSubQueues && n * 2u * 32u <= (uint32_t)cus) n *= 2u;
    return n;
  }();
  const uint32_t n_sub = n_xcc;
  plan->n_sub = n_sub;
  // NESVOR_HASHGRID_CAP_SCALE (tests only; read on every call): shrinks the queue capacities so that the exact
  // overflow fallback is exercised
  const char* cs = getenv("NESVOR_HASHGRID_CAP_SCALE");
  const double cap_scale = cs ? atof(cs) : 0.0;
  uint32_t nb = 0;
  uint64_t off = 0;
  for (int l = 0; l < g->n_levels; ++l) {
    uint32_t shift = shift0;
    if (l >= g->n_levels - fine_levels && shift > 8 && ((g->size[l] + (1u << (shift - 1)) - 1) >> (shift - 1)) <= (uint32_t)kMaxChunks) --shift;
    plan->shift[l] = shift;
    const uint32_t nc = (g->size[l] + (1u << shift) - 1) >> shift;
    if (nc > (uint32_t)kMaxChunks) return false;
    plan->n_chunks[l] = nc;
    plan->bucket_base[l] = nb;
    nb += nc;
    // worst case 8N records per level; a chunk's share is split over kSubQueues producer groups (the XCCs, which
    // the dispatcher feeds round-robin): 1/n_sub each plus slack for the imbalance.  Overflow is handled exactly
    // (atomic fallback in the aggregation pass), so the capacity only has to make it rare.
    // queue_scale[l] in (0, 1] (host array, optional): the fraction of that worst case to provide for level l - a
    // PSF-cloud batch fills well under a tenth of it at most levels; the caller grows a level when its overflow counter
    // (nesvor_hashgrid_backward_overflow_offset) turns non-zero
    const uint64_t per = (uint64_t)(8 * N) / nc / n_sub;
    double qs = queue_scale ? (double)queue_scale[l] : 1.0;
    if (!(qs > 0.0) || qs > 1.0) qs = 1.0;
    uint64_t cap = (uint64_t)((double)(per + per / 8) * qs) + 1024;
    if (cap_scale > 0.0) cap = (uint64_t)((double)cap * cap_scale) + 1;  // test knob: force the overflow path
    if (cap * n_sub > 0x7FFFFFFFull) return false;
    if (cap * n_sub * nc * (uint64_t)(4 * (1 + F)) > 0xFFFFFFFFull) return false;  // a level's queues are addressed by 32-bit byte offsets
    plan->cap[l] = (uint32_t)cap;
    // small (coarse, dense) levels collect very many records on few entries: split their queue over many
    // workgroups (the closing atomics are then only entries x slices); big chunks keep one sole writer
    const uint32_t ents = g->size[l] < (1u << shift) ? g->size[l] : (1u << shift);
    uint64_t sl = (uint64_t)ents * 16;
    if (sl < 8192) sl = 8192;
    if (sl > kOwnerSlice) sl = kOwnerSlice;
    plan->slice[l] = (uint32_t)sl;
    plan->rec_off[l] = off;
    off += cap * n_sub * nc;
  }
  for (int l = g->n_levels; l < NESVOR_MAX_LEVELS; ++l) {
    plan->shift[l] = shift0;
    plan->n_chunks[l] = 0; plan->bucket_base[l] = nb; plan->cap[l] = 0; plan->slice[l] = kOwnerSlice; plan->rec_off[l] = off;
  }
  plan->n_buckets = nb;
  static const uint32_t box = []() { const char* e = getenv("NESVOR_HASHGRID_BOX"); return (e == nullptr || atoi(e) != 0) ? 1u : 0u; }();
  plan->box_slots = box;
  plan->level_begin = 0; plan->level_end = g->n_levels; plan->accumulate_u = 0;
  if (off > 0xFFFFFFFFull) return false;  // records are addressed by 32-bit record numbers
  *n_records = off;
  return true;
}

constexpr uint64_t kTailBytes = (uint64_t)kSubQueues * kTailStride * sizeof(uint32_t);  // one tails region; the workspace starts with two
constexpr uint64_t kDoneBytes = (uint64_t)kTailStride * sizeof(uint32_t);  // then the owner pass's per-chunk tickets (OwnerAdam::done)
constexpr uint64_t kHeadBytes = 2 * kTailBytes + kDoneBytes;               // zero-filled once by the caller; the records follow

// Which of a workspace's two tail regions the current backward uses.  A fresh backward (not a later launch of a split
// one) switches to the region the previous aggregation pass zero-filled; the caller zero-fills both once after
// allocating the workspace (nesvor_hashgrid_backward_workspace_zero_bytes).
inline int tails_parity(void* workspace, bool advance) {
  static std::mutex mu;
  static std::unordered_map<void*, int> parity;
  std::lock_guard<std::mutex> lock(mu);
  int& p = parity[workspace];
  if (advance) p ^= 1;
  return p;
}

// ---- unclustered input (NESVOR_LAYOUT_UNCLUSTERED): the points in the order of a coarse lattice's cells -------------------
// The aggregation pass lives on the lattice vertices that the 256 samples of a workgroup share (PSF clouds: all of them at the
// coarse levels).  256 CONSECUTIVE points of an arbitrary batch share nothing; 256 points of neighbouring coarse cells do.  The
// points are binned by the Morton code of a 2^b-per-axis grid (b chosen so that a cell holds 32..255 points: 32^3 cells at
// N = 2^20), ONE returning atomic per point:
//   place    point i -> slot (cell, k) of the cell's kCap-slot strip, k = its ticket; a ticket >= cap goes to a spill list
//            (a batch that is clustered after all: its dense cells overflow, the spilled points are processed in arrival order)
//   scan     first rank of every cell = exclusive sum of min(count, cap); the spill list follows the last cell (one workgroup,
//            the counters pass through LDS)
//   compact  perm[first rank of the cell + k] = the strip's k-th point
// Workgroup w of the aggregation pass takes points perm[256 w ..]: ~8 neighbouring cells, a box of ~1/16 of the cube's side,
// which the box rounds take up to level 11 of the headline grid (vertices shared 3-70x); the finer levels, where uniform points
// share nothing, write direct records as before.  (First version, gpurun_out/r05s: an exact counting sort - count, scan, place -
// 50 + 51 + 73 us at N = 2^20; both of its atomic passes ran at ~20 G atomics/s and its one-workgroup scan read the counters
// strided.)  The order inside a cell is that of the tickets: sums are formed in fixed point / per chunk owner, so the result does
// not depend on it beyond the owner pass's usual last-bit differences.
constexpr int kSortMaxBits = 6;  // <= 64 cells per axis (2^18 counters)
struct SortPlan {
  int b;            // cells per axis = 2^b
  uint32_t n_cells; // 8^b
  uint32_t cap;     // slots per cell: twice the mean occupancy, rounded up to a power of two
};
inline SortPlan sort_plan(int64_t N) {
  int lg = 0;
  while (((int64_t)2 << lg) <= N) ++lg;  // floor(log2 N)
  int b = (lg - 5) / 3;
  b = b < 1 ? 1 : (b > kSortMaxBits ? kSortMaxBits : b);
  SortPlan p;
  p.b = b; p.n_cells = 1u << (3 * b);
  const int64_t mean = (N + p.n_cells - 1) / p.n_cells;
  uint32_t cap = 2;
  while ((int64_t)cap < 2 * mean) cap *= 2;
  p.cap = cap;
  return p;
}
// scratch behind `order` (all uint32 unless noted): perm [n_pad] | count [n_cells + 1, the last = spilled points] |
// first [n_cells + 2] | spill [n_pad] | where [n_pad] uint64: (cell << 32 | ticket) of every point | strips [n_cells * cap]
struct SortBufs {
  uint32_t *perm, *count, *first, *spill, *strips;
  unsigned long long* where;
};
inline uint64_t sort_max_cells() { return (uint64_t)1 << (3 * kSortMaxBits); }
inline SortBufs sort_bufs(uint32_t* base, int64_t N, const SortPlan& p) {
  const size_t n_pad = (size_t)((N + 255) / 256) * 256;
  SortBufs s;
  s.perm = base;
  s.count = s.perm + n_pad;
  s.first = s.count + sort_max_cells() + 2;
  s.spill = s.first + sort_max_cells() + 2;
  s.where = reinterpret_cast<unsigned long long*>(s.spill + n_pad);
  s.strips = reinterpret_cast<uint32_t*>(s.where + n_pad);
  return s;
}
inline uint64_t sort_bytes(int64_t N) {
  const uint64_t n_pad = (uint64_t)((N + 255) / 256) * 256;
  const SortPlan p = sort_plan(N);
  return (n_pad * 2 + 2 * (sort_max_cells() + 2)) * sizeof(uint32_t) + n_pad * sizeof(unsigned long long) +
         (uint64_t)p.n_cells * p.cap * sizeof(uint32_t);
}
__device__ __forceinline__ uint32_t sort_cell(const float* __restrict__ u, int64_t i, int b) {
  const float s = (float)(1 << b);
  const int top = (1 << b) - 1;
  // (NaN / out-of-range coordinates land in an edge cell: the order is only a matter of speed)
  const int x = min(max((int)(u[3 * i] * s), 0), top), y = min(max((int)(u[3 * i + 1] * s), 0), top), z = min(max((int)(u[3 * i + 2] * s), 0), top);
  return spread3((uint32_t)x) | (spread3((uint32_t)y) << 1) | (spread3((uint32_t)z) << 2);
}
__global__ __launch_bounds__(256) void sort_place_kernel(const float* __restrict__ u, int64_t N, SortPlan p, SortBufs s) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  uint32_t c = sort_cell(u, i, p.b);
  uint32_t k = atomicAdd(&s.count[c], 1u);
  if (k < p.cap) s.strips[(size_t)c * p.cap + k] = (uint32_t)i;
  else {
    k = atomicAdd(&s.count[p.n_cells], 1u);
    c = p.n_cells;
    s.spill[k] = (uint32_t)i;
  }
  s.where[i] = ((unsigned long long)c << 32) | k;
}
// first[c] = sum over c' < c of min(count[c'], cap) for c <= n_cells; first[n_cells + 1] = N.  One workgroup of 1024 threads; the
// counters go through LDS in tiles of 32768 (coalesced loads, every thread then owns 32 consecutive ones).
__global__ __launch_bounds__(1024) void sort_scan_kernel(SortPlan p, SortBufs s) {
  constexpr int kTile = 32768, kPer = kTile / 1024;
  extern __shared__ uint32_t cnt[];  // [kTile + kTile / 32]: one pad word per 32 (a thread's 32 counters would share a bank)
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  const uint32_t total = p.n_cells + 1;  // (the spill counter is the last "cell", uncapped)
  for (uint32_t t0 = 0; t0 < total; t0 += kTile) {
    {
      uint32_t v[kPer];  // (all loads of the tile in flight before the first LDS store)
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const uint32_t c = t0 + (uint32_t)(k * 1024 + tid);
        v[k] = c < total ? s.count[c] : 0u;
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const uint32_t c = t0 + (uint32_t)(k * 1024 + tid);
        const int j = k * 1024 + tid;
        cnt[j + (j >> 5)] = c < p.n_cells ? min(v[k], p.cap) : v[k];
      }
    }
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) sum += cnt[tid * kPer + k + tid];  // (tid * 32 + k + pad(tid))
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
      if ((tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = carry_s + incl - sum;
    for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int j = tid * kPer + k;
      const uint32_t v = cnt[j + tid];
      cnt[j + tid] = before;
      before += v;
    }
    __syncthreads();
    if (tid == 1023) carry_s = before;
    for (int k = 0; k < kPer; ++k) {
      const uint32_t c = t0 + (uint32_t)(k * 1024 + tid);
      const int j = k * 1024 + tid;
      if (c < total) s.first[c] = cnt[j + (j >> 5)];
    }
    __syncthreads();
  }
  if (tid == 0) s.first[total] = carry_s;
}
__global__ __launch_bounds__(256) void sort_compact_kernel(SortPlan p, SortBufs s) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t n_strip = (uint64_t)p.n_cells * p.cap;
  if (t < n_strip) {
    const uint32_t c = (uint32_t)(t / p.cap), k = (uint32_t)(t - (uint64_t)c * p.cap);
    if (k < min(s.count[c], p.cap)) s.perm[s.first[c] + k] = s.strips[t];
  } else {
    const uint64_t k = t - n_strip;
    if (k < (uint64_t)s.count[p.n_cells]) s.perm[s.first[p.n_cells] + (uint32_t)k] = s.spill[k];
  }
}
// Feature-major dpe (E, N) -> rows (N, E) in cell order.  The aggregation pass reads a point's dpe level by level; through perm
// that is a 4-byte access per (point, feature) into rows of N floats - 33 M scattered sectors at N = 2^20, measured 1.80 ms for
// the backward against 0.89 ms on row-major input.  Here the columns are READ in their own order (256 consecutive floats per row
// and workgroup) and every point's E values are WRITTEN as one row of 4 E bytes at its place in the order: both sides whole lines.
constexpr int kGatherPts = 256;
__global__ __launch_bounds__(256) void gather_dy_rows_kernel(const float* __restrict__ dpe, SortBufs s, float* __restrict__ rows, int64_t N, int E) {
  extern __shared__ float tile[];  // [kGatherPts][E + 1]
  __shared__ uint32_t rank[kGatherPts];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kGatherPts;
  const bool ok = i0 + tid < N;
  if (ok) {
    const unsigned long long w = s.where[i0 + tid];
    rank[tid] = s.first[(uint32_t)(w >> 32)] + (uint32_t)w;
  }
  for (int e0 = 0; e0 < E; e0 += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (ok && e0 + q < E) ? dpe[(size_t)(e0 + q) * N + i0 + tid] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (e0 + q < E) tile[tid * (E + 1) + e0 + q] = v[q];
  }
  __syncthreads();
  if ((E & 3) == 0) {  // 16 bytes per store: a row of E floats is E / 4 consecutive lanes
    const int E4 = E >> 2;
    for (int idx = tid; idx < kGatherPts * E4; idx += 256) {
      const int q = idx / E4, e = (idx - q * E4) * 4;
      const float* t = tile + q * (E + 1) + e;
      if (i0 + q < N) *reinterpret_cast<float4*>(rows + (size_t)rank[q] * E + e) = make_float4(t[0], t[1], t[2], t[3]);
    }
  } else {
    for (int idx = tid; idx < kGatherPts * E; idx += 256) {
      const int q = idx / E, e = idx - q * E;
      if (i0 + q < N) rows[(size_t)rank[q] * E + e] = tile[q * (E + 1) + e];
    }
  }
}
// The inverse, for the unclustered FORWARD: rows (N, E) in cell order -> feature-major pe (E, N).  Every point's row is READ
// whole (4 E contiguous bytes at its rank), the columns are WRITTEN in their own order (256 consecutive floats per row and
// workgroup).
__global__ __launch_bounds__(256) void scatter_pe_rows_kernel(const float* __restrict__ rows, SortBufs s, float* __restrict__ pe, int64_t N, int E) {
  extern __shared__ float tile[];  // [kGatherPts][E + 1]
  __shared__ uint32_t rank[kGatherPts];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kGatherPts;
  const bool ok = i0 + tid < N;
  if (ok) {
    const unsigned long long w = s.where[i0 + tid];
    rank[tid] = s.first[(uint32_t)(w >> 32)] + (uint32_t)w;
  }
  __syncthreads();
  if ((E & 3) == 0) {
    const int E4 = E >> 2;
    for (int idx = tid; idx < kGatherPts * E4; idx += 256) {
      const int q = idx / E4, e = (idx - q * E4) * 4;
      if (i0 + q < N) {
        const float4 v = *reinterpret_cast<const float4*>(rows + (size_t)rank[q] * E + e);
        float* t = tile + q * (E + 1) + e;
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
      }
    }
  } else {
    for (int idx = tid; idx < kGatherPts * E; idx += 256) {
      const int q = idx / E, e = idx - q * E;
      if (i0 + q < N) tile[q * (E + 1) + e] = rows[(size_t)rank[q] * E + e];
    }
  }
  __syncthreads();
  if (ok) {
    for (int e = 0; e < E; ++e) pe[(size_t)e * N + i0 + tid] = tile[tid * (E + 1) + e];
  }
}
inline int sort_points(const float* u, int64_t N, uint32_t* base, hipStream_t st) {
  const SortPlan p = sort_plan(N);
  const SortBufs s = sort_bufs(base, N, p);
  hipError_t e = hipMemsetAsync(s.count, 0, sizeof(uint32_t) * (p.n_cells + 1), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(sort_place_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, u, N, p, s);
  static std::once_flag raised;
  constexpr size_t kScanLds = sizeof(uint32_t) * (32768 + 32768 / 32);
  std::call_once(raised, []() { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sort_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScanLds); });
  hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), kScanLds, st, p, s);
  const uint64_t n_threads = (uint64_t)p.n_cells * p.cap + (uint64_t)((N + 255) / 256) * 256;
  hipLaunchKernelGGL(sort_compact_kernel, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, st, p, s);
  return (int)hipGetLastError();
}

template <int F, int LAYOUT>
int launch_bwd_owner(const nesvor_grid_t* g, const float* u, const float* table, const float* dpe, float* gt,
                     float* gu, int64_t N, void* workspace, int stages, int level_begin, int level_end, const float* queue_scale,
                     const float* dy_bound, const OwnerAdam* adam, hipStream_t st, int hints = 0) {
  const bool unclustered = (hints & NESVOR_LAYOUT_UNCLUSTERED) != 0;
  if constexpr (LAYOUT == NESVOR_LAYOUT_FEATURE_MAJOR) {
    // unclustered feature-major dpe with room for its re-ordered copy (NESVOR_LAYOUT_DY_SCRATCH: the workspace was sized by
    // nesvor_hashgrid_backward_workspace_bytes_ex with the same layout): the aggregation pass runs in its row-major form on the copy
    if (unclustered && (hints & NESVOR_LAYOUT_DY_SCRATCH) && (stages & 1))
      return launch_bwd_owner<F, NESVOR_LAYOUT_ROW_MAJOR>(g, u, table, dpe, gt, gu, N, workspace, stages, level_begin, level_end, queue_scale,
                                                          dy_bound, adam, st, hints | 0x10000);
  }
  BwdPlan plan;
  uint64_t n_rec;
  if (!make_plan(g, N, &plan, &n_rec, queue_scale)) return (int)hipErrorInvalidValue;
  plan.level_begin = level_begin; plan.level_end = level_end;
  plan.accumulate_u = (stages & 8) ? 1 : 0;
  // bit 4: a later launch of a split backward - same region, the queue tails of its levels are still zero
  const int par = tails_parity(workspace, (stages & 1) && !(stages & 4));
  uint32_t* tails = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + (par ? kTailBytes : 0));
  uint32_t* tails_next = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + (par ? 0 : kTailBytes));
  uint32_t* records = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + kHeadBytes);
  uint8_t* order = reinterpret_cast<uint8_t*>(records) + n_rec * (1 + F) * sizeof(uint32_t);  // ceil(N / 256) * 256 bytes
  const int order_mode = (stages & 4) ? 2 : 1;
  // unclustered input: the points in cell order (a later launch of a split backward finds the first launch's order in place)
  // (256-byte aligned in absolute terms: the records in front have any multiple of 4 bytes; the size query leaves the slack)
  uint32_t* const perm_buf = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(order) + (uintptr_t)((N + 255) / 256) * 256 + 255) & ~(uintptr_t)255);
  const uint32_t* const perm = unclustered ? perm_buf : nullptr;
  dim3 grid((unsigned)((N + 255) / 256)), block(256);
  hipError_t e;
  int dy_by_slot = 0;
  if (!(stages & 1)) goto owner_stage;
  if (unclustered && order_mode == 1) {
    const int se = sort_points(u, N, perm_buf, st);
    if (se) return se;
  }
  if (hints & 0x10000) {  // (re-entered from the feature-major instantiation: see the top)
    const int E = g->n_levels * F;
    float* rows = reinterpret_cast<float*>(reinterpret_cast<char*>(perm_buf) + ((sort_bytes(N) + 255) & ~(uint64_t)255));
    // (a later launch of a split backward finds the first launch's copy in place)
    if (order_mode == 1) {
      const size_t lds = sizeof(float) * kGatherPts * (E + 1);
      if (lds > 48 * 1024) {
        static std::mutex mu;
        static size_t raised = 0;
        std::lock_guard<std::mutex> lock(mu);
        if (lds > raised) {
          e = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_dy_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (e != hipSuccess) return (int)e;
          raised = lds;
        }
      }
      hipLaunchKernelGGL(gather_dy_rows_kernel, dim3((unsigned)((N + kGatherPts - 1) / kGatherPts)), dim3(256), lds, st, dpe,
                         sort_bufs(perm_buf, N, sort_plan(N)), rows, N, E);
    }
    dpe = rows;
    dy_by_slot = 1;
  }
  {
    static const bool merge = []() { const char* e = getenv("NESVOR_HASHGRID_MERGE"); return e == nullptr || atoi(e) != 0; }();
#define NESVOR_LAUNCH_AGG(IG, MG, BD)                                                                                     \
    hipLaunchKernelGGL((hashgrid_bwd_aggregate<F, LAYOUT, IG, MG, BD>), grid, block, 0, st, *g, plan, u, table, dpe, gt, gu, \
                       tails, tails_next, records, N, dy_bound, order, order_mode, perm, dy_by_slot)
    const bool bounded = dy_bound != nullptr && merge;  // (the bound only scales the merge table's fixed-point sums)
    if (gu != nullptr) { if (bounded) NESVOR_LAUNCH_AGG(true, true, true); else if (merge) NESVOR_LAUNCH_AGG(true, true, false); else NESVOR_LAUNCH_AGG(true, false, false); }
    else { if (bounded) NESVOR_LAUNCH_AGG(false, true, true); else if (merge) NESVOR_LAUNCH_AGG(false, true, false); else NESVOR_LAUNCH_AGG(false, false, false); }
#undef NESVOR_LAUNCH_AGG
  }
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
owner_stage:
  if (stages & 2) {
    static const int coalesced = []() { const char* e = getenv("NESVOR_OWNER_COALESCED"); return e == nullptr ? 1 : atoi(e); }();
    OwnerAdam oa{};
    // NESVOR_OWNER_STRIDE=<odd number> (A/B switch): deal the (level, chunk) pairs to the workgroup ids with that stride instead
    // of level by level; raised to the next number coprime to the grid size
    static const uint32_t want = []() { const char* e = getenv("NESVOR_OWNER_STRIDE"); return e ? (uint32_t)atoi(e) : 1u; }();
    // only the workgroups of this launch's level range (a launch over all ids, the others returning at once, cost the pipelined
    // table update of the training step 30 us per pair of launches: gpurun_out/r06g)
    const uint32_t wg_base = NESVOR_OWNER_FINE_FIRST ? 0u : owner_grid(g, plan, 0, plan.level_begin);
    const uint32_t og = NESVOR_OWNER_FINE_FIRST ? owner_grid(g, plan) : owner_grid(g, plan, plan.level_begin, plan.level_end);
    if (og == 0u) return (int)hipGetLastError();
    uint32_t id_stride = want < 1u ? 1u : want;
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
    while (id_stride > 1u && gcd(id_stride, og) != 1u) ++id_stride;
    if (adam != nullptr) {
      oa = *adam;
      oa.done = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + 2 * kTailBytes);
      hipLaunchKernelGGL((hashgrid_bwd_owner<F, true, true>), dim3(og), dim3(kOwnerThreads), 0, st, *g, plan, tails, records, gt, oa, id_stride, wg_base);
    } else if (coalesced)
      hipLaunchKernelGGL((hashgrid_bwd_owner<F, true>), dim3(og), dim3(kOwnerThreads), 0, st, *g, plan, tails, records, gt, oa, id_stride, wg_base);
    else
      hipLaunchKernelGGL((hashgrid_bwd_owner<F, false>), dim3(og), dim3(kOwnerThreads), 0, st, *g, plan, tails, records, gt, oa, id_stride, wg_base);
  }
  return (int)hipGetLastError();
}

// scratch of the unclustered forward: the order (sort_bytes) | feature-major output: rows (n_pad, E) in that order
inline uint64_t fwd_ws_bytes(const nesvor_grid_t* g, int64_t N, int layout) {
  const uint64_t n_pad = (uint64_t)((N + 255) / 256) * 256;
  uint64_t b = 256 + ((sort_bytes(N) + 255) & ~(uint64_t)255);
  if ((layout & NESVOR_LAYOUT_MASK) == NESVOR_LAYOUT_FEATURE_MAJOR) b += n_pad * (uint64_t)(g->n_levels * g->n_features) * sizeof(float);
  return b;
}
template <int F, int LAYOUT>
int launch_fwd_unclustered(const nesvor_grid_t* g, const float* u, const float* table, float* pe, int64_t N, float* pe_absmax,
                           void* workspace, hipStream_t st) {
  uint32_t* const perm_buf = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  const int se = sort_points(u, N, perm_buf, st);
  if (se) return se;
  float* rows = nullptr;
  const int E = g->n_levels * F;
  if constexpr (LAYOUT == NESVOR_LAYOUT_FEATURE_MAJOR)
    rows = reinterpret_cast<float*>(reinterpret_cast<char*>(perm_buf) + ((sort_bytes(N) + 255) & ~(uint64_t)255));
  hipLaunchKernelGGL((hashgrid_fwd_cloud<F, LAYOUT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, *g, u, table, pe, N, pe_absmax,
                     (const uint32_t*)perm_buf, rows);
  if constexpr (LAYOUT == NESVOR_LAYOUT_FEATURE_MAJOR) {
    const size_t lds = sizeof(float) * kGatherPts * (E + 1);
    if (lds > 48 * 1024) {
      static std::mutex mu;
      static size_t raised = 0;
      std::lock_guard<std::mutex> lock(mu);
      if (lds > raised) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scatter_pe_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        raised = lds;
      }
    }
    hipLaunchKernelGGL(scatter_pe_rows_kernel, dim3((unsigned)((N + kGatherPts - 1) / kGatherPts)), dim3(256), lds, st, rows,
                       sort_bufs(perm_buf, N, sort_plan(N)), pe, N, E);
  }
  return (int)hipGetLastError();
}

template <int F, int LAYOUT>
int launch_fwd(const nesvor_grid_t* g, const float* u, const float* table, float* pe, int64_t N, int clustered, float* pe_absmax, hipStream_t st) {
  // Two kernels, identical results: "cloud" (one workgroup per 256 samples, all levels: the fastest on spatially
  // clustered batches - PSF clouds - and 1.4x slower than the other on unclustered points) is taken when the caller
  // says its batch is clustered (NESVOR_LAYOUT_CLUSTERED); "level" (one block per 256 samples and level) otherwise.
  // NESVOR_HASHGRID_FWD=cloud|level|gather forces one (gather = level without the LDS box copy).
  static const int forced = []() {
    const char* e = getenv("NESVOR_HASHGRID_FWD");
    if (e == nullptr) return -1;
    return e[0] == 'c' ? 0 : (e[0] == 'g' ? 2 : 1);
  }();
  const int mode = forced >= 0 ? forced : (clustered ? 0 : 1);
  if (mode == 0) {
    hipLaunchKernelGGL((hashgrid_fwd_cloud<F, LAYOUT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, *g, u, table, pe, N, pe_absmax);
    return (int)hipGetLastError();
  }
  dim3 grid((unsigned)((N + 255) / 256), g->n_levels), block(256);
  hipLaunchKernelGGL((hashgrid_fwd<F, LAYOUT>), grid, block, 0, st, *g, u, table, pe, N, mode == 1 ? 1 : 0, pe_absmax);
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

extern "C" int nesvor_hashgrid_forward_bounded(const nesvor_grid_t* grid, const float* u, const float* table, float* pe,
                                               int64_t N, int layout, float* pe_absmax, void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  const int clustered = (layout & NESVOR_LAYOUT_CLUSTERED) ? 1 : 0;
  layout &= NESVOR_LAYOUT_MASK;
  DISPATCH_F_LAYOUT(launch_fwd, grid, u, table, pe, N, clustered, pe_absmax, (hipStream_t)stream);
}

extern "C" int nesvor_hashgrid_forward(const nesvor_grid_t* grid, const float* u, const float* table, float* pe,
                                       int64_t N, int layout, void* stream) {
  return nesvor_hashgrid_forward_bounded(grid, u, table, pe, N, layout, nullptr, stream);
}

namespace {
template <int F, int LAYOUT>
int launch_fwd_levels(const nesvor_grid_t* g, const float* u, const float* table, float* pe, int64_t N, float* pe_absmax, int lb, int le,
                      hipStream_t st) {
  hipLaunchKernelGGL((hashgrid_fwd_cloud<F, LAYOUT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, *g, u, table, pe, N, pe_absmax,
                     (const uint32_t*)nullptr, (float*)nullptr, lb, le);
  return (int)hipGetLastError();
}
}  // namespace

extern "C" int nesvor_hashgrid_forward_levels(const nesvor_grid_t* grid, const float* u, const float* table, float* pe, int64_t N,
                                              int layout, float* pe_absmax, int level_begin, int level_end, void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  if (level_begin < 0 || level_end > grid->n_levels || level_begin >= level_end) return (int)hipErrorInvalidValue;
  layout &= NESVOR_LAYOUT_MASK;
  DISPATCH_F_LAYOUT(launch_fwd_levels, grid, u, table, pe, N, pe_absmax, level_begin, level_end, (hipStream_t)stream);
}

extern "C" int64_t nesvor_hashgrid_forward_workspace_bytes(const nesvor_grid_t* grid, int64_t N, int layout) {
  if (grid == nullptr || N <= 0) return 0;
  if (!(layout & NESVOR_LAYOUT_UNCLUSTERED)) return 0;
  return (int64_t)fwd_ws_bytes(grid, N, layout);
}

extern "C" int nesvor_hashgrid_forward_unclustered(const nesvor_grid_t* grid, const float* u, const float* table, float* pe,
                                                   int64_t N, int layout, float* pe_absmax, void* workspace, int64_t workspace_bytes,
                                                   void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  if (N >= ((int64_t)1 << 32)) return (int)hipErrorInvalidValue;  // (32-bit point indices in the order)
  if (workspace == nullptr || workspace_bytes < (int64_t)fwd_ws_bytes(grid, N, layout)) return (int)hipErrorInvalidValue;
  layout &= NESVOR_LAYOUT_MASK;
  DISPATCH_F_LAYOUT(launch_fwd_unclustered, grid, u, table, pe, N, pe_absmax, workspace, (hipStream_t)stream);
}

extern "C" int nesvor_hashgrid_backward_atomic(const nesvor_grid_t* grid, const float* u, const float* table,
                                               const float* dpe, float* grad_table, float* grad_u, int64_t N,
                                               int layout, void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  layout &= NESVOR_LAYOUT_MASK;  // (hints of the other kernels)
  DISPATCH_F_LAYOUT(launch_bwd, grid, u, table, dpe, grad_table, grad_u, N, (hipStream_t)stream);
}

extern "C" int64_t nesvor_hashgrid_backward_workspace_bytes_ex(const nesvor_grid_t* grid, int64_t N, const float* queue_scale, int layout) {
  int64_t base = nesvor_hashgrid_backward_workspace_bytes(grid, N, queue_scale);
  if (base <= 0) return base;
  // (round-5 advisor) the order's scratch - ~26 MB at N = 2^20 - only for callers that will pass NESVOR_LAYOUT_UNCLUSTERED; the
  // plain query above keeps it (its callers may pass any hint later)
  if (!(layout & NESVOR_LAYOUT_UNCLUSTERED)) base -= (int64_t)((sort_bytes(N) + 255) & ~(uint64_t)255);
  if ((layout & NESVOR_LAYOUT_MASK) == NESVOR_LAYOUT_FEATURE_MAJOR && (layout & NESVOR_LAYOUT_UNCLUSTERED) && (layout & NESVOR_LAYOUT_DY_SCRATCH))
    return base + 256 + (int64_t)((N + 255) / 256) * 256 * grid->n_levels * grid->n_features * (int64_t)sizeof(float);
  return base;
}

extern "C" int64_t nesvor_hashgrid_backward_workspace_bytes(const nesvor_grid_t* grid, int64_t N, const float* queue_scale) {
  if (N <= 0) return 0;
  BwdPlan plan;
  uint64_t n_rec;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return -1;
  if (!make_plan(grid, N, &plan, &n_rec, queue_scale)) return -1;
  if (plan.n_buckets > kOverflowBase) return -1;
  return (int64_t)(kHeadBytes + n_rec * (1 + grid->n_features) * sizeof(uint32_t) + (uint64_t)((N + 255) / 256) * 256 + 256 + ((sort_bytes(N) + 255) & ~(uint64_t)255));
}

extern "C" int64_t nesvor_hashgrid_backward_workspace_zero_bytes(void) { return (int64_t)kHeadBytes; }

#if NESVOR_HG_TIMELINE
// (debug builds only; not part of include/nesvor_hip.h)  64 workgroups x 96 marks of (site << 56 | 100 MHz time)
extern "C" int nesvor_debug_hg_timeline(unsigned long long* dst_host) {
  return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_hg_timeline), sizeof(unsigned long long) * kTlWgs * kTlMarks);
}
#endif

extern "C" int64_t nesvor_hashgrid_backward_overflow_offset(void* workspace) {
  return (int64_t)(tails_parity(workspace, false) ? kTailBytes : 0) + (int64_t)kOverflowBase * (int64_t)sizeof(uint32_t);
}

extern "C" int nesvor_hashgrid_backward(const nesvor_grid_t* grid, const float* u, const float* table,
                                        const float* dpe, float* grad_table, float* grad_u, int64_t N, int layout,
                                        void* workspace, int stages, const float* queue_scale, void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  if (workspace == nullptr || (stages & 3) == 0) return (int)hipErrorInvalidValue;
  const int hints = layout & (NESVOR_LAYOUT_UNCLUSTERED | NESVOR_LAYOUT_DY_SCRATCH);
  layout &= NESVOR_LAYOUT_MASK;
  DISPATCH_F_LAYOUT(launch_bwd_owner, grid, u, table, dpe, grad_table, grad_u, N, workspace, stages & 3, 0, grid->n_levels,
                    queue_scale, nullptr, nullptr, (hipStream_t)stream, hints);
}

extern "C" int nesvor_hashgrid_backward_levels(const nesvor_grid_t* grid, const float* u, const float* table,
                                               const float* dpe, float* grad_table, float* grad_u, int64_t N, int layout,
                                               void* workspace, int stages, int level_begin, int level_end,
                                               const float* queue_scale, void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  if (workspace == nullptr || (stages & 3) == 0) return (int)hipErrorInvalidValue;
  if (level_begin < 0 || level_end > grid->n_levels || level_begin >= level_end) return (int)hipErrorInvalidValue;
  const int hints = layout & (NESVOR_LAYOUT_UNCLUSTERED | NESVOR_LAYOUT_DY_SCRATCH);
  layout &= NESVOR_LAYOUT_MASK;
  DISPATCH_F_LAYOUT(launch_bwd_owner, grid, u, table, dpe, grad_table, grad_u, N, workspace, stages, level_begin, level_end,
                    queue_scale, nullptr, nullptr, (hipStream_t)stream, hints);
}

extern "C" int nesvor_hashgrid_backward_bounded(const nesvor_grid_t* grid, const float* u, const float* table,
                                                const float* dpe, float* grad_table, float* grad_u, int64_t N, int layout,
                                                void* workspace, int stages, int level_begin, int level_end,
                                                const float* queue_scale, const float* dy_bound, void* stream) {
  if (N <= 0) return 0;
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  if (workspace == nullptr || (stages & 3) == 0) return (int)hipErrorInvalidValue;
  if (level_begin < 0 || level_end > grid->n_levels || level_begin >= level_end) return (int)hipErrorInvalidValue;
  const int hints = layout & (NESVOR_LAYOUT_UNCLUSTERED | NESVOR_LAYOUT_DY_SCRATCH);
  layout &= NESVOR_LAYOUT_MASK;
  DISPATCH_F_LAYOUT(launch_bwd_owner, grid, u, table, dpe, grad_table, grad_u, N, workspace, stages, level_begin, level_end,
                    queue_scale, dy_bound, nullptr, (hipStream_t)stream, hints);
}

extern "C" int nesvor_hashgrid_backward_adamw(const nesvor_grid_t* grid, const float* u, float* table, const float* dpe,
                                              float* grad_table, float* grad_u, int64_t N, int layout, void* workspace,
                                              int stages, const float* queue_scale, const float* dy_bound, float* exp_avg,
                                              float* exp_avg_sq, const nesvor_adamw_t* adam, void* stream) {
  return nesvor_hashgrid_backward_adamw_levels(grid, u, table, dpe, grad_table, grad_u, N, layout, workspace, stages, 0, grid->n_levels,
                                               queue_scale, dy_bound, exp_avg, exp_avg_sq, adam, stream);
}

extern "C" int nesvor_hashgrid_backward_adamw_levels(const nesvor_grid_t* grid, const float* u, float* table, const float* dpe,
                                                     float* grad_table, float* grad_u, int64_t N, int layout, void* workspace,
                                                     int stages, int level_begin, int level_end, const float* queue_scale,
                                                     const float* dy_bound, float* exp_avg, float* exp_avg_sq, const nesvor_adamw_t* adam,
                                                     void* stream) {
  if (grid->n_levels <= 0 || grid->n_levels > NESVOR_MAX_LEVELS) return (int)hipErrorInvalidValue;
  if (N <= 0 || workspace == nullptr || (stages & ~3) != 0 || (stages & 3) == 0) return (int)hipErrorInvalidValue;
  if (level_begin < 0 || level_end > grid->n_levels || level_begin >= level_end) return (int)hipErrorInvalidValue;
  // (a level range is for the OWNER stage alone - stages == 2: the table's update level range by level range, so that the next
  //  forward can start on the levels that are done; the aggregation stage always covers every level)
  if ((level_begin != 0 || level_end != grid->n_levels) && stages != 2) return (int)hipErrorInvalidValue;
  if (table == nullptr || grad_table == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || adam == nullptr) return (int)hipErrorInvalidValue;
  OwnerAdam oa;
  oa.param = table; oa.exp_avg = exp_avg; oa.exp_avg_sq = exp_avg_sq; oa.done = nullptr;
  oa.a = make_adam_args(adam->lr, adam->beta1, adam->beta2, adam->eps, adam->weight_decay, adam->bias_correction1,
                        adam->bias_correction2, adam->grad_scale);
  const int hints = layout & (NESVOR_LAYOUT_UNCLUSTERED | NESVOR_LAYOUT_DY_SCRATCH);
  layout &= NESVOR_LAYOUT_MASK;
  DISPATCH_F_LAYOUT(launch_bwd_owner, grid, u, table, dpe, grad_table, grad_u, N, workspace, stages, level_begin, level_end, queue_scale,
                    dy_bound, &oa, (hipStream_t)stream, hints);
}
