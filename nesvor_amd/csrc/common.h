// Shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NESVOR_WAVE 64

// Full-wave (64-lane) sum via butterfly shuffles; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ int wave_sum_u32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Full-wave sum on the VALU only (DPP butterflies inside each 16-lane row, then 4 readlanes):
// no LDS crossbar traffic, unlike __shfl_xor which lowers to ds_bpermute.  Result is wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

// Sum over the 16 lanes of a DPP row (VALU only); every lane of the row gets the row total.
__device__ __forceinline__ float row_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// Full-wave min / max of unsigned values on the VALU only (same DPP pattern as wave_sum_dpp).  Result is wave-uniform.
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
  const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(r0, r1), min(r2, r3));
}
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
  const uint32_t r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}

// Full-wave max of floats on the VALU only (same DPP pattern); wave-uniform result.
__device__ __forceinline__ float wave_max_f32_dpp(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)));
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Publish the maximum of the non-negative `m` into a SLOTTED bound: NESVOR_ABSMAX_SLOTS device floats, NESVOR_ABSMAX_STRIDE floats
// (one 256-byte line) apart, whose maximum is the bound (non-negative floats order like their bit patterns: integer atomic max).
// Thousands of waves finish together, and atomics on one cache line serialise at the memory side at ~5.5 ns each (measured,
// round 5: 8 K publishes into one scalar stretched the 18 us loss kernel to 63 us; into 64 adjacent floats - two lines - still
// to 29 us, and the hash-grid forward from 70 to 99 us).  Spread over 16 lines by workgroup, and reduced over the workgroup
// first where every thread reaches the call (publish_absmax_wg), they cost nothing measurable.  Consumers take the maximum
// over the slots (absmax_slots); the caller zero-fills them.
#ifndef NESVOR_ABSMAX_SLOTS
#define NESVOR_ABSMAX_SLOTS 16
#define NESVOR_ABSMAX_STRIDE 64
#endif
__device__ __forceinline__ void publish_absmax_f32(float* slots, float m) {  // one atomic per wave (waves may have left the kernel)
  m = wave_max_f32_dpp(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f)
    atomicMax(reinterpret_cast<unsigned int*>(slots) + (blockIdx.x & (NESVOR_ABSMAX_SLOTS - 1)) * NESVOR_ABSMAX_STRIDE, __float_as_uint(m));
}
__device__ __forceinline__ void publish_absmax_wg(float* slots, float m) {  // one atomic per workgroup: EVERY thread must call (barrier)
  __shared__ float wg_max[16];
  m = wave_max_f32_dpp(m);
  if ((threadIdx.x & 63) == 0) wg_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = wg_max[0];
    for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmaxf(t, wg_max[w]);
    if (t > 0.f) atomicMax(reinterpret_cast<unsigned int*>(slots) + (blockIdx.x & (NESVOR_ABSMAX_SLOTS - 1)) * NESVOR_ABSMAX_STRIDE, __float_as_uint(t));
  }
}
__device__ __forceinline__ float absmax_slots(const float* __restrict__ slots) {  // (uniform address: scalar loads, s_max_u32)
  const uint32_t* s = reinterpret_cast<const uint32_t*>(slots);
  uint32_t m = 0u;
#pragma unroll
  for (int i = 0; i < NESVOR_ABSMAX_SLOTS; ++i) m = max(m, s[i * NESVOR_ABSMAX_STRIDE]);
  return __uint_as_float(m);
}

// The value lane (l ^ J) holds, J a power of two below 64, without a trip through the LDS crossbar (ds_bpermute: ~100
// cycles of latency per exchange): quad permutes, row rotates / shifts with bank masks, and gfx950's lane-swap
// instructions across 16- and 32-lane halves.
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int lane) {
  static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "power of two below the wave size");
  const int iv = (int)v;
  if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
  else if constexpr (J == 4) {
    int t = __builtin_amdgcn_update_dpp(iv, iv, 0x104, 0xf, 0x5, false);  // row_shl:4 into banks 0 and 2 (lanes 0-3, 8-11 of a row)
    t = __builtin_amdgcn_update_dpp(t, iv, 0x114, 0xf, 0xA, false);       // row_shr:4 into banks 1 and 3
    return (uint32_t)t;
  } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(iv, iv, 0x128, 0xf, 0xf, false);  // row_ror:8
  else if constexpr (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
    return (lane & 16) ? r[0] : r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // r[0] = halves [lo,lo], r[1] = [hi,hi]
    return (lane & 32) ? r[0] : r[1];
  }
}

// Global loads / stores in the scalar-base form: address = SGPR pair (wave-uniform base) + zero-extended 32-bit VGPR byte
// offset.  The compiler only selects this form when it can prove the offset fits 32 bits; written out, a gather or a
// feature-major row store needs no 64-bit vector address arithmetic (v_lshl_add_u64 / v_add_co pairs per access).  The loads
// are invisible to the compiler's wait-count insertion: every consumer must sit behind gwait_loads().
typedef float nesvor_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gload_b64_sbase(nesvor_f32x2& dst, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gstore_b32_sbase(const void* sbase, uint32_t voff, float v) {
  asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gwait_loads() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void gpin(nesvor_f32x2& x) { asm volatile("" : "+v"(x)); }

// fp32 add into LDS through an integer compare-and-swap loop.  On gfx950 ds_add_f32 retires ~1 lane
// per 3 cycles (192+ cycles per wave-instruction, measured, independent of conflicts) while
// ds_cmpst_rtn_b32 runs at ~7 cycles per conflict-free wave-instruction, so the CAS loop wins
// whenever few lanes of an instruction collide.
__device__ __forceinline__ void lds_add_f32_cas(float* addr, float v) {
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  unsigned int old = *reinterpret_cast<volatile unsigned int*>(a), assumed;
  do {
    assumed = old;
    old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (old != assumed);
}
