// fp32 -> three bf16 planes (hi, mid, lo; x = hi + mid + lo up to 2^-24 |x|): instruction count and exactness of
// alternative formulations of split3() (csrc/mlp.hip).
//   A  cvt_pk_bf16 / shift+and widen / v_sub                                  (22 VALU per 4 values)
//   B  cvt_pk_bf16 / residual by v_dot2_f32_bf16 (x - hi in ONE instruction:  (14 VALU per 4 values)
//      dot2((hi_lo, hi_hi), (-1, 0)) + x picks and subtracts one half of the packed pair)
//   C  as B with v_dot2c_f32_bf16 (VOP2 accumulate form)
// Every variant's planes are compared bit for bit with A's on random, tiny, huge and special inputs.
// hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/splitp && /tmp/splitp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct Split3 { s16x4 hi, mid, lo; };

__device__ __forceinline__ s16x4 pack_bf16(const f32x4& v) { return __builtin_bit_cast(s16x4, __builtin_convertvector(v, bf16x4)); }
__device__ __forceinline__ f32x4 widen_bf16(const s16x4& v) {
  const uint2 u = __builtin_bit_cast(uint2, v);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
}
__device__ __forceinline__ Split3 split_a(const f32x4& v) {
  Split3 s;
  s.hi = pack_bf16(v);
  const f32x4 r1 = v - widen_bf16(s.hi);
  s.mid = pack_bf16(r1);
  s.lo = pack_bf16(r1 - widen_bf16(s.mid));
  return s;
}
// x - (one half of a packed bf16 pair): v_dot2_f32_bf16 D = S0.lo S1.lo + S0.hi S1.hi + S2 through the compiler's builtin
// (inline asm hides the instruction from the hazard recogniser: a DOT result read by another VALU instruction needs
// wait states on gfx940+, and the first version of this probe, written with asm, read stale registers)
// The selector constants are hidden from the optimiser (an SGPR it cannot see through): folded, the low-half selector
// 0x0000BF80 becomes the inline constant "-1.0", which the hardware reads as the 32-bit pattern 0xBF800000 = the HIGH half
// (measured: the first builtin version subtracted the wrong element).
__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ float sub_lo(float x, uint32_t pair) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pair), __builtin_bit_cast(bf16x2, opaque(0x0000BF80u)), x, false);
}
__device__ __forceinline__ float sub_hi(float x, uint32_t pair) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pair), __builtin_bit_cast(bf16x2, opaque(0xBF800000u)), x, false);
}
__device__ __forceinline__ Split3 split_b(const f32x4& v) {
  Split3 s;
  s.hi = pack_bf16(v);
  uint2 h = __builtin_bit_cast(uint2, s.hi);
  const f32x4 r1 = f32x4{sub_lo(v[0], h.x), sub_hi(v[1], h.x), sub_lo(v[2], h.y), sub_hi(v[3], h.y)};
  s.mid = pack_bf16(r1);
  uint2 m = __builtin_bit_cast(uint2, s.mid);
  const f32x4 r2 = f32x4{sub_lo(r1[0], m.x), sub_hi(r1[1], m.x), sub_lo(r1[2], m.y), sub_hi(r1[3], m.y)};
  s.lo = pack_bf16(r2);
  return s;
}
__device__ __forceinline__ float subc_lo(float x, uint32_t pair) {
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(pair), "v"(0x0000BF80u));
  return x;
}
__device__ __forceinline__ float subc_hi(float x, uint32_t pair) {
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(pair), "v"(0xBF800000u));
  return x;
}
__device__ __forceinline__ Split3 split_c(const f32x4& v) {
  Split3 s;
  s.hi = pack_bf16(v);
  uint2 h = __builtin_bit_cast(uint2, s.hi);
  const f32x4 r1 = f32x4{subc_lo(v[0], h.x), subc_hi(v[1], h.x), subc_lo(v[2], h.y), subc_hi(v[3], h.y)};
  s.mid = pack_bf16(r1);
  uint2 m = __builtin_bit_cast(uint2, s.mid);
  const f32x4 r2 = f32x4{subc_lo(r1[0], m.x), subc_hi(r1[1], m.x), subc_lo(r1[2], m.y), subc_hi(r1[3], m.y)};
  s.lo = pack_bf16(r2);
  return s;
}

// D: as B with the selector constants in VGPRs; E: as B with the operands swapped (S0 = selector, S1 = the pair)
__device__ __forceinline__ float sub_sel(float x, uint32_t pair, uint32_t sel) {
  float d;
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(d) : "v"(pair), "v"(sel), "v"(x));
  return d;
}
__device__ __forceinline__ float sub_sel_swapped(float x, uint32_t pair, uint32_t sel) {
  float d;
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(d) : "v"(sel), "v"(pair), "v"(x));
  return d;
}
template <bool SWAP>
__device__ __forceinline__ Split3 split_de(const f32x4& v) {
  const uint32_t klo = 0x0000BF80u, khi = 0xBF800000u;
  auto sub = [&](float x, uint32_t pair, uint32_t sel) { return SWAP ? sub_sel_swapped(x, pair, sel) : sub_sel(x, pair, sel); };
  Split3 s;
  s.hi = pack_bf16(v);
  uint2 h = __builtin_bit_cast(uint2, s.hi);
  const f32x4 r1 = f32x4{sub(v[0], h.x, klo), sub(v[1], h.x, khi), sub(v[2], h.y, klo), sub(v[3], h.y, khi)};
  s.mid = pack_bf16(r1);
  uint2 m = __builtin_bit_cast(uint2, s.mid);
  const f32x4 r2 = f32x4{sub(r1[0], m.x, klo), sub(r1[1], m.x, khi), sub(r1[2], m.y, klo), sub(r1[3], m.y, khi)};
  s.lo = pack_bf16(r2);
  return s;
}
// F: x - widen(hi) with the widening done by v_perm_b32 / v_lshlrev and v_pk_add_f32 with a negated operand (two values
// per subtraction)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ Split3 split_f(const f32x4& v) {
  Split3 s;
  s.hi = pack_bf16(v);
  const f32x4 w = widen_bf16(s.hi);
  const f32x2 r1a = pk_sub(f32x2{v[0], v[1]}, f32x2{w[0], w[1]}), r1b = pk_sub(f32x2{v[2], v[3]}, f32x2{w[2], w[3]});
  const f32x4 r1 = f32x4{r1a[0], r1a[1], r1b[0], r1b[1]};
  s.mid = pack_bf16(r1);
  const f32x4 w2 = widen_bf16(s.mid);
  const f32x2 r2a = pk_sub(r1a, f32x2{w2[0], w2[1]}), r2b = pk_sub(r1b, f32x2{w2[2], w2[3]});
  s.lo = pack_bf16(f32x4{r2a[0], r2a[1], r2b[0], r2b[1]});
  return s;
}

template <int V> __device__ __forceinline__ Split3 split(const f32x4& v) {
  if constexpr (V == 0) return split_a(v);
  else if constexpr (V == 1) return split_b(v);
  else if constexpr (V == 2) return split_c(v);
  else if constexpr (V == 3) return split_de<false>(v);
  else if constexpr (V == 4) return split_de<true>(v);
  else return split_f(v);
}

template <int V>
__global__ void check(const float* x, uint32_t* out, int n) {  // out: 6 words per 4 inputs
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * i + 3 >= n) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
  const Split3 s = split<V>(v);
  const uint2 a = __builtin_bit_cast(uint2, s.hi), b = __builtin_bit_cast(uint2, s.mid), c = __builtin_bit_cast(uint2, s.lo);
  uint32_t* o = out + 6 * i;
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y;
}

template <int V>
__global__ __launch_bounds__(256) void timeit(float* out, int rounds) {
  f32x4 v[4];
  for (int i = 0; i < 4; ++i) v[i] = f32x4{threadIdx.x * 0.001f + i, 1.f + i, 2.f - i, 0.5f * i + 0.25f};
  uint32_t acc = 0;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const Split3 sp = split<V>(v[s & 3]);
      const uint2 a = __builtin_bit_cast(uint2, sp.hi), b = __builtin_bit_cast(uint2, sp.mid), c = __builtin_bit_cast(uint2, sp.lo);
      acc ^= a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y;              // 6 xor: keeps the planes live
      const uint32_t t = acc & 0x7u;                         // + 5 ops: all four inputs of this register's next split depend on this one
#pragma unroll
      for (int k = 0; k < 4; ++k) v[s & 3][k] = __uint_as_float(__float_as_uint(v[s & 3][k]) ^ t);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = __uint_as_float(acc);
}

template <int V> float time_variant() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 1024);  // 1024 workgroups of 4 waves on 256 CUs: 4 waves per SIMD
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(timeit<V>, dim3(1024), dim3(256), 0, 0, out, 10); hipDeviceSynchronize();
  hipEventRecord(s); hipLaunchKernelGGL(timeit<V>, dim3(1024), dim3(256), 0, 0, out, 4000); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  hipFree(out);
  return ms * 1e6f / (4000.f * 16.f * 4.f);  // ns per split (+ 11 bookkeeping ops) per SIMD at four waves per SIMD (throughput)
}

int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (int i = 0; i < n; ++i) {
    const uint32_t bits = (uint32_t)rnd();
    float f;
    if (i < n / 2) { memcpy(&f, &bits, 4); if (std::isnan(f) || std::isinf(f)) f = 1.f; }  // every exponent, incl. denormals
    else f = (float)((double)(int32_t)bits / 2147483648.0) * std::pow(10.f, (float)((int)(rnd() % 13) - 8));
    h[i] = f;
  }
  h[0] = 0.f; h[1] = -0.f; h[2] = 1.f; h[3] = -1.f; h[4] = 1.00390625f; h[5] = 3.3895314e38f; h[6] = 1.17549435e-38f; h[7] = 1e-45f;
  float* dx; uint32_t *da, *db; hipMalloc(&dx, 4 * n); hipMalloc(&da, 6 * n); hipMalloc(&db, 6 * n);
  hipMemcpy(dx, h.data(), 4 * n, hipMemcpyHostToDevice);
  std::vector<uint32_t> ra(6 * (n / 4)), rb(6 * (n / 4));
  hipLaunchKernelGGL(check<0>, dim3(n / 4 / 256), dim3(256), 0, 0, dx, da, n);
  hipMemcpy(ra.data(), da, 4 * ra.size(), hipMemcpyDeviceToHost);
  const char* names[6] = {"A widen + sub", "B v_dot2_f32_bf16 (sgpr selector)", "C v_dot2c_f32_bf16", "D v_dot2_f32_bf16 (vgpr selector)",
                          "E v_dot2_f32_bf16 (operands swapped)", "F v_pk_add_f32 residuals"};
  auto bf = [](uint32_t w, int half) { uint32_t b = (half ? (w & 0xFFFF0000u) : (w << 16)); float f; memcpy(&f, &b, 4); return f; };
  for (int var = 1; var < 6; ++var) {
    if (var == 1) hipLaunchKernelGGL(check<1>, dim3(n / 4 / 256), dim3(256), 0, 0, dx, db, n);
    else if (var == 2) hipLaunchKernelGGL(check<2>, dim3(n / 4 / 256), dim3(256), 0, 0, dx, db, n);
    else if (var == 3) hipLaunchKernelGGL(check<3>, dim3(n / 4 / 256), dim3(256), 0, 0, dx, db, n);
    else if (var == 4) hipLaunchKernelGGL(check<4>, dim3(n / 4 / 256), dim3(256), 0, 0, dx, db, n);
    else hipLaunchKernelGGL(check<5>, dim3(n / 4 / 256), dim3(256), 0, 0, dx, db, n);
    hipMemcpy(rb.data(), db, 4 * rb.size(), hipMemcpyDeviceToHost);
    long bad = 0, bad_normal = 0, shown = 0;
    double worst = 0;  // largest |x - (hi + mid + lo)| / |x| over normal inputs
    for (size_t quad = 0; quad < (size_t)n / 4; ++quad) {
      bool differs = false;
      for (int w = 0; w < 6; ++w) differs = differs || ra[6 * quad + w] != rb[6 * quad + w];
      bool normal = true;
      for (int k = 0; k < 4; ++k) normal = normal && std::fabs(h[4 * quad + k]) > 1e-30f && std::fabs(h[4 * quad + k]) < 1e30f;
      if (normal) {
        for (int k = 0; k < 4; ++k) {
          const double x = h[4 * quad + k];
          const double sum = (double)bf(rb[6 * quad + k / 2], k & 1) + (double)bf(rb[6 * quad + 2 + k / 2], k & 1) + (double)bf(rb[6 * quad + 4 + k / 2], k & 1);
          worst = std::fmax(worst, std::fabs(x - sum) / std::fabs(x));
        }
      }
      if (!differs) continue;
      ++bad;
      if (normal) {
        ++bad_normal;
        if (shown < 4) {
          ++shown;
          for (int k = 0; k < 4; ++k)
            printf("     x %.9g : A (%.9g, %.9g, %.9g)  this (%.9g, %.9g, %.9g)\n", h[4 * quad + k],
                   bf(ra[6 * quad + k / 2], k & 1), bf(ra[6 * quad + 2 + k / 2], k & 1), bf(ra[6 * quad + 4 + k / 2], k & 1),
                   bf(rb[6 * quad + k / 2], k & 1), bf(rb[6 * quad + 2 + k / 2], k & 1), bf(rb[6 * quad + 4 + k / 2], k & 1));
        }
      }
    }
    printf("%-40s quads differing from A: %ld of %d (%ld with all inputs in [1e-30, 1e30]); worst |x - (hi+mid+lo)| / |x| = %.3g\n", names[var], bad,
           n / 4, bad_normal, worst);
  }
  printf("throughput, ns per split (+ 11 bookkeeping VALU ops) per SIMD, four waves per SIMD:\n");
  printf("  A %.2f   B %.2f   C %.2f   D %.2f   E %.2f   F %.2f\n", time_variant<0>(), time_variant<1>(), time_variant<2>(), time_variant<3>(),
         time_variant<4>(), time_variant<5>());
  return 0;
}
