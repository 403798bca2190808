// How much VALU work hides behind v_mfma_f32_16x16x32_bf16 on one SIMD, and at what engine clock do such loops run?
// Each wave runs R rounds of NM MFMAs (4 independent accumulators) and NS split3() evaluations (the 22-instruction
// fp32 -> 3 x bf16 operand split of csrc/mlp.hip: cvt_pk / shift / and / sub), either as two blocks or interleaved
// (one split per NM / NS MFMAs), with 1, 2 or 3 waves per SIMD; a role-specialised run puts MFMA-only and VALU-only
// waves on the same SIMD.  Cycles are counted with s_memtime (constant 100 MHz) and wall time; the engine clock is
// derived from a dependent-VALU chain of known length.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_overlap.hip -o /tmp/ovl16 && /tmp/ovl16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 pack_bf16(const f32x4& v) { return __builtin_bit_cast(s16x4, __builtin_convertvector(v, bf16x4)); }
__device__ __forceinline__ f32x4 widen_bf16(const s16x4& v) {
  const uint2 u = __builtin_bit_cast(uint2, v);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
}
struct Split3 { s16x4 hi, mid, lo; };
__device__ __forceinline__ Split3 split3(const f32x4& v) {
  Split3 s;
  s.hi = pack_bf16(v);
  const f32x4 r1 = v - widen_bf16(s.hi);
  s.mid = pack_bf16(r1);
  s.lo = pack_bf16(r1 - widen_bf16(s.mid));
  return s;
}
__device__ __forceinline__ bf16x8 join8(const s16x4& a, const s16x4& b) {
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// MODE 0: blocks [NM MFMAs][NS splits]; 1: interleaved; 2: role-specialised: 8-wave workgroups, waves 0-3 MFMA only, waves 4-7
// splits only (wave w and w + 4 share a SIMD)
template <int NM, int NS, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 512 : 256) void k(float* out, int rounds) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 v[4];
  for (int i = 0; i < 4; ++i) v[i] = f32x4{threadIdx.x * 0.001f + i, 1.f + i, 2.f - i, 0.5f * i};
  bf16x8 a = join8(pack_bf16(v[0]), pack_bf16(v[1])), b = join8(pack_bf16(v[2]), pack_bf16(v[3]));
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE != 2 || wave < 4, do_s = MODE != 2 || wave >= 4;
  for (int r = 0; r < rounds; ++r) {
    if (MODE == 1) {
      constexpr int PER = NS > 0 ? NM / NS : NM;
#pragma unroll
      for (int s = 0; s < (NS > 0 ? NS : 1); ++s) {
#pragma unroll
        for (int m = 0; m < PER; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
        if (NS > 0) {
          const Split3 sp = split3(v[s & 3]);
          v[s & 3] = widen_bf16(sp.hi) + widen_bf16(sp.mid) * 1.0001f + widen_bf16(sp.lo);
        }
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
      }
      if (do_s) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const Split3 sp = split3(v[s & 3]);
          v[s & 3] = widen_bf16(sp.hi) + widen_bf16(sp.mid) * 1.0001f + widen_bf16(sp.lo);
        }
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 4; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  out[blockIdx.x * 256 + (threadIdx.x & 255)] = s;
}

// dependent chain of N fp32 fmas per round: 4 cycles each per wave on an otherwise idle SIMD -> engine clock
__global__ __launch_bounds__(64) void clock_probe(float* out, int rounds) {
  float x = threadIdx.x * 1e-3f;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 256; ++i) x = fmaf(x, 1.0001f, 0.5f);
  }
  out[threadIdx.x] = x;
}

template <int NM, int NS, int MODE>
void run(int wgs_per_cu, const char* label) {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  const int grid = 256 * wgs_per_cu, rounds = 2000;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<NM, NS, MODE>), dim3(grid), dim3(MODE == 2 ? 512 : 256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((k<NM, NS, MODE>), dim3(grid), dim3(MODE == 2 ? 512 : 256), 0, 0, out, rounds);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("%-44s NM=%3d NS=%3d %d wave/SIMD: %8.1f ns/round\n", label, NM, NS, wgs_per_cu, ms * 1e6 / rounds);
  hipFree(out);
}
int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, out, 10); hipDeviceSynchronize();
  hipEventRecord(s); hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, out, 20000); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("idle-chip clock from a dependent fma chain (assuming 4 cycles per dependent wave64 fma): %.0f MHz\n", 20000.0 * 256 * 4 / (ms * 1e-3) / 1e6);
  // full chip, same chain on every SIMD (1 wave each)
  hipLaunchKernelGGL(clock_probe, dim3(1024), dim3(64), 0, 0, out, 10); hipDeviceSynchronize();
  hipEventRecord(s); hipLaunchKernelGGL(clock_probe, dim3(1024), dim3(64), 0, 0, out, 20000); hipEventRecord(e); hipEventSynchronize(e);
  hipEventElapsedTime(&ms, s, e);
  printf("full-chip clock from the same chain on 1024 waves: %.0f MHz\n", 20000.0 * 256 * 4 / (ms * 1e-3) / 1e6);
  run<48, 0, 0>(1, "MFMA only");
  run<0, 12, 0>(1, "split only (12 x 22 VALU + 12 recombine)");
  run<48, 12, 0>(1, "blocks, one wave");
  run<48, 12, 1>(1, "interleaved 4 MFMA : 1 split, one wave");
  run<48, 12, 0>(2, "blocks, two waves");
  run<48, 12, 1>(2, "interleaved, two waves");
  run<48, 12, 0>(3, "blocks, three waves");
  run<48, 12, 0>(4, "blocks, four waves");
  run<48, 12, 2>(1, "role-specialised (1 MFMA + 1 VALU wave / SIMD)");
  run<48, 12, 2>(2, "role-specialised (2 + 2 waves / SIMD)");
  run<48, 24, 0>(2, "blocks, two waves, 2x VALU");
  run<48, 24, 1>(2, "interleaved, two waves, 2x VALU");
  run<48, 24, 2>(2, "role-specialised (2 + 2), 2x VALU");
  return 0;
}
