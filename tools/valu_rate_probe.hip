// Issue rate of single VALU / LDS instructions on gfx950, relative to v_fma_f32: every SIMD of the device runs W waves that each
// execute REP x 64 copies of one instruction (independent destinations, so only issue rate counts); time by events.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
  double d = a, e = 1.0001, f = 0.5;
  unsigned u = threadIdx.x, v = 12345u;
  unsigned long long q = threadIdx.x;
  __shared__ unsigned long long lds[2048];
  lds[threadIdx.x] = 0; lds[threadIdx.x + 256] = 0;
  __syncthreads();
  unsigned addr = (threadIdx.x * 8u) & 8191u;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
    if constexpr (OP == 1) { REP64(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"(e), "v"(f));) }
    if constexpr (OP == 2) { REP64(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a));) }
    if constexpr (OP == 3) { REP64(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u) : "v"(v), "v"(u));) }
    if constexpr (OP == 4) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(e), "v"(f));) }
    if constexpr (OP == 5) { REP64(asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u) : "v"(a));) }
    if constexpr (OP == 6) { REP64(asm volatile("v_rndne_f32 %0, %1" : "=v"(a) : "v"(b));) }
    if constexpr (OP == 7) { REP64(asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));) }
    if constexpr (OP == 8) { REP64(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u) : "v"(v), "v"(u));) }
    if constexpr (OP == 9) { REP64(asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(q) : "v"(q));) }
    if constexpr (OP == 10) { REP64(asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(u) : "v"(v), "v"(u));) }
    if constexpr (OP == 11) { REP64(asm volatile("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(e), "v"(f));) }
    if constexpr (OP == 12) { REP64(asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(u) : "v"(v), "v"(u));) }
    if constexpr (OP == 13) { REP64(asm volatile("ds_add_u64 %0, %1" : : "v"(addr), "v"(q) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (OP == 14) { REP64(asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(u) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if constexpr (OP == 15) { REP64(asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a) : "v"(u));) }
    if constexpr (OP == 16) { REP64(asm volatile("v_and_b32 %0, %1, %2" : "=v"(u) : "v"(v), "v"(u));) }
    if constexpr (OP == 17) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(v), "v"(u) : "vcc");) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + (float)d + (float)u + (float)q + (float)lds[threadIdx.x];
}

template <int OP>
float run(const char* name, int waves_per_simd, float base) {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<blocks, 256>>>(out, 10);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    probe<OP><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double per = best * 1e6 / ((double)iters * 64 * waves_per_simd);  // ns per wave-instruction per SIMD
  printf("%-22s waves/SIMD %d: %7.3f ns per wave-instruction and SIMD%s", name, waves_per_simd, per, base > 0 ? "" : "\n");
  if (base > 0) printf("  = %.2f x v_fma_f32\n", per / base);
  hipFree(out);
  return (float)per;
}

int main() {
  for (int w : {1, 4, 8}) {
    const float base = run<0>("v_fma_f32", w, 0.f);
    run<1>("v_fma_f64", w, base);
    run<11>("v_add_f64", w, base);
    run<2>("v_cvt_f64_f32", w, base);
    run<3>("v_mul_lo_u32", w, base);
    run<12>("v_mul_hi_u32", w, base);
    run<10>("v_mul_u32_u24", w, base);
    run<17>("v_mad_u64_u32", w, base);
    run<4>("v_pk_fma_f32", w, base);
    run<5>("v_cvt_i32_f32", w, base);
    run<15>("v_cvt_f32_i32", w, base);
    run<6>("v_rndne_f32", w, base);
    run<7>("v_fmac_f32_dpp row_shr", w, base);
    run<16>("v_and_b32", w, base);
    run<9>("v_lshlrev_b64", w, base);
    run<13>("ds_add_u64 (no confl.)", w, base);
    run<14>("ds_add_u32 (no confl.)", w, base);
  }
  return 0;
}
