// Probe of ds_read_b64_tr_b16 (gfx950): every lane supplies the address of its own 8-byte chunk (lane l -> bytes 8l..8l+7, holding
// the 16-bit values 4l..4l+3); prints which values each lane of the first two 16-lane groups receives.
//   hipcc --offload-arch=gfx950 -O2 tools/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[256];
  for (int e = threadIdx.x; e < 256; e += 64) lds[e] = (short)e;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + threadIdx.x * 4));
  *(s16x4*)(out + threadIdx.x * 4) = v;
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 32; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
