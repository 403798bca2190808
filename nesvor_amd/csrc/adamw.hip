// Fused AdamW + grad zero-fill over one flat fp32 buffer (gfx950).
//
// Replaces torch.optim.AdamW.step() + optimizer.zero_grad() of the reference's
// training loop (nesvor/nesvor/train.py:144-152,195-197).  All parameters of
// the model (hash table, MLPs, per-slice parameters) live in ONE flat buffer
// with matching flat grad / moment buffers, so one launch covers everything.
// Pure HBM streaming: 16 B read + 16 B written per parameter (p, g, m, v in;
// p, m, v, g=0 out), float4 per lane, grid-stride.
#include <hip/hip_runtime.h>
#include "common.h"

#include "adam.h"

namespace {

template <bool ZERO>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const AdamArgs a, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // small launches on the step's critical tail, next to the owner pass (see step_epilogue_kernel)
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    adam1(P.x, G.x, M.x, V.x, a); adam1(P.y, G.y, M.y, V.y, a);
    adam1(P.z, G.z, M.z, V.z, a); adam1(P.w, G.w, M.w, V.w, a);
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
    if (ZERO) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // tail (n not a multiple of 4)
  const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    adam1(p[t], g[t], m[t], v[t], a);
    if (ZERO) g[t] = 0.f;
  }
}

}  // namespace

extern "C" int nesvor_adamw_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                                 float bias_correction2, float grad_scale, int zero_grad, void* stream) {
  if (n <= 0) return 0;
  const AdamArgs a = make_adam_args(lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, grad_scale);
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (zero_grad)
    hipLaunchKernelGGL((adamw_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, a, blocks <= 256 ? 1 : 0);
  else
    hipLaunchKernelGGL((adamw_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, a, blocks <= 256 ? 1 : 0);
  return (int)hipGetLastError();
}

// out[c] = sum_r in[r * ld + c], c < cols (ld = row pitch of `in` in floats, so a column range of a wider matrix can be summed):
// the per-workgroup partial parameter gradients of the MLP backward -> gradient segment.
// 64 columns x 16 row groups per workgroup (1024 threads: the matrix has only ~100 column blocks, so the rows carry the
// parallelism), rows read as 256-byte segments, LDS reduction over the row groups.
namespace {
constexpr int kSumRowGroups = 16;
__global__ __launch_bounds__(64 * kSumRowGroups) void sum_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int ld) {
  __shared__ float red[kSumRowGroups][64];
  const int lane = threadIdx.x & 63, col = blockIdx.x * 64 + lane, rg = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f;
  if (col < cols) {
    int r = rg;
    for (; r + kSumRowGroups < rows; r += 2 * kSumRowGroups) {
      a0 += in[(size_t)r * ld + col]; a1 += in[(size_t)(r + kSumRowGroups) * ld + col];
    }
    for (; r < rows; r += kSumRowGroups) a0 += in[(size_t)r * ld + col];
  }
  red[rg][lane] = a0 + a1;
  __syncthreads();
  if (rg == 0 && col < cols) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kSumRowGroups; ++g) s += red[g][lane];
    out[col] = s;
  }
}
}  // namespace

extern "C" int nesvor_sum_rows(const float* in, float* out, int rows, int cols, int ld, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if (ld < cols) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64 * kSumRowGroups), 0, (hipStream_t)stream, in, out, rows, cols, ld);
  return (int)hipGetLastError();
}

// Up to four of those sums in ONE launch (the partial gradients of the step's networks, summed after the last backward).
namespace {
struct SumJobs {
  const float* in[4];
  float* out[4];
  int cols[4], ld[4], first_block[5];
  int rows;
};
__global__ __launch_bounds__(64 * kSumRowGroups) void sum_rows_multi_kernel(const SumJobs jobs) {
  __shared__ float red[kSumRowGroups][64];
  int job = 0;
  while (job < 3 && (int)blockIdx.x >= jobs.first_block[job + 1]) ++job;
  const float* __restrict__ in = jobs.in[job];
  const int cols = jobs.cols[job], ld = jobs.ld[job], rows = jobs.rows;
  const int lane = threadIdx.x & 63, col = ((int)blockIdx.x - jobs.first_block[job]) * 64 + lane, rg = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f;
  if (col < cols) {
    int r = rg;
    for (; r + kSumRowGroups < rows; r += 2 * kSumRowGroups) {
      a0 += in[(size_t)r * ld + col]; a1 += in[(size_t)(r + kSumRowGroups) * ld + col];
    }
    for (; r < rows; r += kSumRowGroups) a0 += in[(size_t)r * ld + col];
  }
  red[rg][lane] = a0 + a1;
  __syncthreads();
  if (rg == 0 && col < cols) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kSumRowGroups; ++g) s += red[g][lane];
    jobs.out[job][col] = s;
  }
}
}  // namespace

extern "C" int nesvor_sum_rows_multi(const float* const* in, float* const* out, const int* cols, const int* ld, int n_jobs, int rows,
                                     void* stream) {
  if (n_jobs <= 0 || rows <= 0) return 0;
  if (n_jobs > 4) return (int)hipErrorInvalidValue;
  SumJobs j{};
  int blocks = 0;
  for (int k = 0; k < 4; ++k) {
    j.first_block[k] = blocks;
    if (k < n_jobs) {
      if (cols[k] <= 0 || ld[k] < cols[k]) return (int)hipErrorInvalidValue;
      j.in[k] = in[k]; j.out[k] = out[k]; j.cols[k] = cols[k]; j.ld[k] = ld[k];
      blocks += (cols[k] + 63) / 64;
    }
  }
  j.first_block[4] = blocks;
  for (int k = n_jobs; k < 4; ++k) j.first_block[k] = blocks;  // (unused jobs own no block)
  j.rows = rows;
  hipLaunchKernelGGL(sum_rows_multi_kernel, dim3((unsigned)blocks), dim3(64 * kSumRowGroups), 0, (hipStream_t)stream, j);
  return (int)hipGetLastError();
}

extern "C" int nesvor_hip_abi_version(void) { return 35; }
