// Imaging model + losses of NeSVoR.forward, value AND gradient in one launch (gfx950).
//
// Replaces the tail of NeSVoR.forward (nesvor/nesvor/models.py:286-325) and edge_reg / tv_reg /
// l2_reg (models.py:366-384) together with their autograd backward — ~60 elementwise / reduction
// launches over (B,S) tensors per iteration in the reference.  Per pixel b with slice k = idx[b]:
//   density = softplus(z0);  bias = exp(log_bias) | 1;  varp = exp(log_var) | 1
//   v_out = c_k mean_s(bias density);   var = (c_k mean_s(bias varp))^2 [pixel variance] + exp(lvs_k) [slice variance]
//   MSE   = mean_b (v_out - v)^2 / (2 var);   logVar = mean_b 0.5 log var
//   imageReg(edge) = delta (mean_{b,s} sqrt(1 + dd^2 / (dx2 delta^2)) - 1),
//       dd = density_s - density_{S-1-s},  dx2 = |x_s - x_{S-1-s}|^2 + 1e-6          (TV / L2 analogous)
//   biasReg = (mean_{b,s} log_bias)^2   (the global mean is passed in: it needs a prior reduction)
// bias and c enter `var` detached, exactly as the reference writes it.
// Forward launch (gw == NULL): per pixel, the three loss partial sums (loss_pix).  Backward launch (gw = the four
// upstream gradients d total / d {MSE, logVar, imageReg, biasReg}, on the device): per pixel
// d(total)/d(c_k), d(total)/d(lvs_k) and per sample d(total)/d(z0, log_var, log_bias, x); with loss_pix != NULL
// too the launch produces values and gradients together (the autograd-free training step).
// One wave per pixel; two streaming passes over the pixel's S samples (pass 1: the two means and the
// regulariser sums, pass 2: gradients); partner samples S-1-s are re-read from L1/L2.
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/nesvor_hip.h"

namespace {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }

// REG: 0 edge, 1 TV, 2 L2
template <int REG>
__global__ __launch_bounds__(256) void imaging_loss_kernel(const nesvor_loss_t a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const int S = a.S;
  const int64_t k = a.slice_idx[b];
  const size_t base = (size_t)b * S;
  const float c = a.c != nullptr ? a.c[k] : 1.f;
  const bool has_lv = a.log_var != nullptr, has_lb = a.log_bias != nullptr;
  const float inv_d2 = 1.f / (a.delta * a.delta);

  float s1 = 0.f, s2 = 0.f, sreg = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float dens = softplus_f(a.z0[base + s]);
    const float bias = has_lb ? expf(a.log_bias[base + s]) : 1.f;
    s1 += bias * dens;
    if (has_lv) s2 += bias * expf(a.log_var[base + s]);
    const int m = S - 1 - s;
    const float dd = dens - softplus_f(a.z0[base + m]);
    const float* xs = a.x + (base + s) * 3;
    const float* xm = a.x + (base + m) * 3;
    const float ex = xs[0] - xm[0], ey = xs[1] - xm[1], ez = xs[2] - xm[2];
    const float dx2 = (ex * ex + ey * ey + ez * ez) + 1e-6f;
    if (REG == 0) sreg += sqrtf(1.f + dd * dd / dx2 * inv_d2);
    if (REG == 1) sreg += fabsf(dd / sqrtf(dx2));
    if (REG == 2) sreg += dd * dd / dx2;
  }
  s1 = wave_sum_dpp(s1); s2 = wave_sum_dpp(s2); sreg = wave_sum_dpp(sreg);
  const float m1 = s1 / S, m2 = s2 / S;
  const float v_out = c * m1;
  float var = 1.f, pv = 0.f;
  if (has_lv) { pv = c * m2; var = pv * pv; }
  const float slice_var = a.log_var_slice != nullptr ? expf(a.log_var_slice[k]) : 0.f;
  var += slice_var;  // the reference starts from var = 1 when there is no pixel variance (models.py:300-314)
  const bool has_var = has_lv || a.log_var_slice != nullptr;
  const float e = v_out - a.v[b];
  const float invB = 1.f / a.B;
  if (a.loss_pix != nullptr && lane == 0) {  // loss values (forward launch, or a combined value+gradient launch)
    a.loss_pix[3 * b + 0] = e * e / (2.f * var);
    a.loss_pix[3 * b + 1] = has_var ? 0.5f * logf(var) : 0.f;
    a.loss_pix[3 * b + 2] = sreg;  // sum over the pixel's samples of the regulariser term
  }
  if (a.gw == nullptr) return;  // forward launch
  // backward launch: gradients of  gw0 MSE + gw1 logVar + gw2 imageReg + gw3 biasReg
  const float gw_mse = a.gw[0], gw_lv = a.gw[1], gw_img = a.gw[2], gw_bias = a.gw[3];
  const float g_vout = gw_mse * e / var * invB;
  const float g_var = has_var ? (gw_mse * (-e * e / (2.f * var * var)) + gw_lv * 0.5f / var) * invB : 0.f;
  const float g_m2 = has_lv ? g_var * 2.f * pv * c : 0.f;  // c detached inside var
  if (lane == 0) {
    if (a.dc_pix != nullptr) a.dc_pix[b] = g_vout * m1;
    if (a.dlvs_pix != nullptr) a.dlvs_pix[b] = g_var * slice_var;
  }
  const float reg_scale = (REG == 0 ? a.delta : 1.f) * gw_img / ((float)a.B * S);
  const float g_lb_reg = has_lb ? gw_bias * 2.f * a.log_bias_mean[0] / ((float)a.B * S) : 0.f;
  float mx_dz = 0.f, mx_lv = 0.f, mx_lb = 0.f;  // published for the consumers of these gradients (nesvor_mlp_t.prep[2])
  for (int s = lane; s < S; s += 64) {
    const float z = a.z0[base + s];
    const float dens = softplus_f(z);
    const float bias = has_lb ? expf(a.log_bias[base + s]) : 1.f;
    float g_dens = g_vout * c * bias / S;
    // regulariser: the pair (s, S-1-s) appears twice in the mean (as s and as its mirror)
    const int m = S - 1 - s;
    const float dd = dens - softplus_f(a.z0[base + m]);
    const float* xs = a.x + (base + s) * 3;
    const float* xm = a.x + (base + m) * 3;
    const float ex = xs[0] - xm[0], ey = xs[1] - xm[1], ez = xs[2] - xm[2];
    const float dx2 = (ex * ex + ey * ey + ez * ez) + 1e-6f;
    float g_dd, g_dx2;  // d term / d dd, d term / d dx2
    if (REG == 0) {
      const float term = sqrtf(1.f + dd * dd / dx2 * inv_d2);
      g_dd = dd / dx2 * inv_d2 / term;
      g_dx2 = -0.5f * dd * dd / (dx2 * dx2) * inv_d2 / term;
    } else if (REG == 1) {
      const float r = sqrtf(dx2);
      g_dd = (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f)) / r;
      g_dx2 = -0.5f * fabsf(dd) / (dx2 * r);
    } else {
      g_dd = 2.f * dd / dx2;
      g_dx2 = -dd * dd / (dx2 * dx2);
    }
    g_dens += 2.f * reg_scale * g_dd;
    const float o_dz = g_dens * sigmoid_f(z);
    a.dz0[base + s] = o_dz;
    mx_dz = fmaxf(mx_dz, fabsf(o_dz));
    if (a.dx != nullptr) {
      const float gx = 2.f * reg_scale * g_dx2 * 2.f;
      float* o = a.dx + (base + s) * 3;
      o[0] = gx * ex; o[1] = gx * ey; o[2] = gx * ez;
    }
    if (has_lv) { const float o_lv = g_m2 * bias * expf(a.log_var[base + s]) / S; a.dlog_var[base + s] = o_lv; mx_lv = fmaxf(mx_lv, fabsf(o_lv)); }  // bias detached
    if (has_lb) { const float o_lb = g_vout * c * dens * bias / S + g_lb_reg; a.dlog_bias[base + s] = o_lb; mx_lb = fmaxf(mx_lb, fabsf(o_lb)); }
  }
  if (a.dz0_absmax != nullptr) publish_absmax_f32(a.dz0_absmax, mx_dz);
  if (has_lv && a.dlog_var_absmax != nullptr) publish_absmax_f32(a.dlog_var_absmax, mx_lv);
  if (has_lb && a.dlog_bias_absmax != nullptr) publish_absmax_f32(a.dlog_bias_absmax, mx_lb);
}


// The same computation for S = 64 K samples per pixel (K = 1, 2, 4, 8: the training default S = 256 is K = 4): one pass.
// A lane keeps its K samples (z0, density, bias, exp(log_var), x) in registers; the mirror sample S-1-s of sample
// s = lane + 64 i is sample 63-lane, K-1-i of the same wave, so its density and coordinates come through a wave
// permute instead of a second softplus and a second read; the gradient pass re-uses everything.  Same formulas in the
// same order as imaging_loss_kernel: the two kernels agree bit for bit.
template <int REG, int K>
__global__ __launch_bounds__(256) void imaging_loss_cached_kernel(const nesvor_loss_t a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  constexpr int S = 64 * K;
  const int64_t k = a.slice_idx[b];
  const size_t base = (size_t)b * S;
  const float c = a.c != nullptr ? a.c[k] : 1.f;
  const bool has_lv = a.log_var != nullptr, has_lb = a.log_bias != nullptr;
  const float inv_d2 = 1.f / (a.delta * a.delta);
  float z[K], dens[K], bias[K], elv[K], xs[K][3];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const size_t s = base + lane + 64 * i;
    z[i] = a.z0[s];
    bias[i] = has_lb ? a.log_bias[s] : 0.f;
    elv[i] = has_lv ? a.log_var[s] : 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) xs[i][d] = a.x[s * 3 + d];
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    dens[i] = softplus_f(z[i]);
    bias[i] = has_lb ? expf(bias[i]) : 1.f;
    elv[i] = has_lv ? expf(elv[i]) : 0.f;
  }
  float dd[K], ex[K][3], dx2[K];
  float s1 = 0.f, s2 = 0.f, sreg = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const float dm = __shfl(dens[K - 1 - i], 63 - lane, 64);
    dd[i] = dens[i] - dm;
#pragma unroll
    for (int d = 0; d < 3; ++d) ex[i][d] = xs[i][d] - __shfl(xs[K - 1 - i][d], 63 - lane, 64);
    dx2[i] = (ex[i][0] * ex[i][0] + ex[i][1] * ex[i][1] + ex[i][2] * ex[i][2]) + 1e-6f;
    s1 += bias[i] * dens[i];
    if (has_lv) s2 += bias[i] * elv[i];
    if (REG == 0) sreg += sqrtf(1.f + dd[i] * dd[i] / dx2[i] * inv_d2);
    if (REG == 1) sreg += fabsf(dd[i] / sqrtf(dx2[i]));
    if (REG == 2) sreg += dd[i] * dd[i] / dx2[i];
  }
  s1 = wave_sum_dpp(s1); s2 = wave_sum_dpp(s2); sreg = wave_sum_dpp(sreg);
  const float m1 = s1 / S, m2 = s2 / S;
  const float v_out = c * m1;
  float var = 1.f, pv = 0.f;
  if (has_lv) { pv = c * m2; var = pv * pv; }
  const float slice_var = a.log_var_slice != nullptr ? expf(a.log_var_slice[k]) : 0.f;
  var += slice_var;
  const bool has_var = has_lv || a.log_var_slice != nullptr;
  const float e = v_out - a.v[b];
  const float invB = 1.f / a.B;
  if (a.loss_pix != nullptr && lane == 0) {
    a.loss_pix[3 * b + 0] = e * e / (2.f * var);
    a.loss_pix[3 * b + 1] = has_var ? 0.5f * logf(var) : 0.f;
    a.loss_pix[3 * b + 2] = sreg;
  }
  if (a.gw == nullptr) return;
  const float gw_mse = a.gw[0], gw_lv = a.gw[1], gw_img = a.gw[2], gw_bias = a.gw[3];
  const float g_vout = gw_mse * e / var * invB;
  const float g_var = has_var ? (gw_mse * (-e * e / (2.f * var * var)) + gw_lv * 0.5f / var) * invB : 0.f;
  const float g_m2 = has_lv ? g_var * 2.f * pv * c : 0.f;
  if (lane == 0) {
    if (a.dc_pix != nullptr) a.dc_pix[b] = g_vout * m1;
    if (a.dlvs_pix != nullptr) a.dlvs_pix[b] = g_var * slice_var;
  }
  const float reg_scale = (REG == 0 ? a.delta : 1.f) * gw_img / ((float)a.B * S);
  const float g_lb_reg = has_lb ? gw_bias * 2.f * a.log_bias_mean[0] / ((float)a.B * S) : 0.f;
  float mx_dz = 0.f, mx_lv = 0.f, mx_lb = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const size_t s = base + lane + 64 * i;
    float g_dens = g_vout * c * bias[i] / S;
    float g_dd, g_dx2;
    if (REG == 0) {
      const float term = sqrtf(1.f + dd[i] * dd[i] / dx2[i] * inv_d2);
      g_dd = dd[i] / dx2[i] * inv_d2 / term;
      g_dx2 = -0.5f * dd[i] * dd[i] / (dx2[i] * dx2[i]) * inv_d2 / term;
    } else if (REG == 1) {
      const float r = sqrtf(dx2[i]);
      g_dd = (dd[i] > 0.f ? 1.f : (dd[i] < 0.f ? -1.f : 0.f)) / r;
      g_dx2 = -0.5f * fabsf(dd[i]) / (dx2[i] * r);
    } else {
      g_dd = 2.f * dd[i] / dx2[i];
      g_dx2 = -dd[i] * dd[i] / (dx2[i] * dx2[i]);
    }
    g_dens += 2.f * reg_scale * g_dd;
    const float o_dz = g_dens * sigmoid_f(z[i]);
    a.dz0[s] = o_dz;
    mx_dz = fmaxf(mx_dz, fabsf(o_dz));
    if (a.dx != nullptr) {
      const float gx = 2.f * reg_scale * g_dx2 * 2.f;
      float* o = a.dx + s * 3;
      o[0] = gx * ex[i][0]; o[1] = gx * ex[i][1]; o[2] = gx * ex[i][2];
    }
    if (has_lv) { const float o_lv = g_m2 * bias[i] * elv[i] / S; a.dlog_var[s] = o_lv; mx_lv = fmaxf(mx_lv, fabsf(o_lv)); }
    if (has_lb) { const float o_lb = g_vout * c * dens[i] * bias[i] / S + g_lb_reg; a.dlog_bias[s] = o_lb; mx_lb = fmaxf(mx_lb, fabsf(o_lb)); }
  }
  if (a.dz0_absmax != nullptr) publish_absmax_f32(a.dz0_absmax, mx_dz);
  if (has_lv && a.dlog_var_absmax != nullptr) publish_absmax_f32(a.dlog_var_absmax, mx_lv);
  if (has_lb && a.dlog_bias_absmax != nullptr) publish_absmax_f32(a.dlog_bias_absmax, mx_lb);
}

}  // namespace

extern "C" int nesvor_imaging_loss(const nesvor_loss_t* args, void* stream) {
  if (args->B <= 0 || args->S <= 0) return 0;
  dim3 grid((args->B + 3) / 4), block(256);
  if (args->reg_type < 0 || args->reg_type > 2) return (int)hipErrorInvalidValue;
  // S = 64, 128, 256, 512: the single-pass kernel (NESVOR_LOSS_TWO_PASS=1: the general one, an A/B switch - the two agree
  // bit for bit)
  static const bool two_pass = []() { const char* e = getenv("NESVOR_LOSS_TWO_PASS"); return e != nullptr && atoi(e) != 0; }();
  const int K = (args->S % 64 == 0) ? args->S / 64 : 0;
  if (!two_pass && (K == 1 || K == 2 || K == 4 || K == 8)) {
#define NESVOR_LAUNCH_LOSS(R, KK) hipLaunchKernelGGL((imaging_loss_cached_kernel<R, KK>), grid, block, 0, (hipStream_t)stream, *args)
#define NESVOR_LAUNCH_LOSS_K(R)                                   \
  switch (K) {                                                    \
    case 1: NESVOR_LAUNCH_LOSS(R, 1); break;                      \
    case 2: NESVOR_LAUNCH_LOSS(R, 2); break;                      \
    case 4: NESVOR_LAUNCH_LOSS(R, 4); break;                      \
    default: NESVOR_LAUNCH_LOSS(R, 8); break;                     \
  }
    if (args->reg_type == 0) { NESVOR_LAUNCH_LOSS_K(0) }
    else if (args->reg_type == 1) { NESVOR_LAUNCH_LOSS_K(1) }
    else { NESVOR_LAUNCH_LOSS_K(2) }
#undef NESVOR_LAUNCH_LOSS_K
#undef NESVOR_LAUNCH_LOSS
    return (int)hipGetLastError();
  }
  if (args->reg_type == 0) hipLaunchKernelGGL(imaging_loss_kernel<0>, grid, block, 0, (hipStream_t)stream, *args);
  else if (args->reg_type == 1) hipLaunchKernelGGL(imaging_loss_kernel<1>, grid, block, 0, (hipStream_t)stream, *args);
  else if (args->reg_type == 2) hipLaunchKernelGGL(imaging_loss_kernel<2>, grid, block, 0, (hipStream_t)stream, *args);
  else return (int)hipErrorInvalidValue;
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Per-pixel -> per-slice gradient accumulation of the training step (one launch instead of four
// index_add + one (N,k)->(B,k) reduction): for pixel b of slice k = slice_idx[b]
//   dc[k] += dc_pix[b];  dlvs[k] += dlvs_pix[b];  dmat[k] += dpix[b] (12 floats);
//   dse[k] += sum_s dxa[b,s,:]   (dxa: the per-SAMPLE gradient of the slice-embedding input of an MLP)
// One wave per pixel; global fp32 atomics (n is a few hundred slices: ~B/n adds per address).
namespace {

__global__ __launch_bounds__(256) void slice_grads_kernel(const int64_t* __restrict__ slice_idx, const float* __restrict__ dc_pix,
                                                          const float* __restrict__ dlvs_pix, const float* __restrict__ dxa,
                                                          const float* __restrict__ dpix, float* dc, float* dlvs, float* dse,
                                                          float* dmat, int B, int S, int ks) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t k = slice_idx[b];
  if (dc_pix != nullptr && lane == 0) atomicAdd(dc + k, dc_pix[b]);
  if (dlvs_pix != nullptr && lane == 1) atomicAdd(dlvs + k, dlvs_pix[b]);
  if (dpix != nullptr && lane >= 16 && lane < 28) atomicAdd(dmat + k * 12 + (lane - 16), dpix[(size_t)b * 12 + (lane - 16)]);
  if (dxa == nullptr || ks <= 0) return;
  const float* src = dxa + (size_t)b * S * ks;
  const int k4 = ks >> 2;
  if ((ks & 3) == 0 && (64 % k4) == 0 && ((S * k4) & 255) == 0) {
    // float4 rows: lane <-> features 4 (lane % k4) .. +3; 1 KB per wave-load, four independent loads in flight
    const float4* src4 = reinterpret_cast<const float4*>(src);
    float4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int total4 = S * k4;
    for (int e = lane; e < total4; e += 256) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 v = src4[e + 64 * t];
        acc[t].x += v.x; acc[t].y += v.y; acc[t].z += v.z; acc[t].w += v.w;
      }
    }
    float r[4] = {(acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                  (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)};
    for (int off = 32; off >= k4; off >>= 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) r[t] += __shfl_xor(r[t], off, 64);
    }
    if (lane < k4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) atomicAdd(dse + k * ks + 4 * lane + t, r[t]);
    }
  } else if ((64 % ks) == 0) {  // lane <-> feature lane % ks, coalesced 256-byte rows
    float acc = 0.f;
    const int total = S * ks;
    for (int e = lane; e < total; e += 64) acc += src[e];
    for (int off = 32; off >= ks; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane < ks) atomicAdd(dse + k * ks + lane, acc);
  } else {
    for (int f = 0; f < ks; ++f) {
      float acc = 0.f;
      for (int s = lane; s < S; s += 64) acc += src[(size_t)s * ks + f];
      acc = wave_sum(acc);
      if (lane == 0) atomicAdd(dse + k * ks + f, acc);
    }
  }
}

}  // namespace

// The same sums with ONE workgroup per slice and no global atomics: the workgroup scans the batch's slice indices (B x 8
// bytes, from L2), lists the pixels of its slice in LDS - in batch order, by ballots and prefix counts, so that the sums are
// reproducible - and sums their contributions: the 14 scalars of a pixel by one thread each, the slice-embedding gradient
// by feature and row group; then it ADDS the totals to the outputs it alone owns.  B x 30 fp32 atomics on a few hundred
// addresses (23 us at B = 4096, twice that next to the owner pass of the hash-grid backward) become a few microseconds of reads.
// Needs B <= kSliceListMax and 256 % ks == 0 (or no dxa); nesvor_slice_grads_by_slice returns hipErrorInvalidValue otherwise and
// the caller uses the atomic kernel.
namespace {
constexpr int kSliceListMax = 4096;
__global__ __launch_bounds__(256) void slice_grads_by_slice_kernel(const int64_t* __restrict__ slice_idx, const float* __restrict__ dc_pix,
                                                                   const float* __restrict__ dlvs_pix, const float* __restrict__ dxa,
                                                                   const float* __restrict__ dpix, float* dc, float* dlvs, float* dse,
                                                                   float* dmat, int B, int S, int ks) {
  __shared__ int list[kSliceListMax];
  __shared__ int wcount[4];
  __shared__ float red[256];
  const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int M = 0;  // pixels of slice k found so far (uniform)
  for (int base = 0; base < B; base += 256) {
    const int b = base + tid;
    const bool hit = b < B && slice_idx[b] == (int64_t)k;
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = M;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    if (hit) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = b;
    M += wcount[0] + wcount[1] + wcount[2] + wcount[3];
    __syncthreads();
  }
  if (M == 0) return;
  {
    // the 14 scalars of a pixel (dc, dlvs, dmat[12]): thread (component c, pixel lane p) sums every 16th listed pixel, the 16
    // partial sums of a component are then added in a fixed order
    const int c = tid >> 4, p = tid & 15;
    float a = 0.f;
    if (c < 14) {
      const float* src = c == 0 ? dc_pix : (c == 1 ? dlvs_pix : dpix);
      if (src != nullptr) {
#pragma unroll 4
        for (int m = p; m < M; m += 16) a += c < 2 ? src[list[m]] : src[(size_t)list[m] * 12 + (c - 2)];
      }
    }
    red[tid] = a;
    __syncthreads();
    if (tid < 14) {
      float t = 0.f;
      for (int q = 0; q < 16; ++q) t += red[tid * 16 + q];
      if (tid == 0) { if (dc_pix != nullptr) dc[k] += t; }
      else if (tid == 1) { if (dlvs_pix != nullptr) dlvs[k] += t; }
      else if (dpix != nullptr) dmat[(size_t)k * 12 + (tid - 2)] += t;
    }
    __syncthreads();
  }
  if (dxa == nullptr || ks <= 0) return;
  const int f = tid % ks, rg = tid / ks, nrg = 256 / ks;
  float a = 0.f;
  for (int r = rg; r < S; r += nrg) {
#pragma unroll 8
    for (int m = 0; m < M; ++m) a += dxa[((size_t)list[m] * S + r) * ks + f];  // independent loads: many in flight
  }
  red[tid] = a;
  __syncthreads();
  if (tid < ks) {
    float t = 0.f;
    for (int g = 0; g < nrg; ++g) t += red[g * ks + tid];
    dse[(size_t)k * ks + tid] += t;
  }
}
}  // namespace

extern "C" int nesvor_slice_grads_by_slice(const int64_t* slice_idx, const float* dc_pix, const float* dlvs_pix, const float* dxa,
                                           const float* dpix, float* dc, float* dlvs, float* dse, float* dmat, int B, int S, int ks,
                                           int n_slices, void* stream) {
  if (B <= 0 || n_slices <= 0) return 0;
  if (B > kSliceListMax || (dxa != nullptr && ks > 0 && (ks > 256 || 256 % ks != 0))) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(slice_grads_by_slice_kernel, dim3((unsigned)n_slices), dim3(256), 0, (hipStream_t)stream, slice_idx, dc_pix, dlvs_pix,
                     dxa, dpix, dc, dlvs, dse, dmat, B, S, ks);
  return (int)hipGetLastError();
}

extern "C" int nesvor_slice_grads(const int64_t* slice_idx, const float* dc_pix, const float* dlvs_pix, const float* dxa,
                                  const float* dpix, float* dc, float* dlvs, float* dse, float* dmat, int B, int S, int ks,
                                  void* stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(slice_grads_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, slice_idx, dc_pix, dlvs_pix, dxa,
                     dpix, dc, dlvs, dse, dmat, B, S, ks);
  return (int)hipGetLastError();
}
