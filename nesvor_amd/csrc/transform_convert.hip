// Rigid-transform conversion kernels for gfx950 (axis-angle <-> [R|t]).
//
// Behavioural spec: the reference's `transform_convert_cuda` extension
// (nesvor/transform/transform_convert_cuda_kernel.cu:14-440).  Written from
// the math, not from that file: R = c I + (1-c) k k^T + s [k]x with the
// first-order branch R = I + [a]x when |a|^2 <= 1e-6; the inverse goes through
// a 4-branch quaternion (branch masks r22 < 1e-6, r00 > r11, r00 < -r11) with
// the w >= 0 sign fix.  One thread per transform; n is tiny (n_slices or the
// batch size), so these kernels are latency- not bandwidth-bound: the point is
// to keep them on the stream (no host sync) and fuse the surrounding algebra.
#include <hip/hip_runtime.h>
#include "common.h"

namespace {

constexpr double kEps = 1e-6;

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ void t_sincos(T x, T* s, T* c);
template <> __device__ __forceinline__ void t_sincos<float>(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
template <> __device__ __forceinline__ void t_sincos<double>(double x, double* s, double* c) { *s = sin(x); *c = cos(x); }
template <typename T> __device__ __forceinline__ T t_atan2(T y, T x);
template <> __device__ __forceinline__ float t_atan2<float>(float y, float x) { return atan2f(y, x); }
template <> __device__ __forceinline__ double t_atan2<double>(double y, double x) { return atan2(y, x); }

template <typename T>
__device__ __forceinline__ void ax2mat_fwd_one(const T* a, T* m) {
  T x = a[0], y = a[1], z = a[2];
  T th2 = x * x + y * y + z * z;
  T R[9];
  if (th2 > (T)kEps) {
    T th = t_sqrt(th2);
    x /= th; y /= th; z /= th;
    T s, c;
    t_sincos(th, &s, &c);
    T oc = 1 - c;
    R[0] = c + x * x * oc;      R[1] = x * y * oc - z * s;  R[2] = y * s + x * z * oc;
    R[3] = z * s + x * y * oc;  R[4] = c + y * y * oc;      R[5] = -x * s + y * z * oc;
    R[6] = -y * s + x * z * oc; R[7] = x * s + y * z * oc;  R[8] = c + z * z * oc;
  } else {
    R[0] = 1;  R[1] = -z; R[2] = y;
    R[3] = z;  R[4] = 1;  R[5] = -x;
    R[6] = -y; R[7] = x;  R[8] = 1;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    m[r * 4 + 0] = R[r * 3 + 0];
    m[r * 4 + 1] = R[r * 3 + 1];
    m[r * 4 + 2] = R[r * 3 + 2];
    m[r * 4 + 3] = a[3 + r];
  }
}

template <typename T>
__global__ void ax2mat_fwd(const T* __restrict__ ax, T* __restrict__ mat, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ax2mat_fwd_one(ax + (size_t)i * 6, mat + (size_t)i * 12);
}

template <typename T>
__device__ __forceinline__ void ax2mat_bwd_one(const T* g, const T* a, T* o) {
  T G[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) G[r][c] = g[r * 4 + c];
  // vee of the antisymmetric part
  T sk[3] = {G[2][1] - G[1][2], G[0][2] - G[2][0], G[1][0] - G[0][1]};
  T k[3] = {a[0], a[1], a[2]};
  T th2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
  if (th2 > (T)kEps) {
    T th = t_sqrt(th2);
    k[0] /= th; k[1] /= th; k[2] /= th;
    T s, c;
    t_sincos(th, &s, &c);
    T oc = 1 - c;
    // dL/dc, dL/ds, dL/dk with k, c, s treated as independent
    T dc = 0, dk[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        dc += G[r][cc] * ((r == cc ? (T)1 : (T)0) - k[r] * k[cc]);
        dk[r] += oc * (G[r][cc] + G[cc][r]) * k[cc];
      }
    T ds = sk[0] * k[0] + sk[1] * k[1] + sk[2] * k[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) dk[d] += s * sk[d];
    T dth = c * ds - s * dc;
    T kd = dk[0] * k[0] + dk[1] * k[1] + dk[2] * k[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d] = dth * k[d] + (dk[d] - kd * k[d]) / th;
  } else {
    o[0] = sk[0]; o[1] = sk[1]; o[2] = sk[2];
  }
  o[3] = g[3]; o[4] = g[7]; o[5] = g[11];
}

template <typename T>
__global__ void ax2mat_bwd(const T* __restrict__ gmat, const T* __restrict__ ax, T* __restrict__ gax, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ax2mat_bwd_one(gmat + (size_t)i * 12, ax + (size_t)i * 6, gax + (size_t)i * 6);
}

// Quaternion (w,x,y,z) from R with the reference's branch selection; `s` is the
// branch's 2*sqrt(trace-like) value and `br` the branch id.
template <typename T>
__device__ __forceinline__ void quat_from_R(const T* m, T q[4], T* s_out, int* br_out) {
  T r00 = m[0], r01 = m[1], r02 = m[2];
  T r10 = m[4], r11 = m[5], r12 = m[6];
  T r20 = m[8], r21 = m[9], r22 = m[10];
  bool d2 = r22 < (T)kEps, d01 = r00 > r11, d0n1 = r00 < -r11;
  int br = (!d2 && !d0n1) ? 0 : (d2 && d01) ? 1 : (d2 && !d01) ? 2 : 3;
  T s;
  if (br == 0) {
    s = 2 * t_sqrt(r00 + r11 + r22 + 1);
    q[0] = (T)0.25 * s; q[1] = (r21 - r12) / s; q[2] = (r02 - r20) / s; q[3] = (r10 - r01) / s;
  } else if (br == 1) {
    s = 2 * t_sqrt(r00 - r11 - r22 + 1);
    q[0] = (r21 - r12) / s; q[1] = (T)0.25 * s; q[2] = (r01 + r10) / s; q[3] = (r02 + r20) / s;
  } else if (br == 2) {
    s = 2 * t_sqrt(r11 - r00 - r22 + 1);
    q[0] = (r02 - r20) / s; q[1] = (r01 + r10) / s; q[2] = (T)0.25 * s; q[3] = (r12 + r21) / s;
  } else {
    s = 2 * t_sqrt(r22 - r00 - r11 + 1);
    q[0] = (r10 - r01) / s; q[1] = (r02 + r20) / s; q[2] = (r12 + r21) / s; q[3] = (T)0.25 * s;
  }
  *s_out = s;
  *br_out = br;
}

template <typename T>
__device__ __forceinline__ void mat2ax_fwd_one(const T* m, T* o) {
  T q[4], s;
  int br;
  quat_from_R(m, q, &s, &br);
  if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  T n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  T si = t_sqrt(n2);
  T th = 2 * t_atan2(si, q[0]);
  T fac = (n2 > (T)kEps) ? th / si : (T)2 / q[0];
  o[0] = q[1] * fac; o[1] = q[2] * fac; o[2] = q[3] * fac;
  o[3] = m[3]; o[4] = m[7]; o[5] = m[11];
}

template <typename T>
__global__ void mat2ax_fwd(const T* __restrict__ mat, T* __restrict__ ax, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  mat2ax_fwd_one(mat + (size_t)i * 12, ax + (size_t)i * 6);
}

template <typename T>
__device__ __forceinline__ void mat2ax_bwd_one(const T* m, const T* ga, T* o) {
  T q[4], s;
  int br;
  quat_from_R(m, q, &s, &br);
  T sgn = q[0] < 0 ? (T)-1 : (T)1;
#pragma unroll
  for (int d = 0; d < 4; ++d) q[d] *= sgn;
  T n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  T si = t_sqrt(n2);
  T th = 2 * t_atan2(si, q[0]);
  bool big = n2 > (T)kEps;
  // reference regularises the small-angle denominators with +eps
  T sid = big ? si : si + (T)kEps;
  T fac = big ? th / si : (T)2 / q[0];
  T dot = q[1] * ga[0] + q[2] * ga[1] + q[3] * ga[2];
  T inv = (T)2 / (q[0] * q[0] + si * si);
  T t2 = (q[0] * inv - fac) / sid;
  T dq[4];
  dq[0] = -dot * inv;
#pragma unroll
  for (int d = 0; d < 3; ++d) dq[d + 1] = dot * t2 * (q[d + 1] / sid) + fac * ga[d];
#pragma unroll
  for (int d = 0; d < 4; ++d) { q[d] *= sgn; dq[d] *= sgn; }
  // Wire quaternion-component grads back to matrix entries.
  // antisymmetric numerators: A=(r21-r12) B=(r02-r20) C=(r10-r01)
  // symmetric numerators:     P=(r01+r10) Q=(r02+r20) Rr=(r12+r21)
  T gA = 0, gB = 0, gC = 0, gP = 0, gQ = 0, gR = 0, major, acc;
  T ds0, ds1, ds2;  // signs of d s / d r_dd
  if (br == 0) {
    gA = dq[1]; gB = dq[2]; gC = dq[3]; major = dq[0];
    acc = q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
    ds0 = 1; ds1 = 1; ds2 = 1;
  } else if (br == 1) {
    gA = dq[0]; gP = dq[2]; gQ = dq[3]; major = dq[1];
    acc = q[0] * dq[0] + q[2] * dq[2] + q[3] * dq[3];
    ds0 = 1; ds1 = -1; ds2 = -1;
  } else if (br == 2) {
    gB = dq[0]; gP = dq[1]; gR = dq[3]; major = dq[2];
    acc = q[0] * dq[0] + q[1] * dq[1] + q[3] * dq[3];
    ds0 = -1; ds1 = 1; ds2 = -1;
  } else {
    gC = dq[0]; gQ = dq[1]; gR = dq[2]; major = dq[3];
    acc = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2];
    ds0 = -1; ds1 = -1; ds2 = 1;
  }
  T dS = (-acc / s + (T)0.25 * major) * ((T)2 / s);
  o[0] = ds0 * dS;          o[1] = (gP - gC) / s;     o[2] = (gQ + gB) / s;   o[3] = ga[3];
  o[4] = (gP + gC) / s;     o[5] = ds1 * dS;          o[6] = (gR - gA) / s;   o[7] = ga[4];
  o[8] = (gQ - gB) / s;     o[9] = (gR + gA) / s;     o[10] = ds2 * dS;       o[11] = ga[5];
}

template <typename T>
__global__ void mat2ax_bwd(const T* __restrict__ mat, const T* __restrict__ gax, T* __restrict__ gmat, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  mat2ax_bwd_one(mat + (size_t)i * 12, gax + (size_t)i * 6, gmat + (size_t)i * 12);
}

// Pose regulariser of NeSVoR.trans_loss (nesvor/nesvor/models.py:357-363), forward AND gradient in one
// launch (the reference runs axisangle2mat x2, inv, compose, mat2axisangle and their backwards: ~20
// launches incl. batched 3x3 GEMMs, every iteration):
//   err = axisangle( inv(T_init) o T_cur );  loss = mean(err_R^2) + 1e-3 mean(err_T^2)
// per slice: loss_k (its share of the two means) and d loss / d axisangle_k.
__device__ __forceinline__ void trans_loss_one(int i, const float* __restrict__ ax, const float* __restrict__ ax_init,
                                               float* __restrict__ loss_k, float* __restrict__ grad_ax, int n) {
  float a[6], a0[6], X[12], Y[12], M[12], err[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) { a[d] = ax[(size_t)i * 6 + d]; a0[d] = ax_init[(size_t)i * 6 + d]; }
  ax2mat_fwd_one(a, X);
  ax2mat_fwd_one(a0, Y);
  // inv(Y) = [Ry^T | -Ry ty];  compose(inv(Y), X): R = Ry^T Rx,  t = tx + Rx^T (-Ry ty)
  float w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) w[r] = -(Y[r * 4 + 0] * Y[3] + Y[r * 4 + 1] * Y[7] + Y[r * 4 + 2] * Y[11]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) M[r * 4 + c] = Y[0 * 4 + r] * X[0 * 4 + c] + Y[1 * 4 + r] * X[1 * 4 + c] + Y[2 * 4 + r] * X[2 * 4 + c];
    M[r * 4 + 3] = X[r * 4 + 3] + (X[0 * 4 + r] * w[0] + X[1 * 4 + r] * w[1] + X[2 * 4 + r] * w[2]);
  }
  mat2ax_fwd_one(M, err);
  const float cr = 1.f / (3.f * n), ct = 1e-3f / (3.f * n);
  loss_k[i] = cr * (err[0] * err[0] + err[1] * err[1] + err[2] * err[2]) + ct * (err[3] * err[3] + err[4] * err[4] + err[5] * err[5]);
  float gerr[6] = {2 * cr * err[0], 2 * cr * err[1], 2 * cr * err[2], 2 * ct * err[3], 2 * ct * err[4], 2 * ct * err[5]};
  float gM[12], gX[12], g[6];
  mat2ax_bwd_one(M, gerr, gM);
  // dRx = Ry gR  -  w gt^T (from t = tx + Rx^T w: dRx[i][k] += w_i gt_k),  dtx = gt
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      gX[r * 4 + c] = Y[r * 4 + 0] * gM[0 * 4 + c] + Y[r * 4 + 1] * gM[1 * 4 + c] + Y[r * 4 + 2] * gM[2 * 4 + c] + w[r] * gM[c * 4 + 3];
    gX[r * 4 + 3] = gM[r * 4 + 3];
  }
  ax2mat_bwd_one(gX, a, g);
#pragma unroll
  for (int d = 0; d < 6; ++d) grad_ax[(size_t)i * 6 + d] = g[d];
}
__global__ void trans_loss_kernel(const float* __restrict__ ax, const float* __restrict__ ax_init,
                                  float* __restrict__ loss_k, float* __restrict__ grad_ax, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  trans_loss_one(i, ax, ax_init, loss_k, grad_ax, n);
}

// ------------------------------------------------------------------------------------------------
// Small-tensor bookkeeping of one training iteration, two launches instead of ~15 (per-slice tensors of a few
// hundred elements: every separate launch costs more than the work).
//   prologue: c = n softmax(logit_coef) (models.py:296), mat = axisangle2mat(axisangle), zero-fill of the
//             per-slice accumulators;
//   epilogue: d logit_coef from d c (softmax backward), d axisangle = axisangle2mat_backward(dmat) + w_T dtrans,
//             and the loss values {MSE, logVar, MSE+logVar, transReg, imageReg} from the per-pixel partial sums.
// One workgroup each (n and B are small); block-wide sums through LDS.
// Round 6: 256 threads per workgroup, not 1024.  Both launches run NEXT TO the owner pass of the hash-grid backward (side stream),
// whose 512-thread workgroups leave most CUs with 8 free wave slots: a 1024-thread workgroup cannot start before a CU has drained
// to half - the epilogue took 41 us (21 us at 2^17 points) on the chain that decides when the next iteration's forward may start
// (profiles/r05_step_timeline.txt), for a few microseconds of work.
constexpr int kStepBlock = 256;
// (any block size that is a multiple of 64 up to 1024)
__device__ __forceinline__ float block_sum_1024(float v, float* red /* 16 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

__global__ __launch_bounds__(kStepBlock) void step_prologue_kernel(const float* __restrict__ logit_coef, float* __restrict__ c,
                                                            const float* __restrict__ axisangle, float* __restrict__ mat,
                                                            float* __restrict__ zero_buf, int n_zero, int n,
                                                            const float* __restrict__ ax_init, float* __restrict__ trans_k,
                                                            float* __restrict__ trans_grad) {
  __shared__ float red[16];
  __builtin_amdgcn_s_setprio(3);  // (see step_epilogue_kernel)
  // (grid = 3 or 4: softmax | pose matrices | zero-fill | pose regulariser, one workgroup each)
  if (blockIdx.x == 3) {
    // NeSVoR.trans_loss and its gradient (trans_loss_kernel's arithmetic): a function of the parameters alone, so it rides in the
    // iteration's first launch instead of one of its own on a second stream (round 5: the fork's and the join's event markers
    // cost the main stream 6-8 us each, more than the kernel)
    for (int i = threadIdx.x; i < n; i += blockDim.x) trans_loss_one(i, axisangle, ax_init, trans_k, trans_grad, n);
    return;
  }
  if (blockIdx.x == 1) {
    if (axisangle != nullptr)
      for (int i = threadIdx.x; i < n; i += blockDim.x) ax2mat_fwd_one(axisangle + (size_t)i * 6, mat + (size_t)i * 12);
    return;
  }
  if (blockIdx.x == 2) {
    for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero_buf[i] = 0.f;
    return;
  }
  if (logit_coef != nullptr) {
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, logit_coef[i]);
    mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, red[w]);
    float se = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) se += expf(logit_coef[i] - mx);
    se = block_sum_1024(se, red);
    const float scale = (float)n / se;
    for (int i = threadIdx.x; i < n; i += blockDim.x) c[i] = expf(logit_coef[i] - mx) * scale;
  }
}

__global__ __launch_bounds__(kStepBlock) void step_epilogue_kernel(const float* __restrict__ dc, const float* __restrict__ c,
                                                            float* __restrict__ dlogit, const float* __restrict__ dmat,
                                                            const float* __restrict__ axisangle, const float* __restrict__ dtrans,
                                                            float w_trans, float* __restrict__ daxisangle,
                                                            const float* __restrict__ loss_pix, const float* __restrict__ trans_terms,
                                                            float* __restrict__ losses, int n, int B, float inv_B, float img_scale,
                                                            float img_offset) {
  // three independent pieces, one workgroup each (grid = 3): five block reductions in a row were 10-13 us on the branch that
  // decides when the next iteration can start
  __shared__ float red[16];
  // A few dependent steps next to the owner pass's thousands of waves: the SIMD arbiter issues the OLDEST ready wave first, and
  // this kernel's waves are the youngest on the chip - 40 us for a few microseconds of work (profiles/r06_step_timeline.txt).
  // Raised wave priority puts its instructions in front.
  __builtin_amdgcn_s_setprio(3);
  if (blockIdx.x == 0) {
    if (dc != nullptr) {  // c = n softmax(l):  dl = c (dc - <dc, c> / n)
      float dot = 0.f;
      for (int i = threadIdx.x; i < n; i += blockDim.x) dot += dc[i] * c[i];
      dot = block_sum_1024(dot, red) / (float)n;
      for (int i = threadIdx.x; i < n; i += blockDim.x) dlogit[i] = c[i] * (dc[i] - dot);
    }
  } else if (blockIdx.x == 1) {
    float tr = 0.f;
    if (dmat != nullptr) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float g[6];
        ax2mat_bwd_one(dmat + (size_t)i * 12, axisangle + (size_t)i * 6, g);
#pragma unroll
        for (int d = 0; d < 6; ++d) daxisangle[(size_t)i * 6 + d] = g[d] + w_trans * dtrans[(size_t)i * 6 + d];
        tr += trans_terms[i];
      }
      tr = block_sum_1024(tr, red);
    }
    if (threadIdx.x == 0) losses[3] = tr;
  } else {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { s0 += loss_pix[3 * i]; s1 += loss_pix[3 * i + 1]; s2 += loss_pix[3 * i + 2]; }
    s0 = block_sum_1024(s0, red); s1 = block_sum_1024(s1, red); s2 = block_sum_1024(s2, red);
    if (threadIdx.x == 0) {
      losses[0] = s0 * inv_B; losses[1] = s1 * inv_B; losses[2] = (s0 + s1) * inv_B;
      losses[4] = s2 * img_scale + img_offset;
    }
  }
}

template <typename K, typename... Args>
int launch1d(K kernel, int n, void* stream, Args... args) {
  if (n <= 0) return 0;
  constexpr int B = 64;
  hipLaunchKernelGGL(kernel, dim3((n + B - 1) / B), dim3(B), 0, (hipStream_t)stream, args..., n);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int nesvor_step_prologue_pose(const float* logit_coef, float* c, const float* axisangle, float* mat, float* zero_buf,
                                         int n_zero, int n, const float* axisangle_init, float* trans_terms, float* g_trans, void* stream) {
  if (n <= 0) return 0;
  const bool pose = axisangle_init != nullptr;
  if (pose && (axisangle == nullptr || trans_terms == nullptr || g_trans == nullptr)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(step_prologue_kernel, dim3(pose ? 4 : 3), dim3(kStepBlock), 0, (hipStream_t)stream, logit_coef, c, axisangle, mat, zero_buf,
                     n_zero, n, axisangle_init, trans_terms, g_trans);
  return (int)hipGetLastError();
}

extern "C" int nesvor_step_prologue(const float* logit_coef, float* c, const float* axisangle, float* mat, float* zero_buf,
                                    int n_zero, int n, void* stream) {
  return nesvor_step_prologue_pose(logit_coef, c, axisangle, mat, zero_buf, n_zero, n, nullptr, nullptr, nullptr, stream);
}

extern "C" int nesvor_step_epilogue(const float* dc, const float* c, float* dlogit, const float* dmat, const float* axisangle,
                                    const float* dtrans, float w_trans, float* daxisangle, const float* loss_pix,
                                    const float* trans_terms, float* losses, int n, int B, float img_scale, float img_offset,
                                    void* stream) {
  if (n <= 0 || B <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(step_epilogue_kernel, dim3(3), dim3(kStepBlock), 0, (hipStream_t)stream, dc, c, dlogit, dmat, axisangle, dtrans,
                     w_trans, daxisangle, loss_pix, trans_terms, losses, n, B, 1.f / (float)B, img_scale, img_offset);
  return (int)hipGetLastError();
}

extern "C" int nesvor_trans_loss(const float* ax, const float* ax_init, float* loss_per_slice, float* grad_ax, int n,
                                 void* stream) {
  return launch1d(trans_loss_kernel, n, stream, ax, ax_init, loss_per_slice, grad_ax);
}

extern "C" {
int nesvor_axisangle2mat_forward(const float* ax, float* mat, int n, void* st) { return launch1d(ax2mat_fwd<float>, n, st, ax, mat); }
int nesvor_axisangle2mat_backward(const float* g, const float* ax, float* gax, int n, void* st) { return launch1d(ax2mat_bwd<float>, n, st, g, ax, gax); }
int nesvor_mat2axisangle_forward(const float* mat, float* ax, int n, void* st) { return launch1d(mat2ax_fwd<float>, n, st, mat, ax); }
int nesvor_mat2axisangle_backward(const float* mat, const float* gax, float* gmat, int n, void* st) { return launch1d(mat2ax_bwd<float>, n, st, mat, gax, gmat); }
int nesvor_axisangle2mat_forward_f64(const double* ax, double* mat, int n, void* st) { return launch1d(ax2mat_fwd<double>, n, st, ax, mat); }
int nesvor_axisangle2mat_backward_f64(const double* g, const double* ax, double* gax, int n, void* st) { return launch1d(ax2mat_bwd<double>, n, st, g, ax, gax); }
int nesvor_mat2axisangle_forward_f64(const double* mat, double* ax, int n, void* st) { return launch1d(mat2ax_fwd<double>, n, st, mat, ax); }
int nesvor_mat2axisangle_backward_f64(const double* mat, const double* gax, double* gmat, int n, void* st) { return launch1d(mat2ax_bwd<double>, n, st, mat, gax, gmat); }
}
