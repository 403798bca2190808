// AdamW update of one parameter, shared by the flat-buffer optimiser kernel (adamw.hip) and the hash-grid backward's owner
// pass (hashgrid.hip), which can apply it to a table chunk while the chunk's gradient is still in LDS.
// Follows torch.optim.AdamW as the reference's training loop configures it (nesvor/nesvor/train.py:144-152).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

struct AdamArgs {
  float lr, beta1, beta2, eps, decay_mul, step_size, inv_sqrt_bc2, grad_scale;
};

inline AdamArgs make_adam_args(float lr, float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                               float bias_correction2, float grad_scale) {
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.decay_mul = 1.f - lr * weight_decay;
  a.step_size = lr / bias_correction1;
  a.inv_sqrt_bc2 = 1.f / sqrtf(bias_correction2);
  a.grad_scale = grad_scale;
  return a;
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
  const float gs = g * a.grad_scale;
  p *= a.decay_mul;
  m = fmaf(1.f - a.beta1, gs - m, m);
  v = fmaf(1.f - a.beta2, gs * gs, a.beta2 * v);
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  p -= a.step_size * (m / denom);
}
