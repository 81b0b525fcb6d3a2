// Fused Adam over the flat parameter buffer: one launch per optimiser step (the reference steps torch.optim.Adam over
// ~60 parameter tensors, baseline/pipelines/base_ray_pipeline.py:246-269).  HBM-bound: 16 B read + 12 B written per
// parameter, float4 accesses, grid-stride.
#include "common.h"
#include "../../include/snerf_hip.h"

namespace snerf {

struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  unsigned long long n4;   // float4 count (the flat buffers are padded to a multiple of 4)
  float beta1, beta2, one_m_beta1, one_m_beta2, eps, step_size, inv_sqrt_bc2, grad_scale;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
  g *= a.grad_scale;
  m = a.beta1 * m + a.one_m_beta1 * g;          // exp_avg.lerp_(grad, 1 - beta1)
  v = a.beta2 * v + a.one_m_beta2 * (g * g);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  p -= a.step_size * (m / denom);               // param.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    adam1(p.x, g.x, m.x, v.x, a);
    adam1(p.y, g.y, m.y, v.y, a);
    adam1(p.z, g.z, m.z, v.z, a);
    adam1(p.w, g.w, m.w, v.w, a);
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
  }
}

}  // namespace snerf

using namespace snerf;

extern "C" int snerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, unsigned long long n,
                               float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq) { set_error("snerf_adam_step: null buffer"); return SNERF_ERR_NULL; }
  if (n == 0) return SNERF_OK;
  if (n & 3ull) { set_error("snerf_adam_step: n must be a multiple of 4 (pad the flat buffers)"); return SNERF_ERR_BAD_DESC; }
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) {
    set_error("snerf_adam_step: buffers must be 16-byte aligned"); return SNERF_ERR_BAD_DESC; }
  if (step < 1) { set_error("snerf_adam_step: step counts from 1"); return SNERF_ERR_BAD_DESC; }
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) { set_error("snerf_adam_step: betas must lie in [0, 1)"); return SNERF_ERR_BAD_DESC; }
  AdamArgs a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n4 = n / 4;
  a.beta1 = beta1; a.beta2 = beta2; a.one_m_beta1 = 1.f - beta1; a.one_m_beta2 = 1.f - beta2; a.eps = eps;
  // bias corrections in double, as torch.optim.Adam computes them on the host
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.step_size = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  a.grad_scale = grad_scale;
  const unsigned long long want = (a.n4 + 255) / 256;
  const unsigned blocks = (unsigned)(want < 2048ull ? want : 2048ull);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  SNERF_LAUNCH_CHECK();
  return SNERF_OK;
}
