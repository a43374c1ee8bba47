// Dense optimizer steps over one flat fp32 parameter buffer (train/trainer.py:62-68 builds
// torch.optim.{AdamW,Adam,Adagrad}(model.parameters(), lr, weight_decay) — every element of every table is updated
// every step, so the step is a pure HBM stream: AdamW moves 28 B per parameter (read p,g,m,v; write p,m,v)).
// The arithmetic follows torch's single-tensor rules in fp32, in the same operation order.
#include "common.h"

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float b1, float b2, float eps, float wd, float step_size, float bc2_sqrt,
                             int decoupled) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float pe = p[e], ge = g[e];
    if (decoupled) pe *= (1.f - lr * wd);       // AdamW: p.mul_(1 - lr * wd)
    else ge += wd * pe;                         // Adam: grad = grad.add(p, alpha=wd)
    const float me = m[e] + (ge - m[e]) * (1.f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float ve = v[e] * b2 + (1.f - b2) * ge * ge;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(ve) / bc2_sqrt + eps;
    p[e] = pe - step_size * (me / denom);
    m[e] = me;
    v[e] = ve;
  }
}

__global__ void adagrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s, long n, float lr,
                               float eps, float wd) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float pe = p[e];
    const float ge = g[e] + wd * pe;
    const float se = s[e] + ge * ge;
    p[e] = pe - lr * (ge / (sqrtf(se) + eps));
    s[e] = se;
  }
}

static int grid_for(long n) {
  int b = sbr_cdiv(n, 256);
  return b > 8192 ? 8192 : (b < 1 ? 1 : b);
}

// kind: 0 AdamW, 1 Adam. step is the 1-based step count (bias corrections are evaluated on the host in double).
extern "C" int sbr_adam_step(int kind, float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                             double eps, double wd, long step, void* stream) {
  SBR_REQUIRE(kind == 0 || kind == 1, "sbr_adam_step: unknown kind %d", kind);
  SBR_REQUIRE(p && g && m && v, "sbr_adam_step: null operand");
  SBR_REQUIRE(step >= 1, "sbr_adam_step: step must be >= 1");
  if (n == 0) return SBR_OK;
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  adamw_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, (float)lr, (float)b1, (float)b2, (float)eps, (float)wd,
                                                             (float)(lr / bc1), (float)sqrt(bc2), kind == 0);
  SBR_CHECK_LAUNCH("sbr_adam_step");
  return SBR_OK;
}

extern "C" int sbr_adagrad_step(float* p, const float* g, float* state_sum, long n, double lr, double eps, double wd,
                                void* stream) {
  SBR_REQUIRE(p && g && state_sum, "sbr_adagrad_step: null operand");
  if (n == 0) return SBR_OK;
  adagrad_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, g, state_sum, n, (float)lr, (float)eps, (float)wd);
  SBR_CHECK_LAUNCH("sbr_adagrad_step");
  return SBR_OK;
}
