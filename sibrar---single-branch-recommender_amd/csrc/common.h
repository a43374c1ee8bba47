// Shared helpers for the SiBraR HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define SBR_OK 0
#define SBR_ERR_ARG 1
#define SBR_ERR_HIP 2

#define SBR_ACT_NONE 0
#define SBR_ACT_RELU 1
#define SBR_ACT_TANH 2
#define SBR_ACT_SIGMOID 3
#define SBR_ACT_SELU 4

void sbr_set_error(const char* fmt, ...);

#define SBR_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      sbr_set_error(__VA_ARGS__);              \
      return SBR_ERR_ARG;                      \
    }                                          \
  } while (0)

#define SBR_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      sbr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return SBR_ERR_HIP;                                                   \
    }                                                                       \
  } while (0)

// selu constants (torch.nn.SELU)
#define SBR_SELU_ALPHA 1.6732632423543772848170429916717f
#define SBR_SELU_SCALE 1.0507009873554804934193349852946f

__device__ __forceinline__ float sbr_act(float x, int act) {
  switch (act) {
    case SBR_ACT_RELU: return x > 0.f ? x : 0.f;
    case SBR_ACT_TANH: return tanhf(x);
    case SBR_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case SBR_ACT_SELU: return SBR_SELU_SCALE * (x > 0.f ? x : SBR_SELU_ALPHA * (expf(x) - 1.f));
    default: return x;
  }
}

// d act / d pre-activation expressed through the activation OUTPUT y.
__device__ __forceinline__ float sbr_act_grad_from_out(float y, int act) {
  switch (act) {
    case SBR_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case SBR_ACT_TANH: return 1.f - y * y;
    case SBR_ACT_SIGMOID: return y * (1.f - y);
    case SBR_ACT_SELU: return y > 0.f ? SBR_SELU_SCALE : (y + SBR_SELU_SCALE * SBR_SELU_ALPHA);
    default: return 1.f;
  }
}

__device__ __forceinline__ float sbr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double sbr_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sbr_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int sbr_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
