// Shared helpers for the SiBraR HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <stdlib.h>
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

// torch.relu: a NaN stays a NaN ("x > 0 ? x : 0" — one v_max_f32 — would turn the NaN of an overflowed activation into a clean 0 and
// hide the overflow from every later layer and from the loss)
__device__ __forceinline__ float sbr_relu(float x) { return x < 0.f ? 0.f : x; }

__device__ __forceinline__ float sbr_act(float x, int act) {
  switch (act) {
    case SBR_ACT_RELU: return sbr_relu(x);
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

// "This wave's vector-memory operations have been performed" — the wait an agent-scope hand-off needs in front of the workgroup
// barrier that precedes an arrival-counter add (MI355X_MICROARCH.md, Valid forms: every storing wave's s_waitcnt vmcnt(0), the
// barrier, then the counter). Stores and no-return atomics count in vmcnt on gfx950. A workgroup-scope release fence does NOT emit
// this wait (checked in the disassembly: the counter add followed the sum atomics with only lgkmcnt(0) + s_barrier in between), and
// inline asm is invisible to the compiler's wait-count pass, so it cannot be dropped. tools/check_arrival_waits.py greps the built
// kernels for it.
#define SBR_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

static inline int sbr_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute configures the CURRENT device's copy of a kernel: `slot` (one static per kernel instantiation) remembers the device
// it was last configured on, so a process that drives several GPUs configures each of them (a plain "done" flag would skip the second)
static inline bool sbr_attr_stale(int* slot) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || *slot != dev) { *slot = dev; return true; }
  return false;
}


// ---------------------------------------------------------------------------------------------------------------
// Column reductions over an [n, D] row-major matrix with D % 4 == 0 (column sums of bias gradients, BatchNorm statistics).
// A thread owns one float4 column group and walks the block's row range with four independent 16-byte loads in flight,
// accumulating in fp32 (at most a few dozen rows per thread), then the row lanes of the block are combined in double through
// LDS and one double atomic per column and block goes to one of SBR_COLRED_REP replicas of the [K][D] result (blocks b, b + REP,
// ... share a replica: 512 blocks hitting the same D addresses serialise in the L2 atomic units — measured 50 us instead of
// 20 for a 90112 x 128 column sum). sbr_colred_final_kernel then adds the replicas into ws[0 .. K*D).
// Workspace layout (doubles): [K*D totals][REP][K*D]; the replica part must be zero when a reduction starts and the
// finishing kernel (sbr_colred_take) leaves it zero again.
//   K: reduced quantities per element; f(row, cg, v) fills v[K] (float4 each) for columns 4*cg .. 4*cg+3 of `row`.
// Block = 256 threads = RL row lanes x (D/4) column groups (D <= 1024).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sbr_f4_add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

#define SBR_COLRED_REP 16

template <int K, class F>
__device__ __forceinline__ void sbr_col_reduce(long n, int D, double* __restrict__ ws, F f) {
  __shared__ float4 sm[K][256];
  const int C4 = D >> 2, RL = 256 / C4;
  const int t = threadIdx.x, cg = t % C4, rl = t / C4;
  const long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long lo = blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  float4 acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < RL) {
    long j = lo + rl;
    for (; j + 3L * RL < hi; j += 4L * RL) {
      float4 v0[K], v1[K], v2[K], v3[K];
      f(j, cg, v0); f(j + RL, cg, v1); f(j + 2L * RL, cg, v2); f(j + 3L * RL, cg, v3);
#pragma unroll
      for (int k = 0; k < K; ++k) { sbr_f4_add(v0[k], v1[k]); sbr_f4_add(v2[k], v3[k]); sbr_f4_add(v0[k], v2[k]); sbr_f4_add(acc[k], v0[k]); }
    }
    for (; j < hi; j += RL) {
      float4 v[K];
      f(j, cg, v);
#pragma unroll
      for (int k = 0; k < K; ++k) sbr_f4_add(acc[k], v[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) sm[k][t] = acc[k];
  __syncthreads();
  if (t < C4) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int r = 0; r < RL; ++r) {
        const float4 p = sm[k][r * C4 + t];
        s0 += (double)p.x; s1 += (double)p.y; s2 += (double)p.z; s3 += (double)p.w;
      }
      double* o = ws + (long)(1 + (blockIdx.x % SBR_COLRED_REP)) * K * D + (long)k * D + 4 * t;
      atomicAdd(o, s0); atomicAdd(o + 1, s1); atomicAdd(o + 2, s2); atomicAdd(o + 3, s3);
    }
  }
}

// sum of the replicas of entry i, leaving the replicas zeroed for the next call (the workspace contract: zero on first use,
// zero again on return — no memset node per call)
__device__ __forceinline__ double sbr_colred_take(double* __restrict__ ws, int KD, int i) {
  double s = 0.0;
#pragma unroll
  for (int r = 1; r <= SBR_COLRED_REP; ++r) {
    s += ws[(long)r * KD + i];
    ws[(long)r * KD + i] = 0.0;
  }
  return s;
}

static __global__ void sbr_colred_final_kernel(double* __restrict__ ws, int KD) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < KD) ws[i] = sbr_colred_take(ws, KD, i);
}

// launch geometry of sbr_col_reduce kernels: every block gets >= 8 passes of its row lanes
static inline int sbr_col_reduce_blocks(long n, int D) {
  const int RL = 256 / (D >> 2);
  long b = (n + 8L * RL - 1) / (8L * RL);
  const long cap = 512;
  if (b > cap) b = cap;          // one double atomic per block and column: more blocks only add contention on D addresses
  return b < 1 ? 1 : (int)b;
}
static inline bool sbr_col_reduce_ok(const void* p, long ld, int D) {
  return (D & 3) == 0 && D >= 4 && D <= 1024 && (ld & 3) == 0 && (((uintptr_t)p) & 15) == 0;
}
