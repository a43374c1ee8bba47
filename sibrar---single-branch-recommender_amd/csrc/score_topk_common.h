// Device helpers of the fused scorer kernel (score_topk_f16_n.hip).
#pragma once
#include "common.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

__device__ __forceinline__ unsigned int st_f2key(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float st_key2f(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Lanes of one wave exchange candidate entries through LDS without any hardware synchronisation (LDS operations of a wave
// execute in order); the wavefront-scope fence only stops the compiler from caching / forwarding values across the exchange.
__device__ __forceinline__ void st_wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

template <int N>
__device__ __forceinline__ void st_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wave-uniform read of an LDS word that another wave of the workgroup writes
__device__ __forceinline__ int st_peek(lds_int* p) {
  st_wave_fence();
  const int v = *(volatile lds_int*)p;
  return __builtin_amdgcn_readfirstlane(v);
}
