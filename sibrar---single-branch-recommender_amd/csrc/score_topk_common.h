// Shared device helpers of the fused scorer kernels (score_topk_f16.hip, score_topk_f16_n.hip).
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

#define S4_CAPH 64                       // candidate buffer entries per (user, lane half) in the global workspace

// All 64 lanes of the owning wave: the k best of the n0 + n1 (each <= 64, wave-uniform) entries of a user's two buffer halves
// (lane l holds b0[l] and b1[l]) are stored to b0[0 .. k), unsorted; returns the k-th best score (-inf and nothing moved while
// fewer than k entries exist). e / keep: the lane's two entries and whether they survived.
__device__ __forceinline__ float s4_select(unsigned long long* b0, unsigned long long* b1, int n0_any, int n1_any, int k, int lane,
                                           unsigned long long e[2], bool keep[2]) {
  const int n0 = __builtin_amdgcn_readfirstlane(n0_any), n1 = __builtin_amdgcn_readfirstlane(n1_any);
  // written and read by this wave only: same-CU vector memory path, in order (see s3_select)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  e[0] = lane < n0 ? b0[lane] : 0ull;
  e[1] = lane < n1 ? b1[lane] : 0ull;
  keep[0] = lane < n0;
  keep[1] = lane < n1;
  if (n0 + n1 < k) return -INFINITY;
  const unsigned int h0 = (unsigned int)(e[0] >> 32), h1 = (unsigned int)(e[1] >> 32);
  unsigned int T = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned int trial = T | (1u << bit);
    const int cnt = __popcll(__ballot(h0 >= trial)) + __popcll(__ballot(h1 >= trial));
    T = cnt >= k ? trial : T;
  }
  unsigned long long C = (unsigned long long)T << 32;
  const int c_ge = __popcll(__ballot(h0 >= T)) + __popcll(__ballot(h1 >= T));
  if (c_ge != k) {
    const int need = k - (__popcll(__ballot(h0 > T)) + __popcll(__ballot(h1 > T)));
    const unsigned int l0 = (unsigned int)e[0], l1 = (unsigned int)e[1];
    unsigned int Lw = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned int trial = Lw | (1u << bit);
      const int cnt = __popcll(__ballot(h0 == T && l0 >= trial)) + __popcll(__ballot(h1 == T && l1 >= trial));
      Lw = cnt >= need ? trial : Lw;
    }
    C |= (unsigned long long)Lw;
  }
  keep[0] = e[0] >= C;
  keep[1] = e[1] >= C;
  const unsigned long long m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);
  const int p0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m0, 0u));
  const int p1 = __popcll(m0) + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m1, 0u));
  if (keep[0]) b0[p0] = e[0];
  if (keep[1]) b0[p1] = e[1];
  return st_key2f(T);
}

// narrow-wave kernel (score_topk_f16_n.hip)
long s5_workspace_bytes(long Bu, long excl_nnz);
int s5_dispatch(const void* U, const void* It, int D, long Bu, int I, const long* u_idx, const long* eptr, const int* eidx, long excl_nnz,
                int item_offset, int k, float* out_val, int* out_idx, void* workspace, long workspace_bytes, hipStream_t s);
