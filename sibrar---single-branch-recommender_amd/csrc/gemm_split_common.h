// Shared pieces of the bf16-split fp32 GEMM kernels (gemm_split_f32.hip, gemm_split_tn_f32.hip): the exact three-way split of an
// fp32 number into bf16 numbers and the MFMA wrapper. See the header of gemm_split_f32.hip for the arithmetic.
#pragma once
#include "gemm_args.h"

typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) sp_u32x4 sp_lds_u32x4;

__device__ __forceinline__ unsigned sp_pack(float x, float y) {     // two fp32 -> two bf16 (round to nearest even), x in the low half
  sp_f32x2 v = {x, y};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sp_bf16x2));
}

// (x, y) -> three packed bf16 pairs with x = x0 + x1 + x2 exactly
__device__ __forceinline__ void sp_split2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = sp_pack(x, y);
  x -= __uint_as_float(p0 << 16);
  y -= __uint_as_float(p0 & 0xffff0000u);
  p1 = sp_pack(x, y);
  x -= __uint_as_float(p1 << 16);
  y -= __uint_as_float(p1 & 0xffff0000u);
  p2 = sp_pack(x, y);
}

__device__ __forceinline__ void sp_split8(const float4 lo, const float4 hi, sp_u32x4& p0, sp_u32x4& p1, sp_u32x4& p2) {
  unsigned a, b, c;
  sp_split2(lo.x, lo.y, a, b, c); p0[0] = a; p1[0] = b; p2[0] = c;
  sp_split2(lo.z, lo.w, a, b, c); p0[1] = a; p1[1] = b; p2[1] = c;
  sp_split2(hi.x, hi.y, a, b, c); p0[2] = a; p1[2] = b; p2[2] = c;
  sp_split2(hi.z, hi.w, a, b, c); p0[3] = a; p1[3] = b; p2[3] = c;
}

__device__ __forceinline__ sp_f32x16 sp_mfma(const sp_u32x4 a, const sp_u32x4 b, const sp_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sp_bf16x8, a), __builtin_bit_cast(sp_bf16x8, b), c, 0, 0, 0);
}

