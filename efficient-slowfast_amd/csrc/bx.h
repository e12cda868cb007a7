// bx.h — fp32 products on the bf16 matrix pipe: exact three-way operand split and the six-product accumulate, shared
// by the attention (attn_bx.h) and convolution (conv_bx.hip) kernels.
#pragma once
#include "common.h"

// gfx950's f32-input MFMA runs at the vector rate (64 FLOP/clk/SIMD); its bf16 MFMA is 16x that.  An fp32 value is the
// EXACT sum of three bf16 pieces  a = a1 + a2 + a3  (8 significand bits each, round-to-nearest: |a2| <= 2^-8 |a|,
// |a3| <= 2^-17 |a|), every bf16 x bf16 product is exact in fp32, and the MFMA accumulates in fp32 — so
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1) + O(2^-24 |a b|)
// six bf16 MFMAs reproduce the fp32 product to fp32 rounding level (the three dropped terms are below 2^-24 |a||b|,
// the size of one fp32 rounding) at 6/16 of the f32-MFMA cost.  Small terms are accumulated first.
// tools/microbench/attn_precision.py measures both product paths against fp64.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: a -> bits 0..15, b -> 16..31
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
// (a, b) -> element `idx` of the three packed piece vectors (11 vector instructions per pair)
template <class V>
__device__ __forceinline__ void split_pair(float a, float b, V (&dst)[3], int idx) {
  const unsigned p1 = cvt_pk_bf16(a, b);
  a -= __builtin_bit_cast(float, p1 << 16);
  b -= __builtin_bit_cast(float, p1 & 0xffff0000u);
  const unsigned p2 = cvt_pk_bf16(a, b);
  a -= __builtin_bit_cast(float, p2 << 16);
  b -= __builtin_bit_cast(float, p2 & 0xffff0000u);
  dst[0][idx] = p1;
  dst[1][idx] = p2;
  dst[2][idx] = cvt_pk_bf16(a, b);
}
__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}
// acc += A B for split operands (index 0 = leading piece), small terms first
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
  c = mfma_bf(a[0], b[2], c);
  c = mfma_bf(a[2], b[0], c);
  c = mfma_bf(a[1], b[1], c);
  c = mfma_bf(a[0], b[1], c);
  c = mfma_bf(a[1], b[0], c);
  return mfma_bf(a[0], b[0], c);
}

// ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements, delivered column-major —
// lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i of the four rows.  EXEC must be
// all ones.
__device__ __forceinline__ u32x2 lds_read_tr(const unsigned short* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(
      u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}
