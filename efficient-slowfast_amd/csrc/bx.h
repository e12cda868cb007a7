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
// Residual of a split step: a - (low half of p as a float), b - (high half of p).  v_dot2c_f32_bf16 (acc += p.lo * m.lo +
// p.hi * m.hi) with m = (-1, 0) / (0, -1) does the unpack and the subtraction in ONE vector instruction, and the residual
// of a round-to-nearest bf16 conversion is exactly representable, so both forms return the same bits — measured in round
// 6 (profiles/r06_dot2_ab.txt, tools/microbench/split_dot2.hip): bit-identical over 2^25 pairs of every exponent class,
// 1.1x the split rate in isolation (the instruction issues at half rate: 7 instructions cost what 11 did), and inside
// the kernels NOT faster — attention d = 32 forward 3.74 vs 3.45 ms, backward 8.98 vs 8.83, d = 8 equal, step 53.9 vs
// 53.4 ms.  The shift / mask + subtract form stays the default; -DSF_BX_DOT2=1 builds the other.
#ifndef SF_BX_DOT2
#define SF_BX_DOT2 0
#endif
__device__ __forceinline__ float bx_res_lo(float a, unsigned p) {
#if SF_BX_DOT2
  // the multiplier (-1, 0) = 0x0000bf80 must not reach the instruction as the inline constant "-1.0": hipcc 7.2 folds it
  // to that, and the hardware reads the inline constant as the FLOAT -1.0 = (0, -1) (split_dot2.hip: low halves wrong)
  unsigned m;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(m));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), __builtin_bit_cast(bf16x2, m), a, false);
#else
  return a - __builtin_bit_cast(float, p << 16);
#endif
}
__device__ __forceinline__ float bx_res_hi(float b, unsigned p) {
#if SF_BX_DOT2
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), (bf16x2){(__bf16)0.0f, (__bf16)-1.0f}, b, false);
#else
  return b - __builtin_bit_cast(float, p & 0xffff0000u);
#endif
}
// (a, b) -> element `idx` of the three packed piece vectors (11 vector instructions per pair; 7 in the dot2 form)
template <class V>
__device__ __forceinline__ void split_pair(float a, float b, V (&dst)[3], int idx) {
  const unsigned p1 = cvt_pk_bf16(a, b);
  a = bx_res_lo(a, p1);
  b = bx_res_hi(b, p1);
  const unsigned p2 = cvt_pk_bf16(a, b);
  a = bx_res_lo(a, p2);
  b = bx_res_hi(b, p2);
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
