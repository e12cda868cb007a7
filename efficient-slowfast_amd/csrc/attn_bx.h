// attn_bx.h — fp32 attention products on the bf16 matrix pipe: operand split, plane layout, shared by the forward
// (attn_flash.hip) and backward (attn_bwd.hip) kernels.
#pragma once
#include "bx.h"

// ---- split planes (made once per call by sf_attn_bx_split, read by every sweeping workgroup) -------------------------
// row planes    [B][piece][n64 * 64 rows][32 channels] bf16   (rows >= N and channels >= C are zero)
// column planes [B][piece][n64 tiles][32 channels][64 row positions] bf16.  Registers 8s..8s+7 of an accumulator tile
// whose rows are this tile's rows are rows 16s + 8(j>>2) + 4h + (j&3) of a 32-row block (h = lane half), so the rows of
// a block are stored in exactly that order: position 16s + 8h + j.  A lane's fragment of either kind is one 16-byte read.
constexpr int BX_KT = 64;       // rows per tile
constexpr int BX_KP = 32 + 8;   // row-plane pitch in LDS, bf16 (80 B: ds_read_b128 conflict-free)
__device__ __forceinline__ int bx_key_of_pos(int pos) {  // position within a 64-row tile -> row within the tile
  const int j = pos & 7, h = (pos >> 3) & 1, s16 = pos >> 4;  // s16 = 2 * block + s
  return 16 * s16 + 8 * (j >> 2) + 4 * h + (j & 3);
}
// src [B, N, cs] fp32 (first C <= 32 channels), times *mul if mul (a device scalar) else 1: writes the row planes
// and / or the column planes (either may be null).  16-byte aligned src rows (cs % 4 == 0).
int sf_attn_bx_split(const float* src, int cs, const float* mul, int B, int N, int C, unsigned short* rows,
                     unsigned short* cols, hipStream_t s);
static inline long sf_attn_bx_plane_elems(int B, int N) { return (long)B * 3 * sf_cdiv(N, BX_KT) * BX_KT * 32; }

// ---- packed planes, C <= 8 ---------------------------------------------------------------------------------------------
// With at most 8 channels the three pieces of a row fit ONE 32-column row  [p1 ch0..7 | p2 ch0..7 | p3 ch0..7 | 0 x 8]:
//   * a product over the CHANNELS (S = Q K^T, dP = dO V^T; 8 of the MFMA's 16 k slots per lane half) takes two piece
//     pairs per instruction — A = [x|y] (lane half h reads piece x or y of its row: a 16-byte offset), B = [u|v] —
//     so [a1|a1][b1|b2], [a2|a2][b1|b2], [a1|a3][b3|b1] are the six kept terms in THREE MFMAs;
//   * a product whose OUTPUT is the channels (O^T = V^T P^T, dV^T, dK^T, dQ) takes the packed row as its 32 output
//     rows / columns: one MFMA per piece of the other operand yields that piece times all three of this one in row
//     groups 0-7 / 8-15 / 16-23, summed at the end (all nine terms) — three MFMAs per k-step instead of six.
// packed rows    [B][n64 * 64 rows][32]                 (one plane)
// packed columns [B][n64 tiles][32 packed channels][64 row positions]
int sf_attn_bx_split_packed(const float* src, int cs, const float* mul, int B, int N, int C, unsigned short* rows,
                            unsigned short* cols, hipStream_t s);
static inline long sf_attn_bx_packed_elems(int B, int N) { return (long)B * sf_cdiv(N, BX_KT) * BX_KT * 32; }
