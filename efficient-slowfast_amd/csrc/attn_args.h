// attn_args.h — argument block shared by the forward attention kernels (attn_flash.hip, attn_small.hip).
#pragma once
#include "common.h"

#include <stdlib.h>
struct SfAttnArgs {
  const float* q; const float* k; const float* v; const float* x;
  const float* gamma; const float* scale; const float* bias;
  float* out;
  float* o_save;    // optional [B, N, C] dense: O = P v (pre-gamma), saved for the backward pass
  float* lse_save;  // optional [B, N]: log2-domain log-sum-exp of each query row
  int q_cs, k_cs, v_cs, x_cs, out_cs, out_coff;
  int B, T, H, W, C, N, alpha, act, nqt;
  // Key-range parts (sf_sweep_parts): with zs > 1 workgroup (b, z, query tile) sweeps only part z of the keys and
  // leaves its UNNORMALISED O^T, running max and denominator in the workspace; attn_fwd_merge_kernel combines the
  // parts (the usual log-sum-exp merge) and applies the epilogue.
  int zs;
  float* part_o;   // [B][zs][N][CP]
  float* part_ml;  // [B][zs][N][2]   (m, l)
  float soft_t;    // attn_fwd_kernel: log2 headroom of the stale softmax reference (set by its launcher)
  float* bx_planes;  // d = 32 with a workspace: room for the bf16 pieces of K and V^T (attn_bx_split_kernel), else null
};

// log2 headroom of the stale softmax reference (SF_ATTN_SOFT_T, default 64: 2^(s - m_ref) stays far inside fp32).
// Clamped to [8, 80]: a row sums up to N <= 2^17 terms of at most 2^t, so t <= 80 keeps the sum below 2^97 (fp32 max
// 2^128); below 8 nearly every tile would refresh.  Unparsable / NaN values fall back to the default.
static inline float sf_attn_soft_t() {
  static const float t = [] {
    const char* e = getenv("SF_ATTN_SOFT_T");
    if (!e) return 64.f;
    char* end = nullptr;
    const float v = strtof(e, &end);
    if (end == e || !(v == v)) return 64.f;
    return v < 8.f ? 8.f : (v > 80.f ? 80.f : v);
  }();
  return t;
}

// SF_ATTN_BX (default 1): the d = 17..32 attention kernels with a workspace run their fp32 products on the bf16 matrix
// pipe (attn_bx.h: three-way operand split, six products — fp32-level results at 6/16 of the f32-MFMA cost);
// 0 = v_mfma_f32_32x32x2_f32.
static inline int sf_attn_bx_level() {
  static const int v = [] {
    const char* e = getenv("SF_ATTN_BX");
    return e ? atoi(e) : 1;
  }();
  return v;
}

// Combine the zs key parts of every query row and run the attention epilogue (attn_flash.hip).
int sf_attn_fwd_merge(const SfAttnArgs& a, int cp, hipStream_t s);

// Place the part buffers in a workspace of sf_attn_fwd_ws_floats(B, N, C) floats.
static inline void sf_attn_place_parts(SfAttnArgs& a, int cp, float* ws) {
  a.part_o = ws;
  a.part_ml = ws + (long)a.B * a.zs * a.N * cp;
}
