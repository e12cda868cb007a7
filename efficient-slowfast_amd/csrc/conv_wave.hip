// conv_wave.hip — dense 3-D convolution as a per-WAVEFRONT implicit GEMM on the fp32 matrix cores (gfx950).
//
//   out[m, n] = act( scale[n] * sum_{tap, c} in[row(m, tap), c] * w[n, tap, c] + bias[n] + res[m, n] )
//
// The f32 MFMA runs at 1/16 of the bf16 rate (v_mfma_f32_16x16x4_f32: 32 cycles per SIMD), so a wavefront that owns a
// (TM*16) x (TN*16) output tile needs only (TM + TN) 16-byte operand fragments per 4*TM*TN MFMAs — 15 loads per 3328
// MFMA cycles at TM = 13, TN = 2 — and the fragment a lane needs IS a contiguous 16-byte run in NDHWC memory (row =
// output position's tap source, 4 consecutive input channels).  So nothing is staged through LDS and the main loop has
// NO barrier: every lane loads its own A / B fragments straight from L2 into registers one K step ahead
// (buffer_load_dwordx4: the conv's zero padding, ragged rows and channel tails are out-of-range offsets that read as
// zero — no branch in the loop), and the four wavefronts of a workgroup are independent until the epilogue.
//
// Work split: a workgroup = 4 wavefronts = 4 / KS output tiles, each tile's K range cut into KS contiguous parts
// (KS = 4: one tile per workgroup, four wavefronts share its reduction — the res4 / res5 layers with few tiles and
// 2304..6144-long reductions; KS = 1: four neighbouring N tiles of the same rows, whose A fragments then hit L1).
// The parts are summed in part order through LDS (no atomics, bit-reproducible), and the same LDS pass transposes the
// accumulators (channel on the lane) into 16-byte row-major stores with scale / bias / residual / ReLU applied.
// The number of rows per M tile is a runtime value <= TM*16: every activation of this network has 49 * 2^k rows, and
// e.g. 196-row tiles (TM = 13) make 256 * j tiles where 128-row tiles leave a quarter of the last round of CUs idle.
//
// Replaces (with conv_igemm.hip as the fallback for odd alignments): stem_helper.py:157-164 (slow pathway),
// resnet_helper.py:182-223, :326-335, video_model_builder.py:128-135, custom_video_model_builder.py:102-108,
// wdf_attention_helper.py:21-29; and, with desc.transposed / the scatter map, their data gradients.
#include "common.h"
#include <stdio.h>
#include <type_traits>
#include <stdlib.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x80000000u;  // byte offset past every buffer this kernel accepts (< 2 GiB each)

struct WaveArgs {
  sf_conv_desc d;
  const float* in;
  const float* w;
  const float* scale;
  const float* bias;
  const float* res;
  float* out;
  int M, ntaps;
  int rows;            // rows per M tile (<= TM * 16)
  int nb_n, ntiles, nwg;
  int nk;              // K steps of 16 = ntaps * cin_pad / 16
  int cpk;             // K steps per tap = cin_pad / 16
  unsigned in_bytes, w_bytes;
  unsigned out_bytes, res_bytes;  // extents of the output / residual buffers (buffer load / store bounds)
  int res_buf;                    // output and residual fit 32-bit buffer addressing: the one-pass epilogue's branch-free form
  int dbg;             // sf_conv_tune(5, mask), microbenchmarks only: 1 = drop the stores, 2 = every A row reads row 0
  int plain;           // 1: 1x1x1 kernel, stride 1, no padding, same extents -> input row == output row
  float* stats;        // != NULL: [part][Cout / 4][count, K, sum(v - K), sum((v - K)^2)][4 channels] of the stored outputs
  // n / d for 0 <= n < 2^31 as (mulhi(n, mul) >> sh); d == 1 has mul == 0 (identity)
  unsigned wo_mul, wo_sh, ho_mul, ho_sh, to_mul, to_sh;
};

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned mul, unsigned sh) {
  return mul ? (__umulhi(n, mul) >> sh) : n;
}

// Bit k set <=> 0 <= x0 + sg*k < n, for k in [0, kn) (dilation 1): the taps of one dimension that fall inside the input.
__device__ __forceinline__ unsigned tap_run(int x0, int sg, int n, int kn) {
  int lo = sg > 0 ? -x0 : x0 - n + 1;
  int hi = sg > 0 ? n - x0 : x0 + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > kn ? kn : hi;
  return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}

template <int TM, int TN, int KS, int KV>
__global__ __launch_bounds__(256) void conv_wave_kernel(const WaveArgs p) {
  constexpr int KB = 16 * KV;                                   // channels per K step: KV 16-byte fragments per lane
  constexpr int BN = TN * 16;
  constexpr int TPW = 4 / KS;                                   // tiles per workgroup
  constexpr int R = (128 / BN) < 1 ? 1 : ((128 / BN) > 4 ? 4 : (128 / BN));  // 16-row slabs per epilogue round
  constexpr int EP = BN + 4;                                    // slab row pitch (dwords): b128 reads conflict-free
  __shared__ __attribute__((aligned(16))) float smem[4 * R * 16 * EP];

  const sf_conv_desc& d = p.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int wg = xcd_remap(blockIdx.x, p.nwg);
  const int tslot = wave / KS;               // which of the workgroup's tiles
  const int ksub = wave - tslot * KS;        // which part of its reduction
  const int tile = wg * TPW + tslot;
  const bool live = tile < p.ntiles;
  const int tile_m = live ? tile / p.nb_n : 0;
  const int tile_n = live ? tile - tile_m * p.nb_n : 0;
  const int m0 = tile_m * p.rows;
  const int n0 = tile_n * BN;
  const int sg = d.transposed ? -1 : 1;
  const bool unit_dil = d.dT == 1 && d.dH == 1 && d.dW == 1;

  // ---- per-row state: byte offset of the (tap 0) source position and the separable validity masks
  unsigned a_base[TM], a_pm[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rr = i * 16 + fr;
    const int m = m0 + rr;
    const bool ok = live && rr < p.rows && m < p.M;
    const unsigned mm = ok ? (unsigned)m : 0u;
    if (p.plain) {  // 1x1x1, stride 1, no padding: source row == output row, the one tap is always in range.  The
      // decode below costs ~300 vector-ALU cycles per row (three magic divisions, 32-bit multiplies) — a quarter of
      // a K = 64 layer's tile on ALUs the fp32 MFMA shares
      a_pm[i] = ok ? (1u | (1u << 10) | (1u << 20)) : 0u;
      a_base[i] = (mm * (unsigned)d.in_cs + (unsigned)(d.in_coff + fg * 4 * KV)) * 4u;
      continue;
    }
    const unsigned q1 = fast_div(mm, p.wo_mul, p.wo_sh);
    const int wo = (int)(mm - q1 * (unsigned)d.Wo);
    const unsigned q2 = fast_div(q1, p.ho_mul, p.ho_sh);
    const int ho = (int)(q1 - q2 * (unsigned)d.Ho);
    const unsigned q3 = fast_div(q2, p.to_mul, p.to_sh);
    const int to = (int)(q2 - q3 * (unsigned)d.To);
    const int n = (int)q3;
    const int t0 = d.transposed ? to + d.pT : to * d.sT - d.pT;
    const int h0 = d.transposed ? ho + d.pH : ho * d.sH - d.pH;
    const int w0 = d.transposed ? wo + d.pW : wo * d.sW - d.pW;
    unsigned pm = 0;
    if (ok) {
      if (unit_dil) {  // the in-range taps of a dimension are a contiguous run: two clamps instead of a loop
        pm = tap_run(t0, sg, d.Ti, d.kT) | (tap_run(h0, sg, d.Hi, d.kH) << 10) | (tap_run(w0, sg, d.Wi, d.kW) << 20);
      } else {
        for (int k = 0; k < d.kT; ++k) pm |= ((unsigned)(t0 + sg * k * d.dT) < (unsigned)d.Ti ? 1u : 0u) << k;
        for (int k = 0; k < d.kH; ++k) pm |= ((unsigned)(h0 + sg * k * d.dH) < (unsigned)d.Hi ? 1u : 0u) << (10 + k);
        for (int k = 0; k < d.kW; ++k) pm |= ((unsigned)(w0 + sg * k * d.dW) < (unsigned)d.Wi ? 1u : 0u) << (20 + k);
      }
    }
    a_pm[i] = pm;
    const int pos = ((n * d.Ti + t0) * d.Hi + h0) * d.Wi + w0;  // may be negative (window starts in the padding)
    a_base[i] = (unsigned)(pos * d.in_cs + d.in_coff + fg * 4 * KV) * 4u;
  }
  unsigned b_base[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) b_base[j] = ((unsigned)(n0 + j * 16 + fr) * (unsigned)(p.nk * KB) + (unsigned)(fg * 4 * KV)) * 4u;

  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, live ? (int)p.w_bytes : 0, 0x00020000);

  // ---- K iteration state (wave-uniform): step `it` = (tap, 16-channel chunk)
  const int it_begin = (int)(((long)p.nk * ksub) / KS);
  const int it_end = (int)(((long)p.nk * (ksub + 1)) / KS);
  int it = it_begin;
  int tap = it / p.cpk;
  int c0 = (it - tap * p.cpk) * KB;
  int kw = tap % d.kW, kh = (tap / d.kW) % d.kH, kt = tap / (d.kW * d.kH);
  unsigned a_cur[TM];  // byte offset of the current tap's source run per row, OOB when the tap falls outside
  auto set_tap = [&]() {
    const unsigned tapoff = (unsigned)(sg * ((kt * d.dT * d.Hi + kh * d.dH) * d.Wi + kw * d.dW) * d.in_cs) * 4u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned v = (a_pm[i] >> kt) & (a_pm[i] >> (10 + kh)) & (a_pm[i] >> (20 + kw)) & 1u;
      a_cur[i] = v ? a_base[i] + tapoff : OOB;
    }
  };
  set_tap();

  // Loads of the step the iteration state (it, c0, a_cur) currently points at.  A lane's fragment is KV * 16
  // contiguous bytes: with KV = 2 the four lanes of a row cover one whole 128-byte line per load pair (with KV = 1
  // every 64-byte half line is fetched into L1 twice, one K step apart — by then evicted).
  auto load_b = [&](f32x4 (&b)[TN][KV]) {
    const unsigned boff = (unsigned)it * (unsigned)(KB * 4);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < KV; ++v)
        b[j][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_base[j] + boff + v * 16, 0, 0));
  };
  auto load_a = [&](int i, f32x4 (&a)[KV]) {
    const unsigned koff = (unsigned)c0 * 4u;     // OOB + koff stays out of range (koff < 2 GiB)
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const bool kin = (c0 + (fg * KV + v) * 4) < d.Cin;  // channel tail of a tap whose Cin is not a multiple of KB
      a[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, kin ? a_cur[i] + koff + v * 16 : OOB, 0, 0));
    }
  };
  auto advance = [&]() {
    ++it;
    c0 += KB;
    if (c0 >= d.cin_pad) {
      c0 = 0;
      ++tap;
      if (++kw == d.kW) {
        kw = 0;
        if (++kh == d.kH) {
          kh = 0;
          ++kt;
        }
      }
      set_tap();
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // ---- main loop.  Row-pair major: the 8*TN MFMAs of rows (i, i+1) consume a[i], a[i+1] completely, and the loads
  // of the NEXT step's a[i], a[i+1] are issued into the same registers right behind them — one register set, every
  // fragment in flight for a whole step (~3000 MFMA cycles) before it is needed; only the TN B fragments are double
  // buffered.  An accumulator is revisited after 2*TN >= 2 other MFMAs (16x16x4 f32: 32-cycle issue, 40-cycle
  // dependent latency).  Loads past the end of the range are harmless (any offset is either valid memory of the
  // buffer or reads zero) and their results are never used.
  {
    f32x4 fa[TM][KV], fb[TN][KV], fbn[TN][KV];
    const int nsteps = it_end - it_begin;
    load_b(fb);
#pragma unroll
    for (int i = 0; i < TM; ++i) load_a(i, fa[i]);
    advance();
    for (int step = 0; step < nsteps; ++step) {
      load_b(fbn);
#pragma unroll
      for (int i2 = 0; i2 < TM; i2 += 2) {
#pragma unroll
        for (int v = 0; v < KV; ++v)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = i2; i < i2 + 2 && i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][v][s], fb[j][v][s], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = i2; i < i2 + 2 && i < TM; ++i) load_a(i, fa[i]);
        __builtin_amdgcn_sched_barrier(0);  // keep the pair's loads right behind its MFMAs (the scheduler would sink them)
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int v = 0; v < KV; ++v) fb[j][v] = fbn[j][v];
      advance();
    }
  }

  // ---- epilogue: R slabs of 16 rows per round through LDS (transpose to row-major + sum of the KS parts in order)
  const bool scatter = d.os_T > 1 || d.os_H > 1 || d.os_W > 1;
  const bool relu = d.act == SF_ACT_RELU || d.act == SF_ACT_RELU6;
  const float hi = d.act == SF_ACT_RELU6 ? 6.f : 3.0e38f;
  constexpr int LPR = BN / 4;                    // lanes (float4s) per row
  constexpr int NV = R * 16 * LPR;               // float4s per round per tile
  constexpr int PER = (NV + KS * 64 - 1) / (KS * 64);
  float* const slab = smem + wave * (R * 16 * EP);
  const float* const slab0 = smem + (tslot * KS) * (R * 16 * EP);
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t res_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.out), 0, (int)p.res_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)p.out_bytes, 0x00020000);
  // Training-mode BN statistics of the outputs, taken where they are stored (removes the statistics kernel's read
  // pass over z): a lane always handles the same 4 channels (c4 = 4 * (lane % LPR)), so it keeps shifted sums
  // sum(v - K), sum((v - K)^2) with K = the tile's first row; lanes, then the tile's KS wavefronts, are combined at
  // the end and ONE row [count, K, S1, S2] per (M tile, channel) goes to p.stats (merged in fp64 by
  // stats_merge_kernel with Chan's formula, so the shift may differ from tile to tile).
  const bool want_stats = p.stats != nullptr;
  f32x4 st_k = {0.f, 0.f, 0.f, 0.f}, st_1 = {0.f, 0.f, 0.f, 0.f}, st_2 = {0.f, 0.f, 0.f, 0.f};
  // a lane handles the same four channels in every element of every round (64 % LPR == 0): its scale / bias once, not a
  // load + s_waitcnt vmcnt(0) per element between the stores
  static_assert(64 % LPR == 0, "lane -> channel group is fixed");
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, bi4 = {0.f, 0.f, 0.f, 0.f};
  {
    const int n = n0 + (lane % LPR) * 4;
    if (p.scale && n < d.Cout) sc4 = *reinterpret_cast<const f32x4*>(p.scale + n);
    if (p.bias && n < d.Cout) bi4 = *reinterpret_cast<const f32x4*>(p.bias + n);
  }
#pragma unroll
  for (int i0 = 0; i0 < TM; i0 += R) {
#pragma unroll
    for (int ii = 0; ii < R; ++ii) {
      if (i0 + ii < TM) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(ii * 16 + fg * 4 + r) * EP + j * 16 + fr] = acc[i0 + ii][j][r];
      }
    }
    __syncthreads();
    if (want_stats && i0 == 0) {  // the tile's first output row (always a valid row of a live tile)
      const int c4 = (lane % LPR) * 4;
      st_k = *reinterpret_cast<const f32x4*>(slab0 + c4);
#pragma unroll
      for (int s = 1; s < KS; ++s) st_k += *reinterpret_cast<const f32x4*>(slab0 + s * (R * 16 * EP) + c4);
      if (p.bias && n0 + c4 < d.Cout) st_k += *reinterpret_cast<const f32x4*>(p.bias + n0 + c4);
    }
    // The round's residual loads FIRST, branch-free (buffer loads: an invalid element's offset is past the buffer and
    // reads zeros), then the slab reads, sums and stores.  With the load inside the per-element `if` (rounds 2-4) every
    // element was load -> s_waitcnt vmcnt(0) -> store: PER memory round trips in a row per round, and vmcnt(0) also
    // waits for the previous element's STORE to be acknowledged — every accumulating data gradient on this kernel paid
    // that (8 per round x 4 rounds per 112 x 64 tile).
    bool okq[PER];
    unsigned ooff[PER], roff[PER];  // byte offsets of the element's float4 in the output / residual buffer, or past them
    long orowq[PER];
    f32x4 resq[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = (q * KS + ksub) * 64 + lane;
      const int row = idx / LPR;
      const int c4 = (idx - row * LPR) * 4;
      const int rr = i0 * 16 + row;
      const int m = m0 + rr;
      const int n = n0 + c4;
      okq[q] = idx < NV && rr < TM * 16 && live && rr < p.rows && m < p.M && n < d.Cout;
      long orow = m;
      if (scatter) {
        const unsigned mu = okq[q] ? (unsigned)m : 0u;
        const unsigned q1 = fast_div(mu, p.wo_mul, p.wo_sh);
        const int wo = (int)mu - (int)q1 * d.Wo;
        const unsigned q2 = fast_div(q1, p.ho_mul, p.ho_sh);
        const int ho = (int)q1 - (int)q2 * d.Ho;
        const unsigned q3 = fast_div(q2, p.to_mul, p.to_sh);
        const int to = (int)q2 - (int)q3 * d.To;
        const int st = d.os_T > 1 ? d.os_T : 1, sh = d.os_H > 1 ? d.os_H : 1, sw = d.os_W > 1 ? d.os_W : 1;
        orow = (((long)q3 * d.ob_T + to * st + d.oo_T) * d.ob_H + ho * sh + d.oo_H) * d.ob_W + wo * sw + d.oo_W;
      }
      orowq[q] = orow;
      ooff[q] = okq[q] ? (unsigned)((orow * d.out_cs + d.out_coff + n) * 4) : 0x80000000u;
      roff[q] = okq[q] ? (unsigned)((orow * d.res_cs + d.res_coff + n) * 4) : 0x80000000u;
      resq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (p.res && p.res_buf) {  // (the uniform tests OUTSIDE the element loop: inside it every load met a merge point —
#pragma unroll                 // and a wait — right behind it)
      for (int q = 0; q < PER; ++q)
        resq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, roff[q], 0, 0));
    } else if (p.res) {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int idx = (q * KS + ksub) * 64 + lane;
        const int n = n0 + (idx - (idx / LPR) * LPR) * 4;
        if (okq[q]) resq[q] = *reinterpret_cast<const f32x4*>(p.res + orowq[q] * d.res_cs + d.res_coff + n);
      }
    }
    if (p.res_buf) {  // straight-line: buffer stores drop what lies past the buffer, so the waits stay counted
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int idx = (q * KS + ksub) * 64 + lane;
        const int row = idx < NV ? idx / LPR : 0;
        const int c4 = idx < NV ? (idx - row * LPR) * 4 : 0;
        const int n = n0 + c4 < d.Cout ? n0 + c4 : 0;
        f32x4 v = *reinterpret_cast<const f32x4*>(slab0 + row * EP + c4);
#pragma unroll
        for (int s = 1; s < KS; ++s) v += *reinterpret_cast<const f32x4*>(slab0 + s * (R * 16 * EP) + row * EP + c4);
        v = v * sc4 + bi4;
        v += resq[q];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), hi);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), out_rs, ooff[q], 0, 0);
        if (want_stats) {
          const f32x4 dv = v - st_k;
          const float keep = okq[q] ? 1.f : 0.f;
          st_1 += dv * keep;
          st_2 += dv * dv * keep;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int idx = (q * KS + ksub) * 64 + lane;
        const int row = idx / LPR;
        const int c4 = (idx - row * LPR) * 4;
        const int n = n0 + c4;
        if (okq[q]) {
          f32x4 v = *reinterpret_cast<const f32x4*>(slab0 + row * EP + c4);
#pragma unroll
          for (int s = 1; s < KS; ++s) v += *reinterpret_cast<const f32x4*>(slab0 + s * (R * 16 * EP) + row * EP + c4);
          const long orow = orowq[q];
          v = v * sc4 + bi4;
          v += resq[q];
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), hi);
          }
          *reinterpret_cast<f32x4*>(p.out + orow * d.out_cs + d.out_coff + n) = v;
          if (want_stats) {
            const f32x4 dv = v - st_k;
            st_1 += dv;
            st_2 += dv * dv;
          }
        }
      }
    }
    if (i0 + R < TM) __syncthreads();
  }
  if (want_stats) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st_1[e] += __shfl_xor(st_1[e], off, 64);
        st_2[e] += __shfl_xor(st_2[e], off, 64);
      }
    __syncthreads();  // the slabs are free now: [wave][LPR][16] = K, S1, S2, count
    float* const red = smem;
    const int left = p.M - m0;
    const float cnt = live ? (float)(left < p.rows ? left : p.rows) : 0.f;
    if (lane < LPR) {
      float* const o = red + (wave * LPR + lane) * 16;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(o) = st_k;
      *reinterpret_cast<f32x4*>(o + 4) = live ? st_1 : zero;
      *reinterpret_cast<f32x4*>(o + 8) = live ? st_2 : zero;
      o[12] = cnt;
    }
    __syncthreads();
    // KS > 1: the tile's KS wavefronts hold disjoint rows of ONE tile (same K).  KS == 1 with a single channel tile
    // (nb_n == 1): the workgroup's 4 wavefronts hold 4 consecutive M tiles of the same channels — they are merged
    // here (shifted to the first tile's K), which leaves a quarter of the rows for the merge kernel.
    const bool whole = KS == 1 && p.nb_n == 1;
    const bool writer = whole ? wave == 0 : ksub == 0;
    const int n = n0 + lane * 4;
    if (writer && lane < LPR && live && n < d.Cout) {
      const float* const b = red + (wave * LPR + lane) * 16;
      const f32x4 k0 = *reinterpret_cast<const f32x4*>(b);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(b + 4);
      f32x4 a2 = *reinterpret_cast<const f32x4*>(b + 8);
      float c = cnt;
      constexpr int NW = KS == 1 ? 4 : KS;
#pragma unroll
      for (int s = 1; s < NW; ++s) {
        const float* const q = b + s * LPR * 16;
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(q + 4);
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(q + 8);
        if (KS > 1) {
          a1 += s1;
          a2 += s2;
        } else if (whole) {
          const float cs = q[12];
          const f32x4 dk = *reinterpret_cast<const f32x4*>(q) - k0;
          if (cs > 0.f) {
            a2 += s2 + 2.f * dk * s1 + cs * dk * dk;
            a1 += s1 + cs * dk;
            c += cs;
          }
        }
      }
      const int part = whole ? (int)blockIdx.x : tile_m;
      float* const o = p.stats + ((long)part * (d.Cout >> 2) + (n >> 2)) * 16;
      *reinterpret_cast<f32x4*>(o) = (f32x4){c, c, c, c};
      *reinterpret_cast<f32x4*>(o + 4) = k0;
      *reinterpret_cast<f32x4*>(o + 8) = a1;
      *reinterpret_cast<f32x4*>(o + 12) = a2;
    }
  }
}

// ---- persistent form (KS = 1) ----------------------------------------------------------------------------------
// The kernel above is one straight pass per workgroup: row decode -> first loads (a full memory latency with nothing to
// do) -> K steps -> LDS transposition -> stores, and equal workgroups march through those phases together.  For the
// short reductions of this network (1x1x1 layers with K = 64 .. 256: 4 .. 16 steps) that fixed part is as long as the
// MFMA phase: 64 -> 256 at 56^2 ran at 0.40 of its own roofline with the MFMA pipe busy 46 % of the time.
// This form changes three things (KS = 1 only: no cross-wavefront sum):
//  * OPERANDS SWAPPED: the weights are the MFMA's A operand and the activations its B operand, so a lane's four
//    accumulator registers are 4 CONSECUTIVE CHANNELS of one output position — exactly a 16-byte NDHWC store.  No LDS
//    transposition, no barrier anywhere in the kernel; scale / bias / residual are per-lane float4 operations.
//  * PERSISTENT wavefronts: a grid of (CUs x occupancy) workgroups whose wavefronts walk the tile list with a fixed
//    stride.  Wavefronts never synchronise, so they drift apart and one's stores overlap another's MFMA phase.
//  * CROSS-TILE PREFETCH: in a tile's LAST K step the row state is switched to the wavefront's next tile, and the loads
//    that step issues "one step ahead" fetch that tile's first fragments — they are in flight during the epilogue.
// BN batch statistics (sf_conv_fwd_stats): a lane owns fixed channels of a tile; shifted sums (K = the tile's first row,
// row_newbcast) over the lane's rows, 16-lane DPP reduction per channel, then the tile is Chan-merged into a RUNNING
// record that lane (j*4+e) of each lane-row keeps for channel (j, e) across all tiles of the wavefront (its channel
// tile never changes: the stride is a multiple of nb_n).  One record per (wavefront, channel) at the end.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, result in every lane
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);   // row_half_mirror
  v += dpp_f<0x140>(v);   // row_mirror
  return v;
}

// PLAIN (1x1x1, stride 1, no padding — every short-K layer): the source row of an output row is that row for the one
// tap, so the per-row state is ONE offset (no validity masks, no tap switching): 2*TM registers less.
template <int TM, int TN, int KV, bool PLAIN>
__global__ __launch_bounds__(256, 2) void conv_wave_p_kernel(const WaveArgs p) {
  constexpr int KB = 16 * KV;
  constexpr int BN = TN * 16;
  const sf_conv_desc& d = p.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int gw = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // this wavefront's slot in the tile walk
  const int stride = (int)gridDim.x * 4;
  const int sg = d.transposed ? -1 : 1;
  const bool unit_dil = d.dT == 1 && d.dH == 1 && d.dW == 1;
  const bool want_stats = p.stats != nullptr;

  constexpr int TS = PLAIN ? 1 : TM;
  unsigned a_base[TS], a_pm[TS], a_cur[TM], b_base[TN];
  int m0 = 0, n0 = 0;
  auto setup = [&](int tile) {  // row state of `tile` (all-OOB when it is past the end)
    const bool live = tile < p.ntiles;
    const int tile_m = live ? tile / p.nb_n : 0;
    const int tile_n = live ? tile - tile_m * p.nb_n : 0;
    m0 = tile_m * p.rows;
    n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rr = i * 16 + fr;
      const int m = m0 + rr;
      const bool ok = live && rr < p.rows && m < p.M;
      const unsigned mm = ok ? (unsigned)m : 0u;
      if constexpr (PLAIN) {
        a_cur[i] = ok ? (((p.dbg & 2) ? (unsigned)fr : mm) * (unsigned)d.in_cs + (unsigned)(d.in_coff + fg * 4 * KV)) * 4u : OOB;
        continue;
      }
      if (p.plain) {
        a_pm[i % TS] = ok ? (1u | (1u << 10) | (1u << 20)) : 0u;
        a_base[i % TS] = (mm * (unsigned)d.in_cs + (unsigned)(d.in_coff + fg * 4 * KV)) * 4u;
        continue;
      }
      const unsigned q1 = fast_div(mm, p.wo_mul, p.wo_sh);
      const int wo = (int)(mm - q1 * (unsigned)d.Wo);
      const unsigned q2 = fast_div(q1, p.ho_mul, p.ho_sh);
      const int ho = (int)(q1 - q2 * (unsigned)d.Ho);
      const unsigned q3 = fast_div(q2, p.to_mul, p.to_sh);
      const int to = (int)(q2 - q3 * (unsigned)d.To);
      const int n = (int)q3;
      const int t0 = d.transposed ? to + d.pT : to * d.sT - d.pT;
      const int h0 = d.transposed ? ho + d.pH : ho * d.sH - d.pH;
      const int w0 = d.transposed ? wo + d.pW : wo * d.sW - d.pW;
      unsigned pm = 0;
      if (ok) {
        if (unit_dil) {
          pm = tap_run(t0, sg, d.Ti, d.kT) | (tap_run(h0, sg, d.Hi, d.kH) << 10) | (tap_run(w0, sg, d.Wi, d.kW) << 20);
        } else {
          for (int k = 0; k < d.kT; ++k) pm |= ((unsigned)(t0 + sg * k * d.dT) < (unsigned)d.Ti ? 1u : 0u) << k;
          for (int k = 0; k < d.kH; ++k) pm |= ((unsigned)(h0 + sg * k * d.dH) < (unsigned)d.Hi ? 1u : 0u) << (10 + k);
          for (int k = 0; k < d.kW; ++k) pm |= ((unsigned)(w0 + sg * k * d.dW) < (unsigned)d.Wi ? 1u : 0u) << (20 + k);
        }
      }
      a_pm[i % TS] = pm;
      const int pos = ((n * d.Ti + t0) * d.Hi + h0) * d.Wi + w0;
      a_base[i % TS] = (unsigned)(pos * d.in_cs + d.in_coff + fg * 4 * KV) * 4u;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
      b_base[j] = live ? ((unsigned)(n0 + j * 16 + fr) * (unsigned)(p.nk * KB) + (unsigned)(fg * 4 * KV)) * 4u : OOB;
  };

  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.out), 0, (int)p.res_bytes, 0x00020000);

  int it = 0, c0 = 0, kw = 0, kh = 0, kt = 0;
  auto set_tap = [&]() {
    if constexpr (PLAIN) return;  // a_cur is the row state
    const unsigned tapoff = (unsigned)(sg * ((kt * d.dT * d.Hi + kh * d.dH) * d.Wi + kw * d.dW) * d.in_cs) * 4u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned v = (a_pm[i % TS] >> kt) & (a_pm[i % TS] >> (10 + kh)) & (a_pm[i % TS] >> (20 + kw)) & 1u;
      a_cur[i] = v ? a_base[i % TS] + tapoff : OOB;
    }
  };
  auto rewind = [&]() { it = 0; c0 = 0; kw = 0; kh = 0; kt = 0; set_tap(); };
  auto load_b = [&](f32x4 (&b)[TN][KV]) {
    const unsigned boff = (unsigned)it * (unsigned)(KB * 4);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < KV; ++v)
        b[j][v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_base[j] + boff + v * 16, 0, 0));
  };
  auto load_a = [&](int i, f32x4 (&a)[KV]) {
    const unsigned koff = (unsigned)c0 * 4u;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const bool kin = (c0 + (fg * KV + v) * 4) < d.Cin;
      a[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, kin ? a_cur[i] + koff + v * 16 : OOB, 0, 0));
    }
  };
  auto advance = [&]() {
    ++it;
    c0 += KB;
    if constexpr (PLAIN) return;  // one tap: c0 runs over its channels
    if (c0 >= d.cin_pad) {
      c0 = 0;
      if (++kw == d.kW) {
        kw = 0;
        if (++kh == d.kH) {
          kh = 0;
          ++kt;
        }
      }
      set_tap();
    }
  };

  // running statistics of this wavefront: lane (fr = j*4 + e) of every lane-row keeps channel n0 + j*16 + 4*fg + e
  float run_n = 0.f, run_k = 0.f, run_1 = 0.f, run_2 = 0.f;
  const int stat_n0 = ((gw % p.nb_n) * BN);  // constant per wavefront (stride % nb_n == 0 when statistics are taken)

  int tile = gw;
  if (tile < p.ntiles) {
    const bool scatter = d.os_T > 1 || d.os_H > 1 || d.os_W > 1;
    const bool relu = d.act == SF_ACT_RELU || d.act == SF_ACT_RELU6;
    const float hi = d.act == SF_ACT_RELU6 ? 6.f : 3.0e38f;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 fa[TM][KV], fb[TN][KV], fbn[TN][KV];
    setup(tile);
    rewind();
    load_b(fb);
#pragma unroll
    for (int i = 0; i < TM; ++i) load_a(i, fa[i]);
    advance();
    for (;;) {
      const int m0c = __builtin_amdgcn_readfirstlane(m0), n0c = __builtin_amdgcn_readfirstlane(n0);
      const int next = tile + stride;
      for (int step = 0; step < p.nk; ++step) {
        if (step == p.nk - 1) {  // everything this step loads "one step ahead" belongs to the next tile
          setup(next);
          rewind();
        }
        load_b(fbn);
#pragma unroll
        for (int i2 = 0; i2 < TM; i2 += 2) {
#pragma unroll
          for (int v = 0; v < KV; ++v)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int i = i2; i < i2 + 2 && i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)  // A = weights (rows = channels), B = activations (columns = positions)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][v][s], fa[i][v][s], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = i2; i < i2 + 2 && i < TM; ++i) load_a(i, fa[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int v = 0; v < KV; ++v) fb[j][v] = fbn[j][v];
        advance();
      }
      // ---- epilogue of the tile at (m0c, n0c): lane = position (i*16 + fr), registers = channels 4*fg .. 4*fg+3 of
      //      channel tile j; the next tile's first fragments are in flight meanwhile.  Branch-free inside a variant
      //      (statistics / residual / activation are compile-time flags of the lambda): invalid rows / channels get
      //      an out-of-range buffer offset (loads read zero, stores are dropped), so the residual loads of a column of
      //      row blocks are issued together instead of one load -> wait -> store per element.
      auto epilogue = [&](auto STATS, auto RES, auto RELU) {
        const int left = p.M - m0c;
        const float cnt = (float)(left < p.rows ? left : p.rows);
        unsigned orow[TM];  // output row of this lane's position in row block i, 0xffffffff when there is none
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int rr = i * 16 + fr;
          const int m = m0c + rr;
          unsigned r = (unsigned)m;
          if (scatter) {
            const unsigned q1 = fast_div((unsigned)m, p.wo_mul, p.wo_sh);
            const int wo = m - (int)q1 * d.Wo;
            const unsigned q2 = fast_div(q1, p.ho_mul, p.ho_sh);
            const int ho = (int)q1 - (int)q2 * d.Ho;
            const unsigned q3 = fast_div(q2, p.to_mul, p.to_sh);
            const int to = (int)q2 - (int)q3 * d.To;
            const int st = d.os_T > 1 ? d.os_T : 1, sh = d.os_H > 1 ? d.os_H : 1, sw = d.os_W > 1 ? d.os_W : 1;
            r = (unsigned)((((int)q3 * d.ob_T + to * st + d.oo_T) * d.ob_H + ho * sh + d.oo_H) * d.ob_W + wo * sw + d.oo_W);
          }
          orow[i] = (rr < p.rows && m < p.M) ? r : 0xffffffffu;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0c + j * 16 + 4 * fg;
          const bool nok = n < d.Cout;
          const int nc = nok ? n : 0;  // clamped: the vectors below are read unconditionally
          const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
          const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + nc) : one;
          const f32x4 bi = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nc) : zero;
          f32x4 st_k = zero, st_1 = zero, st_2 = zero;
          constexpr int CH = RES.value ? (TM > 7 ? 5 : 7) : 1;  // row blocks per batch of residual loads
#pragma unroll
          for (int ib = 0; ib < TM; ib += CH) {
            f32x4 rv[CH];
            if constexpr (RES.value) {
#pragma unroll
              for (int u = 0; u < CH; ++u) {
                if (ib + u < TM) {
                  const bool ok = nok && orow[ib + u] != 0xffffffffu;
                  const unsigned off = ok ? (orow[ib + u] * (unsigned)d.res_cs + (unsigned)(d.res_coff + n)) * 4u : OOB;
                  rv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, off, 0, 0));
                }
              }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
              const int i = ib + u;
              if (i < TM) {
                const bool ok = nok && orow[i] != 0xffffffffu;
                f32x4 v = acc[i][j] * sc + bi;
                acc[i][j] = zero;
                if constexpr (STATS.value) {
                  if (i == 0) {  // K = the tile's first output row (fr == 0), shared by the lane-row
#pragma unroll
                    for (int e = 0; e < 4; ++e) st_k[e] = dpp_f<0x150>(v[e]);  // row_newbcast:0
                  }
                }
                if constexpr (RES.value) v += rv[u];
                if constexpr (RELU.value) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), hi);
                }
                const unsigned off = (ok && !(p.dbg & 1)) ? (orow[i] * (unsigned)d.out_cs + (unsigned)(d.out_coff + n)) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rs, off, 0, 0);
                if constexpr (STATS.value) {
                  const f32x4 dv = v - st_k;
                  const float w = ok ? 1.f : 0.f;
                  st_1 += w * dv;
                  st_2 += w * dv * dv;
                }
              }
            }
          }
          if constexpr (STATS.value) {  // tile totals of channel tile j -> the running record of lane fr == j*4 + e
            float tk = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float s1 = row16_sum(st_1[e]), s2 = row16_sum(st_2[e]);
              const bool mine = fr == j * 4 + e;
              tk = mine ? st_k[e] : tk;
              t1 = mine ? s1 : t1;
              t2 = mine ? s2 : t2;
            }
            const bool owner = fr >= j * 4 && fr < j * 4 + 4;
            const bool first = run_n == 0.f;
            const float dk = first ? 0.f : tk - run_k;
            const float n2 = run_2 + t2 + 2.f * dk * t1 + cnt * dk * dk;
            const float n1 = run_1 + t1 + cnt * dk;
            run_k = owner && first ? tk : run_k;
            run_2 = owner ? n2 : run_2;
            run_1 = owner ? n1 : run_1;
            run_n = owner ? run_n + cnt : run_n;
          }
        }
      };
      using T_ = std::true_type;
      using F_ = std::false_type;
      if (want_stats) epilogue(T_{}, F_{}, F_{});  // raw conv output (+ bias): no residual, no activation
      else if (p.res) { if (relu) epilogue(F_{}, T_{}, T_{}); else epilogue(F_{}, T_{}, F_{}); }
      else { if (relu) epilogue(F_{}, F_{}, T_{}); else epilogue(F_{}, F_{}, F_{}); }
      if (next >= p.ntiles) break;
      tile = next;
    }
  }
  if (want_stats && fr < TN * 4) {
    const int j = fr >> 2, e = fr & 3;
    const int n = stat_n0 + j * 16 + 4 * fg;
    if (n < d.Cout) {
      float* const o = p.stats + ((long)(gw / p.nb_n) * (d.Cout >> 2) + (n >> 2)) * 16;
      o[e] = run_n;
      o[4 + e] = run_k;
      o[8 + e] = run_1;
      o[12 + e] = run_2;
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------
static void magic(unsigned dv, unsigned* mul, unsigned* sh) {
  if (dv <= 1) { *mul = 0; *sh = 0; return; }
  unsigned l = 0;
  while ((1u << l) < dv) ++l;                       // l = ceil(log2 d) >= 1
  const unsigned long long num = 1ull << (31 + l);
  *mul = (unsigned)((num + dv - 1) / dv);           // < 2^32 because d > 2^(l-1)
  *sh = l - 1;                                      // n / d = (n * mul) >> (31 + l) = mulhi(n, mul) >> (l - 1)
}

struct WaveCfg { int tm, tn, ks, occ; };  // occ: workgroups a CU holds at once (registers)
static const WaveCfg CFGS[] = {{13, 2, 4, 2}, {13, 2, 1, 2}, {7, 4, 4, 2}, {7, 4, 1, 2},
                               {7, 2, 4, 4},  {7, 2, 1, 4},  {13, 1, 4, 3}, {13, 1, 1, 3}};
constexpr int NCFG = (int)(sizeof(CFGS) / sizeof(CFGS[0]));

int g_force_cfg = -1;    // sf_conv_tune(1, c): force configuration c (microbenchmarks); -1 = planner
int g_force_rows = 0;    // sf_conv_tune(2, r): force rows per M tile
int g_enable = 1;        // sf_conv_tune(0, e): 0 = never take the wave path
int g_k32 = 0;           // sf_conv_tune(3, 1): 32-channel K steps where the packed rows allow (measured slower)

int g_dbg = 0;           // sf_conv_tune(5, mask): persistent kernel ablations (microbenchmarks)
int g_persist = 1;       // sf_conv_tune(4, e): 0 = KS == 1 layers on the one-pass kernel (A/B runs)

// 0: never; 1: every KS == 1 layer the instantiations cover; 2: plain (1x1x1) layers only; 3 (default): plain layers
// with at most 4 K steps
static int persist_level() {
  static const int env_on = [] {
    const char* e = getenv("SF_CONV_WAVE_P");
    return e ? atoi(e) : 3;
  }();
  return g_persist >= 10 ? g_persist - 10 : (g_persist ? env_on : 0);
}

// launch == false: only report the workgroups of this instantiation a CU holds (registers), asked of the runtime once
template <int TM, int TN, bool PLAIN>
static int wave_p(const WaveArgs& a, int nwg, hipStream_t s, bool launch) {
  if (!launch) {
    static const int occ = [] {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_wave_p_kernel<TM, TN, 1, PLAIN>, 256, 0) != hipSuccess || n < 1)
        n = 1;
      return n > 4 ? 4 : n;
    }();
    return occ;
  }
  hipLaunchKernelGGL((conv_wave_p_kernel<TM, TN, 1, PLAIN>), dim3(nwg), dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

static int wave_p_dispatch(int best, const WaveArgs& a, int nwg, hipStream_t s, bool launch) {
  // (the 13x2 and 7x4 tiles are not instantiated in the persistent form at all: at 256 registers they spill — 16 B and
  //  188 B of scratch, tools/isa_audit.py — and the caller keeps those picks on the one-pass kernel)
  if (a.plain) {
    switch (best) {
      case 5: return wave_p<7, 2, true>(a, nwg, s, launch);
      default: return wave_p<13, 1, true>(a, nwg, s, launch);
    }
  }
  switch (best) {  // the 7x4 / 13x2 tiles have no room for a second row-state set: one-pass kernel
    case 5: return wave_p<7, 2, false>(a, nwg, s, launch);
    default: return wave_p<13, 1, false>(a, nwg, s, launch);
  }
}

template <int TM, int TN, int KS>
static int launch_wave(const WaveArgs& a, int kv, hipStream_t s) {
  if (kv == 2)
    hipLaunchKernelGGL((conv_wave_kernel<TM, TN, KS, 2>), dim3(a.nwg), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_wave_kernel<TM, TN, KS, 1>), dim3(a.nwg), dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// Time model of one configuration in shader cycles (tools/plan/fit_plan.py holds the same formula and the fit against
// tools/microbench/conv_wave_bench.py on MI355X: the planner's picks sum to within 3 % of the best configuration per
// layer).  The busiest CU runs `rounds` workgroups; a workgroup's wavefronts issue steps * 4*TM*TN MFMAs of 32 cycles
// while the CU's load path moves (TM + TN) 1 KiB fragment loads per wavefront and step at ~60 cycles each; prologue
// and epilogue are a fixed cost per set of co-resident workgroups; the layer cannot beat its HBM bytes, and the
// epilogue's stores arrive as one burst when every workgroup finishes together.
static double plan_score(const sf_conv_desc* d, long M, int nk, bool has_res, const WaveCfg& c, int* rows_out) {
  // fitted constants (tools/plan/fit_plan.py); SF_PLAN="ld,f0,f1,bw,ep" overrides them for calibration runs
  static const struct PlanP { double ld, f0, f1, bw, ep; } PP = [] {
    PlanP q = {60.3, 2229.0, 381.0, 6.64e6, 0.96};  // refit on profiles/r02b_conv_wave_ab.txt (was 60, 10000, 150, 4.5e6, 0.6)
    if (const char* e = getenv("SF_PLAN")) sscanf(e, "%lf,%lf,%lf,%lf,%lf", &q.ld, &q.f0, &q.f1, &q.bw, &q.ep);
    return q;
  }();
  const double LD = PP.ld, F0 = PP.f0, F1 = PP.f1, BW = PP.bw / 2.1e3 /* bytes per cycle at 2.1 GHz */, EP = PP.ep;
  const int N = d->Cout, ntaps = d->kT * d->kH * d->kW;
  const int nbn = sf_cdiv(N, c.tn * 16);
  const int nbm0 = sf_cdiv(M, c.tm * 16);
  const double out_bytes = (double)M * N * 4.0 * (has_res ? 2 : 1);
  const double in_bytes = (double)M * d->Cin * 4.0 * (ntaps == 1 ? 1.0 : 1.3);
  const double hbm = (out_bytes + in_bytes) / BW;
  double best = 1e30;
  for (int nbm = nbm0; nbm < nbm0 + 48; ++nbm) {
    const int rows = sf_cdiv(M, nbm);
    const long tiles = (long)nbm * nbn;
    const long wgs = (tiles + (4 / c.ks) - 1) / (4 / c.ks);
    const long rounds = (wgs + 255) / 256;
    const int steps = (nk + c.ks - 1) / c.ks;
    const double mfma = (double)steps * c.tm * c.tn * 128.0;
    const double load = (double)steps * (c.tm + c.tn) * LD * 4.0;
    const double body = (double)rounds * (mfma > load ? mfma : load);
    const double fixed = (double)((rounds + c.occ - 1) / c.occ) * (F0 + F1 * c.tm * c.tn);
    const double t = (body + fixed > hbm ? body + fixed : hbm) + out_bytes / BW * EP;
    if (t < best) { best = t; *rows_out = rows; }
  }
  return best;
}

}  // namespace

int sf_wgrad_wave_tune(int knob, int value);  // conv_wgrad_wave.hip (knobs 10..)
int sf_conv_small_tune(int value);            // conv_small.hip
int sf_conv_bx_tune(int value);               // conv_bx.hip
int sf_conv_bx_dbg(int value);                // conv_bx.hip
int sf_conv_bxw_tune(int value);              // conv_bx.hip
int sf_wgrad_rows_tune(int value);            // conv_wgrad_rows.hip
int sf_conv_pw_tune(int value);               // conv_bx.hip
int sf_conv_rows_tune(int value);             // conv_rows.hip
int sf_wgrad_tring_tune(int value);           // conv_wgrad_rows.hip

// Runtime knobs for microbenchmarks / A-B runs (not used by the model code).
int sf_dwm_tune(int value);  // dwconv_march.hip
int sf_dwm2_tune(int value);

extern "C" int sf_conv_tune(int knob, int value) {
  if (knob == 30) return sf_dwm_tune(value);
  if (knob == 31) return sf_dwm2_tune(value);
  if (knob == 20) return sf_wgrad_rows_tune(value);
  if (knob == 21) return sf_conv_pw_tune(value);
  if (knob == 22) return sf_conv_rows_tune(value);
  if (knob == 23) return sf_wgrad_tring_tune(value);
  if (knob >= 10) return sf_wgrad_wave_tune(knob, value);
  if (knob == 0) g_enable = value;
  else if (knob == 1) g_force_cfg = value;
  else if (knob == 2) g_force_rows = value;
  else if (knob == 3) g_k32 = value;
  else if (knob == 4) g_persist = value;
  else if (knob == 5) g_dbg = value;
  else if (knob == 6) return sf_conv_small_tune(value);
  else if (knob == 7) return sf_conv_bx_tune(value);
  else if (knob == 8) return sf_conv_bx_dbg(value);
  else if (knob == 9) return sf_conv_bxw_tune(value);
  else return SF_EINVAL;
  return SF_OK;
}

static bool wave_enabled() {
  static const int env_on = [] {
    const char* e = getenv("SF_CONV_WAVE");
    return e ? atoi(e) : 1;
  }();
  return env_on && g_enable;
}

// Shape-only part of the decision (pointer alignment is checked at launch): used to size workspaces.
int sf_conv_wave_takes(const sf_conv_desc* d) {
  if (!wave_enabled()) return 0;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if ((d->Cin % 4) || (d->in_cs % 4) || (d->in_coff % 4)) return 0;
  if (d->out_cmul != 1 || (d->Cout % 4) || (d->out_cs % 4) || (d->out_coff % 4)) return 0;
  if (d->kT > 10 || d->kH > 10 || d->kW > 10) return 0;
  if (d->transposed && (d->sT != 1 || d->sH != 1 || d->sW != 1)) return 0;
  if (M <= 16) return 0;  // head / squeeze-excite rows: the GEMV kernel
  if ((long)d->N * d->Ti * d->Hi * d->Wi * d->in_cs * 4 > 0x7ffffff0L) return 0;
  if ((long)d->Cout * d->kT * d->kH * d->kW * d->cin_pad * 4 > 0x7ffffff0L) return 0;
  return 1;
}

// Upper bound of the M tiles any configuration makes of M rows (rows per tile >= M / (ceil(M / 112) + 47)).
// (the persistent form leaves one record per wavefront of its grid and channel tile: at most 256 CUs x 4 workgroups x 4)
long sf_conv_wave_max_parts(long M) {
  const long tiles = (M + 111) / 112 + 48;
  return tiles > 4096 ? tiles : 4096;
}

// Returns 1 when the shape is not taken (the caller falls through to conv_igemm), else SF_OK / an error code.
int sf_conv_wave_try(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                     const float* bias, const float* res, float* out, hipStream_t stream, float* stats,
                     int* stat_parts) {
  if (stat_parts) *stat_parts = 0;
  if (!sf_conv_wave_takes(d)) return 1;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  const int ntaps = d->kT * d->kH * d->kW;
  const bool vec_in = (d->Cin % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && sf_aligned16(in);
  const bool vec_out = (d->out_cmul == 1) && (d->Cout % 4 == 0) && (d->out_cs % 4 == 0) && (d->out_coff % 4 == 0) &&
                       sf_aligned16(out) && (!scale || sf_aligned16(scale)) && (!bias || sf_aligned16(bias)) &&
                       (!res || ((d->res_cs % 4 == 0) && (d->res_coff % 4 == 0) && sf_aligned16(res)));
  if (!vec_in || !vec_out) return 1;
  const long in_bytes = (long)d->N * d->Ti * d->Hi * d->Wi * d->in_cs * 4;
  const long w_bytes = (long)d->Cout * ntaps * d->cin_pad * 4;
  WaveArgs a;
  a.d = *d;
  a.in = in; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.M = (int)M;
  a.ntaps = ntaps;
  // 16-channel K steps; sf_conv_tune(3, 1) selects 32 (whole 128-byte lines per row and load pair) where the packed
  // rows allow — measured slower on every layer of cfg #3 (one wavefront per SIMD at 256+ registers)
  const int kv = (d->cin_pad % 32 == 0 && g_k32) ? 2 : 1;
  a.cpk = d->cin_pad / (16 * kv);
  a.nk = ntaps * a.cpk;
  a.in_bytes = (unsigned)in_bytes;
  a.w_bytes = (unsigned)w_bytes;
  magic((unsigned)d->Wo, &a.wo_mul, &a.wo_sh);
  magic((unsigned)d->Ho, &a.ho_mul, &a.ho_sh);
  magic((unsigned)d->To, &a.to_mul, &a.to_sh);
  int best = 0, best_rows = 0;
  if (g_force_cfg >= 0 && g_force_cfg < NCFG) {
    best = g_force_cfg;
    plan_score(d, M, a.nk * kv, res != nullptr, CFGS[best], &best_rows);
  } else {
    double bs = 1e30;
    for (int c = 0; c < NCFG; ++c) {
      if (CFGS[c].tn * 16 > ((d->Cout + 15) / 16) * 16 && CFGS[c].tn > 1) continue;  // tile wider than the layer
      int rows = 0;
      const double s = plan_score(d, M, a.nk * kv, res != nullptr, CFGS[c], &rows);
      if (s < bs) { bs = s; best = c; best_rows = rows; }
    }
  }
  const WaveCfg& c = CFGS[best];
  // the microbenchmark knob (sf_conv_tune(2, rows)) must not shrink tiles below what sf_conv_wave_max_parts sized the
  // statistics workspace for: ignored when this launch leaves statistics
  if (g_force_rows > 0 && g_force_rows <= c.tm * 16 && !(stats && stat_parts)) best_rows = g_force_rows;
  a.rows = best_rows;
  a.nb_n = sf_cdiv(d->Cout, c.tn * 16);
  a.ntiles = sf_cdiv(M, a.rows) * a.nb_n;
  a.nwg = sf_cdiv(a.ntiles, 4 / c.ks);
  a.plain = (d->kT == 1 && d->kH == 1 && d->kW == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 0 && d->pH == 0 &&
             d->pW == 0 && d->Ti == d->To && d->Hi == d->Ho && d->Wi == d->Wo) ? 1 : 0;
  // statistics of the stored outputs: only for the plain epilogue (raw conv output + bias), dense stores
  const bool scatter = d->os_T > 1 || d->os_H > 1 || d->os_W > 1;
  a.stats = (stats && stat_parts && !scale && !res && d->act == SF_ACT_NONE && !scatter && sf_aligned16(stats) &&
             d->Cout % 4 == 0)
                ? stats : nullptr;
  // persistent form: every plain (1x1x1) layer, and the other KS == 1 layers on the tiles whose register budget has
  // room for a second row-state set (13x2 stays on the one-pass kernel there)
  const long rows_out = scatter ? (long)d->N * d->ob_T * d->ob_H * d->ob_W : M;
  const long out_b = rows_out * d->out_cs * 4, res_b = res ? rows_out * d->res_cs * 4 : 0;
  a.dbg = g_dbg;
  a.out_bytes = (unsigned)out_b;
  a.res_bytes = (unsigned)res_b;
  a.res_buf = (out_b > 0 && out_b < 0x7ffffff0L && res_b < 0x7ffffff0L) ? 1 : 0;
  // Where the persistent form pays (tools/microbench/conv_pw_probe.py, MI355X): plain layers with at most 4 K steps
  // (K <= 64: 64 -> 256 at 56^2 79 us on 7x2 tiles against 86-94 us one-pass on any tile; the C <= 32 Fast-pathway
  // projections 10-25 % faster) — a tile there is ~7 k MFMA cycles, as long as the one-pass kernel's fixed part.  From
  // 8 K steps on the two forms are within 3 % of each other and the planner's picks stay on the one-pass kernel.
  // SF_CONV_WAVE_P=1 sends every KS == 1 layer the instantiations cover to the persistent form (A/B runs).
  const bool can_p = kv == 1 && persist_level() > 0 && out_b < 0x7ffffff0L && res_b < 0x7ffffff0L;
  const bool short_k = a.plain && a.nk <= 5;
  if (can_p && short_k && g_force_cfg < 0 && d->Cout >= 32) {
    best = 5;  // 7x2 tiles: 3 workgroups per CU at 157 registers
    plan_score(d, M, a.nk * kv, res != nullptr, CFGS[best], &best_rows);
    a.rows = best_rows;
    a.nb_n = sf_cdiv(d->Cout, CFGS[best].tn * 16);
    a.ntiles = sf_cdiv(M, a.rows) * a.nb_n;
    a.nwg = sf_cdiv(a.ntiles, 4);
  }
  const WaveCfg& cp = CFGS[best];
  const bool take_p = persist_level() == 1 ? true : (persist_level() == 2 ? a.plain : short_k);
  if (cp.ks == 1 && can_p && take_p && best != 1 && best != 3) {
    const int occ = wave_p_dispatch(best, a, 0, stream, false);
    int nwg = a.nwg < 256 * occ ? a.nwg : 256 * occ;
    if (a.stats) {
      const int g4 = (a.nb_n % 4 == 0) ? 4 : ((a.nb_n % 2 == 0) ? 2 : 1);
      const int unit = a.nb_n / g4;
      nwg = (nwg + unit - 1) / unit * unit;
      *stat_parts = nwg * 4 / a.nb_n;
      if (*stat_parts > sf_conv_wave_max_parts(M)) { a.stats = nullptr; *stat_parts = 0; }
    }
    return wave_p_dispatch(best, a, nwg, stream, true);
  }
  if (a.stats) *stat_parts = (c.ks == 1 && a.nb_n == 1) ? a.nwg : sf_cdiv(M, a.rows);
  switch (best) {
    case 0: return launch_wave<13, 2, 4>(a, kv, stream);
    case 1: return launch_wave<13, 2, 1>(a, kv, stream);
    case 2: return launch_wave<7, 4, 4>(a, kv, stream);
    case 3: return launch_wave<7, 4, 1>(a, kv, stream);
    case 4: return launch_wave<7, 2, 4>(a, kv, stream);
    case 5: return launch_wave<7, 2, 1>(a, kv, stream);
    case 6: return launch_wave<13, 1, 4>(a, kv, stream);
    default: return launch_wave<13, 1, 1>(a, kv, stream);
  }
}
