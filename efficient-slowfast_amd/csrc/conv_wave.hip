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
  // Training-mode BN statistics of the outputs, taken where they are stored (removes the statistics kernel's read
  // pass over z): a lane always handles the same 4 channels (c4 = 4 * (lane % LPR)), so it keeps shifted sums
  // sum(v - K), sum((v - K)^2) with K = the tile's first row; lanes, then the tile's KS wavefronts, are combined at
  // the end and ONE row [count, K, S1, S2] per (M tile, channel) goes to p.stats (merged in fp64 by
  // stats_merge_kernel with Chan's formula, so the shift may differ from tile to tile).
  const bool want_stats = p.stats != nullptr;
  f32x4 st_k = {0.f, 0.f, 0.f, 0.f}, st_1 = {0.f, 0.f, 0.f, 0.f}, st_2 = {0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = (q * KS + ksub) * 64 + lane;
      const int row = idx / LPR;
      const int c4 = (idx - row * LPR) * 4;
      const int rr = i0 * 16 + row;
      const int m = m0 + rr;
      const int n = n0 + c4;
      if (idx < NV && rr < TM * 16 && live && rr < p.rows && m < p.M && n < d.Cout) {
        f32x4 v = *reinterpret_cast<const f32x4*>(slab0 + row * EP + c4);
#pragma unroll
        for (int s = 1; s < KS; ++s) v += *reinterpret_cast<const f32x4*>(slab0 + s * (R * 16 * EP) + row * EP + c4);
        long orow = m;
        if (scatter) {
          const unsigned q1 = fast_div((unsigned)m, p.wo_mul, p.wo_sh);
          const int wo = m - (int)q1 * d.Wo;
          const unsigned q2 = fast_div(q1, p.ho_mul, p.ho_sh);
          const int ho = (int)q1 - (int)q2 * d.Ho;
          const unsigned q3 = fast_div(q2, p.to_mul, p.to_sh);
          const int to = (int)q2 - (int)q3 * d.To;
          const int st = d.os_T > 1 ? d.os_T : 1, sh = d.os_H > 1 ? d.os_H : 1, sw = d.os_W > 1 ? d.os_W : 1;
          orow = (((long)q3 * d.ob_T + to * st + d.oo_T) * d.ob_H + ho * sh + d.oo_H) * d.ob_W + wo * sw + d.oo_W;
        }
        if (p.scale) v *= *reinterpret_cast<const f32x4*>(p.scale + n);
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + orow * d.res_cs + d.res_coff + n);
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

// Runtime knobs for microbenchmarks / A-B runs (not used by the model code).
extern "C" int sf_conv_tune(int knob, int value) {
  if (knob >= 10) return sf_wgrad_wave_tune(knob, value);
  if (knob == 0) g_enable = value;
  else if (knob == 1) g_force_cfg = value;
  else if (knob == 2) g_force_rows = value;
  else if (knob == 3) g_k32 = value;
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
long sf_conv_wave_max_parts(long M) { return (M + 111) / 112 + 48; }

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
