// conv_igemm.hip — dense 3-D convolution as an implicit GEMM on the fp32 matrix cores (gfx950).
//
//   out[m, n] = act( scale[n] * sum_{tap, c} in[row(m, tap), c] * w[n, tap, c] + bias[n] + res[m, n] )
//
// m runs over the B*To*Ho*Wo output positions (NDHWC), n over output channels, the K dimension over
// (kT*kH*kW taps) x (input channels).  In NDHWC every tap of every output position is a CONTIGUOUS run
// of Cin floats, so the A operand is gathered row-by-row with 16-byte loads and never materialised
// (no im2col buffer); the zero padding of the convolution is a per-(row, tap) predicate.
//
// Tiling: 256 threads = 4 wavefronts; block tile BM x BN, K step 16.  A and B tiles are staged
// global -> registers -> LDS as [row][16 k] with a 24-dword row pitch (conflict-free ds_read_b128),
// double buffered, one barrier per K step; the next tile's global loads are issued before the MFMAs
// of the current one (guide T14).  Each lane reads ONE 16-byte fragment per 16-row tile per K step:
// lane quarter g = lane>>4 takes k = 4g..4g+3 and feeds them to four v_mfma_f32_16x16x4_f32 — the k
// order inside a step is permuted identically for A and B, which a sum over k does not care about.
// Accumulators: (BM/WM/16) x (BN/WN/16) tiles of 4 VGPRs.  C/D map: col = lane&15, row = 4*(lane>>4)+reg.
//
// Replaces: stem_helper.py:157-164, resnet_helper.py:182-223, :326-335, video_model_builder.py:128-135,
// custom_video_model_builder.py:102-108, wdf_attention_helper.py:21-29, head_helper.py:181 (as 1x1x1).
#include "common.h"
#include <stdlib.h>

namespace {

struct ConvArgs {
  sf_conv_desc d;
  const float* in;
  const float* w;
  const float* scale;
  const float* bias;
  const float* res;
  float* out;
  int M;        // B*To*Ho*Wo
  int ntaps;    // kT*kH*kW
  int nb_n;     // number of N tiles
  int nblocks;  // total tiles
  int vec_epi;  // 1: outputs/residual are 16-byte addressable -> LDS-transposed float4 epilogue
  int ksplit;   // > 1: blockIdx.y takes a contiguous share of the K steps and stores a raw partial tile to ws
  float* ws;    // [ksplit][M][Cout] partials (split-K), finished by conv_splitk_finish_kernel
  // grouped convolution (sf_conv_fwd_grouped): blockIdx.z = group; d.Cin / d.Cout are ONE group's widths and the group
  // index moves the channel windows and the weight / scale / bias rows.  All zero for a dense launch (gridDim.z == 1).
  int g_in = 0;     // input channels per group  (in_coff  += g * g_in)
  int g_out = 0;    // output channel step       (out_coff += g * g_out: Cout per group, or 1 for a shuffled store)
  int g_res = 0;    // residual channel step     (res_coff += g * g_res)
  int g_sb = 0;     // scale / bias rows per group
  long g_w = 0;     // packed weight floats per group = Cout_g * taps * cin_pad
  int groups = 1;   // gridDim.z
};

constexpr int BK = 16;
constexpr int LDK = 24;  // LDS row pitch in dwords (16 data + 8 pad): ds_read_b128 conflict-free

template <int BM, int BN, int WM, int WN, int VEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
  static_assert(WM * WN == 4, "4 wavefronts per workgroup");
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int A_IT = (BM + 63) / 64;  // BM = 112 (7 x 16 rows): the second pass stages rows 64..111 only
  constexpr int B_IT = (BN + 63) / 64;
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "whole 16x16 MFMA tiles per wavefront");

  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];
  float* const As = smem;
  float* const Bs = smem + 2 * BM * LDK;

  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, p.nblocks);
  const int tile_m = bid / p.nb_n;
  const int tile_n = bid - tile_m * p.nb_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int grp = (int)blockIdx.z;   // 0 for a dense conv
  const int g_in_coff = d.in_coff + grp * p.g_in;
  const float* const g_w = p.w + (long)grp * p.g_w;

  // ---- staging role: thread owns float4 column lc of rows lr + 64*i
  const int lr = tid >> 2;
  const int lc = (tid & 3) * 4;
  int a_n[A_IT], a_t0[A_IT], a_h0[A_IT], a_w0[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + lr + 64 * i;
    a_ok[i] = (BM % 64 == 0 || lr + 64 * i < BM) && m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int wo = mm % d.Wo;
    const int t1 = mm / d.Wo;
    const int ho = t1 % d.Ho;
    const int t2 = t1 / d.Ho;
    const int to = t2 % d.To;
    a_n[i] = t2 / d.To;
    // transposed (data-gradient) mode: row = input position of the forward conv, the tap's source row is
    // the forward OUTPUT position (t + pT - kt*dT) / sT when that division is exact (see set_tap)
    a_t0[i] = d.transposed ? to + d.pT : to * d.sT - d.pT;
    a_h0[i] = d.transposed ? ho + d.pH : ho * d.sH - d.pH;
    a_w0[i] = d.transposed ? wo + d.pW : wo * d.sW - d.pW;
  }
  const long kpad = (long)p.ntaps * d.cin_pad;
  const float* b_ptr[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int rn = lr + 64 * j;
    const int n = n0 + rn;
    b_ok[j] = (rn < BN) && (n < d.Cout);
    b_ptr[j] = g_w + (long)(b_ok[j] ? n : 0) * kpad + lc;
  }

  // ---- K iteration state: tap (kt,kh,kw) outer, 16-channel chunk inner; a split-K workgroup starts at its share
  const int nk_all = p.ntaps * (d.cin_pad / BK);
  const int nk_per = (nk_all + p.ksplit - 1) / p.ksplit;
  const int it_begin = (int)blockIdx.y * nk_per;
  const int it_end = (it_begin + nk_per < nk_all) ? it_begin + nk_per : nk_all;
  int tap = it_begin / (d.cin_pad / BK);
  int c0 = (it_begin - tap * (d.cin_pad / BK)) * BK;
  int kw = tap % d.kW, kh = (tap / d.kW) % d.kH, kt = tap / (d.kW * d.kH);
  long a_off[A_IT];
  bool a_v[A_IT];
  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int ti, hi, wi;
      bool ok = a_ok[i];
      if (d.transposed) {
        ti = a_t0[i] - kt * d.dT;
        hi = a_h0[i] - kh * d.dH;
        wi = a_w0[i] - kw * d.dW;
        ok = ok && ti >= 0 && hi >= 0 && wi >= 0;
        if (d.sT > 1) { ok = ok && (ti % d.sT) == 0; ti /= d.sT; }
        if (d.sH > 1) { ok = ok && (hi % d.sH) == 0; hi /= d.sH; }
        if (d.sW > 1) { ok = ok && (wi % d.sW) == 0; wi /= d.sW; }
      } else {
        ti = a_t0[i] + kt * d.dT;
        hi = a_h0[i] + kh * d.dH;
        wi = a_w0[i] + kw * d.dW;
      }
      a_v[i] = ok && (unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi &&
               (unsigned)wi < (unsigned)d.Wi;
      a_off[i] = ((((long)a_n[i] * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.in_cs + g_in_coff + lc;
    }
  };
  set_tap();

  f32x4 ra[A_IT], rb[B_IT];
  auto load_global = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (VEC == 4) {
        if (a_v[i] && (c0 + lc) < d.Cin) v = *reinterpret_cast<const f32x4*>(p.in + a_off[i] + c0);
      } else {
        if (a_v[i]) {
          const float* s = p.in + a_off[i] + c0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((c0 + lc + e) < d.Cin) v[e] = s[e];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ok[j]) v = *reinterpret_cast<const f32x4*>(b_ptr[j] + (long)tap * d.cin_pad + c0);
      rb[j] = v;
    }
  };
  auto advance = [&]() {
    c0 += BK;
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
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (BM % 64 == 0 || lr + 64 * i < BM) *reinterpret_cast<f32x4*>(As + (buf * BM + lr + 64 * i) * LDK + lc) = ra[i];
#pragma unroll
    for (int j = 0; j < B_IT; ++j)
      if (lr + 64 * j < BN) *reinterpret_cast<f32x4*>(Bs + (buf * BN + lr + 64 * j) * LDK + lc) = rb[j];
  };

  // ---- compute role
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave - wm * WN;
  const int fr = lane & 15;
  const int fg = lane >> 4;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = it_end - it_begin;  // >= 16 for every split by construction of splitk_factor; 0 is still safe
  if (nk > 0) {
    load_global();
    store_lds(0);
  }
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    const bool more = (it + 1) < nk;
    if (more) {
      advance();
      load_global();
    }
    const float* as = As + (buf * BM + wm * (BM / WM) + fr) * LDK + fg * 4;
    const float* bs = Bs + (buf * BN + wn * (BN / WN) + fr) * LDK + fg * 4;
    f32x4 a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(as + i * 16 * LDK);
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bs + j * 16 * LDK);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    if (more) store_lds(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue
  // Scattered stores (strided data-gradient classes): output position (n,t,h,w) of this launch lands at
  // (t*os_T + oo_T, h*os_H + oo_H, w*os_W + oo_W) of a destination of dims ob_T x ob_H x ob_W.
  const bool split = p.ksplit > 1;  // raw partial tile to the workspace; the finish kernel applies the epilogue
  const float* const e_scale = (split || !p.scale) ? nullptr : p.scale + grp * p.g_sb;
  const float* const e_bias = (split || !p.bias) ? nullptr : p.bias + grp * p.g_sb;
  const float* const e_res = split ? nullptr : p.res;
  const int e_res_coff = d.res_coff + grp * p.g_res;
  float* const e_out = split ? p.ws + (long)blockIdx.y * p.M * d.Cout : p.out;
  const int e_out_cs = split ? d.Cout : d.out_cs, e_out_coff = split ? 0 : d.out_coff + grp * p.g_out;
  const int e_out_cmul = split ? 1 : d.out_cmul;
  const bool scatter = !split && (d.os_T > 1 || d.os_H > 1 || d.os_W > 1);
  auto out_row = [&](int m) -> long {
    if (!scatter) return (long)m;
    const int wo = m % d.Wo;
    const int t1 = m / d.Wo;
    const int ho = t1 % d.Ho;
    const int t2 = t1 / d.Ho;
    const int to = t2 % d.To;
    const int n = t2 / d.To;
    const int st = d.os_T > 1 ? d.os_T : 1, sh = d.os_H > 1 ? d.os_H : 1, sw = d.os_W > 1 ? d.os_W : 1;
    return (((long)n * d.ob_T + to * st + d.oo_T) * d.ob_H + ho * sh + d.oo_H) * d.ob_W + wo * sw + d.oo_W;
  };
  const bool has_res = e_res != nullptr;
  const bool relu = !split && (d.act == SF_ACT_RELU || d.act == SF_ACT_RELU6);
  const float hi = d.act == SF_ACT_RELU6 ? 6.f : 3.0e38f;
  if (split ? (d.Cout % 4 == 0) : p.vec_epi) {
    // The accumulator layout has the channel on the lane (16 lanes = 64 B per row): stored directly, a wave
    // store touches 4 rows x 64 B.  Transpose each 16-row slab through this wave's private LDS region and
    // emit 16-byte stores (and residual loads) that cover whole WN_COLS*4-byte row segments instead.
    constexpr int WN_COLS = TN * 16;
    constexpr int EP = WN_COLS + 4;            // slab row pitch (dwords)
    constexpr int LPR = WN_COLS / 4;           // lanes per row
    constexpr int RPP = 64 / LPR;              // rows per pass
    float* const slab = smem + wave * (16 * EP);
    const int c4 = (lane % LPR) * 4;
    const int n = n0 + wn * (BN / WN) + c4;
    const bool n_ok = n < d.Cout;              // Cout % 4 == 0 on this path
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (n_ok) {
      if (e_scale) sc = *reinterpret_cast<const f32x4*>(e_scale + n);
      if (e_bias) bi = *reinterpret_cast<const f32x4*>(e_bias + n);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(fg * 4 + r) * EP + j * 16 + fr] = acc[i][j][r];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16 / RPP; ++q) {
        const int row = lane / LPR + q * RPP;
        const int m = m0 + wm * (BM / WM) + i * 16 + row;
        if (n_ok && m < p.M) {
          const long orow = out_row(m);
          f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * EP + c4);
          v = v * sc + bi;
          if (has_res) v += *reinterpret_cast<const f32x4*>(e_res + orow * d.res_cs + e_res_coff + n);
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), hi);
          }
          *reinterpret_cast<f32x4*>(e_out + orow * e_out_cs + e_out_coff + n) = v;
        }
      }
      __syncthreads();
    }
    return;
  }
  // scalar epilogue (odd channel counts / channel-multiplier stores): lane holds channel .. + fr, rows 4*fg + reg
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / WN) + j * 16 + fr;
    if (n >= d.Cout) continue;
    const float sc = e_scale ? e_scale[n] : 1.f;
    const float bi = e_bias ? e_bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * (BM / WM) + i * 16 + fg * 4 + r;
        if (m >= p.M) continue;
        const long orow = out_row(m);
        float v = acc[i][j][r] * sc + bi;
        if (has_res) v += e_res[orow * d.res_cs + e_res_coff + n];
        v = relu ? fminf(fmaxf(v, 0.f), hi) : v;
        e_out[orow * e_out_cs + e_out_coff + (long)n * e_out_cmul] = v;
      }
    }
  }
}


// Tiny-M pointwise GEMM (head Linear, SqueezeExcite FCs: M = batch rows): one wavefront per output
// channel, lanes split K with 16-byte loads, rows accumulated in registers, one shuffle reduction.  The
// 128-row MFMA tile would run 4 workgroups for the 2304 -> 400 head and serialise 144 K steps in each.
constexpr int GEMV_MMAX = 16;
__global__ __launch_bounds__(256) void gemv_rows_kernel(const ConvArgs p) {
  const sf_conv_desc& d = p.d;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= d.Cout) return;
  float acc[GEMV_MMAX];
#pragma unroll
  for (int m = 0; m < GEMV_MMAX; ++m) acc[m] = 0.f;
  const float* wrow = p.w + (long)n * d.cin_pad;
  for (int k = lane * 4; k < d.Cin; k += 256) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + k);
#pragma unroll
    for (int m = 0; m < GEMV_MMAX; ++m) {
      if (m < p.M) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(p.in + (long)m * d.in_cs + d.in_coff + k);
        acc[m] += x4[0] * w4[0] + x4[1] * w4[1] + x4[2] * w4[2] + x4[3] * w4[3];
      }
    }
  }
  const float sc = p.scale ? p.scale[n] : 1.f;
  const float bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int m = 0; m < GEMV_MMAX; ++m) {
    float v = acc[m];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0 && m < p.M) {
      v = v * sc + bi;
      if (p.res) v += p.res[(long)m * d.res_cs + d.res_coff + n];
      v = sf_act(v, d.act);
      p.out[(long)m * d.out_cs + d.out_coff + (long)n * d.out_cmul] = v;
    }
  }
}

// Split-K finish: out[m, n] = act(scale[n] * sum_s ws[s][m][n] + bias[n] + res[m, n]) — the partials summed in split
// order (no float atomics), then the same epilogue (incl. the scattered-store map) as the single-pass kernel.
__global__ void conv_splitk_finish_kernel(const ConvArgs p) {
  const sf_conv_desc& d = p.d;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.M * d.Cout;
  if (idx >= total) return;
  const int n = (int)(idx % d.Cout);
  const int m = (int)(idx / d.Cout);
  float v = 0.f;
  for (int s = 0; s < p.ksplit; ++s) v += p.ws[(long)s * total + idx];
  long orow = m;
  if (d.os_T > 1 || d.os_H > 1 || d.os_W > 1) {
    const int wo = m % d.Wo;
    const int t1 = m / d.Wo;
    const int ho = t1 % d.Ho;
    const int t2 = t1 / d.Ho;
    const int to = t2 % d.To;
    const int nn = t2 / d.To;
    const int st = d.os_T > 1 ? d.os_T : 1, sh = d.os_H > 1 ? d.os_H : 1, sw = d.os_W > 1 ? d.os_W : 1;
    orow = (((long)nn * d.ob_T + to * st + d.oo_T) * d.ob_H + ho * sh + d.oo_H) * d.ob_W + wo * sw + d.oo_W;
  }
  v = v * (p.scale ? p.scale[n] : 1.f) + (p.bias ? p.bias[n] : 0.f);
  if (p.res) v += p.res[orow * d.res_cs + d.res_coff + n];
  v = sf_act(v, d.act);
  p.out[orow * d.out_cs + d.out_coff + (long)n * d.out_cmul] = v;
}

// The same for dense 16-byte addressable outputs: four channels per thread, the S partial loads in flight together.
__global__ void conv_splitk_finish4_kernel(const ConvArgs p) {
  const sf_conv_desc& d = p.d;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int c4 = d.Cout >> 2;
  const long total4 = (long)p.M * c4;
  if (idx >= total4) return;
  // (M * Cout / 4 < 2^31 for every launch of this library: one 32-bit division instead of a 64-bit one per thread)
  const long m = total4 < 0x7fffffffL ? (long)((unsigned)idx / (unsigned)c4) : idx / c4;
  const int n = (int)(idx - m * c4) * 4;
  const long stride = (long)p.M * d.Cout;
  const float* src = p.ws + m * d.Cout + n;
  // the S partials summed in split order (bit-reproducible), four loads in flight at a time (measured neutral: the
  // ~28 us this launch costs a res4 3x3 layer — 77 MB at 2.8 TB/s, a quarter of the layer — are not load latency)
  f32x4 v = *reinterpret_cast<const f32x4*>(src);
  int s = 1;
  for (; s + 3 < p.ksplit; s += 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + (long)s * stride);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + (long)(s + 1) * stride);
    const f32x4 c = *reinterpret_cast<const f32x4*>(src + (long)(s + 2) * stride);
    const f32x4 e = *reinterpret_cast<const f32x4*>(src + (long)(s + 3) * stride);
    v += a;
    v += b;
    v += c;
    v += e;
  }
  for (; s < p.ksplit; ++s) v += *reinterpret_cast<const f32x4*>(src + (long)s * stride);
  if (p.scale) v *= *reinterpret_cast<const f32x4*>(p.scale + n);
  if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
  if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * d.res_cs + d.res_coff + n);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = sf_act(v[e], d.act);
  *reinterpret_cast<f32x4*>(p.out + m * d.out_cs + d.out_coff + n) = v;
}

template <int BM, int BN, int WM, int WN>
int launch(const ConvArgs& a, bool vec4, hipStream_t s) {
  ConvArgs p = a;
  p.nb_n = sf_cdiv(p.d.Cout, BN);
  p.nblocks = sf_cdiv(p.M, BM) * p.nb_n;
  const dim3 grid(p.nblocks, p.ksplit, p.groups);
  if (vec4)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 4>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), 0, s, p);
  if (p.ksplit > 1) {
    const long total = (long)p.M * p.d.Cout;
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, p);
  }
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// Split-K factor: the M <= 12544 layers of res4 / res5 with long reductions (3x1x1 and 1x3x3 over 256..2048 channels)
// give 25..200 tiles of 128x128 — less than one per CU — but 144..384 K steps each: share the K steps of a tile between
// `S` workgroups instead of shrinking the tile (the vendor GEMM reaches 108-123 TFLOP/s on these shapes the same way).
static int splitk_factor(const sf_conv_desc* d, long M) {
  if (d->Cout < 64) return 1;
  const long tiles = (long)sf_cdiv(M, 128) * sf_cdiv(d->Cout, d->Cout <= 64 ? 64 : 128);
  const int nk = d->kT * d->kH * d->kW * (d->cin_pad / BK);
  if (tiles >= 384 || nk < 48) return 1;
  // Equal-sized workgroups run in "rounds" of 256 (one per CU): pick the split whose workgroup count wastes the
  // least of its last round, with a small penalty per split for the partial-tile traffic.  Measured on MI355X
  // (tools/microbench/splitk_sweep.py): 196 tiles -> S = 5 (980 = 3.83 rounds) 86 TFLOP/s vs S = 4 (784 = 3.06
  // rounds) 76; 100 tiles -> S = 5 (500 = 1.95 rounds) 96 vs S = 4 (1.56 rounds) 83.
  int best = 1;
  double best_score = (double)tiles / (double)(sf_cdiv(tiles, 256) * 256L);
  for (int S = 2; S <= 12 && nk / S >= 16; ++S) {
    const long wg = tiles * S;
    const double score = (double)wg / (double)(sf_cdiv(wg, 256) * 256L) - 0.01 * S;
    if (score > best_score + 1e-9) {
      best_score = score;
      best = S;
    }
  }
  return best;
}

// Tile configuration of a layer.  Every activation of this network has 49 * 2^k rows, so power-of-two tiles give
// 49 * 2^j workgroups — 392, 784, 1568: 0.77, 0.77, 0.875 of their last round of 256 CUs filled, and the launch ends
// with its busiest CU (tools/microbench/conv_tail.py: the 3x3 128->128 layer runs at 81 TFLOP/s at M = 50 176 and at
// 111 at M = 65 536).  Tiles of 112 = 7 x 16 rows give 7 * 2^j workgroups instead (448, 896: 0.875; 1792, 3584: whole
// rounds), so among the MFMA tilings that fit the layer take the one with the best (last-round fill) x (tile
// efficiency prior); the priors are the measured TFLOP/s of each tiling divided by its fill on the res3 layers
// (tools/microbench/tile_sweep.py: 3x3 128->128 at M = 50 176: 112x128 93.9, 224x128 86.1, 128x128 85.0, 112x64
// 84.3, 64x64 81.3, 128x64 78.3 TFLOP/s).
enum ConvCfg { CFG_256x16, CFG_256x32, CFG_64x64, CFG_128x64, CFG_128x128, CFG_112x64, CFG_112x128, CFG_224x128,
               CFG_COUNT };
static const int CFG_BM[] = {256, 256, 64, 128, 128, 112, 112, 224};
static const int CFG_BN[] = {16, 32, 64, 64, 128, 64, 128, 128};
static const double CFG_EFF[] = {0, 0, 0.86, 0.94, 1.00, 0.90, 0.965, 0.97};

static ConvCfg conv_cfg(const sf_conv_desc* d, long M, int ksplit) {
  static const int forced = [] {
    const char* e = getenv("SF_CONV_CFG");  // microbenchmark aid: force one tiling for Cout >= 64 layers (-2: no 7x16-row tiles)
    return e ? atoi(e) : -1;
  }();
  if (d->Cout <= 16) return CFG_256x16;
  if (d->Cout <= 32) return CFG_256x32;
  if (ksplit > 1) return d->Cout <= 64 ? CFG_128x64 : CFG_128x128;
  if (forced >= CFG_64x64 && forced < CFG_COUNT) return (ConvCfg)forced;
  // Short reductions into wide outputs (the bottleneck "c" convs, K = Cin <= 256 -> Cout >= 128, with the residual in
  // the epilogue) are prologue / epilogue bound: 64x64 tiles (4x the workgroups) measured 81 vs 67 TFLOP/s on
  // 256 -> 1024 at M = 12544 and 74 vs 66 on 128 -> 512 at M = 50176 (tools/microbench/tile_sweep.py).
  // (also 256 -> 64 at M = 200704: 78.7 vs 74.5-76.7 for the larger tiles)
  const int nk = d->kT * d->kH * d->kW * (d->cin_pad / BK);
  if (nk <= 16) return CFG_64x64;
  ConvCfg best = CFG_64x64;
  double best_score = -1.0;
  for (int c = CFG_64x64; c < CFG_COUNT; ++c) {
    if (CFG_BN[c] == 128 && d->Cout <= 64) continue;
    if (forced == -2 && CFG_BM[c] % 112 == 0) continue;  // A/B aid: power-of-two tiles only
    const long nbm = sf_cdiv(M, CFG_BM[c]), nbn = sf_cdiv(d->Cout, CFG_BN[c]);
    const long tiles = nbm * nbn;
    const double fill = (double)tiles / (double)(sf_cdiv(tiles, 256) * 256L);
    const double used = ((double)M * d->Cout) / ((double)nbm * CFG_BM[c] * nbn * CFG_BN[c]);  // padded rows / columns
    const double lone = tiles <= 256 ? 0.9 : 1.0;  // one workgroup per CU: nothing covers its barriers
    const double score = fill * used * CFG_EFF[c] * lone;
    if (score > best_score + 1e-9) { best_score = score; best = (ConvCfg)c; }
  }
  return best;
}

}  // namespace

// Sum S raw partial tiles ws[S][M][Cout] in split order and apply the conv epilogue (conv_bx.hip's finish step).
int sf_conv_splitk_finish(const sf_conv_desc* d, const float* ws, int S, const float* scale, const float* bias,
                          const float* res, float* out, hipStream_t s) {
  ConvArgs p;
  p.d = *d;
  p.in = nullptr; p.w = nullptr; p.scale = scale; p.bias = bias; p.res = res; p.out = out;
  p.M = (int)((long)d->N * d->To * d->Ho * d->Wo);
  p.ntaps = d->kT * d->kH * d->kW;
  p.nb_n = 0; p.nblocks = 0; p.vec_epi = 0;
  p.ksplit = S;
  p.ws = const_cast<float*>(ws);
  const long total = (long)p.M * d->Cout;
  const bool scatter = d->os_T > 1 || d->os_H > 1 || d->os_W > 1;
  const bool vec = !scatter && d->out_cmul == 1 && (d->Cout % 4 == 0) && (d->out_cs % 4 == 0) && (d->out_coff % 4 == 0) &&
                   sf_aligned16(out) && sf_aligned16(ws) && (!scale || sf_aligned16(scale)) &&
                   (!bias || sf_aligned16(bias)) &&
                   (!res || ((d->res_cs % 4 == 0) && (d->res_coff % 4 == 0) && sf_aligned16(res)));
  if (vec)
    hipLaunchKernelGGL(conv_splitk_finish4_kernel, dim3(sf_cdiv(total / 4, 256)), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, p);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

int sf_conv_bx_takes(const sf_conv_desc* d);                                                  // conv_bx.hip
int sf_conv_pw_takes(const sf_conv_desc* d);                                                  // conv_bx.hip
extern "C" long sf_conv_pw_ws_floats(const sf_conv_desc* d, int have_w_planes);               // conv_bx.hip
int sf_conv_pw_try(const sf_conv_desc* d, const float* in, const float* w_packed, const unsigned short* w_planes,
                   const float* scale, const float* bias, const float* res, float* out, float* ws, float* stats,
                   int* parts, hipStream_t stream);
extern "C" long sf_conv_bx_ws_floats(const sf_conv_desc* d, int have_in_planes, int have_w_planes);  // conv_bx.hip
int sf_conv_bx_try(const sf_conv_desc* d, const float* in, const unsigned short* in_planes, const float* w,
                   const unsigned short* w_planes, const float* scale, const float* bias, const float* res, float* out,
                   float* ws, hipStream_t stream);                                           // conv_bx.hip
int sf_conv_wave_try(const sf_conv_desc* d, const float* in, const float* w, const float* scale,
                     const float* bias, const float* res, float* out, hipStream_t stream, float* stats,
                     int* stat_parts);                                                       // conv_wave.hip
int sf_conv_wave_takes(const sf_conv_desc* d);                                                // conv_wave.hip
long sf_conv_wave_max_parts(long M);                                                          // conv_wave.hip

extern "C" long sf_conv_fwd_ws_floats(const sf_conv_desc* d) {
  if (!d) return 0;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M <= 0 || d->Cout <= 0 || d->cin_pad <= 0) return 0;
  if (sf_conv_pw_takes(d)) return sf_conv_pw_ws_floats(d, 0);     // pointwise layers: the weight planes
  if (sf_conv_bx_takes(d)) return sf_conv_bx_ws_floats(d, 0, 0);  // operand planes + split-K partial tiles
  if (sf_conv_wave_takes(d)) return 0;  // conv_wave.hip splits long reductions inside the workgroup
  const int S = splitk_factor(d, M);
  return S > 1 ? (long)S * M * d->Cout : 0;
}

int sf_conv_stem_fwd_try(const sf_conv_desc* d, const float* in, const float* w, const float* scale,
                         const float* bias, const float* res, float* out, hipStream_t stream);  // conv_stem.hip
int sf_conv_small_try(const sf_conv_desc* d, const float* in, const float* w, const float* scale, const float* bias,
                      const float* res, float* out, hipStream_t stream, float* stats, int* stat_parts);  // conv_small.hip
int sf_conv_rows_try(const sf_conv_desc* d, const float* in, const float* w, const float* scale, const float* bias,
                     const float* res, float* out, hipStream_t stream, float* stats, int* stat_parts);   // conv_rows.hip

static int conv_fwd_impl(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                         const float* bias, const float* res, float* out, float* ws, void* stream,
                         float* stats = nullptr, int* stat_parts = nullptr) {
  if (stat_parts) *stat_parts = 0;
  if (!d || !in || !w_packed || !out) return SF_EINVAL;
  if (d->Cin <= 0 || d->Cout <= 0 || d->cin_pad < d->Cin || (d->cin_pad % BK) != 0) return SF_EINVAL;
  if (d->kT <= 0 || d->kH <= 0 || d->kW <= 0 || d->sT <= 0 || d->sH <= 0 || d->sW <= 0) return SF_EINVAL;
  if (d->out_cmul <= 0) return SF_EINVAL;
  if (d->act != SF_ACT_NONE && d->act != SF_ACT_RELU && d->act != SF_ACT_RELU6) return SF_EINVAL;
  if (!sf_aligned16(w_packed)) return SF_EALIGN;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M <= 0 || M > 0x7fffffffL) return SF_EINVAL;
  const bool scatter = d->os_T > 1 || d->os_H > 1 || d->os_W > 1;
  if (scatter) {  // every scattered position must lie inside the destination
    const int st = d->os_T > 1 ? d->os_T : 1, sh = d->os_H > 1 ? d->os_H : 1, sw = d->os_W > 1 ? d->os_W : 1;
    if (d->oo_T < 0 || d->oo_H < 0 || d->oo_W < 0 || (d->To - 1) * st + d->oo_T >= d->ob_T ||
        (d->Ho - 1) * sh + d->oo_H >= d->ob_H || (d->Wo - 1) * sw + d->oo_W >= d->ob_W)
      return SF_EINVAL;
  }
  {  // the Fast pathway's stem: LDS-ring kernel instead of one L2 fetch per tap
    const int rc = sf_conv_stem_fwd_try(d, in, w_packed, scale, bias, res, out, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  {  // small-channel stride-1 "same" layers (the Fast pathway, lateral / q|k|v projections) and their data gradients:
     // whole rows through LDS, swapped-operand MFMAs, statistics in the epilogue (conv_rows.hip)
    const int rc = sf_conv_rows_try(d, in, w_packed, scale, bias, res, out, (hipStream_t)stream, stats, stat_parts);
    if (rc != 1) return rc;
  }
  {  // the tiniest channel counts (8 -> 8 spatial layers of the Fast pathway): LDS-staged input, scalar-register weights, vector FMAs
    const int rc = sf_conv_small_try(d, in, w_packed, scale, bias, res, out, (hipStream_t)stream, stats, stat_parts);
    if (rc != 1) return rc;
  }
  if (ws && !stats) {  // pointwise layers on the bf16 matrix pipe, activations split in registers (conv_bx.hip)
    const int rc = sf_conv_pw_try(d, in, w_packed, nullptr, scale, bias, res, out, ws, nullptr, nullptr,
                                  (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  if (ws && !stats) {  // long reductions on the bf16 matrix pipe (fp32-exact operand pieces), conv_bx.hip
    const int rc = sf_conv_bx_try(d, in, nullptr, w_packed, nullptr, scale, bias, res, out, ws, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  {  // the per-wavefront implicit GEMM (no LDS staging, no barrier in the main loop) for every 16-byte aligned shape
    const int rc = sf_conv_wave_try(d, in, w_packed, scale, bias, res, out, (hipStream_t)stream, stats, stat_parts);
    if (rc != 1) return rc;
  }
  ConvArgs a;
  a.d = *d;
  a.in = in; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.M = (int)M;
  a.ntaps = d->kT * d->kH * d->kW;
  a.nb_n = 0; a.nblocks = 0; a.vec_epi = 0;
  // split-K partial tiles need S * M * Cout floats of workspace: sf_conv_fwd_ws_floats sizes it for that ONLY for shapes
  // none of the pw / bx / wave paths takes by shape.  A shape they take by shape but refused at run time (an operand
  // pointer that is not 16-byte aligned) arrives here with a workspace sized for weight planes: one pass, no partials.
  const bool ws_is_for_splitk = ws && !sf_conv_pw_takes(d) && !sf_conv_bx_takes(d) && !sf_conv_wave_takes(d);
  a.ksplit = ws_is_for_splitk ? splitk_factor(d, M) : 1;
  a.ws = ws;
  const bool vec4 = (d->Cin % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && sf_aligned16(in);
  a.vec_epi = (d->out_cmul == 1) && (d->Cout % 4 == 0) && (d->out_cs % 4 == 0) && (d->out_coff % 4 == 0) &&
              sf_aligned16(out) && (!scale || sf_aligned16(scale)) && (!bias || sf_aligned16(bias)) &&
              (!res || ((d->res_cs % 4 == 0) && (d->res_coff % 4 == 0) && sf_aligned16(res)));
  hipStream_t s = (hipStream_t)stream;
  if (M <= GEMV_MMAX && !scatter && a.ntaps == 1 && vec4 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 0 &&
      d->pH == 0 && d->pW == 0) {
    hipLaunchKernelGGL(gemv_rows_kernel, dim3(sf_cdiv(d->Cout, 4)), dim3(256), 0, s, a);
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
  switch (conv_cfg(d, M, a.ksplit)) {
    case CFG_256x16: return launch<256, 16, 4, 1>(a, vec4, s);
    case CFG_256x32: return launch<256, 32, 4, 1>(a, vec4, s);
    case CFG_64x64: return launch<64, 64, 2, 2>(a, vec4, s);
    case CFG_128x64: return launch<128, 64, 2, 2>(a, vec4, s);
    case CFG_112x64: return launch<112, 64, 1, 4>(a, vec4, s);
    case CFG_112x128: return launch<112, 128, 1, 4>(a, vec4, s);
    case CFG_224x128: return launch<224, 128, 2, 2>(a, vec4, s);
    default: return launch<128, 128, 2, 2>(a, vec4, s);
  }
}

extern "C" int sf_conv_fwd(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                           const float* bias, const float* res, float* out, void* stream) {
  return conv_fwd_impl(d, in, w_packed, scale, bias, res, out, nullptr, stream);
}

// As sf_conv_fwd with a caller-provided workspace of sf_conv_fwd_ws_floats(d) floats (0: none needed): enables the
// split-K schedule for short-M / long-K layers.  ws == NULL behaves exactly like sf_conv_fwd.
extern "C" int sf_conv_fwd_ws(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                              const float* bias, const float* res, float* out, float* ws, void* stream) {
  if (ws && !sf_aligned16(ws)) return SF_EALIGN;
  return conv_fwd_impl(d, in, w_packed, scale, bias, res, out, ws, stream);
}

// nn.Conv3d(groups = G), 1 < G < channels, as ONE launch of the LDS-tiled kernel: the block-diagonal GEMM's group is the
// grid's z index.  `d` carries the WHOLE layer (Cin, Cout = all groups; cin_pad = the packed width of ONE group's
// Cin / G channels); w_packed = [Cout][taps][cin_pad], which is the G per-group packs one after the other (forward), or
// with d->transposed the G transposed packs [Cin_fwd][taps][pad(Cout_fwd / G)] (data gradient: d->Cin = forward Cout).
// shuffle != 0: group g's channel j is stored at channel j * G + g (channel_shuffle(., G) in the store index).
extern "C" int sf_conv_fwd_grouped(const sf_conv_desc* d, int groups, int shuffle, const float* in,
                                   const float* w_packed, const float* scale, const float* bias, const float* res,
                                   float* out, void* stream) {
  if (!d || groups < 1) return SF_EINVAL;
  if (groups == 1 && !shuffle) return conv_fwd_impl(d, in, w_packed, scale, bias, res, out, nullptr, stream);
  if (!in || !w_packed || !out) return SF_EINVAL;
  if (d->Cin <= 0 || d->Cout <= 0 || d->Cin % groups || d->Cout % groups) return SF_EINVAL;
  const int cin_g = d->Cin / groups, cout_g = d->Cout / groups;
  if (d->cin_pad < cin_g || (d->cin_pad % BK) != 0) return SF_EINVAL;
  if (d->kT <= 0 || d->kH <= 0 || d->kW <= 0 || d->sT <= 0 || d->sH <= 0 || d->sW <= 0) return SF_EINVAL;
  if (d->out_cmul != 1) return SF_EINVAL;   // the shuffle owns the channel multiplier
  if (d->act != SF_ACT_NONE && d->act != SF_ACT_RELU && d->act != SF_ACT_RELU6) return SF_EINVAL;
  if (d->os_T > 1 || d->os_H > 1 || d->os_W > 1) return SF_EINVAL;
  if (!sf_aligned16(w_packed)) return SF_EALIGN;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M <= 0 || M > 0x7fffffffL || groups > 65535) return SF_EINVAL;
  ConvArgs a;
  a.d = *d;
  a.d.Cin = cin_g;
  a.d.Cout = cout_g;
  a.d.out_cmul = shuffle ? groups : 1;
  a.in = in; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.M = (int)M;
  a.ntaps = d->kT * d->kH * d->kW;
  a.nb_n = 0; a.nblocks = 0;
  a.ksplit = 1;
  a.ws = nullptr;
  a.groups = groups;
  a.g_in = cin_g;
  a.g_out = shuffle ? 1 : cout_g;
  a.g_res = cout_g;
  a.g_sb = cout_g;
  a.g_w = (long)cout_g * a.ntaps * d->cin_pad;
  // 16-byte forms only when every group's window starts on a 16-byte boundary
  const bool vec4 = (cin_g % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && sf_aligned16(in);
  a.vec_epi = !shuffle && (cout_g % 4 == 0) && (d->out_cs % 4 == 0) && (d->out_coff % 4 == 0) && sf_aligned16(out) &&
              (!scale || sf_aligned16(scale)) && (!bias || sf_aligned16(bias)) &&
              (!res || ((d->res_cs % 4 == 0) && (d->res_coff % 4 == 0) && sf_aligned16(res)));
  hipStream_t s = (hipStream_t)stream;
  // tiles sized for ONE group's GEMM; the G groups multiply the workgroup count
  if (cout_g <= 16) return launch<256, 16, 4, 1>(a, vec4, s);
  if (cout_g <= 32) return launch<256, 32, 4, 1>(a, vec4, s);
  if (cout_g <= 64 || (long)sf_cdiv(M, 128) * sf_cdiv(cout_g, 128) * groups < 256) {
    if ((long)sf_cdiv(M, 128) * sf_cdiv(cout_g, 64) * groups < 256) return launch<64, 64, 2, 2>(a, vec4, s);
    return launch<128, 64, 2, 2>(a, vec4, s);
  }
  return launch<128, 128, 2, 2>(a, vec4, s);
}

// Training-mode BN statistics out of the conv's epilogue (conv_wave.hip): sf_conv_stats_ws_floats(d) floats of scratch
// (0: this shape never produces them); sf_conv_fwd_stats runs the conv and sets *parts to the number of [count, K, S1,
// S2] rows per channel it left in stats_ws (0: none were produced — run sf_bn_train_stats on the output instead).
extern "C" long sf_conv_stats_ws_floats(const sf_conv_desc* d) {
  if (!d || !sf_conv_wave_takes(d) || d->transposed) return 0;
  if (sf_conv_pw_takes(d)) return 0;  // statistics of these come from sf_conv_fwd_pw (sf_conv_pw_stats_floats)
  if (sf_conv_bx_takes(d)) return 0;  // conv_bx.hip leaves no statistics: the caller's statistics pass runs
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  return sf_conv_wave_max_parts(M) * 4 * d->Cout;
}

extern "C" int sf_conv_fwd_stats(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                                 const float* bias, const float* res, float* out, float* stats_ws, int* parts,
                                 void* stream) {
  if (!parts) return SF_EINVAL;
  return conv_fwd_impl(d, in, w_packed, scale, bias, res, out, nullptr, stream, stats_ws, parts);
}
