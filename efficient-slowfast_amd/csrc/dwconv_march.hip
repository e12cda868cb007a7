// dwconv_march.hip — depthwise kT x 3 x 3 convolutions (kT = 1 | 3, stride 1, "same" padding) of GhostNet's cheap
// operations and ShuffleNetV2's branches as ROW MARCHES (gfx950): forward, data gradient and weight gradient.
//
// The generic kernels (elementwise.hip dwconv_kernel, backward.hip dwconv_dgrad_* / dwconv_wgrad_partial4_kernel) give
// a thread one output position: 27 taps = 27 loads, each behind its own bounds test and 64-bit address product —
// ~20 vector instructions of index arithmetic per FMA.  They are bound by exactly that arithmetic: 0.055 of HBM on the
// backward of SlowFastGhostNet at 8 clips (profiles/r05_dwconv_per_shape_before.txt), worst where the Fast pathway's
// channel counts (2, 6, 10: no multiples of 4) force one channel per thread over 32 x 112 x 112 positions.
//
// Here a thread owns ONE column (n, t, w, channel group) and marches down `HC` consecutive rows h of it with the
// 3 x 3 (x kT) window of its inputs in registers: each step shifts the window by a row and loads only the new row's
// 3 x kT values — a third of the loads — at addresses that advance by a constant (one add per load), the t / w bounds
// are decided once per column, and the weights (forward / data gradient) or the kT x 9 tap sums (weight gradient) stay
// in registers for the whole march.  Loads are raw buffer loads: a tap outside the tensor gets an out-of-range offset
// and reads 0, so the march has no branch.  Lanes of a wavefront walk (channel group, w) fastest: a wave-load covers
// consecutive addresses of one input row, for every channel pitch (the ghost modules read and write channel SLICES of a
// wider buffer) and any channel count (V = 1 when channels, pitches or offsets are no multiples of 4).
//
//   forward        out[m, c] = act(scale[c] * sum_tap x[m + tap, c] w[tap, c] + bias[c] + res[m, c])     (c < Cout)
//   data gradient  dx[m, c] (+)= sum_tap dz[m - tap, c] w[tap, c]            = the forward with the taps mirrored
//   weight grad.   partial[blk][tap][c] = sum over the block's columns of dz[m, c] x[m + tap, c]; the four / .. lanes
//                  of a channel group are summed in lane order through LDS (no atomics: bit-reproducible) and
//                  dwconv_wgrad_final_kernel (backward.hip) sums the blocks in fp64.
//
// Replaces for these shapes: nn.Conv3d(C, C, (kT, 3, 3), 1, (kT/2, 1, 1), groups = C) + BatchNorm3d (+ ReLU) of
// ghostnet_helper.py:72-100 (cheap_operation), :130-140 (conv_dw stride 1) and shufflenetv2_helper.py:60-92, and
// their autograd gradients.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int DWM_TPB = 256;
constexpr unsigned DWM_OOB = 0x80000000u;

struct DwmArgs {
  const float* x;        // marched input: forward = in, data gradient = dz, weight gradient = x
  const float* g;        // weight gradient: dz
  const float* w;        // [taps][wpitch]
  const float* scale;
  const float* bias;
  const float* res;
  float* out;            // forward: out, data gradient: dx
  float* partial;        // weight gradient: [nblk][taps][C]
  unsigned x_bytes, g_bytes, res_bytes, out_bytes;
  int x_cs, x_coff, g_cs, g_coff, res_cs, res_coff, out_cs, out_coff, wpitch;
  int N, T, H, W, C;     // C = channels computed (forward: Cout <= Cin)
  int Ho, Wo;            // strided kernels: output rows / columns (H, W = the input's)
  int act, accumulate, flip;
  int HC, nhc;           // rows per march, marches per column
  int CQ;                // channel groups per workgroup (power of two <= 64)
  long units;            // N * T * nhc * W columns pieces
  int nblk;              // weight gradient: workgroups along the units
};

template <int V> struct Vec;
template <> struct Vec<4> {
  typedef f32x4 T;
  static __device__ __forceinline__ T zero() { return (T){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  }
  static __device__ __forceinline__ T ldp(const float* p) { return *reinterpret_cast<const T*>(p); }
  static __device__ __forceinline__ void stp(float* p, T v) { *reinterpret_cast<T*>(p) = v; }
};
template <> struct Vec<1> {
  typedef float T;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  }
  static __device__ __forceinline__ T ldp(const float* p) { return *p; }
  static __device__ __forceinline__ void stp(float* p, T v) { *p = v; }
};

__device__ __forceinline__ float elem(float v, int) { return v; }
__device__ __forceinline__ float elem(f32x4 v, int e) { return v[e]; }
__device__ __forceinline__ void set_elem(float& v, int, float x) { v = x; }
__device__ __forceinline__ void set_elem(f32x4& v, int e, float x) { v[e] = x; }

// Column piece `u` -> (n, t, first row, w); w runs fastest so that the lanes of a wavefront read consecutive addresses.
struct Col { int n, t, h0, w; };
__device__ __forceinline__ Col decode(const DwmArgs& p, unsigned u) {
  Col c;
  const unsigned q1 = u / (unsigned)p.W;
  c.w = (int)(u - q1 * (unsigned)p.W);
  const unsigned q2 = q1 / (unsigned)p.nhc;
  c.h0 = (int)(q1 - q2 * (unsigned)p.nhc) * p.HC;
  const unsigned q3 = q2 / (unsigned)p.T;
  c.t = (int)(q2 - q3 * (unsigned)p.T);
  c.n = (int)q3;
  return c;
}

// Byte offsets of the window's 3 x KT columns at row 0 of their planes (DWM_OOB where t or w falls outside).
template <int KT>
__device__ __forceinline__ void window_cols(const DwmArgs& p, const Col& c, int ch, unsigned (&col)[KT][3]) {
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int tt = c.t + kt - KT / 2;
    const bool okt = (unsigned)tt < (unsigned)p.T;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = c.w + kw - 1;
      const bool ok = okt && (unsigned)ww < (unsigned)p.W;
      const unsigned row0 = (unsigned)((c.n * p.T + tt) * p.H) * (unsigned)p.W;
      col[kt][kw] = ok ? ((row0 + (unsigned)ww) * (unsigned)p.x_cs + (unsigned)(p.x_coff + ch)) * 4u : DWM_OOB;
    }
  }
}

// ------------------------------------------------------------------------------------------- forward / data gradient
template <int KT, int V>
__global__ __launch_bounds__(DWM_TPB) void dwm_fwd_kernel(const DwmArgs p) {
  typedef typename Vec<V>::T VT;
  const int ql = threadIdx.x % p.CQ, rl = threadIdx.x / p.CQ, rpi = DWM_TPB / p.CQ;
  const int ch = ((int)blockIdx.y * p.CQ + ql) * V;
  const long uu = (long)blockIdx.x * rpi + rl;
  if (ch >= p.C || uu >= p.units) return;
  const Col c = decode(p, (unsigned)uu);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  unsigned col[KT][3];
  window_cols<KT>(p, c, ch, col);
  VT wt[KT][3][3];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = (kt * 3 + kh) * 3 + kw;
        wt[kt][kh][kw] = Vec<V>::ldp(p.w + (long)(p.flip ? KT * 9 - 1 - tap : tap) * p.wpitch + ch);
      }
  const unsigned rowb = (unsigned)p.W * (unsigned)p.x_cs * 4u;
  VT win[KT][3][3];
  auto load_row = [&](int h, int slot) {
    const bool okh = (unsigned)h < (unsigned)p.H;
    const unsigned ro = (unsigned)h * rowb;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        win[kt][slot][kw] = Vec<V>::ld(xr, (okh && col[kt][kw] != DWM_OOB) ? col[kt][kw] + ro : DWM_OOB);
  };
  load_row(c.h0 - 1, 1);
  load_row(c.h0, 2);
  const int h1 = c.h0 + p.HC < p.H ? c.h0 + p.HC : p.H;
  VT sc = Vec<V>::zero(), bi = Vec<V>::zero();
  if (p.scale) {
    sc = Vec<V>::ldp(p.scale + ch);
    bi = Vec<V>::ldp(p.bias + ch);
  }
  const long m0 = (((long)c.n * p.T + c.t) * p.H) * p.W + c.w;
  for (int h = c.h0; h < h1; ++h) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        win[kt][0][kw] = win[kt][1][kw];
        win[kt][1][kw] = win[kt][2][kw];
      }
    load_row(h + 1, 2);
    VT acc = Vec<V>::zero();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc += win[kt][kh][kw] * wt[kt][kh][kw];
    const long m = m0 + (long)h * p.W;
    if (p.scale) acc = acc * sc + bi;
    float* const o = p.out + m * p.out_cs + p.out_coff + ch;
    if (p.res) acc += Vec<V>::ldp(p.res + m * p.res_cs + p.res_coff + ch);
    if (p.accumulate) acc += Vec<V>::ldp(o);
#pragma unroll
    for (int e = 0; e < V; ++e) set_elem(acc, e, sf_act(elem(acc, e), p.act));
    Vec<V>::stp(o, acc);
  }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
template <int KT, int V>
__global__ __launch_bounds__(DWM_TPB) void dwm_wgrad_kernel(const DwmArgs p) {
  typedef typename Vec<V>::T VT;
  __shared__ VT red[9][DWM_TPB];
  const int ql = threadIdx.x % p.CQ, rl = threadIdx.x / p.CQ, rpi = DWM_TPB / p.CQ;
  const int ch = ((int)blockIdx.y * p.CQ + ql) * V;
  const bool cok = ch < p.C;
  const long per = (p.units + p.nblk - 1) / p.nblk;
  const long u0 = (long)blockIdx.x * per;
  const long u1 = (u0 + per < p.units) ? u0 + per : p.units;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, (int)p.g_bytes, 0x00020000);
  const unsigned rowb = (unsigned)p.W * (unsigned)p.x_cs * 4u;
  const unsigned growb = (unsigned)p.W * (unsigned)p.g_cs * 4u;
  VT acc[KT][3][3];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) acc[kt][kh][kw] = Vec<V>::zero();
  if (cok) {
    for (long uu = u0 + rl; uu < u1; uu += rpi) {
      const Col c = decode(p, (unsigned)uu);
      unsigned col[KT][3];
      window_cols<KT>(p, c, ch, col);
      VT win[KT][3][3];
      auto load_row = [&](int h, int slot) {
        const bool okh = (unsigned)h < (unsigned)p.H;
        const unsigned ro = (unsigned)h * rowb;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            win[kt][slot][kw] = Vec<V>::ld(xr, (okh && col[kt][kw] != DWM_OOB) ? col[kt][kw] + ro : DWM_OOB);
      };
      load_row(c.h0 - 1, 1);
      load_row(c.h0, 2);
      const int h1 = c.h0 + p.HC < p.H ? c.h0 + p.HC : p.H;
      unsigned goff = ((unsigned)(((c.n * p.T + c.t) * p.H + c.h0) * p.W + c.w) * (unsigned)p.g_cs +
                       (unsigned)(p.g_coff + ch)) * 4u;
      for (int h = c.h0; h < h1; ++h, goff += growb) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            win[kt][0][kw] = win[kt][1][kw];
            win[kt][1][kw] = win[kt][2][kw];
          }
        load_row(h + 1, 2);
        const VT g = Vec<V>::ld(gr, goff);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[kt][kh][kw] += g * win[kt][kh][kw];
      }
    }
  }
  // the lanes rl = 0 .. rpi-1 of a channel group, summed in lane order, 9 taps per round through LDS
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) red[kh * 3 + kw][threadIdx.x] = acc[kt][kh][kw];
    __syncthreads();
    if (rl == 0 && cok) {
      for (int u = 0; u < 9; ++u) {
        VT tot = Vec<V>::zero();
        for (int i = 0; i < rpi; ++i) tot += red[u][i * p.CQ + ql];
        Vec<V>::stp(p.partial + ((long)blockIdx.x * (KT * 9) + kt * 9 + u) * p.C + ch, tot);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ stride (1, 2, 2): 1 x K x K, K = 3 | 5, padding K / 2
// GhostNet's down-sampling depthwise layers (ghostnet_helper.py:114-120: conv_dw with stride 2, kernel 3 or 5) and
// ShuffleNetV2's stride-2 branches.  Same idea as above — a thread owns a column and marches down it with its window in
// registers — with the row step of the window doubled:
//   forward / weight gradient: the thread owns an OUTPUT column (n, t, wo, channel group); per output row the K x K input
//     window moves down two rows, so two rows of K values are loaded and K - 2 rows are kept;
//   data gradient: the thread owns an INPUT column (n, t, wi, channel group).  Input rows 2b - p and 2b - p + 1 read the
//     SAME rows b, b - 1, .. of dz — the first with the even taps kh = 2i, the second with the odd ones — so a step loads
//     one row of dz and writes two rows of dx; which columns of dz (and which kw) a thread needs follows from the parity
//     of wi alone and is fixed for the march: (K + 1) / 2 columns, taps outside the kernel carry a zero weight.
// Unit -> (n, t, first row of the march, column); the column runs fastest.
struct Col2 { int n, t, r0, w; };
__device__ __forceinline__ Col2 decode2(const DwmArgs& p, unsigned u, int wn) {
  Col2 c;
  const unsigned q1 = u / (unsigned)wn;
  c.w = (int)(u - q1 * (unsigned)wn);
  const unsigned q2 = q1 / (unsigned)p.nhc;
  c.r0 = (int)(q1 - q2 * (unsigned)p.nhc) * p.HC;
  const unsigned q3 = q2 / (unsigned)p.T;
  c.t = (int)(q2 - q3 * (unsigned)p.T);
  c.n = (int)q3;
  return c;
}

template <int K, int V>
__global__ __launch_bounds__(DWM_TPB) void dwm2_fwd_kernel(const DwmArgs p) {
  typedef typename Vec<V>::T VT;
  constexpr int P = K / 2;
  const int ql = threadIdx.x % p.CQ, rl = threadIdx.x / p.CQ, rpi = DWM_TPB / p.CQ;
  const int ch = ((int)blockIdx.y * p.CQ + ql) * V;
  const long uu = (long)blockIdx.x * rpi + rl;
  if (ch >= p.C || uu >= p.units) return;
  const Col2 c = decode2(p, (unsigned)uu, p.Wo);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  unsigned col[K];
  const unsigned plane0 = (unsigned)((c.n * p.T + c.t) * p.H) * (unsigned)p.W;
#pragma unroll
  for (int kw = 0; kw < K; ++kw) {
    const int wi = 2 * c.w - P + kw;
    col[kw] = (unsigned)wi < (unsigned)p.W ? ((plane0 + (unsigned)wi) * (unsigned)p.x_cs + (unsigned)(p.x_coff + ch)) * 4u
                                           : DWM_OOB;
  }
  VT wt[K][K];
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) wt[kh][kw] = Vec<V>::ldp(p.w + (long)(kh * K + kw) * p.wpitch + ch);
  const unsigned rowb = (unsigned)p.W * (unsigned)p.x_cs * 4u;
  VT win[K][K];
  auto load_row = [&](int h, int slot) {
    const bool okh = (unsigned)h < (unsigned)p.H;
    const unsigned ro = (unsigned)h * rowb;
#pragma unroll
    for (int kw = 0; kw < K; ++kw) win[slot][kw] = Vec<V>::ld(xr, (okh && col[kw] != DWM_OOB) ? col[kw] + ro : DWM_OOB);
  };
#pragma unroll
  for (int kh = 0; kh < K - 2; ++kh) load_row(2 * c.r0 - P + kh, kh + 2);
  const int h1 = c.r0 + p.HC < p.Ho ? c.r0 + p.HC : p.Ho;
  VT sc = Vec<V>::zero(), bi = Vec<V>::zero();
  if (p.scale) {
    sc = Vec<V>::ldp(p.scale + ch);
    bi = Vec<V>::ldp(p.bias + ch);
  }
  const long m0 = (((long)c.n * p.T + c.t) * p.Ho) * p.Wo + c.w;
  for (int ho = c.r0; ho < h1; ++ho) {
#pragma unroll
    for (int kh = 0; kh < K - 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < K; ++kw) win[kh][kw] = win[kh + 2][kw];
    load_row(2 * ho - P + K - 2, K - 2);
    load_row(2 * ho - P + K - 1, K - 1);
    VT acc = Vec<V>::zero();
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
      for (int kw = 0; kw < K; ++kw) acc += win[kh][kw] * wt[kh][kw];
    const long m = m0 + (long)ho * p.Wo;
    if (p.scale) acc = acc * sc + bi;
    float* const o = p.out + m * p.out_cs + p.out_coff + ch;
    if (p.res) acc += Vec<V>::ldp(p.res + m * p.res_cs + p.res_coff + ch);
#pragma unroll
    for (int e = 0; e < V; ++e) set_elem(acc, e, sf_act(elem(acc, e), p.act));
    Vec<V>::stp(o, acc);
  }
}

// dx[n, t, hi, wi, c] (+)= sum over (kh, kw) with hi + p - kh and wi + p - kw even of
//                          dz[n, t, (hi + p - kh) / 2, (wi + p - kw) / 2, c] w[kh, kw, c]
template <int K, int V>
__global__ __launch_bounds__(DWM_TPB) void dwm2_dgrad_kernel(const DwmArgs p) {
  typedef typename Vec<V>::T VT;
  constexpr int P = K / 2, J = (K + 1) / 2;
  const int ql = threadIdx.x % p.CQ, rl = threadIdx.x / p.CQ, rpi = DWM_TPB / p.CQ;
  const int ch = ((int)blockIdx.y * p.CQ + ql) * V;
  const long uu = (long)blockIdx.x * rpi + rl;
  if (ch >= p.C || uu >= p.units) return;
  const Col2 c = decode2(p, (unsigned)uu, p.W);   // p.x = dz [N][T][Ho][Wo], p.out = dx [N][T][H][W]
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const int par = (c.w + P) & 1;
  const unsigned plane0 = (unsigned)((c.n * p.T + c.t) * p.Ho) * (unsigned)p.Wo;
  unsigned col[J];
  VT we[J][J], wo[J][J];  // rows: dz row b - i; even taps kh = 2i, odd taps kh = 2i + 1
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int kw = par + 2 * j;
    const int num = c.w + P - kw;  // even
    const bool ok = kw < K && num >= 0 && (num >> 1) < p.Wo;
    col[j] = ok ? ((plane0 + (unsigned)(num >> 1)) * (unsigned)p.x_cs + (unsigned)(p.x_coff + ch)) * 4u : DWM_OOB;
#pragma unroll
    for (int i = 0; i < J; ++i) {
      we[i][j] = (kw < K) ? Vec<V>::ldp(p.w + (long)((2 * i) * K + kw) * p.wpitch + ch) : Vec<V>::zero();
      wo[i][j] = (kw < K && 2 * i + 1 < K) ? Vec<V>::ldp(p.w + (long)((2 * i + 1) * K + kw) * p.wpitch + ch)
                                           : Vec<V>::zero();
    }
  }
  const unsigned rowb = (unsigned)p.Wo * (unsigned)p.x_cs * 4u;
  VT win[J][J];
  auto load_row = [&](int b, int slot) {
    const bool okh = (unsigned)b < (unsigned)p.Ho;
    const unsigned ro = (unsigned)b * rowb;
#pragma unroll
    for (int j = 0; j < J; ++j) win[slot][j] = Vec<V>::ld(xr, (okh && col[j] != DWM_OOB) ? col[j] + ro : DWM_OOB);
  };
#pragma unroll
  for (int i = 0; i < J - 1; ++i) load_row(c.r0 - 1 - i, i);
  const int nb = (p.H - 1 + P) / 2 + 1;
  const int b1 = c.r0 + p.HC < nb ? c.r0 + p.HC : nb;
  const long m0 = (((long)c.n * p.T + c.t) * p.H) * p.W + c.w;
  for (int b = c.r0; b < b1; ++b) {
#pragma unroll
    for (int i = J - 1; i > 0; --i)
#pragma unroll
      for (int j = 0; j < J; ++j) win[i][j] = win[i - 1][j];
    load_row(b, 0);
    VT ae = Vec<V>::zero(), ao = Vec<V>::zero();
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        ae += win[i][j] * we[i][j];
        if (2 * i + 1 < K) ao += win[i][j] * wo[i][j];
      }
    const int he = 2 * b - P;
    if ((unsigned)he < (unsigned)p.H) {
      float* const o = p.out + (m0 + (long)he * p.W) * p.out_cs + p.out_coff + ch;
      if (p.accumulate) ae += Vec<V>::ldp(o);
      Vec<V>::stp(o, ae);
    }
    if ((unsigned)(he + 1) < (unsigned)p.H) {
      float* const o = p.out + (m0 + (long)(he + 1) * p.W) * p.out_cs + p.out_coff + ch;
      if (p.accumulate) ao += Vec<V>::ldp(o);
      Vec<V>::stp(o, ao);
    }
  }
}

template <int K, int V>
__global__ __launch_bounds__(DWM_TPB) void dwm2_wgrad_kernel(const DwmArgs p) {
  typedef typename Vec<V>::T VT;
  constexpr int P = K / 2;
  __shared__ VT red[K][DWM_TPB];
  const int ql = threadIdx.x % p.CQ, rl = threadIdx.x / p.CQ, rpi = DWM_TPB / p.CQ;
  const int ch = ((int)blockIdx.y * p.CQ + ql) * V;
  const bool cok = ch < p.C;
  const long per = (p.units + p.nblk - 1) / p.nblk;
  const long u0 = (long)blockIdx.x * per;
  const long u1 = (u0 + per < p.units) ? u0 + per : p.units;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, (int)p.g_bytes, 0x00020000);
  const unsigned rowb = (unsigned)p.W * (unsigned)p.x_cs * 4u;
  const unsigned growb = (unsigned)p.Wo * (unsigned)p.g_cs * 4u;
  VT acc[K][K];
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) acc[kh][kw] = Vec<V>::zero();
  if (cok) {
    for (long uu = u0 + rl; uu < u1; uu += rpi) {
      const Col2 c = decode2(p, (unsigned)uu, p.Wo);
      unsigned col[K];
      const unsigned plane0 = (unsigned)((c.n * p.T + c.t) * p.H) * (unsigned)p.W;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int wi = 2 * c.w - P + kw;
        col[kw] = (unsigned)wi < (unsigned)p.W
                      ? ((plane0 + (unsigned)wi) * (unsigned)p.x_cs + (unsigned)(p.x_coff + ch)) * 4u : DWM_OOB;
      }
      VT win[K][K];
      auto load_row = [&](int h, int slot) {
        const bool okh = (unsigned)h < (unsigned)p.H;
        const unsigned ro = (unsigned)h * rowb;
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
          win[slot][kw] = Vec<V>::ld(xr, (okh && col[kw] != DWM_OOB) ? col[kw] + ro : DWM_OOB);
      };
#pragma unroll
      for (int kh = 0; kh < K - 2; ++kh) load_row(2 * c.r0 - P + kh, kh + 2);
      const int h1 = c.r0 + p.HC < p.Ho ? c.r0 + p.HC : p.Ho;
      unsigned goff = ((unsigned)(((c.n * p.T + c.t) * p.Ho + c.r0) * p.Wo + c.w) * (unsigned)p.g_cs +
                       (unsigned)(p.g_coff + ch)) * 4u;
      for (int ho = c.r0; ho < h1; ++ho, goff += growb) {
#pragma unroll
        for (int kh = 0; kh < K - 2; ++kh)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) win[kh][kw] = win[kh + 2][kw];
        load_row(2 * ho - P + K - 2, K - 2);
        load_row(2 * ho - P + K - 1, K - 1);
        const VT g = Vec<V>::ld(gr, goff);
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) acc[kh][kw] += g * win[kh][kw];
      }
    }
  }
  // the lanes rl = 0 .. rpi-1 of a channel group, summed in lane order, one kernel row (K taps) per round through LDS
#pragma unroll
  for (int kh = 0; kh < K; ++kh) {
#pragma unroll
    for (int kw = 0; kw < K; ++kw) red[kw][threadIdx.x] = acc[kh][kw];
    __syncthreads();
    if (rl == 0 && cok) {
      for (int u = 0; u < K; ++u) {
        VT tot = Vec<V>::zero();
        for (int i = 0; i < rpi; ++i) tot += red[u][i * p.CQ + ql];
        Vec<V>::stp(p.partial + ((long)blockIdx.x * (K * K) + kh * K + u) * p.C + ch, tot);
      }
    }
    __syncthreads();
  }
}

inline int pow2ceil_m(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// rows per march: long enough that the two halo rows of a march are a small part of its loads, short enough that the
// layer still has a few thousand wavefronts
int march_rows(int H, long columns, int groups) {
  int hc = H;
  while (hc > 8 && columns * sf_cdiv(H, hc) * groups < 64L * 4 * 256 * 4) hc = (hc + 1) / 2;
  // (shorter marches for tiny layers — down to 2 rows below 32 768 lanes — measured neutral on cfg #1: 9.70 vs 9.71 ms)
  return hc < 1 ? 1 : hc;
}

bool g_dwm_on = [] {
  const char* e = getenv("SF_DW_MARCH");
  return !(e && e[0] == '0');
}();

bool shape_ok(const sf_conv_desc* d) {
  if (!g_dwm_on) return false;
  if (!(d->kT == 1 || d->kT == 3) || d->kH != 3 || d->kW != 3) return false;
  if (d->sT != 1 || d->sH != 1 || d->sW != 1 || d->dT != 1 || d->dH != 1 || d->dW != 1) return false;
  if (d->pT != d->kT / 2 || d->pH != 1 || d->pW != 1) return false;
  if (d->To != d->Ti || d->Ho != d->Hi || d->Wo != d->Wi) return false;
  if (d->N <= 0 || d->Ti <= 0 || d->Hi <= 0 || d->Wi <= 0) return false;
  return true;
}

// 1 x K x K (K = 3 | 5), stride (1, 2, 2), padding K / 2: the down-sampling depthwise layers
bool g_dwm2_on = [] {
  const char* e = getenv("SF_DW_MARCH_S2");
  return !(e && e[0] == '0');
}();
bool shape2_ok(const sf_conv_desc* d) {
  if (!g_dwm_on || !g_dwm2_on) return false;
  if (d->kT != 1 || d->kH != d->kW || !(d->kH == 3 || d->kH == 5)) return false;
  if (d->sT != 1 || d->sH != 2 || d->sW != 2 || d->dT != 1 || d->dH != 1 || d->dW != 1) return false;
  if (d->pT != 0 || d->pH != d->kH / 2 || d->pW != d->kW / 2) return false;
  if (d->To != d->Ti || d->Ho != (d->Hi + 2 * d->pH - d->kH) / 2 + 1 || d->Wo != (d->Wi + 2 * d->pW - d->kW) / 2 + 1)
    return false;
  if (d->N <= 0 || d->Ti <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  return true;
}
bool fits32o(const sf_conv_desc* d, int cs) {
  return (long)d->N * d->To * d->Ho * d->Wo * cs * 4 < 0x7fffffffL;
}

bool fits32(const sf_conv_desc* d, int cs) {
  return (long)d->N * d->Ti * d->Hi * d->Wi * cs * 4 < 0x7fffffffL;
}

void fill_geometry(DwmArgs& a, const sf_conv_desc* d, int C, int V) {
  a.N = d->N; a.T = d->Ti; a.H = d->Hi; a.W = d->Wi; a.C = C;
  const int groups = sf_cdiv(C, V);
  a.CQ = pow2ceil_m(groups) < 64 ? pow2ceil_m(groups) : 64;
  a.HC = march_rows(a.H, (long)a.N * a.T * a.W, groups);
  a.nhc = sf_cdiv(a.H, a.HC);
  a.units = (long)a.N * a.T * a.nhc * a.W;
}

// marched rows / columns of a strided launch: forward and weight gradient march over the OUTPUT rows of output columns,
// the data gradient over the row pairs b of input columns
void fill_geometry2(DwmArgs& a, const sf_conv_desc* d, int C, int V, bool dgrad) {
  a.N = d->N; a.T = d->Ti; a.H = d->Hi; a.W = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo; a.C = C;
  const int groups = sf_cdiv(C, V);
  a.CQ = pow2ceil_m(groups) < 64 ? pow2ceil_m(groups) : 64;
  const int rows = dgrad ? (d->Hi - 1 + d->pH) / 2 + 1 : d->Ho;
  const int wn = dgrad ? d->Wi : d->Wo;
  a.HC = march_rows(rows, (long)a.N * a.T * wn, groups);
  a.nhc = sf_cdiv(rows, a.HC);
  a.units = (long)a.N * a.T * a.nhc * wn;
}

template <int V>
int launch2_fwd(const DwmArgs& a, int K, bool dgrad, hipStream_t s) {
  const int rpi = DWM_TPB / a.CQ;
  const dim3 grid(sf_cdiv(a.units, rpi), sf_cdiv(sf_cdiv(a.C, V), a.CQ));
  if (dgrad) {
    if (K == 5) hipLaunchKernelGGL((dwm2_dgrad_kernel<5, V>), grid, dim3(DWM_TPB), 0, s, a);
    else hipLaunchKernelGGL((dwm2_dgrad_kernel<3, V>), grid, dim3(DWM_TPB), 0, s, a);
  } else {
    if (K == 5) hipLaunchKernelGGL((dwm2_fwd_kernel<5, V>), grid, dim3(DWM_TPB), 0, s, a);
    else hipLaunchKernelGGL((dwm2_fwd_kernel<3, V>), grid, dim3(DWM_TPB), 0, s, a);
  }
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int V>
int launch_fwd(const DwmArgs& a, int kT, hipStream_t s) {
  const int rpi = DWM_TPB / a.CQ;
  const dim3 grid(sf_cdiv(a.units, rpi), sf_cdiv(sf_cdiv(a.C, V), a.CQ));
  if (kT == 3)
    hipLaunchKernelGGL((dwm_fwd_kernel<3, V>), grid, dim3(DWM_TPB), 0, s, a);
  else
    hipLaunchKernelGGL((dwm_fwd_kernel<1, V>), grid, dim3(DWM_TPB), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

}  // namespace

// sf_conv_tune(30, 0 | 1): the row-march depthwise kernels off / on (A/B runs and the tests' comparison with the generic
// kernels)
int sf_dwm_tune(int value) {
  g_dwm_on = value != 0;
  return SF_OK;
}
// sf_conv_tune(31, 0 | 1): the stride-2 marches alone
int sf_dwm2_tune(int value) {
  g_dwm2_on = value != 0;
  return SF_OK;
}

// 1: not taken (the caller runs its generic kernel).  Forward with the conv epilogue.
int sf_dwm_fwd_try(const sf_conv_desc* d, const float* in, const float* w, const float* scale, const float* bias,
                   const float* res, float* out, hipStream_t s) {
  const bool s2 = shape2_ok(d);
  if ((!shape_ok(d) && !s2) || d->out_cmul != 1 || d->Cout > d->Cin) return 1;
  if (!fits32(d, d->in_cs) || !fits32(d, d->out_cs) || (res && !fits32(d, d->res_cs))) return 1;
  if (d->N * d->Ti * d->Hi * d->Wi <= 0) return 1;
  DwmArgs a = {};
  a.x = in; a.w = w; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.x_cs = d->in_cs; a.x_coff = d->in_coff; a.out_cs = d->out_cs; a.out_coff = d->out_coff;
  a.res_cs = d->res_cs; a.res_coff = d->res_coff; a.wpitch = d->cin_pad;
  a.x_bytes = (unsigned)((long)d->N * d->Ti * d->Hi * d->Wi * d->in_cs * 4);
  a.act = d->act; a.accumulate = 0; a.flip = 0;
  const bool v4 = (d->Cin % 4 == 0) && (d->Cout % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) &&
                  (d->out_cs % 4 == 0) && (d->out_coff % 4 == 0) && (d->cin_pad % 4 == 0) && sf_aligned16(in) &&
                  sf_aligned16(w) && sf_aligned16(out) && (!scale || (sf_aligned16(scale) && sf_aligned16(bias))) &&
                  (!res || ((d->res_cs % 4 == 0) && (d->res_coff % 4 == 0) && sf_aligned16(res)));
  if (s2) {
    fill_geometry2(a, d, d->Cout, v4 ? 4 : 1, false);
    return v4 ? launch2_fwd<4>(a, d->kH, false, s) : launch2_fwd<1>(a, d->kH, false, s);
  }
  fill_geometry(a, d, d->Cout, v4 ? 4 : 1);   // channels >= Cout are not stored: not computed either
  return v4 ? launch_fwd<4>(a, d->kT, s) : launch_fwd<1>(a, d->kT, s);
}

// dx (+)= the transposed conv of dz.  `d` is the forward descriptor.
int sf_dwm_dgrad_try(const sf_conv_desc* d, const float* dz, int dz_cs, int dz_coff, const float* w, float* dx,
                     int dx_cs, int dx_coff, int C, int accumulate, hipStream_t s) {
  const bool s2 = shape2_ok(d);
  if ((!shape_ok(d) && !s2) || !fits32(d, dz_cs) || !fits32(d, dx_cs)) return 1;
  DwmArgs a = {};
  a.x = dz; a.w = w; a.out = dx;
  a.x_cs = dz_cs; a.x_coff = dz_coff; a.out_cs = dx_cs; a.out_coff = dx_coff; a.wpitch = d->cin_pad;
  a.x_bytes = (unsigned)((long)d->N * d->To * d->Ho * d->Wo * dz_cs * 4);
  a.act = SF_ACT_NONE; a.accumulate = accumulate; a.flip = 1;
  const bool v4 = (C % 4 == 0) && (dz_cs % 4 == 0) && (dz_coff % 4 == 0) && (dx_cs % 4 == 0) && (dx_coff % 4 == 0) &&
                  (d->cin_pad % 4 == 0) && sf_aligned16(dz) && sf_aligned16(dx) && sf_aligned16(w);
  if (s2) {
    fill_geometry2(a, d, C, v4 ? 4 : 1, true);
    return v4 ? launch2_fwd<4>(a, d->kH, true, s) : launch2_fwd<1>(a, d->kH, true, s);
  }
  fill_geometry(a, d, C, v4 ? 4 : 1);
  return v4 ? launch_fwd<4>(a, d->kT, s) : launch_fwd<1>(a, d->kT, s);
}

// partial[*nblk][taps][C] of the weight gradient into ws (room for `max_blk` blocks); the caller sums the blocks.
int sf_dwm_wgrad_try(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, int C, float* ws,
                     int max_blk, int* nblk, hipStream_t s) {
  const bool s2 = shape2_ok(d);
  if ((!shape_ok(d) && !s2) || !fits32(d, d->in_cs) || !fits32(d, dz_cs) || max_blk < 1) return 1;
  DwmArgs a = {};
  a.x = x; a.g = dz; a.partial = ws;
  a.x_cs = d->in_cs; a.x_coff = d->in_coff; a.g_cs = dz_cs; a.g_coff = dz_coff;
  a.x_bytes = (unsigned)((long)d->N * d->Ti * d->Hi * d->Wi * d->in_cs * 4);
  a.g_bytes = (unsigned)((long)d->N * d->To * d->Ho * d->Wo * dz_cs * 4);
  const bool v4 = (C % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && (dz_cs % 4 == 0) &&
                  (dz_coff % 4 == 0) && sf_aligned16(x) && sf_aligned16(dz) && sf_aligned16(ws);
  if (s2) fill_geometry2(a, d, C, v4 ? 4 : 1, false);
  else fill_geometry(a, d, C, v4 ? 4 : 1);
  const int V = v4 ? 4 : 1;
  const int ncb = sf_cdiv(sf_cdiv(C, V), a.CQ);
  const int rpi = DWM_TPB / a.CQ;
  long nb = 2048 / ncb;                               // ~8 workgroups per CU in all
  if (nb > max_blk) nb = max_blk;
  if (nb > a.units / rpi) nb = a.units / rpi;         // at least one column piece per lane
  if (nb < 1) nb = 1;
  a.nblk = (int)nb;
  *nblk = a.nblk;
  const dim3 grid(a.nblk, ncb);
  if (s2) {
    if (d->kH == 5) {
      if (v4) hipLaunchKernelGGL((dwm2_wgrad_kernel<5, 4>), grid, dim3(DWM_TPB), 0, s, a);
      else hipLaunchKernelGGL((dwm2_wgrad_kernel<5, 1>), grid, dim3(DWM_TPB), 0, s, a);
    } else {
      if (v4) hipLaunchKernelGGL((dwm2_wgrad_kernel<3, 4>), grid, dim3(DWM_TPB), 0, s, a);
      else hipLaunchKernelGGL((dwm2_wgrad_kernel<3, 1>), grid, dim3(DWM_TPB), 0, s, a);
    }
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
  if (d->kT == 3) {
    if (v4) hipLaunchKernelGGL((dwm_wgrad_kernel<3, 4>), grid, dim3(DWM_TPB), 0, s, a);
    else hipLaunchKernelGGL((dwm_wgrad_kernel<3, 1>), grid, dim3(DWM_TPB), 0, s, a);
  } else {
    if (v4) hipLaunchKernelGGL((dwm_wgrad_kernel<1, 4>), grid, dim3(DWM_TPB), 0, s, a);
    else hipLaunchKernelGGL((dwm_wgrad_kernel<1, 1>), grid, dim3(DWM_TPB), 0, s, a);
  }
  SF_CHECK_LAUNCH();
  return SF_OK;
}
