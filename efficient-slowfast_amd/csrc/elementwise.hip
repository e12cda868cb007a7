// elementwise.hip — the bandwidth-bound side of the SlowFast / CMDA path on gfx950 (NDHWC, fp32):
// layout conversion, pooling, depthwise convolution, the ECA / SqueezeExcite channel gates, the head's
// activation+mean and channel-slice copies.  All of these are HBM-bound (<= 10 FLOP/B): the design rule
// is one pass over the data, channel-contiguous (coalesced) accesses, 16-byte vectors whenever the
// channel count / pitch / offset are multiples of 4, and outputs written straight into the consumer's
// (possibly wider, concatenated) tensor.
#include "common.h"

namespace {

constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------------
// NCTHW -> padded NDHWC.  One thread per destination pixel; reads are coalesced along W per channel
// plane, writes are cpad contiguous floats (one 16-byte store when cpad == 4).
__global__ void ncthw_to_ndhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C,
                                      int T, int H, int W, int cpad, int ph, int pw, int Hp, int Wp, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int wp = (int)(idx % Wp);
  long r = idx / Wp;
  const int hp = (int)(r % Hp);
  r /= Hp;
  const int t = (int)(r % T);
  const int n = (int)(r / T);
  const int h = hp - ph, w = wp - pw;
  const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
  float* o = dst + idx * cpad;
  const long plane = (long)T * H * W;
  const float* s = src + ((long)n * C * T + t) * H * W + (long)h * W + w;
  if (cpad == 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (in) {
      v[0] = s[0];
      if (C > 1) v[1] = s[plane];
      if (C > 2) v[2] = s[2 * plane];
      if (C > 3) v[3] = s[3 * plane];
    }
    *reinterpret_cast<f32x4*>(o) = v;
  } else {
    for (int c = 0; c < cpad; ++c) o[c] = (in && c < C) ? s[c * plane] : 0.f;
  }
}

__global__ void ndhwc_to_ncthw_kernel(const float* __restrict__ src, int cs, int coff, float* __restrict__ dst,
                                      int C, long thw, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const long pos = idx % thw;
  const long r = idx / thw;
  const int c = (int)(r % C);
  const long n = r / C;
  dst[idx] = src[(n * thw + pos) * cs + coff + c];
}

// ------------------------------------------------------------------------------------------------
// Generic 3-D max / average pooling, thread per (output position, VEC channels).
template <int VEC>
__global__ void pool_kernel(const sf_pool_desc d, const float* __restrict__ in, float* __restrict__ out,
                            long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int cv = (d.C + VEC - 1) / VEC;
  const int c = (int)(idx % cv) * VEC;
  long r = idx / cv;
  const int wo = (int)(r % d.Wo);
  r /= d.Wo;
  const int ho = (int)(r % d.Ho);
  r /= d.Ho;
  const int to = (int)(r % d.To);
  const int n = (int)(r / d.To);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = d.is_avg ? 0.f : -3.0e38f;
  for (int kt = 0; kt < d.kT; ++kt) {
    const int ti = to * d.sT - d.pT + kt;
    if ((unsigned)ti >= (unsigned)d.Ti) continue;
    for (int kh = 0; kh < d.kH; ++kh) {
      const int hi = ho * d.sH - d.pH + kh;
      if ((unsigned)hi >= (unsigned)d.Hi) continue;
      for (int kw = 0; kw < d.kW; ++kw) {
        const int wi = wo * d.sW - d.pW + kw;
        if ((unsigned)wi >= (unsigned)d.Wi) continue;
        const float* s = in + ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.in_cs + d.in_coff + c;
        if (VEC == 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(s);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = d.is_avg ? acc[e] + v[e] : fmaxf(acc[e], v[e]);
        } else {
          acc[0] = d.is_avg ? acc[0] + s[0] : fmaxf(acc[0], s[0]);
        }
      }
    }
  }
  const float inv = d.is_avg ? 1.f / (float)(d.kT * d.kH * d.kW) : 1.f;
  float* o = out + ((((long)n * d.To + to) * d.Ho + ho) * d.Wo + wo) * d.out_cs + d.out_coff + c;
  if (VEC == 4) {
    *reinterpret_cast<f32x4*>(o) = (f32x4){acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
  } else {
    o[0] = acc[0] * inv;
  }
}

// Max pooling that also records WHICH tap won (first maximum in (kt, kh, kw) scan order, nn.MaxPool3d's rule) as one
// byte per output element, [positions][C] dense: the backward pass then needs neither x nor y and no tie search
// (sf_maxpool_bwd_arg: 335 -> 70 us on the Slow pathway's pool1 at 8 clips).  Four channels per thread.
__global__ void pool_arg_kernel(const sf_pool_desc d, const float* __restrict__ in, float* __restrict__ out,
                                unsigned char* __restrict__ arg, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int cv = d.C >> 2;
  const int c = (int)(idx % cv) * 4;
  const long row = idx / cv;
  long r = row;
  const int wo = (int)(r % d.Wo);
  r /= d.Wo;
  const int ho = (int)(r % d.Ho);
  r /= d.Ho;
  const int to = (int)(r % d.To);
  const int n = (int)(r / d.To);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned win = 0;  // four tap indices, one byte each
  bool first = true;
  for (int kt = 0; kt < d.kT; ++kt) {
    const int ti = to * d.sT - d.pT + kt;
    if ((unsigned)ti >= (unsigned)d.Ti) continue;
    for (int kh = 0; kh < d.kH; ++kh) {
      const int hi = ho * d.sH - d.pH + kh;
      if ((unsigned)hi >= (unsigned)d.Hi) continue;
      for (int kw = 0; kw < d.kW; ++kw) {
        const int wi = wo * d.sW - d.pW + kw;
        if ((unsigned)wi >= (unsigned)d.Wi) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            in + ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.in_cs + d.in_coff + c);
        const unsigned tap = (unsigned)((kt * d.kH + kh) * d.kW + kw);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (first || v[e] > acc[e]) {
            acc[e] = v[e];
            win = (win & ~(0xffu << (8 * e))) | (tap << (8 * e));
          }
        first = false;
      }
    }
  }
  *reinterpret_cast<f32x4*>(out + row * d.out_cs + d.out_coff + c) = acc;
  *reinterpret_cast<unsigned*>(arg + row * d.C + c) = win;
}

// dx[input element] (+)= sum of dy over the windows that recorded this element as their winner.
__global__ __launch_bounds__(256) void maxpool_bwd_arg_kernel(const sf_pool_desc d, const unsigned char* __restrict__ arg,
                                                              const float* __restrict__ dy, int dy_cs, int dy_coff,
                                                              float* __restrict__ dx, int dx_cs, int dx_coff, int acc,
                                                              unsigned total) {
  const unsigned idx = blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const unsigned cv = d.C >> 2;
  const unsigned rin = idx / cv;
  const int c = (int)(idx - rin * cv) * 4;
  unsigned r = rin;
  const int wi = (int)(r % (unsigned)d.Wi);
  r /= (unsigned)d.Wi;
  const int hi = (int)(r % (unsigned)d.Hi);
  r /= (unsigned)d.Hi;
  const int ti = (int)(r % (unsigned)d.Ti);
  const int n = (int)(r / (unsigned)d.Ti);
  auto lo = [](int i, int p, int k, int s) { const int a = i + p - k + 1; return a <= 0 ? 0 : (a + s - 1) / s; };
  const int t0 = lo(ti, d.pT, d.kT, d.sT), t1 = min((ti + d.pT) / d.sT, d.To - 1);
  const int h0 = lo(hi, d.pH, d.kH, d.sH), h1 = min((hi + d.pH) / d.sH, d.Ho - 1);
  const int w0 = lo(wi, d.pW, d.kW, d.sW), w1 = min((wi + d.pW) / d.sW, d.Wo - 1);
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  for (int to = t0; to <= t1; ++to)
    for (int ho = h0; ho <= h1; ++ho)
      for (int wo = w0; wo <= w1; ++wo) {
        const long ro = (((long)n * d.To + to) * d.Ho + ho) * d.Wo + wo;
        const unsigned tap = (unsigned)(((ti - (to * d.sT - d.pT)) * d.kH + (hi - (ho * d.sH - d.pH))) * d.kW +
                                        (wi - (wo * d.sW - d.pW)));
        const unsigned a = *reinterpret_cast<const unsigned*>(arg + ro * d.C + c);
        const unsigned x4 = a ^ (tap * 0x01010101u);  // a zero byte = this element won that channel of the window
        if (((x4 - 0x01010101u) & ~x4 & 0x80808080u) != 0u) {
          const f32x4 gv = *reinterpret_cast<const f32x4*>(dy + ro * dy_cs + dy_coff + c);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (((x4 >> (8 * e)) & 0xffu) == 0u) g[e] += gv[e];
        }
      }
  f32x4* o = reinterpret_cast<f32x4*>(dx + (long)rin * dx_cs + dx_coff + c);
  *o = acc ? *o + g : g;
}

// ------------------------------------------------------------------------------------------------
// Depthwise convolution + folded BN + residual + ReLU, thread per (output position, VEC channels).
// Neighbouring threads share taps through L1/L2; the op moves ~2 floats per MAC and is HBM-bound.
template <int VEC>
__global__ void dwconv_kernel(const sf_conv_desc d, const float* __restrict__ in, const float* __restrict__ w,
                              const float* __restrict__ scale, const float* __restrict__ bias,
                              const float* __restrict__ res, float* __restrict__ out, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int cv = (d.Cout + VEC - 1) / VEC;
  const int c = (int)(idx % cv) * VEC;
  long r = idx / cv;
  const long m = r;
  const int wo = (int)(r % d.Wo);
  r /= d.Wo;
  const int ho = (int)(r % d.Ho);
  r /= d.Ho;
  const int to = (int)(r % d.To);
  const int n = (int)(r / d.To);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  int tap = 0;
  for (int kt = 0; kt < d.kT; ++kt) {
    const int ti = to * d.sT - d.pT + kt * d.dT;
    for (int kh = 0; kh < d.kH; ++kh) {
      const int hi = ho * d.sH - d.pH + kh * d.dH;
      for (int kw = 0; kw < d.kW; ++kw, ++tap) {
        const int wi = wo * d.sW - d.pW + kw * d.dW;
        if ((unsigned)ti >= (unsigned)d.Ti || (unsigned)hi >= (unsigned)d.Hi || (unsigned)wi >= (unsigned)d.Wi)
          continue;
        const float* s = in + ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.in_cs + d.in_coff + c;
        const float* wp = w + (long)tap * d.cin_pad + c;
        if (VEC == 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(s);
          const f32x4 ww = *reinterpret_cast<const f32x4*>(wp);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(v[e], ww[e], acc[e]);
        } else {
          acc[0] = fmaf(s[0], wp[0], acc[0]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    float v = acc[e];
    if (scale) v = v * scale[c + e] + bias[c + e];
    if (res) v += res[m * d.res_cs + d.res_coff + c + e];
    v = sf_act(v, d.act);
    out[m * d.out_cs + d.out_coff + (long)(c + e) * d.out_cmul] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// ECA / SE squeeze: partial[b][blk][c] = sum over the block's rows of max_{r<alpha} x[b, t*alpha+r, hw, c]
constexpr int POOL_P = 64;  // partial blocks per batch element (fixed => bit-reproducible sums)

__global__ void tmax_partial_kernel(const float* __restrict__ x, int cs, int coff, int T, int HW, int C,
                                    int alpha, int CB, float* __restrict__ partial) {
  __shared__ float red[TPB];
  const int blk = blockIdx.x, cb = blockIdx.y, b = blockIdx.z;
  const int cl = threadIdx.x % CB, rl = threadIdx.x / CB, rpi = TPB / CB;
  const int c = cb * CB + cl;
  const int To = T / alpha;
  const long rows = (long)To * HW;
  const long per = (rows + POOL_P - 1) / POOL_P;
  const long r0 = (long)blk * per;
  const long r1 = (r0 + per < rows) ? r0 + per : rows;
  float sum = 0.f;
  if (c < C) {
    for (long r = r0 + rl; r < r1; r += rpi) {
      const long t = r / HW, hw = r - t * HW;
      const float* s = x + (((long)b * T + t * alpha) * HW + hw) * cs + coff + c;
      float mx = s[0];
      for (int a = 1; a < alpha; ++a) mx = fmaxf(mx, s[(long)a * HW * cs]);
      sum += mx;
    }
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  if (rl == 0 && c < C) {
    float tot = 0.f;
    for (int i = 0; i < rpi; ++i) tot += red[i * CB + cl];
    partial[((long)b * POOL_P + blk) * C + c] = tot;
  }
}

__global__ void tmax_final_kernel(const float* __restrict__ partial, int C, float inv_count,
                                  float* __restrict__ pooled) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= C) return;
  float tot = 0.f;
  for (int i = 0; i < POOL_P; ++i) tot += partial[((long)b * POOL_P + i) * C + c];
  pooled[(long)b * C + c] = tot * inv_count;
}

// out = act(scale * (max_r x * gate) + bias); gate from ECA's 3-tap conv over channels or SE's hard-sigmoid.
template <int VEC>
__global__ void gate_apply_kernel(const float* __restrict__ x, int cs, int coff, int T, int HW, int C, int alpha,
                                  const float* __restrict__ pooled, const float* __restrict__ w3,
                                  const float* __restrict__ scale, const float* __restrict__ bias, int act,
                                  float* __restrict__ out, int out_cs, int out_coff) {
  extern __shared__ float gsm[];  // gate[C], eff_scale[C], eff_bias[C]
  float* gate = gsm;
  const int b = blockIdx.y;
  const float* pb = pooled + (long)b * C;
  for (int c = threadIdx.x; c < C; c += TPB) {
    float g;
    if (w3) {
      const float l = c > 0 ? pb[c - 1] : 0.f;
      const float r = c + 1 < C ? pb[c + 1] : 0.f;
      g = 1.f / (1.f + __expf(-(w3[0] * l + w3[1] * pb[c] + w3[2] * r)));
    } else {
      g = fminf(fmaxf(pb[c] + 3.f, 0.f), 6.f) * (1.f / 6.f);
    }
    gate[c] = g;
  }
  __syncthreads();
  const int cv = (C + VEC - 1) / VEC;
  const int To = T / alpha;
  const long total = (long)To * HW * cv;
  for (long idx = (long)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (long)gridDim.x * TPB) {
    const int c = (int)(idx % cv) * VEC;
    const long r = idx / cv;
    const long t = r / HW, hw = r - t * HW;
    const float* s = x + (((long)b * T + t * alpha) * HW + hw) * cs + coff + c;
    float* o = out + ((long)b * To * HW + r) * out_cs + out_coff + c;
    if (VEC == 4) {
      f32x4 mx = *reinterpret_cast<const f32x4*>(s);
      for (int a = 1; a < alpha; ++a) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(s + (long)a * HW * cs);
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[e]);
      }
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = mx[e] * gate[c + e];
        if (scale) v = v * scale[c + e] + bias[c + e];
        y[e] = (act == SF_ACT_RELU) ? fmaxf(v, 0.f) : v;
      }
      *reinterpret_cast<f32x4*>(o) = y;
    } else {
      float mx = s[0];
      for (int a = 1; a < alpha; ++a) mx = fmaxf(mx, s[(long)a * HW * cs]);
      float v = mx * gate[c];
      if (scale) v = v * scale[c] + bias[c];
      o[0] = (act == SF_ACT_RELU) ? fmaxf(v, 0.f) : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// head tail: out[b,k] = mean_p act(logits[b,p,:])[k]
__global__ void head_act_mean_kernel(const float* __restrict__ logits, int P, int K, int act,
                                     float* __restrict__ out) {
  __shared__ float red[TPB];
  const int b = blockIdx.x;
  const float invP = 1.f / (float)P;
  for (int k = threadIdx.x; k < K; k += TPB) out[(long)b * K + k] = 0.f;
  for (int pi = 0; pi < P; ++pi) {
    const float* l = logits + ((long)b * P + pi) * K;
    float mx = -3.0e38f, sum = 1.f;
    if (act == SF_ACT_SOFTMAX) {
      for (int k = threadIdx.x; k < K; k += TPB) mx = fmaxf(mx, l[k]);
      red[threadIdx.x] = mx;
      __syncthreads();
      for (int s = TPB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
      }
      mx = red[0];
      __syncthreads();
      float part = 0.f;
      for (int k = threadIdx.x; k < K; k += TPB) part += expf(l[k] - mx);
      red[threadIdx.x] = part;
      __syncthreads();
      for (int s = TPB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
      }
      sum = red[0];
      __syncthreads();
    }
    for (int k = threadIdx.x; k < K; k += TPB) {
      float v = l[k];
      if (act == SF_ACT_SOFTMAX) v = expf(v - mx) / sum;
      else if (act == SF_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
      else if (act == SF_ACT_RELU) v = fmaxf(v, 0.f);
      out[(long)b * K + k] += v * invP;
    }
  }
}

__global__ void copy_channels_kernel(const float* __restrict__ in, int in_cs, int in_coff, float* __restrict__ out,
                                     int out_cs, int out_coff, int out_cmul, int C, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long r = idx / C;
  out[r * out_cs + out_coff + (long)c * out_cmul] = in[r * in_cs + in_coff + c];
}

// channel_shuffle(x, G): out[.., c] (+)= in[.., (c % G) * (C / G) + c / G] — one thread per OUTPUT element (coalesced
// stores; the reads stay inside the row's C * 4 bytes).  The inverse permutation is the same kernel with G' = C / G.
__global__ void channel_shuffle_kernel(const float* __restrict__ in, int in_cs, int in_coff, float* __restrict__ out,
                                       int out_cs, int out_coff, int G, int C, int accumulate, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long r = idx / C;
  const float v = in[r * in_cs + in_coff + (c % G) * (C / G) + c / G];
  float* const o = out + r * out_cs + out_coff + c;
  *o = accumulate ? *o + v : v;
}


// ------------------------------------------------------------------------------------------------
// Training-mode BN statistics: partial[blk][{sum,sumsq}][c] over the block's rows (fp32), combined in fp64.
// Sums are taken of (x - K[c]) with K = the channel's first sample (shifted-data variance): E[d^2] - E[d]^2 then
// cancels at the scale of (mean-K)^2 / var ~ O(1) instead of mean^2 / var, which for a channel with |mean| >> std
// would cost several digits of the variance in fp32.
constexpr int STAT_MAX_P = 1024;

// Final step of the statistics for output o = split * C + channel from the fp64 sums of (x - K), (x - K)^2
// (nn.BatchNorm3d training semantics: biased variance to normalise, unbiased for running_var).
struct StatsFinal {
  double inv_rows, unbias;
  float* mean; float* var;
  const float* gamma; const float* beta;
  float eps, momentum;
  float* run_mean; float* run_var;
  float* invstd; float* scale; float* shift;
  unsigned* tickets;   // != NULL: fused launch — one counter per channel group
};

__device__ __forceinline__ void stats_finish(const StatsFinal& f, int o, double s1, double s2, float k) {
  const double md = s1 * f.inv_rows;  // mean of (x - K)
  double v = s2 * f.inv_rows - md * md;
  if (v < 0.0) v = 0.0;
  const double m = md + (double)k;
  f.mean[o] = (float)m;
  f.var[o] = (float)v;
  if (f.gamma) {
    const float is = (float)(1.0 / sqrt(v + (double)f.eps));
    const float sc = f.gamma[o] * is;
    f.invstd[o] = is;
    f.scale[o] = sc;
    f.shift[o] = f.beta[o] - (float)m * sc;
    if (f.run_mean) {
      f.run_mean[o] = (1.f - f.momentum) * f.run_mean[o] + f.momentum * (float)m;
      f.run_var[o] = (1.f - f.momentum) * f.run_var[o] + f.momentum * (float)(v * f.unbias);
    }
  }
}

template <int VEC>
__global__ void stats_partial_kernel(const float* __restrict__ x, int cs, int coff, long group_rows, int chunks,
                                     int S, int C, int CB, float* __restrict__ partial, const StatsFinal fin) {
  // block -> (row group n, chunk): a group is the whole tensor (S == 1) or one sample (Sub-BN, split = n % S)
  __shared__ float red[2 * TPB * VEC];
  const int blk = blockIdx.x, cb = blockIdx.y;
  const int cl = threadIdx.x % CB, rl = threadIdx.x / CB, rpi = TPB / CB;
  const int c = (cb * CB + cl) * VEC;
  const int n = blk / chunks, ck = blk - n * chunks;
  const long per = (group_rows + chunks - 1) / chunks;
  const long gend = (long)(n + 1) * group_rows;
  const long r0 = (long)n * group_rows + (long)ck * per;
  const long r1 = (r0 + per < gend) ? r0 + per : gend;
  const long krow = (long)(n % S) * group_rows;  // first row of this split
  float s1[VEC], s2[VEC], k[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; k[e] = 0.f; }
  if (c < C) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) k[e] = x[krow * cs + coff + c + e];
    long r = r0 + rl;
    if constexpr (VEC == 4) {  // four rows per trip, all four loads issued before the first use
      for (; r + 3 * (long)rpi < r1; r += 4 * (long)rpi) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (r + (long)u * rpi) * cs + coff + c);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = v[u][e] - k[e];
            s1[e] += d;
            s2[e] = fmaf(d, d, s2[e]);
          }
      }
    }
    for (; r < r1; r += rpi) {
      if (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * cs + coff + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - k[e];
          s1[e] += d;
          s2[e] = fmaf(d, d, s2[e]);
        }
      } else {
        const float d = x[r * cs + coff + c] - k[0];
        s1[0] += d;
        s2[0] = fmaf(d, d, s2[0]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    red[threadIdx.x * VEC + e] = s1[e];
    red[(TPB + threadIdx.x) * VEC + e] = s2[e];
  }
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t1 = 0.f, t2 = 0.f;
      for (int i = 0; i < rpi; ++i) {
        t1 += red[(i * CB + cl) * VEC + e];
        t2 += red[(TPB + i * CB + cl) * VEC + e];
      }
      if (fin.tickets) {
        sf_store_sc1(partial + ((long)blk * 2 + 0) * C + c + e, t1);
        sf_store_sc1(partial + ((long)blk * 2 + 1) * C + c + e, t2);
      } else {
        partial[((long)blk * 2 + 0) * C + c + e] = t1;
        partial[((long)blk * 2 + 1) * C + c + e] = t2;
      }
    }
  }
  if (!fin.tickets) return;
  // ---- fused final step: the last workgroup of this channel group sums the P partials of its channels in fp64.
  // Four lanes per output (split, channel) take the partials i = q, q + 4, ..; the four sub-sums are combined in lane
  // order — a fixed summation order, as bit-reproducible as the two-launch form.
  if (!sf_ticket_last(fin.tickets + cb, gridDim.x)) return;
  const int P = gridDim.x;
  const int cbase = cb * CB * VEC;
  const int ncg = (C - cbase) < CB * VEC ? (C - cbase) : CB * VEC;
  const int q = threadIdx.x & 3;
  for (int item = threadIdx.x >> 2; item < S * ncg; item += TPB / 4) {
    const int sp = item / ncg, cc = cbase + (item - sp * ncg);
    double s1 = 0.0, s2 = 0.0;
    for (int i = q; i < P; i += 4) {
      if ((i / chunks) % S != sp) continue;
      s1 += (double)partial[((long)i * 2 + 0) * C + cc];
      s2 += (double)partial[((long)i * 2 + 1) * C + cc];
    }
    s1 += __shfl_xor(s1, 1, 64);
    s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 2, 64);
    if (q == 0) stats_finish(fin, sp * C + cc, s1, s2, x[(long)sp * group_rows * cs + coff + cc]);
  }
}

// One 64-lane block per channel: lanes stride over the P partials in fp64, tree-reduce through LDS.  With
// BN parameters given it also emits the normalisation affine and updates the running statistics in place
// (the two-launch form: SF_BN_TICKET=0).
__global__ void stats_final_kernel(const float* __restrict__ x, int cs, int coff, long group_rows, int chunks, int S,
                                   const float* __restrict__ partial, int C, int P, double inv_rows, double unbias,
                                   float* __restrict__ mean, float* __restrict__ var,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ run_mean, float* __restrict__ run_var,
                                   float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double r1[64], r2[64];
  const int o = blockIdx.x;            // output index = split * C + channel
  const int sp = o / C, c = o - sp * C;
  double s1 = 0.0, s2 = 0.0;
  int i = threadIdx.x;
  if (S == 1) {  // four partials per trip, loads first (see pair_final_kernel, backward.hip)
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    for (; i + 192 < P; i += 256) {
      float t1[4], t2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        t1[u] = partial[((long)(i + 64 * u) * 2 + 0) * C + c];
        t2[u] = partial[((long)(i + 64 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a1[u] += (double)t1[u];
        a2[u] += (double)t2[u];
      }
    }
    s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  }
  for (; i < P; i += 64) {
    if ((i / chunks) % S != sp) continue;
    s1 += (double)partial[((long)i * 2 + 0) * C + c];
    s2 += (double)partial[((long)i * 2 + 1) * C + c];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {  // the workgroup is one wavefront
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  r1[0] = s1;
  r2[0] = s2;
  if (threadIdx.x == 0) {
    const double md = r1[0] * inv_rows;  // mean of (x - K)
    double v = r2[0] * inv_rows - md * md;
    if (v < 0.0) v = 0.0;
    const double m = md + (double)x[(long)sp * group_rows * cs + coff + c];
    mean[o] = (float)m;
    var[o] = (float)v;
    if (gamma) {
      const float is = (float)(1.0 / sqrt(v + (double)eps));
      const float sc = gamma[o] * is;
      invstd[o] = is;
      scale[o] = sc;
      shift[o] = beta[o] - (float)m * sc;
      if (run_mean) {
        run_mean[o] = (1.f - momentum) * run_mean[o] + momentum * (float)m;
        run_var[o] = (1.f - momentum) * run_var[o] + momentum * (float)(v * unbias);
      }
    }
  }
}

// out[n, t*rep + r, hw, c] = act(x[n,t,hw,c] * scale[(n%S)*C + c] + bias[(n%S)*C + c] + res[n,t,hw,c])
template <int VEC>
__global__ void affine_kernel(const float* __restrict__ x, int cs, int coff, long THW_in, int HW, int C, int S,
                              const float* __restrict__ scale, const float* __restrict__ bias,
                              const float* __restrict__ res, int res_cs, int res_coff, int act, int rep,
                              float* __restrict__ out, int out_cs, int out_coff, int out_cmul, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int cv = (C + VEC - 1) / VEC;
  const int c = (int)(idx % cv) * VEC;
  const long r = idx / cv;                 // input row: (n, t, hw)
  const long n = r / THW_in;
  const long thw = r - n * THW_in;
  const long t = thw / HW, hw = thw - t * HW;
  const int so = (int)(n % S) * C;
  float v[VEC];
  if (VEC == 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * cs + coff + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = a[e];
  } else {
    v[0] = x[r * cs + coff + c];
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    float y = v[e];
    if (scale) y = y * scale[so + c + e] + bias[so + c + e];
    if (res) y += res[r * res_cs + res_coff + c + e];
    v[e] = sf_act(y, act);
  }
  for (int q = 0; q < rep; ++q) {
    float* o = out + ((n * (THW_in / HW) * rep + t * rep + q) * HW + hw) * out_cs + out_coff + (long)c * out_cmul;
    if (VEC == 4) {
      *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
    } else {
      o[0] = v[0];
    }
  }
}

// The common case of the above (one scale/bias block, no T repeat, float4 channels): four float4 per thread, all
// loads issued before the first use, 32-bit index arithmetic (the general kernel pays three 64-bit divisions per
// float4 and keeps one load in flight: 3.1 TB/s against 4.5+ here on the 8x8x56x56x256 activations).
__global__ __launch_bounds__(256) void affine_flat_kernel(const float* __restrict__ x, int cs, int coff, int cv,
                                                          int cv_shift, const float* __restrict__ scale,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ res, int res_cs, int res_coff,
                                                          int act, float* __restrict__ out, int out_cs, int out_coff,
                                                          unsigned char* __restrict__ mask, unsigned total) {
  constexpr int U = 4;
  const unsigned base = blockIdx.x * (TPB * U) + threadIdx.x;
  unsigned r[U], c[U];
  f32x4 v[U], q[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned idx = min(base + u * TPB, total - 1);  // clamped lanes re-read the last element, never store
    r[u] = cv_shift >= 0 ? idx >> cv_shift : idx / (unsigned)cv;
    c[u] = (idx - r[u] * cv) * 4;
    v[u] = *reinterpret_cast<const f32x4*>(x + (long)r[u] * cs + coff + c[u]);
    if (res) q[u] = *reinterpret_cast<const f32x4*>(res + (long)r[u] * res_cs + res_coff + c[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (scale) v[u] = v[u] * *reinterpret_cast<const f32x4*>(scale + c[u]) + *reinterpret_cast<const f32x4*>(bias + c[u]);
    if (res) v[u] += q[u];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[u][e] = sf_act(v[u][e], act);
    if (base + u * TPB < total) {
      *reinterpret_cast<f32x4*>(out + (long)r[u] * out_cs + out_coff + c[u]) = v[u];
      if (mask) {  // which of the 4 channels pass a gradient through the activation (bn_bwd's relu == 3 mode)
        unsigned mk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          mk |= ((v[u][e] > 0.f && !(act == SF_ACT_RELU6 && !(v[u][e] < 6.f))) ? 1u : 0u) << e;
        mask[base + u * TPB] = (unsigned char)mk;
      }
    }
  }
}

// Row softmax over the channel dimension (Nonlocal "softmax" instantiation, nonlocal_helper.py:132-134): one
// wavefront per row of C = N_k scores; x <- softmax(scale * x) in place.  bwd: dp <- scale * p * (dp - <p, dp>).
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ x, int cs, int coff, long rows, int C,
                                                          float scale) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  float* row = x + r * cs + coff;
  float m = -3.0e38f;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c] * scale);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float e = __expf(row[c] * scale - m);
    row[c] = e;
    sum += e;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.f / sum;
  for (int c = lane; c < C; c += 64) row[c] *= inv;
}

__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float* __restrict__ p, int p_cs, int p_coff,
                                                              float* __restrict__ dp, int dp_cs, int dp_coff,
                                                              long rows, int C, float scale) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* pr = p + r * p_cs + p_coff;
  float* dr = dp + r * dp_cs + dp_coff;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot = fmaf(pr[c], dr[c], dot);
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  for (int c = lane; c < C; c += 64) dr[c] = scale * pr[c] * (dr[c] - dot);
}

// nn.Conv3d weight [Cout][Cin][taps] -> the two layouts the implicit-GEMM kernels consume, in ONE launch:
//   wp  [Cout][taps][cin_pad]  (forward / weight-gradient order)   and, when asked,
//   wtp [Cin][taps][cout_pad]  (data-gradient order: the transposed conv's "output channel" is Cin),
// zero padded.  Runs once per optimizer step per conv (the packed copies are keyed on the parameter's version).
__global__ void pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, float* __restrict__ wp,
                                   int cin_pad, float* __restrict__ wtp, int cout_pad, long n_wp, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  if (idx < n_wp) {
    const int c = (int)(idx % cin_pad);
    const long r = idx / cin_pad;
    const int tap = (int)(r % taps);
    const int co = (int)(r / taps);
    wp[idx] = c < Cin ? w[((long)co * Cin + c) * taps + tap] : 0.f;
  } else {
    const long j = idx - n_wp;
    const int co = (int)(j % cout_pad);
    const long r = j / cout_pad;
    const int tap = (int)(r % taps);
    const int ci = (int)(r / taps);
    wtp[j] = co < Cout ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
  }
}

// The same for MANY weights in one launch (a training step re-packs every conv once: ~108 launches of ~6 us otherwise).
// items[i] describes one weight; blk0[i] is its first workgroup (blk0[n] = grid size).
struct PackItem {
  const float* w;
  float* wp;
  float* wtp;
  int Cout, Cin, taps, cin_pad, cout_pad, pad_;
  long n_wp, total;
};
// Workgroups of item i (the binding computes the same number, sfhip.h): the forward layout's (below), then one per
// 32 x 32 tile of the transposed layout.
// * wp [Cout][taps][cin_pad]: taps == 1 is a padded copy; 2 <= taps <= 9 (the 3x1x1 / 1x3x3 layers: most of the bytes)
//   goes through LDS per (8 output channels, 32 input channels): ceil(Cout / 8) * ceil(cin_pad / 32) workgroups;
//   the stems' 7 / 35 taps gather element-wise.  32-bit index arithmetic.
// * wtp [Cin][taps][cout_pad] is the plain transpose of the weight seen as W[Cout][Q = Cin * taps] (q = ci * taps + tap
//   is contiguous in the parameter, and (q, co) -> q * cout_pad + co in the packed copy): 32 x 32 tiles through LDS,
//   reads and writes both in 128-byte runs.  (One element per thread with the source at stride Cin * taps * 4 bytes
//   — every lane its own cache line and DRAM page — ran the 137 MB of cfg #3's weights at 1.6 TB/s: 0.26 ms at the
//   head of every training step, in front of the first conv.)
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackItem* __restrict__ items,
                                                                   const int* __restrict__ blk0, int n) {
  __shared__ float lds_buf[8 * (32 * 9 + 1)];  // the forward layout's 8 x (32 * taps + 1) rows, or the 32 x 33 transpose tile
  float (*const tile)[33] = reinterpret_cast<float (*)[33]>(lds_buf);
  int lo = 0, hi = n;  // the item whose block range holds blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const PackItem it = items[lo];
  const unsigned b = blockIdx.x - (unsigned)blk0[lo];
  const unsigned n_wp = (unsigned)it.n_wp;
  const bool rows = it.taps > 1 && it.taps <= 9;
  const unsigned ctiles = ((unsigned)it.cin_pad + 31u) / 32u;
  const unsigned nbw = rows ? (((unsigned)it.Cout + 7u) / 8u) * ctiles : (n_wp + 255u) / 256u;
  if (b < nbw) {
    if (!rows) {  // taps == 1: a padded copy; stems (taps 7 / 35): element-wise gather
      const unsigned idx = b * 256u + threadIdx.x;
      if (idx >= n_wp) return;
      const unsigned r = idx / (unsigned)it.cin_pad, c = idx - r * (unsigned)it.cin_pad;
      const unsigned co = r / (unsigned)it.taps, tap = r - co * (unsigned)it.taps;
      it.wp[idx] = c < (unsigned)it.Cin ? it.w[((long)co * it.Cin + c) * it.taps + tap] : 0.f;
      return;
    }
    // 8 output channels x 32 input channels x taps: the 32 * taps source floats of an output channel are ONE contiguous
    // run; through LDS (row pitch 32 * taps + 1) they leave as `taps` runs of 32 floats
    float* const seg = lds_buf;
    const unsigned cob = b / ctiles, ct = b - cob * ctiles;
    const unsigned lx = threadIdx.x & 31u, ly = threadIdx.x >> 5;
    const unsigned co = cob * 8u + ly, c0 = ct * 32u, taps = (unsigned)it.taps;
    const unsigned pitch = 32u * taps + 1u;
    const unsigned nsrc = (c0 < (unsigned)it.Cin ? ((unsigned)it.Cin - c0 < 32u ? (unsigned)it.Cin - c0 : 32u) : 0u) * taps;
    for (unsigned e = lx; e < 32u * taps; e += 32u)
      seg[ly * pitch + e] = (co < (unsigned)it.Cout && e < nsrc) ? it.w[((long)co * it.Cin + c0) * taps + e] : 0.f;
    __syncthreads();
    if (co < (unsigned)it.Cout && c0 + lx < (unsigned)it.cin_pad)
      for (unsigned tap = 0; tap < taps; ++tap)
        it.wp[((long)co * taps + tap) * it.cin_pad + c0 + lx] = seg[ly * pitch + lx * taps + tap];
    return;
  }
  const unsigned Q = (unsigned)it.Cin * (unsigned)it.taps;
  const unsigned tq = (Q + 31u) / 32u;
  const unsigned t = b - nbw;
  const unsigned tco = t / tq, tqi = t - tco * tq;
  const unsigned q0 = tqi * 32u, co0 = tco * 32u;
  const unsigned lx = threadIdx.x & 31u, ly = threadIdx.x >> 5;  // 32 x 8 threads, four rows each
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned co = co0 + ly + 8u * k, q = q0 + lx;
    tile[ly + 8 * k][lx] = (co < (unsigned)it.Cout && q < Q) ? it.w[(long)co * Q + q] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned q = q0 + ly + 8u * k, co = co0 + lx;
    if (q < Q && co < (unsigned)it.cout_pad) it.wtp[(long)q * it.cout_pad + co] = tile[lx][ly + 8 * k];
  }
}

inline int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

// =================================================================================================
// Ticket ring (common.h): 64 Ki zero-initialised counters per device, handed out in consecutive runs; a run is reused
// only after 64 Ki counters' worth of later launches, and every last arriver leaves its counter at zero.
#include <atomic>
// OFF by default (SF_BN_TICKET=1 switches it on): measured on MI355X (profiles/README.md, round 2) the fused launches
// are SLOWER than partial + final kernels — stats 2.5 -> 5.8 ms and BN-backward reductions 4.1 -> 7.2 ms per train
// step (serial profile), 71.1 -> 73.8 ms per step: the last workgroup's walk over <= 512 write-through partials per
// output is latency-bound on one CU, where the separate final kernel spreads it over S*C workgroups.
bool sf_tickets_enabled() {
  static const bool on = [] { const char* e = getenv("SF_BN_TICKET"); return e && e[0] == '1'; }();
  return on;
}
// SF_BN_TICKET=2 (round 3, default; 0 = off): tickets only where the last arriver's walk is short — the BN backward
// reduction of layers with >= 256 channels, cut into <= 64 row blocks (>= 4 channel groups of 64 keep >= 256 workgroups in
// flight): one launch instead of partial + final on ~50 of the 114 BN layers of cfg #3, 67.8 -> 67.5 ms per step.
int sf_tickets_level() {
  static const int lv = [] { const char* e = getenv("SF_BN_TICKET"); return e ? atoi(e) : 2; }();  // default since round 3
  return lv;
}
namespace {
constexpr int TICKET_RING = 1 << 16, TICKET_MAXDEV = 16;
unsigned* g_ticket_ring[TICKET_MAXDEV] = {};
std::atomic<unsigned> g_ticket_next[TICKET_MAXDEV];
}  // namespace

// The ring is allocated and zeroed on first use (hipMalloc + a synchronous hipMemset): never inside a stream capture.
// False = no ring on this device and `stream` is being captured: the caller takes its two-launch form.
bool sf_ticket_ring_ready(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TICKET_MAXDEV) return false;
  if (g_ticket_ring[dev]) return true;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  unsigned* p = nullptr;
  if (hipMalloc(&p, TICKET_RING * sizeof(unsigned)) != hipSuccess) return false;
  if (hipMemset(p, 0, TICKET_RING * sizeof(unsigned)) != hipSuccess) return false;
  g_ticket_ring[dev] = p;
  return true;
}

unsigned* sf_ticket_slots(int n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TICKET_MAXDEV || n <= 0 || n > 4096) return nullptr;
  if (!g_ticket_ring[dev]) return nullptr;  // sf_ticket_ring_ready() first
  const unsigned take = ((unsigned)n + 63u) & ~63u;
  unsigned off;
  do {  // a range that would straddle the end of the ring is skipped, never folded onto slot 0 (another launch's)
    off = g_ticket_next[dev].fetch_add(take) % TICKET_RING;
  } while (off + take > (unsigned)TICKET_RING);
  return g_ticket_ring[dev] + off;
}

extern "C" int sf_abi_version(void) { return 1; }
extern "C" const char* sf_build_arch(void) { return "gfx950"; }

extern "C" int sf_ncthw_to_ndhwc(const float* src, float* dst, int N, int C, int T, int H, int W, int cpad,
                                 int ph, int pw, int Wp, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || cpad < C || ph < 0 || pw < 0 ||
      Wp < W + 2 * pw)
    return SF_EINVAL;
  if (cpad == 4 && !sf_aligned16(dst)) return SF_EALIGN;
  const int Hp = H + 2 * ph;
  const long total = (long)N * T * Hp * Wp;
  hipLaunchKernelGGL(ncthw_to_ndhwc_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, src,
                     dst, N, C, T, H, W, cpad, ph, pw, Hp, Wp, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_ndhwc_to_ncthw(const float* src, int cs, int coff, float* dst, int N, int C, int T, int H,
                                 int W, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || cs < coff + C) return SF_EINVAL;
  const long thw = (long)T * H * W;
  const long total = (long)N * C * thw;
  hipLaunchKernelGGL(ndhwc_to_ncthw_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, src,
                     cs, coff, dst, C, thw, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_pool_fwd(const sf_pool_desc* d, const float* in, float* out, void* stream) {
  if (!d || !in || !out || d->C <= 0 || d->kT <= 0 || d->kH <= 0 || d->kW <= 0) return SF_EINVAL;
  const bool vec4 = (d->C % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && (d->out_cs % 4 == 0) &&
                    (d->out_coff % 4 == 0) && sf_aligned16(in) && sf_aligned16(out);
  const long pos = (long)d->N * d->To * d->Ho * d->Wo;
  if (vec4) {
    const long total = pos * (d->C / 4);
    hipLaunchKernelGGL(pool_kernel<4>, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, in, out,
                       total);
  } else {
    const long total = pos * d->C;
    hipLaunchKernelGGL(pool_kernel<1>, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, in, out,
                       total);
  }
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_maxpool_fwd_arg(const sf_pool_desc* d, const float* in, float* out, unsigned char* arg, void* stream) {
  if (!d || !in || !out || !arg || d->C <= 0 || d->kT <= 0 || d->kH <= 0 || d->kW <= 0 || d->is_avg) return SF_EINVAL;
  const bool vec4 = (d->C % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && (d->out_cs % 4 == 0) &&
                    (d->out_coff % 4 == 0) && sf_aligned16(in) && sf_aligned16(out) && ((uintptr_t)arg % 4 == 0);
  if (!vec4 || (long)d->kT * d->kH * d->kW > 255) return SF_EINVAL;
  const long total = (long)d->N * d->To * d->Ho * d->Wo * (d->C / 4);
  hipLaunchKernelGGL(pool_arg_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, in, out, arg,
                     total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_maxpool_bwd_arg(const sf_pool_desc* d, const unsigned char* arg, const float* dy, int dy_cs,
                                  int dy_coff, float* dx, int dx_cs, int dx_coff, int overwrite, void* stream) {
  if (!d || !arg || !dy || !dx || d->is_avg || d->C <= 0) return SF_EINVAL;
  const long total = (long)d->N * d->Ti * d->Hi * d->Wi * (d->C / 4);
  const bool vec4 = (d->C % 4 == 0) && (dy_cs % 4 == 0) && (dy_coff % 4 == 0) && (dx_cs % 4 == 0) &&
                    (dx_coff % 4 == 0) && sf_aligned16(dy) && sf_aligned16(dx) && ((uintptr_t)arg % 4 == 0) &&
                    total < 0x7fffffffL;
  if (!vec4 || (long)d->kT * d->kH * d->kW > 255) return SF_EINVAL;
  hipLaunchKernelGGL(maxpool_bwd_arg_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, arg, dy,
                     dy_cs, dy_coff, dx, dx_cs, dx_coff, overwrite ? 0 : 1, (unsigned)total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

int sf_dwm_fwd_try(const sf_conv_desc* d, const float* in, const float* w, const float* scale, const float* bias,
                   const float* res, float* out, hipStream_t s);  // dwconv_march.hip

extern "C" int sf_dwconv_fwd(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                             const float* bias, const float* res, float* out, void* stream) {
  if (!d || !in || !w_packed || !out || d->Cout <= 0 || d->Cout > d->Cin || d->cin_pad < d->Cin) return SF_EINVAL;
  if ((scale == nullptr) != (bias == nullptr)) return SF_EINVAL;
  if (d->act != SF_ACT_NONE && d->act != SF_ACT_RELU && d->act != SF_ACT_RELU6) return SF_EINVAL;
  {  // kT x 3 x 3 stride-1 "same" layers: the row-march kernels
    const int rc = sf_dwm_fwd_try(d, in, w_packed, scale, bias, res, out, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  const bool vec4 = (d->Cout % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && (d->cin_pad % 4 == 0) &&
                    sf_aligned16(in) && sf_aligned16(w_packed);
  const long pos = (long)d->N * d->To * d->Ho * d->Wo;
  if (vec4) {
    const long total = pos * (d->Cout / 4);
    hipLaunchKernelGGL(dwconv_kernel<4>, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, in,
                       w_packed, scale, bias, res, out, total);
  } else {
    const long total = pos * d->Cout;
    hipLaunchKernelGGL(dwconv_kernel<1>, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, in,
                       w_packed, scale, bias, res, out, total);
  }
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" long sf_tmax_mean_ws_floats(int N, int C) { return (long)N * POOL_P * C; }

extern "C" int sf_tmax_mean(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha,
                            float* pooled, float* ws, void* stream) {
  if (!x || !pooled || !ws || N <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || alpha <= 0 || (T % alpha) != 0)
    return SF_EINVAL;
  const int CB = pow2ceil(C) < TPB ? pow2ceil(C) : TPB;
  const int ncb = sf_cdiv(C, CB);
  hipLaunchKernelGGL(tmax_partial_kernel, dim3(POOL_P, ncb, N), dim3(TPB), 0, (hipStream_t)stream, x, cs, coff, T,
                     H * W, C, alpha, CB, ws);
  const float inv = 1.f / (float)((long)(T / alpha) * H * W);
  hipLaunchKernelGGL(tmax_final_kernel, dim3(sf_cdiv(C, TPB), N), dim3(TPB), 0, (hipStream_t)stream, ws, C, inv,
                     pooled);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_gate_apply(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha,
                             const float* pooled, const float* w3, const float* scale, const float* bias, int act,
                             float* out, int out_cs, int out_coff, void* stream) {
  if (!x || !pooled || !out || N <= 0 || C <= 0 || alpha <= 0 || (T % alpha) != 0) return SF_EINVAL;
  if ((scale == nullptr) != (bias == nullptr)) return SF_EINVAL;
  if (act != SF_ACT_NONE && act != SF_ACT_RELU) return SF_EINVAL;
  const bool vec4 = (C % 4 == 0) && (cs % 4 == 0) && (coff % 4 == 0) && (out_cs % 4 == 0) && (out_coff % 4 == 0) &&
                    sf_aligned16(x) && sf_aligned16(out);
  const long rows = (long)(T / alpha) * H * W;
  const long total = rows * (vec4 ? C / 4 : C);
  int gx = sf_cdiv(total, TPB);
  if (gx > 2048) gx = 2048;
  const size_t shm = (size_t)C * sizeof(float);
  if (vec4)
    hipLaunchKernelGGL(gate_apply_kernel<4>, dim3(gx, N), dim3(TPB), shm, (hipStream_t)stream, x, cs, coff, T,
                       H * W, C, alpha, pooled, w3, scale, bias, act, out, out_cs, out_coff);
  else
    hipLaunchKernelGGL(gate_apply_kernel<1>, dim3(gx, N), dim3(TPB), shm, (hipStream_t)stream, x, cs, coff, T,
                       H * W, C, alpha, pooled, w3, scale, bias, act, out, out_cs, out_coff);
  SF_CHECK_LAUNCH();
  return SF_OK;
}


extern "C" long sf_channel_stats_ws_floats(int C) { return (long)STAT_MAX_P * 2 * C; }

// Statistics over `groups` row groups of `group_rows` rows each; group n belongs to split n % S and every
// per-channel output has S*C entries.  Plain BN: groups = 1, S = 1.  Sub-BN: groups = N samples, S = NUM_SPLITS.
static int stats_launch(const float* x, int cs, int coff, int groups, long group_rows, int S, int C, float* mean,
                        float* var, float* ws, const float* gamma, const float* beta, float eps, float momentum,
                        float* run_mean, float* run_var, float* invstd, float* scale, float* shift, hipStream_t s) {
  if (groups > STAT_MAX_P || S <= 0 || groups % S != 0) return SF_EINVAL;
  const bool vec4 = (C % 4 == 0) && (cs % 4 == 0) && (coff % 4 == 0) && sf_aligned16(x);
  const int cv = sf_cdiv(C, vec4 ? 4 : 1);
  const bool ring = sf_ticket_ring_ready(s);
  // small tensors (the launch-bound models: cfg #1's 16 384 .. 32 rows): the natural geometry has <= 16 row blocks, the
  // last arriver's walk is 16 partials per output at most and the final launch it replaces is a sixth of the layer's
  // launches — fused whenever the tickets are not switched off (SF_BN_TICKET=0)
  const bool small = ring && sf_tickets_level() >= 1 && S == 1 && groups == 1 &&
                     group_rows / ((long)(TPB / (pow2ceil(cv) < TPB ? pow2ceil(cv) : TPB)) * 8) <= 16;
  const bool fused = (sf_tickets_enabled() && ring) || small;
  // fused launch of a LARGE tensor: <= 16 lanes (64 channels) per channel group, so that the group's last workgroup
  // finishes few outputs while the groups finish in parallel, and <= 512 partials per output
  const int cb_max = (fused && !small) ? 16 : TPB;
  const int CB = pow2ceil(cv) < cb_max ? pow2ceil(cv) : cb_max;
  const int rpi = TPB / CB;
  const int max_p = fused ? 512 : STAT_MAX_P;
  if (groups > max_p) return SF_EINVAL;
  long ch = group_rows / ((long)rpi * 8);
  if (ch < 1) ch = 1;
  if (ch * groups > max_p) ch = max_p / groups;
  const int chunks = (int)ch, P = chunks * groups;
  const long srows = group_rows * (groups / S);  // rows per split
  const int ncb = sf_cdiv(cv, CB);
  StatsFinal fin;
  fin.inv_rows = 1.0 / (double)srows;
  fin.unbias = srows > 1 ? (double)srows / (double)(srows - 1) : 1.0;
  fin.mean = mean; fin.var = var; fin.gamma = gamma; fin.beta = beta; fin.eps = eps; fin.momentum = momentum;
  fin.run_mean = run_mean; fin.run_var = run_var; fin.invstd = invstd; fin.scale = scale; fin.shift = shift;
  fin.tickets = fused ? sf_ticket_slots(ncb) : nullptr;
  if (fused && !fin.tickets) return SF_ELAUNCH;
  if (vec4)
    hipLaunchKernelGGL(stats_partial_kernel<4>, dim3(P, ncb), dim3(TPB), 0, s, x, cs, coff, group_rows, chunks, S, C, CB,
                       ws, fin);
  else
    hipLaunchKernelGGL(stats_partial_kernel<1>, dim3(P, ncb), dim3(TPB), 0, s, x, cs, coff, group_rows, chunks, S, C, CB,
                       ws, fin);
  if (!fused)
    hipLaunchKernelGGL(stats_final_kernel, dim3(S * C), dim3(64), 0, s, x, cs, coff, group_rows, chunks, S, ws, C, P,
                       fin.inv_rows, fin.unbias, mean, var, gamma, beta, eps, momentum, run_mean, run_var, invstd, scale,
                       shift);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// Merge of the per-part statistics a conv epilogue left (conv_wave.hip): [part][C / 4][count, K, S1 = sum(v - K),
// S2 = sum((v - K)^2)][4 channels], 64 contiguous bytes per (part, channel quad).  One workgroup per channel quad, one
// pass: every part is shifted to the reference R = K of part 0 (a sample of the channel, so |mean - R| is a few
// standard deviations at most and E[(v-R)^2] - E[v-R]^2 loses nothing in fp64):
//   S1' = S1 + n (K - R),  S2' = S2 + 2 (K - R) S1 + n (K - R)^2.
// Lanes stride over the parts with 4 rows of loads in flight; lanes, then wavefronts, are combined in a fixed order.
__global__ void stats_merge_kernel(const float* __restrict__ parts, int P, int C, const StatsFinal f) {
  __shared__ double red[16][9];
  const int quad = blockIdx.x, nq = C >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const f32x4 R = *reinterpret_cast<const f32x4*>(parts + (long)quad * 16 + 4);
  double n = 0.0, s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  auto add = [&](const f32x4& cnt, const f32x4& k, const f32x4& a1, const f32x4& a2) {
    const double ni = (double)cnt[0];
    n += ni;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double dk = (double)k[e] - (double)R[e], t1 = (double)a1[e];
      s1[e] += t1 + ni * dk;
      s2[e] += (double)a2[e] + dk * (2.0 * t1 + ni * dk);
    }
  };
  const int step = blockDim.x;
  int i = tid;
  for (; i + 3 * step < P; i += 4 * step) {
    f32x4 v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* q = parts + ((long)(i + u * step) * nq + quad) * 16;
#pragma unroll
      for (int w = 0; w < 4; ++w) v[u][w] = *reinterpret_cast<const f32x4*>(q + 4 * w);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) add(v[u][0], v[u][1], v[u][2], v[u][3]);
  }
  for (; i < P; i += step) {
    const float* q = parts + ((long)i * nq + quad) * 16;
    add(*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4),
        *reinterpret_cast<const f32x4*>(q + 8), *reinterpret_cast<const f32x4*>(q + 12));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n += __shfl_xor(n, off, 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] += __shfl_xor(s1[e], off, 64);
      s2[e] += __shfl_xor(s2[e], off, 64);
    }
  }
  if (lane == 0) {
    red[wave][0] = n;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wave][1 + e] = s1[e]; red[wave][5 + e] = s2[e]; }
  }
  __syncthreads();
  if (tid < 4) {
    double N = 0.0, t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < nwave; ++w) { N += red[w][0]; t1 += red[w][1 + tid]; t2 += red[w][5 + tid]; }
    StatsFinal g = f;
    g.inv_rows = 1.0 / N;
    g.unbias = N > 1.0 ? N / (N - 1.0) : 1.0;
    stats_finish(g, quad * 4 + tid, t1, t2, R[tid]);
  }
}

extern "C" int sf_bn_train_stats_merge(const float* parts, int P, int C, const float* gamma, const float* beta, float eps,
                                       float momentum, float* run_mean, float* run_var, float* mean, float* var,
                                       float* invstd, float* scale, float* shift, void* stream) {
  if (!parts || P <= 0 || C <= 0 || (C % 4) != 0 || !gamma || !beta || !mean || !var || !invstd || !scale || !shift)
    return SF_EINVAL;
  if ((run_mean == nullptr) != (run_var == nullptr) || !sf_aligned16(parts)) return SF_EINVAL;
  StatsFinal fin;
  fin.inv_rows = 1.0; fin.unbias = 1.0;
  fin.mean = mean; fin.var = var; fin.gamma = gamma; fin.beta = beta; fin.eps = eps; fin.momentum = momentum;
  fin.run_mean = run_mean; fin.run_var = run_var; fin.invstd = invstd; fin.scale = scale; fin.shift = shift;
  fin.tickets = nullptr;
  hipLaunchKernelGGL(stats_merge_kernel, dim3(C / 4), dim3(P > 1024 ? 1024 : 256), 0, (hipStream_t)stream, parts, P, C,
                     fin);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_channel_stats(const float* x, int cs, int coff, long rows, int C, float* mean, float* var,
                                float* ws, void* stream) {
  if (!x || !mean || !var || !ws || rows <= 0 || C <= 0) return SF_EINVAL;
  return stats_launch(x, cs, coff, 1, rows, 1, C, mean, var, ws, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr,
                      nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int sf_bn_train_stats(const float* x, int cs, int coff, long rows, int C, const float* gamma,
                                 const float* beta, float eps, float momentum, float* run_mean, float* run_var,
                                 float* mean, float* var, float* invstd, float* scale, float* shift, float* ws,
                                 void* stream) {
  if (!x || !gamma || !beta || !mean || !var || !invstd || !scale || !shift || !ws || rows <= 0 || C <= 0)
    return SF_EINVAL;
  if ((run_mean == nullptr) != (run_var == nullptr)) return SF_EINVAL;
  return stats_launch(x, cs, coff, 1, rows, 1, C, mean, var, ws, gamma, beta, eps, momentum, run_mean, run_var,
                      invstd, scale, shift, (hipStream_t)stream);
}

extern "C" int sf_bn_train_stats_split(const float* x, int cs, int coff, int N, long rows_per_sample, int C,
                                       int nsplit, const float* gamma, const float* beta, float eps, float momentum,
                                       float* run_mean, float* run_var, float* mean, float* var, float* invstd,
                                       float* scale, float* shift, float* ws, void* stream) {
  if (!x || !gamma || !beta || !mean || !var || !invstd || !scale || !shift || !ws || N <= 0 ||
      rows_per_sample <= 0 || C <= 0 || nsplit <= 0 || N % nsplit != 0)
    return SF_EINVAL;
  if ((run_mean == nullptr) != (run_var == nullptr)) return SF_EINVAL;
  return stats_launch(x, cs, coff, N, rows_per_sample, nsplit, C, mean, var, ws, gamma, beta, eps, momentum, run_mean,
                      run_var, invstd, scale, shift, (hipStream_t)stream);
}

static int affine_launch(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int S,
                         const float* scale, const float* bias, const float* res, int res_cs, int res_coff,
                         int act, int rep, float* out, int out_cs, int out_coff, int out_cmul, void* stream,
                         unsigned char* mask = nullptr) {
  if (!x || !out || N <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || rep <= 0 || out_cmul <= 0 || S <= 0)
    return SF_EINVAL;
  if ((scale == nullptr) != (bias == nullptr)) return SF_EINVAL;
  if (act != SF_ACT_NONE && act != SF_ACT_RELU && act != SF_ACT_RELU6) return SF_EINVAL;
  const bool vec4 = (out_cmul == 1) && (C % 4 == 0) && (cs % 4 == 0) && (coff % 4 == 0) && (out_cs % 4 == 0) &&
                    (out_coff % 4 == 0) && sf_aligned16(x) && sf_aligned16(out) &&
                    (!res || ((res_cs % 4 == 0) && (res_coff % 4 == 0) && sf_aligned16(res)));
  const long rows = (long)N * T * H * W;
  const long total = rows * (vec4 ? C / 4 : C);
  if (vec4 && S == 1 && rep == 1 && total < 0x7fffffffL && (!scale || (sf_aligned16(scale) && sf_aligned16(bias)))) {
    const int cv = C / 4;
    int shift = -1;
    if ((cv & (cv - 1)) == 0)
      for (shift = 0; (1 << shift) < cv; ++shift) {}
    hipLaunchKernelGGL(affine_flat_kernel, dim3(sf_cdiv(total, TPB * 4)), dim3(TPB), 0, (hipStream_t)stream, x, cs,
                       coff, cv, shift, scale, bias, res, res_cs, res_coff, act, out, out_cs, out_coff, mask,
                       (unsigned)total);
  } else if (mask) {
    return SF_EINVAL;  // the byte mask is a product of the flat float4 kernel only
  } else if (vec4)
    hipLaunchKernelGGL(affine_kernel<4>, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, cs, coff,
                       (long)T * H * W, H * W, C, S, scale, bias, res, res_cs, res_coff, act, rep, out, out_cs,
                       out_coff, out_cmul, total);
  else
    hipLaunchKernelGGL(affine_kernel<1>, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, x, cs, coff,
                       (long)T * H * W, H * W, C, S, scale, bias, res, res_cs, res_coff, act, rep, out, out_cs,
                       out_coff, out_cmul, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_affine_fwd(const float* x, int cs, int coff, int N, int T, int H, int W, int C,
                             const float* scale, const float* bias, const float* res, int res_cs, int res_coff,
                             int act, int rep, float* out, int out_cs, int out_coff, int out_cmul, void* stream) {
  return affine_launch(x, cs, coff, N, T, H, W, C, 1, scale, bias, res, res_cs, res_coff, act, rep, out, out_cs,
                       out_coff, out_cmul, stream);
}

// sf_affine_fwd that also leaves a byte per 4 channels saying which of them pass a gradient through the activation
// (ReLU: y > 0; ReLU6: 0 < y < 6): mask[rows][C/4], read by sf_bn_bwd_reduce / sf_bn_bwd_apply with relu = 3 in place
// of the activation itself (1/16 of its bytes, twice per backward).  Needs the flat float4 case: C % 4 == 0,
// 16-byte addressable views, rep = 1, no channel multiplier; SF_EINVAL otherwise.
extern "C" int sf_affine_fwd_mask(const float* x, int cs, int coff, int N, int T, int H, int W, int C,
                                  const float* scale, const float* bias, const float* res, int res_cs, int res_coff,
                                  int act, float* out, int out_cs, int out_coff, unsigned char* mask, void* stream) {
  if (!mask || (act != SF_ACT_RELU && act != SF_ACT_RELU6)) return SF_EINVAL;
  return affine_launch(x, cs, coff, N, T, H, W, C, 1, scale, bias, res, res_cs, res_coff, act, 1, out, out_cs,
                       out_coff, 1, stream, mask);
}

extern "C" int sf_affine_fwd_split(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int nsplit,
                                   const float* scale, const float* bias, const float* res, int res_cs,
                                   int res_coff, int act, int rep, float* out, int out_cs, int out_coff,
                                   int out_cmul, void* stream) {
  return affine_launch(x, cs, coff, N, T, H, W, C, nsplit, scale, bias, res, res_cs, res_coff, act, rep, out, out_cs,
                       out_coff, out_cmul, stream);
}

extern "C" int sf_pack_conv_weight(const float* w, int Cout, int Cin, int taps, float* wp, int cin_pad, float* wtp,
                                   int cout_pad, void* stream) {
  if (!w || !wp || Cout <= 0 || Cin <= 0 || taps <= 0 || cin_pad < Cin || (wtp && cout_pad < Cout)) return SF_EINVAL;
  const long n_wp = (long)Cout * taps * cin_pad;
  const long total = n_wp + (wtp ? (long)Cin * taps * cout_pad : 0);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, w, Cout, Cin,
                     taps, wp, cin_pad, wtp, cout_pad, n_wp, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_pack_conv_weights(const void* items, const int* blk0, int n, int nblocks, void* stream) {
  if (!items || !blk0 || n <= 0 || nblocks <= 0) return SF_EINVAL;
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(nblocks), dim3(TPB), 0, (hipStream_t)stream,
                     reinterpret_cast<const PackItem*>(items), blk0, n);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_row_softmax_fwd(float* x, int cs, int coff, long rows, int C, float scale, void* stream) {
  if (!x || rows <= 0 || C <= 0) return SF_EINVAL;
  hipLaunchKernelGGL(row_softmax_kernel, dim3(sf_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, cs, coff, rows,
                     C, scale);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_row_softmax_bwd(const float* p, int p_cs, int p_coff, float* dp, int dp_cs, int dp_coff, long rows,
                                  int C, float scale, void* stream) {
  if (!p || !dp || rows <= 0 || C <= 0) return SF_EINVAL;
  hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3(sf_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, p, p_cs,
                     p_coff, dp, dp_cs, dp_coff, rows, C, scale);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_head_act_mean(const float* logits, int B, int P, int K, int act, float* out, void* stream) {
  if (!logits || !out || B <= 0 || P <= 0 || K <= 0) return SF_EINVAL;
  hipLaunchKernelGGL(head_act_mean_kernel, dim3(B), dim3(TPB), 0, (hipStream_t)stream, logits, P, K, act, out);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_channel_shuffle(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff,
                                  int groups, long rows, int C, int accumulate, void* stream) {
  if (!in || !out || rows <= 0 || C <= 0 || groups <= 0 || C % groups) return SF_EINVAL;
  const long total = rows * C;
  hipLaunchKernelGGL(channel_shuffle_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, in, in_cs,
                     in_coff, out, out_cs, out_coff, groups, C, accumulate, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_copy_channels(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff,
                                int out_cmul, long rows, int C, void* stream) {
  if (!in || !out || rows <= 0 || C <= 0 || out_cmul <= 0) return SF_EINVAL;
  const long total = rows * C;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, in,
                     in_cs, in_coff, out, out_cs, out_coff, out_cmul, C, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}
