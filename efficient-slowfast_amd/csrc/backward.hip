// backward.hip — bandwidth-bound backward passes of the SlowFast / CMDA path (NDHWC, fp32, gfx950):
// training-mode BatchNorm backward fused with the ReLU mask, the residual fan-out and the nearest-upsample
// reduction; max-pool backward; ECA (temporal max + channel gate) backward; broadcast / row-dot helpers.
// Per-channel reductions go through a fixed number of fp32 partials combined in fp64 (bit-reproducible).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TPB = 256;


inline int pow2ceil_b(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// g(n,t,hw,c..c+VEC) = sum_{q<rep} dy[n, t*rep+q, hw, c] * (relu ? y[n, t*rep, hw, c] > 0 : 1)
// relu == 3 (SF_BN_MASK_BYTES): `y` is not the activation but the byte mask sf_affine_fwd_mask left, one byte per
// 4 channels ([rows][C/4], y_cs = C/4; bit e = channel c+e passes): 1/16 of the activation's bytes per read.
template <int VEC>
__device__ __forceinline__ void bn_g(const float* __restrict__ dy, int dy_cs, int dy_coff,
                                     const float* __restrict__ y, int y_cs, int y_coff, long r, long THW, int HW,
                                     int rep, int relu, int c, float (&g)[VEC]) {
  long r0 = r;
  if (rep > 1) {
    const long n = r / THW, rem = r - n * THW;
    const long t = rem / HW, hw = rem - t * HW;
    r0 = (n * (THW / HW) * rep + t * rep) * HW + hw;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) g[e] = 0.f;
  for (int q = 0; q < rep; ++q) {
    const float* d = dy + (r0 + (long)q * HW) * dy_cs + dy_coff + c;
    if (VEC == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(d);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] += v[e];
    } else {
      g[0] += d[0];
    }
  }
  if (relu == 3) {
    if (VEC == 4) {
      const unsigned mk = reinterpret_cast<const unsigned char*>(y)[r0 * y_cs + (c >> 2)];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!((mk >> e) & 1u)) g[e] = 0.f;
    }
  } else if (relu) {
    const float* yp = y + r0 * y_cs + y_coff + c;
    if (VEC == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(yp);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(v[e] > 0.f) || (relu == 2 && !(v[e] < 6.f))) g[e] = 0.f;
    } else if (!(yp[0] > 0.f) || (relu == 2 && !(yp[0] < 6.f))) {
      g[0] = 0.f;
    }
  }
}

// Row-block layout shared by the per-channel reductions: CB lanes cover the C/VEC channel vectors of a row,
// TPB/CB rows are processed per iteration, gridDim.x = P row blocks (P partials, combined in fp64).
template <int VEC>
__global__ void bn_bwd_partial_kernel(const float* __restrict__ dy, int dy_cs, int dy_coff,
                                      const float* __restrict__ y, int y_cs, int y_coff,
                                      const float* __restrict__ z, int z_cs, int z_coff, long group_rows, int chunks,
                                      int S, long THW, int HW, int C, int rep, int relu,
                                      const float* __restrict__ mean, const float* __restrict__ invstd, int CB,
                                      float* __restrict__ partial, unsigned* tickets, float* __restrict__ o1,
                                      float* __restrict__ o2, float* __restrict__ acc1, float* __restrict__ acc2) {
  // block -> (row group n, chunk); group = whole tensor (S == 1) or one sample of split n % S (Sub-BN)
  __shared__ float red[2 * TPB * VEC];
  const int blk = blockIdx.x, cb = blockIdx.y;
  const int cl = threadIdx.x % CB, rl = threadIdx.x / CB, rpi = TPB / CB;
  const int c = (cb * CB + cl) * VEC;
  const int n = blk / chunks, ck = blk - n * chunks;
  const long per = (group_rows + chunks - 1) / chunks;
  const long gend = (long)(n + 1) * group_rows;
  const long r0 = (long)n * group_rows + (long)ck * per;
  const long r1 = (r0 + per < gend) ? r0 + per : gend;
  const int so = (n % S) * C;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (c < C) {
    float mu[VEC], is[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mu[e] = mean[so + c + e]; is[e] = invstd[so + c + e]; }
    long r = r0 + rl;
    if constexpr (VEC == 4) if (rep == 1 && (relu == 0 || relu == 3)) {
      // four rows per trip, every load issued before the first use: a thread keeps 8 x 16 bytes (+ 4 mask bytes) in
      // flight instead of 2 x 16 (<= 1024 workgroups = 16 wavefronts per CU cannot hide HBM latency with less)
      const unsigned char* const mk = reinterpret_cast<const unsigned char*>(y);
      for (; r + 3 * (long)rpi < r1; r += 4 * (long)rpi) {
        f32x4 gv[4], zq[4];
        unsigned mb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long ru = r + (long)u * rpi;
          gv[u] = *reinterpret_cast<const f32x4*>(dy + ru * dy_cs + dy_coff + c);
          zq[u] = *reinterpret_cast<const f32x4*>(z + ru * z_cs + z_coff + c);
          mb[u] = relu == 3 ? mk[ru * y_cs + (c >> 2)] : 0xFu;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ge = ((mb[u] >> e) & 1u) ? gv[u][e] : 0.f;
            s1[e] += ge;
            s2[e] = fmaf(ge, (zq[u][e] - mu[e]) * is[e], s2[e]);
          }
      }
    }
    for (; r < r1; r += rpi) {
      float g[VEC], zv[VEC];
      bn_g<VEC>(dy, dy_cs, dy_coff, y, y_cs, y_coff, r, THW, HW, rep, relu, c, g);
      if (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(z + r * z_cs + z_coff + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) zv[e] = v[e];
      } else {
        zv[0] = z[r * z_cs + z_coff + c];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        s1[e] += g[e];
        s2[e] = fmaf(g[e], (zv[e] - mu[e]) * is[e], s2[e]);
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
      if (tickets) {
        sf_store_sc1(partial + ((long)blk * 2 + 0) * C + c + e, t1);
        sf_store_sc1(partial + ((long)blk * 2 + 1) * C + c + e, t2);
      } else {
        partial[((long)blk * 2 + 0) * C + c + e] = t1;
        partial[((long)blk * 2 + 1) * C + c + e] = t2;
      }
    }
  }
  if (!tickets) return;
  // fused final step (common.h): the last workgroup of this channel group sums the P partials of its channels in
  // fp64, four lanes per output in a fixed order, and accumulates into the parameter gradients when asked to
  if (!sf_ticket_last(tickets + cb, gridDim.x)) return;
  const int P = gridDim.x;
  const int cbase = cb * CB * VEC;
  const int ncg = (C - cbase) < CB * VEC ? (C - cbase) : CB * VEC;
  const int q = threadIdx.x & 3;
  for (int item = threadIdx.x >> 2; item < S * ncg; item += TPB / 4) {
    const int sp = item / ncg, cc = cbase + (item - sp * ncg);
    double a1 = 0.0, a2 = 0.0;
    for (int i = q; i < P; i += 4) {
      if ((i / chunks) % S != sp) continue;
      a1 += (double)partial[((long)i * 2 + 0) * C + cc];
      a2 += (double)partial[((long)i * 2 + 1) * C + cc];
    }
    a1 += __shfl_xor(a1, 1, 64);
    a2 += __shfl_xor(a2, 1, 64);
    a1 += __shfl_xor(a1, 2, 64);
    a2 += __shfl_xor(a2, 2, 64);
    if (q == 0) {
      const int o = sp * C + cc;
      o1[o] = (float)a1;
      o2[o] = (float)a2;
      if (acc1) {
        acc1[o] += (float)a1;
        acc2[o] += (float)a2;
      }
    }
  }
}

__global__ void pair_final_kernel(const float* __restrict__ partial, int C, int P, int chunks, int S,
                                  float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ acc1,
                                  float* __restrict__ acc2) {
  __shared__ double r1[64], r2[64];
  const int o = blockIdx.x;  // split * C + channel
  const int sp = o / C, c = o - sp * C;
  double s1 = 0.0, s2 = 0.0;
  int i = threadIdx.x;
  if (S == 1) {
    // four partials of this lane per trip, the eight loads issued before the first add (one pair per trip left this
    // 64-thread workgroup at one L2 round trip per 64 partials: ~7 us for P = 1024, on the path between the two BN
    // backward passes of every layer)
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
  // lanes combined in a fixed order without leaving the wavefront (the workgroup is one wavefront)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  r1[0] = s1;
  r2[0] = s2;
  if (threadIdx.x == 0) {
    o1[o] = (float)r1[0];
    o2[o] = (float)r2[0];
    if (acc1) {  // parameter-gradient sink: dL/dbeta, dL/dgamma accumulate straight into .grad
      acc1[o] += (float)r1[0];
      acc2[o] += (float)r2[0];
    }
  }
}

// dz = gamma*invstd * (g - dbeta/M - xhat*dgamma/M)   [written to dz, may alias z];   dres += g
template <int VEC>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, int dy_cs, int dy_coff,
                                    const float* __restrict__ y, int y_cs, int y_coff, const float* z, int z_cs,
                                    int z_coff, long rows, long THW, int HW, int C, int S, int rep, int relu,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ dbeta,
                                    const float* __restrict__ dgamma, float inv_m, float* dz, int dz_cs, int dz_coff,
                                    float* __restrict__ dres, int dres_cs, int dres_coff, int dres_acc, long total,
                                    int unroll4) {
  if constexpr (VEC == 4) if (unroll4) {
    // four elements per thread: the SAME channel vector of four rows a quarter of the tensor apart (the launcher takes
    // this path when rows % 4 == 0), so each wave-load is one contiguous run, the per-channel coefficients are loaded
    // and folded once per thread, and every load (dy, z, mask byte, dres) is issued before the first use: 8-12 x 16
    // bytes in flight per thread instead of 2-3.  32-bit index arithmetic (below 2^31 elements): three 64-bit
    // divisions per element were what the one-element form spent its issue slots on.
    const unsigned cv = (unsigned)C / 4u;
    const unsigned qrows = (unsigned)rows / 4u;
    const unsigned i0 = blockIdx.x * (unsigned)TPB + threadIdx.x;
    if (i0 >= qrows * cv) return;
    const int cv_shift = (cv & (cv - 1u)) == 0u ? __builtin_ctz(cv) : -1;
    const unsigned row0 = cv_shift >= 0 ? i0 >> cv_shift : i0 / cv;
    const int c = (int)(i0 - row0 * cv) * 4;
    const unsigned char* const mk = reinterpret_cast<const unsigned char*>(y);
    f32x4 gv[4], zq[4], rv[4];
    unsigned mb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long r = (long)(row0 + (unsigned)u * qrows);
      gv[u] = *reinterpret_cast<const f32x4*>(dy + r * dy_cs + dy_coff + c);
      zq[u] = *reinterpret_cast<const f32x4*>(z + r * z_cs + z_coff + c);
      mb[u] = relu == 3 ? mk[r * y_cs + (c >> 2)] : 0xFu;
      if (dres && dres_acc) rv[u] = *reinterpret_cast<const f32x4*>(dres + r * dres_cs + dres_coff + c);
      else rv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned row = row0 + (unsigned)u * qrows;
      const long r = (long)row;
      const int so = (S > 1 ? (int)((row / (unsigned)THW) % (unsigned)S) * C : 0) + c;
      const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + so), mu = *reinterpret_cast<const f32x4*>(mean + so);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + so), db = *reinterpret_cast<const f32x4*>(dbeta + so);
      const f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + so);
      f32x4 ov, gg;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gg[e] = ((mb[u] >> e) & 1u) ? gv[u][e] : 0.f;
        const float xh = (zq[u][e] - mu[e]) * is[e];
        ov[e] = ga[e] * is[e] * (gg[e] - db[e] * inv_m - xh * dg[e] * inv_m);
      }
      *reinterpret_cast<f32x4*>(dz + r * dz_cs + dz_coff + c) = ov;
      if (dres) *reinterpret_cast<f32x4*>(dres + r * dres_cs + dres_coff + c) = rv[u] + gg;
    }
    return;
  }
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int cv = C / VEC;
  const int c = (int)(idx % cv) * VEC;
  const long r = idx / cv;
  float g[VEC], zv[VEC], o[VEC];
  const int so = (int)((r / THW) % S) * C + c;
  bn_g<VEC>(dy, dy_cs, dy_coff, y, y_cs, y_coff, r, THW, HW, rep, relu, c, g);
  if (VEC == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(z + r * z_cs + z_coff + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) zv[e] = v[e];
  } else {
    zv[0] = z[r * z_cs + z_coff + c];
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float is = invstd[so + e];
    const float xh = (zv[e] - mean[so + e]) * is;
    o[e] = gamma[so + e] * is * (g[e] - dbeta[so + e] * inv_m - xh * dgamma[so + e] * inv_m);
  }
  float* dp = dz + r * dz_cs + dz_coff + c;
  if (VEC == 4) {
    *reinterpret_cast<f32x4*>(dp) = (f32x4){o[0], o[1], o[2], o[3]};
  } else {
    dp[0] = o[0];
  }
  if (dres) {
    float* rp = dres + r * dres_cs + dres_coff + c;
    if (VEC == 4) {  // dres_acc == 0: first writer of the residual branch's gradient (no zero fill, no read)
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (dres_acc) v = *reinterpret_cast<f32x4*>(rp);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += g[e];
      *reinterpret_cast<f32x4*>(rp) = v;
    } else {
      rp[0] = dres_acc ? rp[0] + g[0] : g[0];
    }
  }
}

// max-pool backward by equality gather: dx[p] += sum_{windows w containing p} dy[w] * [p is w's first maximum]
__global__ void maxpool_bwd_kernel(const sf_pool_desc d, const float* __restrict__ x, const float* __restrict__ y,
                                   const float* __restrict__ dy, int dy_cs, int dy_coff, float* __restrict__ dx,
                                   int dx_cs, int dx_coff, int acc, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % d.C);
  long r = idx / d.C;
  const long rin = r;
  const int wi = (int)(r % d.Wi);
  r /= d.Wi;
  const int hi = (int)(r % d.Hi);
  r /= d.Hi;
  const int ti = (int)(r % d.Ti);
  const int n = (int)(r / d.Ti);
  const float xv = x[rin * d.in_cs + d.in_coff + c];
  auto lo = [](int i, int p, int k, int s) { const int a = i + p - k + 1; return a <= 0 ? 0 : (a + s - 1) / s; };
  const int t0 = lo(ti, d.pT, d.kT, d.sT), t1 = min((ti + d.pT) / d.sT, d.To - 1);
  const int h0 = lo(hi, d.pH, d.kH, d.sH), h1 = min((hi + d.pH) / d.sH, d.Ho - 1);
  const int w0 = lo(wi, d.pW, d.kW, d.sW), w1 = min((wi + d.pW) / d.sW, d.Wo - 1);
  float g = 0.f;
  for (int to = t0; to <= t1; ++to)
    for (int ho = h0; ho <= h1; ++ho)
      for (int wo = w0; wo <= w1; ++wo) {
        const long ro = (((long)n * d.To + to) * d.Ho + ho) * d.Wo + wo;
        const float yv = y[ro * d.out_cs + d.out_coff + c];
        if (yv != xv) continue;
        // nn.MaxPool3d credits a window's gradient to its FIRST maximum in (t, h, w) scan order only: this element
        // takes it unless an earlier element of the window holds the same value (ties: exact zeros after a ReLU)
        bool first = true;
        const int ts = to * d.sT - d.pT, hs = ho * d.sH - d.pH, ws = wo * d.sW - d.pW;
        for (int tt = max(ts, 0); tt <= ti && first; ++tt)
          for (int hh = max(hs, 0); hh < min(hs + d.kH, d.Hi) && first; ++hh) {
            if (tt == ti && hh > hi) break;
            for (int ww = max(ws, 0); ww < min(ws + d.kW, d.Wi); ++ww) {
              if (tt == ti && hh == hi && ww >= wi) break;
              if (x[((((long)n * d.Ti + tt) * d.Hi + hh) * d.Wi + ww) * d.in_cs + d.in_coff + c] == yv) {
                first = false;
                break;
              }
            }
          }
        if (first) g += dy[ro * dy_cs + dy_coff + c];
      }
  float* o = dx + rin * dx_cs + dx_coff + c;
  *o = acc ? *o + g : g;
}

// ECA backward, reduction: partial[b][blk][c] = sum_{t',hw} dz[b,t',hw,c] * max_r x[b,t'*alpha+r,hw,c]
constexpr int POOL_P = 64;
__global__ void tmax_dot_partial_kernel(const float* __restrict__ x, int cs, int coff, int T, int HW, int C,
                                        int alpha, const float* __restrict__ dz, int dz_cs, int dz_coff, int CB,
                                        float* __restrict__ partial) {
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
      sum = fmaf(mx, dz[((long)b * rows + r) * dz_cs + dz_coff + c], sum);
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

__global__ void pool_final_kernel(const float* __restrict__ partial, int C, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= C) return;
  float tot = 0.f;
  for (int i = 0; i < POOL_P; ++i) tot += partial[((long)b * POOL_P + i) * C + c];
  out[(long)b * C + c] = tot;
}

// ECA backward, apply: dm = dz * gate[b,c] + dpool[b,c]; routed to the FIRST frame holding the temporal max
// (nn.MaxPool3d((alpha,1,1)) keeps the first maximum's index, custom_video_model_builder.py:131)
__global__ void eca_bwd_apply_kernel(const float* __restrict__ x, int cs, int coff, int T, int HW, int C, int alpha,
                                     const float* __restrict__ dz, int dz_cs, int dz_coff,
                                     const float* __restrict__ gate, const float* __restrict__ dpool,
                                     float* __restrict__ dx, int dx_cs, int dx_coff, long total_per_b) {
  const int b = blockIdx.y;
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total_per_b) return;
  const int c = (int)(idx % C);
  const long r = idx / C;  // (t', hw)
  const long t = r / HW, hw = r - t * HW;
  const long rows = (long)(T / alpha) * HW;
  const long base = ((long)b * T + t * alpha) * HW + hw;
  const float* s = x + base * cs + coff + c;
  float mx = s[0];
  for (int a = 1; a < alpha; ++a) mx = fmaxf(mx, s[(long)a * HW * cs]);
  const float dm = dz[((long)b * rows + r) * dz_cs + dz_coff + c] * gate[(long)b * C + c] + dpool[(long)b * C + c];
  for (int a = 0; a < alpha; ++a)
    if (s[(long)a * HW * cs] == mx) {
      dx[(base + (long)a * HW) * dx_cs + dx_coff + c] += dm;
      break;
    }
}

__global__ void bcast_add_kernel(float* __restrict__ g, int cs, int coff, long rows_per_n, int C,
                                 const float* __restrict__ v, float scale, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long r = idx / C;
  const long n = r / rows_per_n;
  g[r * cs + coff + c] += v[n * C + c] * scale;
}

__global__ void rowdot_kernel(const float* __restrict__ a, int a_cs, int a_coff, const float* __restrict__ b,
                              int b_cs, int b_coff, long rows, int C, float scale, float* __restrict__ out) {
  const long r = (long)blockIdx.x * TPB + threadIdx.x;
  if (r >= rows) return;
  const float* ap = a + r * a_cs + a_coff;
  const float* bp = b + r * b_cs + b_coff;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s = fmaf(ap[c], bp[c], s);
  out[r] = s * scale;
}

// C/4 = 1..64 (a power of two) lanes per row, one float4 each, shuffle reduction inside the lane group: coalesced
// 16-byte loads instead of one thread walking a row (row-strided scalar loads).
__global__ __launch_bounds__(256) void rowdot_vec4_kernel(const float* __restrict__ a, int a_cs, int a_coff,
                                                          const float* __restrict__ b, int b_cs, int b_coff, long rows,
                                                          int cv_shift, float scale, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  const long r = idx >> cv_shift;
  const int c = (int)(idx - (r << cv_shift)) * 4;
  float s = 0.f;
  if (r < rows) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(a + r * a_cs + a_coff + c);
    const f32x4 y = *reinterpret_cast<const f32x4*>(b + r * b_cs + b_coff + c);
    s = x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
  }
  for (int o = (1 << cv_shift) >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (r < rows && c == 0) out[r] = s * scale;
}

// out[r, c] (+)= alpha * a[r, c]   (gradient fan-in / scaled copies between channel slices)
__global__ void axpy_kernel(const float* __restrict__ a, int a_cs, int a_coff, float alpha, float* __restrict__ out,
                            int out_cs, int out_coff, int C, int accumulate, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long r = idx / C;
  const float v = alpha * a[r * a_cs + a_coff + c];
  float* o = out + r * out_cs + out_coff + c;
  *o = accumulate ? *o + v : v;
}

// dx[r, c] (+)= dy[r, c] * [0 < y[r, c] (< 6)]   (backward of a bare ReLU / ReLU6, mask taken from its output)
__global__ void act_bwd_kernel(const float* __restrict__ dy, int dy_cs, int dy_coff, const float* __restrict__ y,
                               int y_cs, int y_coff, int act, float* __restrict__ dx, int dx_cs, int dx_coff, int C,
                               int accumulate, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long r = idx / C;
  const float yv = y[r * y_cs + y_coff + c];
  const bool pass = (yv > 0.f) && (act != SF_ACT_RELU6 || yv < 6.f);
  const float v = pass ? dy[r * dy_cs + dy_coff + c] : 0.f;
  float* o = dx + r * dx_cs + dx_coff + c;
  *o = accumulate ? *o + v : v;
}

}  // namespace

constexpr int MAX_P = 1024;
// lanes per row (CB), rows per iteration and number of row blocks for a per-channel reduction
static inline void red_geometry(long rows, int C, int vec, int* CB, int* P) {
  const int cv = sf_cdiv(C, vec);
  *CB = pow2ceil_b(cv) < TPB ? pow2ceil_b(cv) : TPB;
  const int rpi = TPB / *CB;
  long p = rows / ((long)rpi * 8);
  if (p < 1) p = 1;
  if (p > MAX_P) p = MAX_P;
  *P = (int)p;
}
extern "C" long sf_bn_bwd_ws_floats(int C) { return (long)MAX_P * 2 * C; }

static int bn_bwd_reduce_launch(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                                const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int S,
                                int rep, int relu, const float* mean, const float* invstd, float* dbeta,
                                float* dgamma, float* ws, void* stream, float* dbeta_acc = nullptr,
                                float* dgamma_acc = nullptr) {
  if ((dbeta_acc == nullptr) != (dgamma_acc == nullptr)) return SF_EINVAL;
  if (!dy || !z || !mean || !invstd || !dbeta || !dgamma || !ws || (relu && !y)) return SF_EINVAL;
  if (relu == 3 && (rep != 1 || (C % 4) != 0 || y_cs != C / 4 || y_coff != 0)) return SF_EINVAL;
  if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || rep <= 0 || S <= 0 || N % S != 0) return SF_EINVAL;
  if (S > 1 && N > MAX_P) return SF_EINVAL;
  const long rows = (long)N * T * H * W;
  const bool vec4 = (C % 4 == 0) && (dy_cs % 4 == 0) && (dy_coff % 4 == 0) && (z_cs % 4 == 0) && (z_coff % 4 == 0) &&
                    sf_aligned16(dy) && sf_aligned16(z) &&
                    (!relu || relu == 3 || ((y_cs % 4 == 0) && (y_coff % 4 == 0) && sf_aligned16(y)));
  if (relu == 3 && !vec4) return SF_EINVAL;  // the byte mask only exists on the float4 path
  const int groups = S > 1 ? N : 1;
  const long group_rows = rows / groups;
  int CB, P;
  // the ticket ring is allocated on first use; while a stream capture is under way before that, the two-launch form runs
  const bool ring = sf_ticket_ring_ready((hipStream_t)stream);
  // short last-arriver walk: <= 64 partials.  That is also at most 64 x C/64 workgroups, which streams the big early
  // tensors at a third of the bandwidth when the kernel runs ALONE (round 5, tools/microbench/bn_passes.py: 200 704 rows
  // x 256 channels 317 us with tickets, 123 us as partial + final launches) — but in the step these launches run beside
  // the other pathway's kernels, and restricting the tickets to <= 16 384 rows measured 54.47-54.55 ms against
  // 54.26-54.54 with them everywhere (three alternations on one box): the thin launch leaves the chip to its neighbour
  // and the final launch it saves is on the stream's critical path.  Kept as it was.
  red_geometry(group_rows, C, vec4 ? 4 : 1, &CB, &P);
  // small tensors (the launch-bound models): <= 16 row blocks in the natural geometry — the last arriver's walk is
  // short and the final launch it replaces is a third of the reduction's launches
  const bool small = ring && sf_tickets_level() >= 1 && S == 1 && P <= 16;
  const bool few = !small && ring && sf_tickets_level() == 2 && S == 1 && vec4 && C >= 256;
  const bool fused = (ring && sf_tickets_enabled()) || few || small;
  const int max_p = few ? 64 : (fused ? 512 : MAX_P);
  if (fused && !small && CB > 16) {  // <= 64 channels per channel group: the groups' last workgroups finish in parallel
    CB = 16;
    long pp = group_rows / ((long)(TPB / CB) * 8);
    P = (int)(pp < 1 ? 1 : (pp > max_p ? max_p : pp));
  }
  if (P > max_p) P = max_p;
  if ((long)P * groups > max_p) P = max_p / groups;
  if (P < 1) return SF_EINVAL;
  const int chunks = P;
  P = chunks * groups;
  const int ncb = sf_cdiv(sf_cdiv(C, vec4 ? 4 : 1), CB);
  unsigned* tickets = fused ? sf_ticket_slots(ncb) : nullptr;
  if (fused && !tickets) return SF_ELAUNCH;
  if (vec4)
    hipLaunchKernelGGL(bn_bwd_partial_kernel<4>, dim3(P, ncb), dim3(TPB), 0, (hipStream_t)stream, dy, dy_cs, dy_coff,
                       y, y_cs, y_coff, z, z_cs, z_coff, group_rows, chunks, S, (long)T * H * W, H * W, C, rep, relu,
                       mean, invstd, CB, ws, tickets, dbeta, dgamma, dbeta_acc, dgamma_acc);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel<1>, dim3(P, ncb), dim3(TPB), 0, (hipStream_t)stream, dy, dy_cs, dy_coff,
                       y, y_cs, y_coff, z, z_cs, z_coff, group_rows, chunks, S, (long)T * H * W, H * W, C, rep, relu,
                       mean, invstd, CB, ws, tickets, dbeta, dgamma, dbeta_acc, dgamma_acc);
  if (!fused)
    hipLaunchKernelGGL(pair_final_kernel, dim3(S * C), dim3(64), 0, (hipStream_t)stream, ws, C, P, chunks, S, dbeta,
                       dgamma, dbeta_acc, dgamma_acc);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_bn_bwd_reduce(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                                const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep,
                                int relu, const float* mean, const float* invstd, float* dbeta, float* dgamma,
                                float* ws, void* stream) {
  return bn_bwd_reduce_launch(dy, dy_cs, dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, N, T, H, W, C, 1, rep, relu, mean,
                              invstd, dbeta, dgamma, ws, stream);
}

extern "C" int sf_bn_bwd_reduce_acc(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                                    const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep,
                                    int relu, const float* mean, const float* invstd, float* dbeta, float* dgamma,
                                    float* ws, float* dbeta_acc, float* dgamma_acc, void* stream) {
  return bn_bwd_reduce_launch(dy, dy_cs, dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, N, T, H, W, C, 1, rep, relu, mean,
                              invstd, dbeta, dgamma, ws, stream, dbeta_acc, dgamma_acc);
}

extern "C" int sf_bn_bwd_reduce_split(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                                      const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C,
                                      int nsplit, int rep, int relu, const float* mean, const float* invstd,
                                      float* dbeta, float* dgamma, float* ws, void* stream) {
  return bn_bwd_reduce_launch(dy, dy_cs, dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, N, T, H, W, C, nsplit, rep, relu,
                              mean, invstd, dbeta, dgamma, ws, stream);
}

static int bn_bwd_apply_launch(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                               const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int S, int rep,
                               int relu, const float* mean, const float* invstd, const float* gamma,
                               const float* dbeta, const float* dgamma, float* dz, int dz_cs, int dz_coff,
                               float* dres, int dres_cs, int dres_coff, int dres_acc, void* stream) {
  if (!dy || !z || !mean || !invstd || !gamma || !dbeta || !dgamma || !dz || (relu && !y)) return SF_EINVAL;
  if (relu == 3 && (rep != 1 || (C % 4) != 0 || y_cs != C / 4 || y_coff != 0)) return SF_EINVAL;
  if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || rep <= 0 || S <= 0 || N % S != 0) return SF_EINVAL;
  const long rows = (long)N * T * H * W;
  const bool vec4 = (C % 4 == 0) && (dy_cs % 4 == 0) && (dy_coff % 4 == 0) && (z_cs % 4 == 0) && (z_coff % 4 == 0) &&
                    (dz_cs % 4 == 0) && (dz_coff % 4 == 0) && sf_aligned16(dy) && sf_aligned16(z) && sf_aligned16(dz) &&
                    (!relu || relu == 3 || ((y_cs % 4 == 0) && (y_coff % 4 == 0) && sf_aligned16(y))) &&
                    (!dres || ((dres_cs % 4 == 0) && (dres_coff % 4 == 0) && sf_aligned16(dres)));
  if (relu == 3 && !vec4) return SF_EINVAL;  // the byte mask only exists on the float4 path
  const long total = rows * (vec4 ? C / 4 : C);
  // four elements per thread where the tensor is large enough to keep every CU busy that way (SF_BN_APPLY_UNROLL=0: off).
  // dz may alias z and dres may alias dy element for element: a thread reads its four elements before it writes any, and
  // no other thread touches them
  static const int unroll_env = [] { const char* e = getenv("SF_BN_APPLY_UNROLL"); return e ? atoi(e) : 1; }();
  static const long unroll_min = [] { const char* e = getenv("SF_BN_APPLY_UNROLL_MIN"); return e ? atol(e) : 262144L; }();
  const int unroll4 = (unroll_env && vec4 && rep == 1 && (relu == 0 || relu == 3) && total >= unroll_min &&
                       total < 0x7fffffffL && rows < 0x7fffffffL && (rows % 4) == 0 &&
                       sf_aligned16(mean) && sf_aligned16(invstd) && sf_aligned16(gamma) && sf_aligned16(dbeta) &&
                       sf_aligned16(dgamma)) ? 1 : 0;
  if (vec4)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(sf_cdiv(unroll4 ? (total + 3) / 4 : total, TPB)), dim3(TPB), 0,
                       (hipStream_t)stream, dy, dy_cs, dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, rows, (long)T * H * W,
                       H * W, C, S, rep, relu, mean, invstd, gamma, dbeta, dgamma, (float)S / (float)rows, dz, dz_cs,
                       dz_coff, dres, dres_cs, dres_coff, dres_acc, total, unroll4);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, dy, dy_cs,
                       dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, rows, (long)T * H * W, H * W, C, S, rep, relu, mean,
                       invstd, gamma, dbeta, dgamma, (float)S / (float)rows, dz, dz_cs, dz_coff, dres, dres_cs,
                       dres_coff, dres_acc, total, 0);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_bn_bwd_apply(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                               const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep,
                               int relu, const float* mean, const float* invstd, const float* gamma,
                               const float* dbeta, const float* dgamma, float* dz, int dz_cs, int dz_coff,
                               float* dres, int dres_cs, int dres_coff, void* stream) {
  return bn_bwd_apply_launch(dy, dy_cs, dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, N, T, H, W, C, 1, rep, relu, mean,
                             invstd, gamma, dbeta, dgamma, dz, dz_cs, dz_coff, dres, dres_cs, dres_coff, 1, stream);
}

// As sf_bn_bwd_apply, but the residual branch's gradient is WRITTEN (dres = g), not accumulated: for the first
// writer of that buffer, which then needs neither a zero fill nor a read.
extern "C" int sf_bn_bwd_apply_first(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                                     const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep,
                                     int relu, const float* mean, const float* invstd, const float* gamma,
                                     const float* dbeta, const float* dgamma, float* dz, int dz_cs, int dz_coff,
                                     float* dres, int dres_cs, int dres_coff, void* stream) {
  if (!dres) return SF_EINVAL;
  return bn_bwd_apply_launch(dy, dy_cs, dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, N, T, H, W, C, 1, rep, relu, mean,
                             invstd, gamma, dbeta, dgamma, dz, dz_cs, dz_coff, dres, dres_cs, dres_coff, 0, stream);
}

extern "C" int sf_bn_bwd_apply_split(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                                     const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C,
                                     int nsplit, int rep, int relu, const float* mean, const float* invstd,
                                     const float* gamma, const float* dbeta, const float* dgamma, float* dz,
                                     int dz_cs, int dz_coff, float* dres, int dres_cs, int dres_coff, void* stream) {
  return bn_bwd_apply_launch(dy, dy_cs, dy_coff, y, y_cs, y_coff, z, z_cs, z_coff, N, T, H, W, C, nsplit, rep, relu,
                             mean, invstd, gamma, dbeta, dgamma, dz, dz_cs, dz_coff, dres, dres_cs, dres_coff, 1,
                             stream);
}

// Four channels per thread, 32-bit index arithmetic (the scalar kernel above pays four 64-bit divisions per element
// and runs at a third of the HBM rate): taken when every view is float4-addressable and the element count fits.
__global__ __launch_bounds__(256) void maxpool_bwd_vec4_kernel(const sf_pool_desc d, const float* __restrict__ x,
                                                               const float* __restrict__ y,
                                                               const float* __restrict__ dy, int dy_cs, int dy_coff,
                                                               float* __restrict__ dx, int dx_cs, int dx_coff,
                                                               int acc, unsigned total) {
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
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)rin * d.in_cs + d.in_coff + c);
  auto lo = [](int i, int p, int k, int s) { const int a = i + p - k + 1; return a <= 0 ? 0 : (a + s - 1) / s; };
  const int t0 = lo(ti, d.pT, d.kT, d.sT), t1 = min((ti + d.pT) / d.sT, d.To - 1);
  const int h0 = lo(hi, d.pH, d.kH, d.sH), h1 = min((hi + d.pH) / d.sH, d.Ho - 1);
  const int w0 = lo(wi, d.pW, d.kW, d.sW), w1 = min((wi + d.pW) / d.sW, d.Wo - 1);
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  for (int to = t0; to <= t1; ++to)
    for (int ho = h0; ho <= h1; ++ho)
      for (int wo = w0; wo <= w1; ++wo) {
        const long ro = (((long)n * d.To + to) * d.Ho + ho) * d.Wo + wo;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + ro * d.out_cs + d.out_coff + c);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dy + ro * dy_cs + dy_coff + c);
        unsigned m = (yv[0] == xv[0] ? 1u : 0u) | (yv[1] == xv[1] ? 2u : 0u) | (yv[2] == xv[2] ? 4u : 0u) |
                     (yv[3] == xv[3] ? 8u : 0u);
        if (m) {  // first-maximum rule of nn.MaxPool3d (see the scalar kernel): drop the channels an earlier element ties
          const int ts = to * d.sT - d.pT, hs = ho * d.sH - d.pH, ws = wo * d.sW - d.pW;
          for (int tt = max(ts, 0); tt <= ti && m; ++tt)
            for (int hh = max(hs, 0); hh < min(hs + d.kH, d.Hi) && m; ++hh) {
              if (tt == ti && hh > hi) break;
              for (int ww = max(ws, 0); ww < min(ws + d.kW, d.Wi) && m; ++ww) {
                if (tt == ti && hh == hi && ww >= wi) break;
                const f32x4 xe = *reinterpret_cast<const f32x4*>(
                    x + ((((long)n * d.Ti + tt) * d.Hi + hh) * d.Wi + ww) * d.in_cs + d.in_coff + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (xe[e] == yv[e]) m &= ~(1u << e);
              }
            }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((m >> e) & 1u) g[e] += gv[e];
        }
      }
  f32x4* o = reinterpret_cast<f32x4*>(dx + (long)rin * dx_cs + dx_coff + c);
  *o = acc ? *o + g : g;
}

static int maxpool_bwd_launch(const sf_pool_desc* d, const float* x, const float* y, const float* dy, int dy_cs,
                              int dy_coff, float* dx, int dx_cs, int dx_coff, int acc, void* stream) {
  if (!d || !x || !y || !dy || !dx || d->is_avg) return SF_EINVAL;
  const long total = (long)d->N * d->Ti * d->Hi * d->Wi * d->C;
  const bool vec4 = (d->C % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && (d->out_cs % 4 == 0) &&
                    (d->out_coff % 4 == 0) && (dy_cs % 4 == 0) && (dy_coff % 4 == 0) && (dx_cs % 4 == 0) &&
                    (dx_coff % 4 == 0) && sf_aligned16(x) && sf_aligned16(y) && sf_aligned16(dy) && sf_aligned16(dx) &&
                    total / 4 < 0x7fffffffL;
  if (vec4)
    hipLaunchKernelGGL(maxpool_bwd_vec4_kernel, dim3(sf_cdiv(total / 4, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, x,
                       y, dy, dy_cs, dy_coff, dx, dx_cs, dx_coff, acc, (unsigned)(total / 4));
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, x, y, dy,
                       dy_cs, dy_coff, dx, dx_cs, dx_coff, acc, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_maxpool_bwd(const sf_pool_desc* d, const float* x, const float* y, const float* dy, int dy_cs,
                              int dy_coff, float* dx, int dx_cs, int dx_coff, void* stream) {
  return maxpool_bwd_launch(d, x, y, dy, dy_cs, dy_coff, dx, dx_cs, dx_coff, 1, stream);
}

// dx is WRITTEN (every input position gets its gathered gradient, zero where it won no window), not accumulated: for
// the first writer of that buffer.
extern "C" int sf_maxpool_bwd_first(const sf_pool_desc* d, const float* x, const float* y, const float* dy, int dy_cs,
                                    int dy_coff, float* dx, int dx_cs, int dx_coff, void* stream) {
  return maxpool_bwd_launch(d, x, y, dy, dy_cs, dy_coff, dx, dx_cs, dx_coff, 0, stream);
}

extern "C" int sf_tmax_dot(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha,
                           const float* dz, int dz_cs, int dz_coff, float* out, float* ws, void* stream) {
  if (!x || !dz || !out || !ws || N <= 0 || C <= 0 || alpha <= 0 || (T % alpha) != 0) return SF_EINVAL;
  const int CB = pow2ceil_b(C) < TPB ? pow2ceil_b(C) : TPB;
  hipLaunchKernelGGL(tmax_dot_partial_kernel, dim3(POOL_P, sf_cdiv(C, CB), N), dim3(TPB), 0, (hipStream_t)stream, x, cs,
                     coff, T, H * W, C, alpha, dz, dz_cs, dz_coff, CB, ws);
  hipLaunchKernelGGL(pool_final_kernel, dim3(sf_cdiv(C, TPB), N), dim3(TPB), 0, (hipStream_t)stream, ws, C, out);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_eca_bwd_apply(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha,
                                const float* dz, int dz_cs, int dz_coff, const float* gate, const float* dpool,
                                float* dx, int dx_cs, int dx_coff, void* stream) {
  if (!x || !dz || !gate || !dpool || !dx || N <= 0 || C <= 0 || alpha <= 0 || (T % alpha) != 0) return SF_EINVAL;
  const long per_b = (long)(T / alpha) * H * W * C;
  hipLaunchKernelGGL(eca_bwd_apply_kernel, dim3(sf_cdiv(per_b, TPB), N), dim3(TPB), 0, (hipStream_t)stream, x, cs, coff,
                     T, H * W, C, alpha, dz, dz_cs, dz_coff, gate, dpool, dx, dx_cs, dx_coff, per_b);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// ECA's gate algebra on the [N, C] vectors (wdf_attention_helper.py:77-91, backward of conv1d(k=3, pad=1, no bias)
// ALONG THE CHANNEL AXIS + sigmoid):  a = w (*) pooled,  gate = sigmoid(a),  da = dg * gate * (1 - gate),
// dpool[c] = w0*da[c+1] + w1*da[c] + w2*da[c-1]  (conv_transpose1d),  dw[k] += sum_{b,c} da[b,c] * pooled[b,c+k-1].
// One workgroup: N*C <= a few thousand elements; the three dw sums go through LDS in thread order (bit-reproducible).
namespace {
__device__ __forceinline__ float eca_da(const float* __restrict__ dg, const float* __restrict__ pooled, int b, int c,
                                        int C, float w0, float w1, float w2, float* gate_out) {
  if (c < 0 || c >= C) return 0.f;
  const float* p = pooled + (long)b * C;
  const float a = (c > 0 ? w0 * p[c - 1] : 0.f) + w1 * p[c] + (c + 1 < C ? w2 * p[c + 1] : 0.f);
  const float g = 1.f / (1.f + expf(-a));
  if (gate_out) *gate_out = g;
  return dg[(long)b * C + c] * g * (1.f - g);
}

__global__ void eca_gate_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ pooled,
                                    const float* __restrict__ w3, int N, int C, float dpool_scale,
                                    float* __restrict__ gate, float* __restrict__ dpool, float* __restrict__ dw3) {
  __shared__ float red[3][TPB];
  const float w0 = w3[0], w1 = w3[1], w2 = w3[2];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < N * C; i += TPB) {
    const int b = i / C, c = i - b * C;
    float g;
    const float da = eca_da(dg, pooled, b, c, C, w0, w1, w2, &g);
    const float dam = eca_da(dg, pooled, b, c - 1, C, w0, w1, w2, nullptr);
    const float dap = eca_da(dg, pooled, b, c + 1, C, w0, w1, w2, nullptr);
    gate[i] = g;
    dpool[i] = (w0 * dap + w1 * da + w2 * dam) * dpool_scale;
    const float* p = pooled + (long)b * C;
    s0 += c > 0 ? da * p[c - 1] : 0.f;
    s1 += da * p[c];
    s2 += c + 1 < C ? da * p[c + 1] : 0.f;
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  red[2][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = 0.f;
    for (int i = 0; i < TPB; ++i) t += red[threadIdx.x][i];
    dw3[threadIdx.x] += t;
  }
}
}  // namespace

extern "C" int sf_eca_gate_bwd(const float* dg, const float* pooled, const float* w3, int N, int C, float dpool_scale,
                               float* gate, float* dpool, float* dw3, void* stream) {
  if (!dg || !pooled || !w3 || !gate || !dpool || !dw3 || N <= 0 || C <= 0) return SF_EINVAL;
  hipLaunchKernelGGL(eca_gate_bwd_kernel, dim3(1), dim3(TPB), 0, (hipStream_t)stream, dg, pooled, w3, N, C, dpool_scale,
                     gate, dpool, dw3);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_bcast_add(float* g, int cs, int coff, int N, long rows_per_n, int C, const float* v, float scale,
                            void* stream) {
  if (!g || !v || N <= 0 || rows_per_n <= 0 || C <= 0) return SF_EINVAL;
  const long total = (long)N * rows_per_n * C;
  hipLaunchKernelGGL(bcast_add_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, g, cs, coff,
                     rows_per_n, C, v, scale, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_rowdot(const float* a, int a_cs, int a_coff, const float* b, int b_cs, int b_coff, long rows, int C,
                         float scale, float* out, void* stream) {
  if (!a || !b || !out || rows <= 0 || C <= 0) return SF_EINVAL;
  const int cv = C / 4;
  if (C % 4 == 0 && cv <= 64 && (cv & (cv - 1)) == 0 && a_cs % 4 == 0 && a_coff % 4 == 0 && b_cs % 4 == 0 &&
      b_coff % 4 == 0 && sf_aligned16(a) && sf_aligned16(b)) {
    int shift = 0;
    while ((1 << shift) < cv) ++shift;
    hipLaunchKernelGGL(rowdot_vec4_kernel, dim3(sf_cdiv(rows * cv, TPB)), dim3(TPB), 0, (hipStream_t)stream, a, a_cs,
                       a_coff, b, b_cs, b_coff, rows, shift, scale, out);
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
  hipLaunchKernelGGL(rowdot_kernel, dim3(sf_cdiv(rows, TPB)), dim3(TPB), 0, (hipStream_t)stream, a, a_cs, a_coff, b,
                     b_cs, b_coff, rows, C, scale, out);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_axpy(const float* a, int a_cs, int a_coff, float alpha, float* out, int out_cs, int out_coff,
                       long rows, int C, int accumulate, void* stream) {
  if (!a || !out || rows <= 0 || C <= 0) return SF_EINVAL;
  const long total = rows * C;
  hipLaunchKernelGGL(axpy_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, a, a_cs, a_coff, alpha,
                     out, out_cs, out_coff, C, accumulate, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_act_bwd(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff, int act,
                          float* dx, int dx_cs, int dx_coff, long rows, int C, int accumulate, void* stream) {
  if (!dy || !y || !dx || rows <= 0 || C <= 0) return SF_EINVAL;
  if (act != SF_ACT_RELU && act != SF_ACT_RELU6) return SF_EINVAL;
  const long total = rows * C;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, dy, dy_cs, dy_coff,
                     y, y_cs, y_coff, act, dx, dx_cs, dx_coff, C, accumulate, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// ================================================================================================
// Depthwise convolution backward (GhostNet / ShuffleNetV2) and strided channel gathers.
namespace {

// dx[p, c] += sum_taps dz[q(p, tap), c] * w[tap, c]   (transposed gather, one thread per (position, channel))
__global__ void dwconv_dgrad_kernel(const sf_conv_desc d, const float* __restrict__ dz, int dz_cs, int dz_coff,
                                    const float* __restrict__ w, float* __restrict__ dx, int dx_cs, int dx_coff,
                                    int C, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  long r = idx / C;
  const long rin = r;
  const int wi = (int)(r % d.Wi);
  r /= d.Wi;
  const int hi = (int)(r % d.Hi);
  r /= d.Hi;
  const int ti = (int)(r % d.Ti);
  const int n = (int)(r / d.Ti);
  float acc = 0.f;
  int tap = 0;
  for (int kt = 0; kt < d.kT; ++kt) {
    const int nt = ti + d.pT - kt * d.dT;
    for (int kh = 0; kh < d.kH; ++kh) {
      const int nh = hi + d.pH - kh * d.dH;
      for (int kw = 0; kw < d.kW; ++kw, ++tap) {
        const int nw = wi + d.pW - kw * d.dW;
        if (nt < 0 || nh < 0 || nw < 0 || (nt % d.sT) || (nh % d.sH) || (nw % d.sW)) continue;
        const int to = nt / d.sT, ho = nh / d.sH, wo = nw / d.sW;
        if (to >= d.To || ho >= d.Ho || wo >= d.Wo) continue;
        const long ro = (((long)n * d.To + to) * d.Ho + ho) * d.Wo + wo;
        acc = fmaf(dz[ro * dz_cs + dz_coff + c], w[(long)tap * d.cin_pad + c], acc);
      }
    }
  }
  dx[rin * dx_cs + dx_coff + c] += acc;
}

// The same with 4 channels per thread (16-byte loads / stores) and 32-bit position arithmetic.
__global__ void dwconv_dgrad_vec4_kernel(const sf_conv_desc d, const float* __restrict__ dz, int dz_cs, int dz_coff,
                                         const float* __restrict__ w, float* __restrict__ dx, int dx_cs, int dx_coff,
                                         int CV, unsigned total) {
  const unsigned idx = blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const unsigned rin = idx / (unsigned)CV;
  const int c = (int)(idx - rin * (unsigned)CV) * 4;
  unsigned r = rin;
  const unsigned q1 = r / (unsigned)d.Wi;
  const int wi = (int)(r - q1 * (unsigned)d.Wi);
  const unsigned q2 = q1 / (unsigned)d.Hi;
  const int hi = (int)(q1 - q2 * (unsigned)d.Hi);
  const unsigned n = q2 / (unsigned)d.Ti;
  const int ti = (int)(q2 - n * (unsigned)d.Ti);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int tap = 0;
  for (int kt = 0; kt < d.kT; ++kt) {
    const int nt = ti + d.pT - kt * d.dT;
    for (int kh = 0; kh < d.kH; ++kh) {
      const int nh = hi + d.pH - kh * d.dH;
      for (int kw = 0; kw < d.kW; ++kw, ++tap) {
        const int nw = wi + d.pW - kw * d.dW;
        if (nt < 0 || nh < 0 || nw < 0 || (nt % d.sT) || (nh % d.sH) || (nw % d.sW)) continue;
        const int to = nt / d.sT, ho = nh / d.sH, wo = nw / d.sW;
        if (to >= d.To || ho >= d.Ho || wo >= d.Wo) continue;
        const long ro = (((long)n * d.To + to) * d.Ho + ho) * d.Wo + wo;
        acc += *reinterpret_cast<const f32x4*>(dz + ro * dz_cs + dz_coff + c) *
               *reinterpret_cast<const f32x4*>(w + (long)tap * d.cin_pad + c);
      }
    }
  }
  f32x4* const o = reinterpret_cast<f32x4*>(dx + (long)rin * dx_cs + dx_coff + c);
  *o += acc;
}

// dw[tap, c] = sum_m dz[m, c] * x[row(m, tap), c]: partial[blk][tap][c] over the block's output rows
constexpr int DW_P = 256;
__global__ void dwconv_wgrad_partial_kernel(const sf_conv_desc d, const float* __restrict__ x,
                                            const float* __restrict__ dz, int dz_cs, int dz_coff, int C, int CB,
                                            long rows, float* __restrict__ partial) {
  __shared__ float red[TPB];
  const int blk = blockIdx.x, cb = blockIdx.y;
  const int cl = threadIdx.x % CB, rl = threadIdx.x / CB, rpi = TPB / CB;
  const int c = cb * CB + cl;
  const long per = (rows + DW_P - 1) / DW_P;
  const long r0 = (long)blk * per;
  const long r1 = (r0 + per < rows) ? r0 + per : rows;
  const int ntaps = d.kT * d.kH * d.kW;
  for (int tap = 0; tap < ntaps; ++tap) {
    const int kw = tap % d.kW, kh = (tap / d.kW) % d.kH, kt = tap / (d.kW * d.kH);
    float s = 0.f;
    if (c < C) {
      for (long m = r0 + rl; m < r1; m += rpi) {
        const int wo = (int)(m % d.Wo);
        const long t1 = m / d.Wo;
        const int ho = (int)(t1 % d.Ho);
        const long t2 = t1 / d.Ho;
        const int to = (int)(t2 % d.To);
        const int n = (int)(t2 / d.To);
        const int ti = to * d.sT - d.pT + kt * d.dT;
        const int hi = ho * d.sH - d.pH + kh * d.dH;
        const int wi = wo * d.sW - d.pW + kw * d.dW;
        if ((unsigned)ti >= (unsigned)d.Ti || (unsigned)hi >= (unsigned)d.Hi || (unsigned)wi >= (unsigned)d.Wi)
          continue;
        s = fmaf(dz[m * dz_cs + dz_coff + c],
                 x[((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.in_cs + d.in_coff + c], s);
      }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
      float tot = 0.f;
      for (int i = 0; i < rpi; ++i) tot += red[i * CB + cl];
      partial[((long)blk * ntaps + tap) * C + c] = tot;
    }
    __syncthreads();
  }
}

// One pass over the rows instead of one per tap: a thread owns 4 channels (16-byte loads) of a strided subset of the
// block's output rows, decodes each row ONCE, loads dz once and x at every tap (neighbouring taps hit L1), and keeps
// all taps' sums in registers (NT x 4 accumulators); the row lanes of a channel quad are then summed in lane order
// through LDS.  The per-tap kernel above re-read dz and re-decoded the position (four 64-bit divisions) once per tap and
// row: 82 launches of SlowFastGhostNet at 8 clips took 98 ms of a 219 ms step (profiles/r03_ghostnet_b8_*).
template <int V> struct DwVec;
template <> struct DwVec<4> {
  typedef f32x4 T;
  static __device__ __forceinline__ T zero() { return (T){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ T load(const float* p) { return *reinterpret_cast<const T*>(p); }
  static __device__ __forceinline__ void store(float* p, T v) { *reinterpret_cast<T*>(p) = v; }
};
template <> struct DwVec<1> {
  typedef float T;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, T v) { *p = v; }
};

template <int NT, int V>
__global__ __launch_bounds__(256) void dwconv_wgrad_partial4_kernel(const sf_conv_desc d, const float* __restrict__ x,
                                                                    const float* __restrict__ dz, int dz_cs, int dz_coff,
                                                                    int C, int CQ, long rows, int nblk,
                                                                    float* __restrict__ partial) {
  typedef typename DwVec<V>::T VT;
  __shared__ VT red[9][TPB];
  const int blk = blockIdx.x, cb = blockIdx.y;
  const int ql = threadIdx.x % CQ, rl = threadIdx.x / CQ, rpi = TPB / CQ;
  const int c = (cb * CQ + ql) * V;
  const bool cok = c < C;
  const long per = (rows + nblk - 1) / nblk;
  const long r0 = (long)blk * per;
  const long r1 = (r0 + per < rows) ? r0 + per : rows;
  const int ntaps = d.kT * d.kH * d.kW;
  VT acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = DwVec<V>::zero();
  if (cok) {
    for (long m = r0 + rl; m < r1; m += rpi) {
      const unsigned mu = (unsigned)m;                       // rows < 2^31 (checked by the launcher)
      const unsigned q1 = mu / (unsigned)d.Wo;
      const int wo = (int)(mu - q1 * (unsigned)d.Wo);
      const unsigned q2 = q1 / (unsigned)d.Ho;
      const int ho = (int)(q1 - q2 * (unsigned)d.Ho);
      const unsigned n = q2 / (unsigned)d.To;
      const int to = (int)(q2 - n * (unsigned)d.To);
      const VT g = DwVec<V>::load(dz + m * dz_cs + dz_coff + c);
      const int t0 = to * d.sT - d.pT, h0 = ho * d.sH - d.pH, w0 = wo * d.sW - d.pW;
      const float* const xb = x + (((long)n * d.Ti) * d.Hi * d.Wi) * d.in_cs + d.in_coff + c;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < ntaps) {
          const int kw = t % d.kW, kh = (t / d.kW) % d.kH, kt = t / (d.kW * d.kH);
          const int ti = t0 + kt * d.dT, hi = h0 + kh * d.dH, wi = w0 + kw * d.dW;
          if ((unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi) {
            const VT xv = DwVec<V>::load(xb + (((long)ti * d.Hi + hi) * d.Wi + wi) * d.in_cs);
            acc[t] += g * xv;
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < (NT + 8) / 9; ++r) {                    // 9 taps per round through LDS
    const int t0 = r * 9;
    if (t0 < ntaps) {                                         // uniform
#pragma unroll
      for (int u = 0; u < 9; ++u)
        if (t0 + u < NT) red[u][threadIdx.x] = acc[t0 + u];
      __syncthreads();
      if (rl == 0 && cok) {
        for (int u = 0; u < 9 && t0 + u < ntaps; ++u) {
          VT tot = DwVec<V>::zero();
          for (int i = 0; i < rpi; ++i) tot += red[u][i * CQ + ql];
          DwVec<V>::store(partial + ((long)blk * ntaps + t0 + u) * C + c, tot);
        }
      }
      __syncthreads();
    }
  }
}

__global__ void dwconv_wgrad_final_kernel(const float* __restrict__ partial, int n, int nblk, float* __restrict__ out,
                                          int C, int ntaps, int to_param, int accumulate) {
  // 16 lanes per output (i over ntaps*C): lane q sums the blocks q, q + 16, .. in fp64, the 16 sub-sums are combined by
  // a fixed xor tree — one thread per output walked up to 512 partials in 64 dependent rounds (10 us per launch, 82
  // launches per cfg #5 step)
  const int i = (blockIdx.x * TPB + threadIdx.x) >> 4;
  const int q = threadIdx.x & 15;
  double s = 0.0;
  if (i < n)
    for (int b = q; b < nblk; b += 16) s += (double)partial[(long)b * n + i];
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
  if (i >= n || q != 0) return;
  // to_param: nn.Conv3d's own depthwise layout [C][1][kT][kH][kW] = [c][tap] instead of the packed [tap][c]
  float* const o = out + (to_param ? (long)(i % C) * ntaps + i / C : (long)i);
  *o = accumulate ? *o + (float)s : (float)s;
}

// Row blocks of the one-pass kernel: enough workgroups to fill the chip when the layer has few channel blocks
// (C = 32: one channel block — 256 row blocks were one workgroup per CU, every load latency exposed).
static inline int dw_quads_per_block(int nq) {  // nq = channel groups (quads, or single channels when C % 4 != 0)
  int CQ = 1;
  while (CQ < nq && CQ < 64) CQ <<= 1;
  return CQ;
}
static inline int dw_row_blocks(int C, long rows) {
  const int nq = (C % 4 == 0) ? C / 4 : C;
  const int ncb = sf_cdiv(nq, dw_quads_per_block(nq));
  long nb = 1024 / ncb;                                    // ~1024 workgroups in all: 4 per CU
  if (nb > 512) nb = 512;                                  // what the final kernel sums per output
  if (nb > rows / 64) nb = rows / 64;
  if (nb < 64) nb = 64;
  return (int)nb;
}

// out[r, c] (+)= in[r, in_coff + c * in_cmul]   (backward of a channel-strided store / shuffle)
__global__ void gather_add_kernel(const float* __restrict__ in, int in_cs, int in_coff, int in_cmul,
                                  float* __restrict__ out, int out_cs, int out_coff, int C, int accumulate,
                                  long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long r = idx / C;
  const float v = in[r * in_cs + in_coff + (long)c * in_cmul];
  float* o = out + r * out_cs + out_coff + c;
  *o = accumulate ? *o + v : v;
}

}  // namespace

int sf_dwm_dgrad_try(const sf_conv_desc* d, const float* dz, int dz_cs, int dz_coff, const float* w, float* dx,
                     int dx_cs, int dx_coff, int C, int accumulate, hipStream_t s);  // dwconv_march.hip
int sf_dwm_wgrad_try(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, int C, float* ws,
                     int max_blk, int* nblk, hipStream_t s);

extern "C" int sf_dwconv_dgrad(const sf_conv_desc* d, const float* dz, int dz_cs, int dz_coff, const float* w_packed,
                               float* dx, int dx_cs, int dx_coff, int C, void* stream) {
  if (!d || !dz || !w_packed || !dx || C <= 0) return SF_EINVAL;
  {  // kT x 3 x 3 stride-1 "same" layers: the row-march kernel with mirrored taps
    const int rc = sf_dwm_dgrad_try(d, dz, dz_cs, dz_coff, w_packed, dx, dx_cs, dx_coff, C, 1, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  const long total = (long)d->N * d->Ti * d->Hi * d->Wi * C;
  if (C % 4 == 0 && dz_cs % 4 == 0 && dz_coff % 4 == 0 && dx_cs % 4 == 0 && dx_coff % 4 == 0 && d->cin_pad % 4 == 0 &&
      sf_aligned16(dz) && sf_aligned16(dx) && sf_aligned16(w_packed) && total / 4 < 0x7fffffffL) {
    hipLaunchKernelGGL(dwconv_dgrad_vec4_kernel, dim3(sf_cdiv(total / 4, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, dz,
                       dz_cs, dz_coff, w_packed, dx, dx_cs, dx_coff, C / 4, (unsigned)(total / 4));
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
  hipLaunchKernelGGL(dwconv_dgrad_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, *d, dz, dz_cs,
                     dz_coff, w_packed, dx, dx_cs, dx_coff, C, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" long sf_dwconv_wgrad_ws_floats(const sf_conv_desc* d, int C) {
  if (!d) return 0;
  const long rows = (long)d->N * d->To * d->Ho * d->Wo;
  const int nb = dw_row_blocks(C, rows);
  return (long)(nb > DW_P ? nb : DW_P) * d->kT * d->kH * d->kW * C;
}

static int dwconv_wgrad_impl(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, int C,
                             float* dw, float* ws, int to_param, int accumulate, void* stream) {
  if (!d || !x || !dz || !dw || !ws || C <= 0) return SF_EINVAL;
  const long rows = (long)d->N * d->To * d->Ho * d->Wo;
  const int CB = pow2ceil_b(C) < TPB ? pow2ceil_b(C) : TPB;
  const int ntaps = d->kT * d->kH * d->kW;
  const bool vec4 = (C % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && (dz_cs % 4 == 0) &&
                    (dz_coff % 4 == 0) && sf_aligned16(x) && sf_aligned16(dz) && sf_aligned16(ws);
  int nblk = DW_P;
  const int march = sf_dwm_wgrad_try(d, x, dz, dz_cs, dz_coff, C, ws, dw_row_blocks(C, rows) > DW_P ? dw_row_blocks(C, rows) : DW_P,
                                     &nblk, (hipStream_t)stream);
  if (march != 1) {                                          // kT x 3 x 3 stride-1 "same": the row-march kernel
    if (march != SF_OK) return march;
  } else if (rows < 0x7fffffffL && ntaps <= 27) {            // one pass over the rows, all taps in registers
    const int nq = (C % 4 == 0) ? C / 4 : C;                 // (the row-block count follows C % 4 alone: see ws_floats)
    const int CQ = dw_quads_per_block(nq);
    nblk = dw_row_blocks(C, rows);
    const dim3 grid(nblk, sf_cdiv(nq, CQ));
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && vec4) {
      if (ntaps <= 9)
        hipLaunchKernelGGL((dwconv_wgrad_partial4_kernel<9, 4>), grid, dim3(TPB), 0, st, *d, x, dz, dz_cs, dz_coff, C, CQ, rows, nblk, ws);
      else
        hipLaunchKernelGGL((dwconv_wgrad_partial4_kernel<27, 4>), grid, dim3(TPB), 0, st, *d, x, dz, dz_cs, dz_coff, C, CQ, rows, nblk, ws);
    } else {
      const int CQ1 = dw_quads_per_block(C);
      const dim3 grid1(nblk, sf_cdiv(C, CQ1));
      if (ntaps <= 9)
        hipLaunchKernelGGL((dwconv_wgrad_partial4_kernel<9, 1>), grid1, dim3(TPB), 0, st, *d, x, dz, dz_cs, dz_coff, C, CQ1, rows, nblk, ws);
      else
        hipLaunchKernelGGL((dwconv_wgrad_partial4_kernel<27, 1>), grid1, dim3(TPB), 0, st, *d, x, dz, dz_cs, dz_coff, C, CQ1, rows, nblk, ws);
    }
  } else
    hipLaunchKernelGGL(dwconv_wgrad_partial_kernel, dim3(DW_P, sf_cdiv(C, CB)), dim3(TPB), 0, (hipStream_t)stream, *d, x,
                       dz, dz_cs, dz_coff, C, CB, rows, ws);
  hipLaunchKernelGGL(dwconv_wgrad_final_kernel, dim3(sf_cdiv((long)ntaps * C * 16, TPB)), dim3(TPB), 0, (hipStream_t)stream, ws,
                     ntaps * C, nblk, dw, C, ntaps, to_param, accumulate);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

extern "C" int sf_dwconv_wgrad(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, int C,
                               float* dw /* [taps][C] */, float* ws, void* stream) {
  return dwconv_wgrad_impl(d, x, dz, dz_cs, dz_coff, C, dw, ws, 0, 0, stream);
}

// The same with the result stored (accumulate == 0) or accumulated in nn.Conv3d's own layout [C][1][kT][kH][kW]: the
// gradient lands in the parameter's .grad without a transpose and an add.
extern "C" int sf_dwconv_wgrad_param(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff,
                                     int C, float* dw_param, int accumulate, float* ws, void* stream) {
  return dwconv_wgrad_impl(d, x, dz, dz_cs, dz_coff, C, dw_param, ws, 1, accumulate, stream);
}

extern "C" int sf_gather_add(const float* in, int in_cs, int in_coff, int in_cmul, float* out, int out_cs,
                             int out_coff, long rows, int C, int accumulate, void* stream) {
  if (!in || !out || rows <= 0 || C <= 0 || in_cmul <= 0) return SF_EINVAL;
  const long total = rows * C;
  hipLaunchKernelGGL(gather_add_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, in, in_cs, in_coff,
                     in_cmul, out, out_cs, out_coff, C, accumulate, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}
