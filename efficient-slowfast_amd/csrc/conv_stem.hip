// conv_stem.hip — forward of the Fast pathway's stem convolution from an LDS ring (gfx950).
//
// The stem trick (engine.stem_conv_bn_relu; stem_helper.py:157-164 in the reference) turns the 5x7x7 / stride (1,2,2)
// conv over the border-padded NDHWC4 clip into a 5x7x1 conv over pixels of 8 floats that reads 28 contiguous floats
// per tap.  Through the implicit-GEMM kernel every one of the 35 taps fetches its input row from L2 again (1.7 ms,
// Cout = 8 of a 16-wide tile).  Here — as in conv_wgrad_stem_kernel — a workgroup owns one output row h of a clip
// and walks t: the 5 frames x 7 input rows it needs sit in a 129 KB LDS ring that takes ONE new 7-row slab per step.
//   z[pos, co] = sum_{tap, j} x[ring row(tap)][pos * ps + j] * w[co][tap][j]
//   A[m = position][k]: the K order inside a 16-float chunk is permuted to j = 4 g + s (lane quarter g, MFMA s), so
//       a lane's four K values are 16 contiguous bytes: ONE ds_read_b128 feeds 4 MFMAs;
//   B[k][n = co]: the weights of this wavefront's taps, in registers for the whole launch (the 35 taps are dealt to
//       the 4 wavefronts: 9 taps x 2 chunks x 4 = 72 values per lane); packed channels 28..31 carry zero weights.
// The wavefronts' partial rows are summed through LDS in wavefront order, then scale / bias / ReLU and the store.
//
// PAIR form (Cout <= 8, pixel stride 8 floats — the Fast stem itself): Cout = 8 fills half of the 16 MFMA columns, so
// a row of the tile is a PAIR of neighbouring output positions and the columns are (h, co) = position 2p + h, channel
// co.  The pair reads ONE window of 28 + 8 = 36 floats; position h sees it shifted by 8, i.e. the weights of column
// (h, co) are w[co][tap][j - 8 h] (zero outside 0..27).  K per tap = 36 = two 16-float chunks + one 4-float chunk
// (a ds_read_b32 feeding one MFMA): 9 MFMAs per tap per 32 positions instead of 16.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int KT = 5, KH = 7, NTAP = KT * KH, TPW = (NTAP + 3) / 4;  // taps per wavefront
constexpr int MAXBLK = 8;                                            // 16-position blocks per output row (Wo <= 128)

struct StemFwdArgs {
  sf_conv_desc d;
  const float* in; const float* w; const float* scale; const float* bias;
  float* out;
  int rowf, ps, nblk, tparts, units;
};

template <bool PAIR>
__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(const StemFwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const sf_conv_desc& d = q.d;
  const int slab = KH * q.rowf;
  float* const ring = lds;              // [KT][slab]
  float* const red = lds + KT * slab;   // [4][nblk * 16][16]: per-wavefront partial rows (co on the fast axis)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int slab4 = slab >> 2;
  constexpr int LD4 = 8;

  // this wavefront's weights: tap = wave + 4 u; chunk c; MFMA s  ->  w[co = fr][tap][16 c + 4 fg + s]
  // PAIR: column fr = (h, co) = (fr >> 3, fr & 7) holds w[co][tap][j - ps h]; wx = the 4-float chunk j = 32 + fg
  float wreg[TPW][2][4], wx[TPW];
  const int wco = PAIR ? (fr & 7) : fr, wsh = PAIR ? (fr >> 3) * q.ps : 0;
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int tap = wave + 4 * u;
    const bool okw = tap < NTAP && wco < d.Cout;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int j = 16 * c + 4 * fg + s - wsh;
        wreg[u][c][s] = (okw && j >= 0 && j < d.Cin) ? q.w[((long)wco * NTAP + tap) * d.cin_pad + j] : 0.f;
      }
    const int jx = 32 + fg - wsh;
    wx[u] = (PAIR && okw && jx >= 0 && jx < d.Cin) ? q.w[((long)wco * NTAP + tap) * d.cin_pad + jx] : 0.f;
  }
  int t_kt[TPW], t_off[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int tap = min(wave + 4 * u, NTAP - 1);
    t_kt[u] = tap / KH;
    t_off[u] = (tap % KH) * q.rowf + 4 * fg;
  }

  f32x4 rs[LD4];
  auto load_slab = [&](int n, int ti, int h) {
    const bool ok = (unsigned)ti < (unsigned)d.Ti;
    const f32x4* src =
        reinterpret_cast<const f32x4*>(q.in + (((long)n * d.Ti + (ok ? ti : 0)) * d.Hi + (long)h * d.sH) * q.rowf);
#pragma unroll
    for (int u = 0; u < LD4; ++u) {
      const int f = tid + u * 256;
      rs[u] = (ok && f < slab4) ? src[f] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_slab = [&](int slot) {
    f32x4* dst = reinterpret_cast<f32x4*>(ring + slot * slab);
#pragma unroll
    for (int u = 0; u < LD4; ++u) {
      const int f = tid + u * 256;
      if (f < slab4) dst[f] = rs[u];
    }
  };
  auto slot_of = [](int ti) { return ((ti % KT) + KT) % KT; };

  const bool relu = d.act == SF_ACT_RELU || d.act == SF_ACT_RELU6;
  const float hi = d.act == SF_ACT_RELU6 ? 6.f : 3.0e38f;
  const int nrow = q.nblk * 16;

  for (int unit = blockIdx.x; unit < q.units; unit += gridDim.x) {
    const int tz = unit % q.tparts;
    const int nh = unit / q.tparts;
    const int h = nh % d.Ho, n = nh / d.Ho;
    const int tper = (d.To + q.tparts - 1) / q.tparts;
    const int t0 = tz * tper, t1 = min(d.To, t0 + tper);
    if (t0 >= t1) continue;
    __syncthreads();
    for (int kt = 0; kt < KT - 1; ++kt) {
      const int ti = t0 - d.pT + kt;
      load_slab(n, ti, h);
      store_slab(slot_of(ti));
    }
    load_slab(n, t0 - d.pT + KT - 1, h);
    for (int t = t0; t < t1; ++t) {
      const int tnew = t - d.pT + KT - 1;
      store_slab(slot_of(tnew));
      __syncthreads();
      if (t + 1 < t1) load_slab(n, tnew + 1, h);
      const int base = t - d.pT;
      int aoff[TPW];
#pragma unroll
      for (int u = 0; u < TPW; ++u) aoff[u] = slot_of(base + t_kt[u]) * slab + t_off[u];
      for (int blk = 0; blk < q.nblk; ++blk) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // rows past the last position (pair) recompute the last one and are never stored
        const int pos = PAIR ? min(blk * 16 + fr, ((d.Wo + 1) >> 1) - 1) * 2 * q.ps : min(blk * 16 + fr, d.Wo - 1) * q.ps;
        // the pair's 4-float chunk: j = 32 + fg; past the row's end only for the unstored half of an odd last pair,
        // whose weights are not zero-padded there -> read a zero instead of the neighbouring row
        const bool xin = PAIR && pos + 32 + fg < q.rowf;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
          if (wave + 4 * u >= NTAP) break;  // wave-uniform
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ring + aoff[u] + pos + 16 * c);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wreg[u][c][s], acc, 0, 0, 0);
          }
          if (PAIR) {
            const float ax = xin ? ring[aoff[u] + pos + 32 - 3 * fg] : 0.f;  // aoff carries 4 fg; this chunk wants fg
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, wx[u], acc, 0, 0, 0);
          }
        }
        // C layout: col n = fr (co), rows 4 fg + r (position within the block)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * nrow + blk * 16 + 4 * fg + r) * 16 + fr] = acc[r];
      }
      __syncthreads();  // partial rows complete; nobody reads the ring's oldest slab any more
      const long orow0 = (((long)n * d.To + t) * d.Ho + h) * d.Wo;
      for (int e = tid; e < d.Wo * d.Cout; e += 256) {
        const int w = e / d.Cout, co = e - w * d.Cout;
        const int idx = PAIR ? w * 8 + co : w * 16 + co;  // PAIR: row w >> 1, column 8 (w & 1) + co
        float v = red[idx] + red[nrow * 16 + idx] + red[2 * nrow * 16 + idx] + red[3 * nrow * 16 + idx];
        v = v * (q.scale ? q.scale[co] : 1.f) + (q.bias ? q.bias[co] : 0.f);
        if (relu) v = fminf(fmaxf(v, 0.f), hi);
        q.out[(orow0 + w) * d.out_cs + d.out_coff + co] = v;
      }
      // the next step's partial rows are written only after its own barrier, which also orders these reads
    }
  }
}


// PAIR form with the taps NOT dealt to the wavefronts: a wavefront owns a block of 16 position pairs and runs all 35
// taps for it with every weight in registers (35 x 9 values per lane; the workgroup is alone on its CU, so a
// wavefront has the SIMD's whole 512-register file).  No cross-wavefront sum, no partial rows in LDS, the epilogue
// goes straight from the accumulator to HBM, and a SIX-slot ring (the slab being stored is never one of the five
// being read) leaves one barrier per step of t.
constexpr int RING6 = KT + 1;

__global__ __launch_bounds__(256) void conv_stem_pair_kernel(const StemFwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const sf_conv_desc& d = q.d;
  const int slab = KH * q.rowf;
  float* const ring = lds;  // [RING6][slab]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int slab4 = slab >> 2;
  constexpr int LD4 = 8;

  // column fr = (h, co): w[co][tap][j - ps h]; chunks j = 16 c + 4 fg + s (c = 0, 1) and j = 32 + fg
  float wreg[NTAP][2][4], wx[NTAP];
  const int wco = fr & 7, wsh = (fr >> 3) * q.ps;
#pragma unroll
  for (int tap = 0; tap < NTAP; ++tap) {
    const float* wt = q.w + ((long)wco * NTAP + tap) * d.cin_pad;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int j = 16 * c + 4 * fg + s - wsh;
        wreg[tap][c][s] = (wco < d.Cout && j >= 0 && j < d.Cin) ? wt[j] : 0.f;
      }
    const int jx = 32 + fg - wsh;
    wx[tap] = (wco < d.Cout && jx >= 0 && jx < d.Cin) ? wt[jx] : 0.f;
  }
  const float sc = (q.scale && wco < d.Cout) ? q.scale[wco] : 1.f;
  const float bi = (q.bias && wco < d.Cout) ? q.bias[wco] : 0.f;

  f32x4 rs[LD4];
  auto load_slab = [&](int n, int ti, int h) {
    const bool ok = (unsigned)ti < (unsigned)d.Ti;
    const f32x4* src =
        reinterpret_cast<const f32x4*>(q.in + (((long)n * d.Ti + (ok ? ti : 0)) * d.Hi + (long)h * d.sH) * q.rowf);
#pragma unroll
    for (int u = 0; u < LD4; ++u) {
      const int f = tid + u * 256;
      rs[u] = (ok && f < slab4) ? src[f] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_slab = [&](int slot) {
    f32x4* dst = reinterpret_cast<f32x4*>(ring + slot * slab);
#pragma unroll
    for (int u = 0; u < LD4; ++u) {
      const int f = tid + u * 256;
      if (f < slab4) dst[f] = rs[u];
    }
  };
  auto slot_of = [](int ti) { return ((ti % RING6) + RING6) % RING6; };

  const bool relu = d.act == SF_ACT_RELU || d.act == SF_ACT_RELU6;
  const float hi = d.act == SF_ACT_RELU6 ? 6.f : 3.0e38f;
  const int npair = (d.Wo + 1) >> 1;

  for (int unit = blockIdx.x; unit < q.units; unit += gridDim.x) {
    const int tz = unit % q.tparts;
    const int nh = unit / q.tparts;
    const int h = nh % d.Ho, n = nh / d.Ho;
    const int tper = (d.To + q.tparts - 1) / q.tparts;
    const int t0 = tz * tper, t1 = min(d.To, t0 + tper);
    if (t0 >= t1) continue;
    __syncthreads();  // the previous unit's last step may still read the ring
    for (int kt = 0; kt < KT - 1; ++kt) {
      const int ti = t0 - d.pT + kt;
      load_slab(n, ti, h);
      store_slab(slot_of(ti));
    }
    load_slab(n, t0 - d.pT + KT - 1, h);
    for (int t = t0; t < t1; ++t) {
      const int tnew = t - d.pT + KT - 1;
      store_slab(slot_of(tnew));  // not one of the five slots the step before reads
      __syncthreads();
      if (t + 1 < t1) load_slab(n, tnew + 1, h);
      int soff[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) soff[kt] = slot_of(t - d.pT + kt) * slab;
      const long orow0 = (((long)n * d.To + t) * d.Ho + h) * d.Wo;
      for (int blk = wave; blk < q.nblk; blk += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: chunk 0 / chunks 1 + x
        const int pos = min(blk * 16 + fr, npair - 1) * 2 * q.ps;  // rows past the last pair are never stored
        const bool xin = pos + 32 + fg < q.rowf;                    // see conv_stem_fwd_kernel
        const float* const a0 = ring + pos + 4 * fg;
        // the next tap's fragments are read from LDS while this tap's 9 MFMAs run (one wavefront per SIMD: nothing
        // else hides the ds_read latency); the scheduling barriers keep the compiler from sinking the reads
        f32x4 a_cur[2];
        float x_cur;
        {
          const float* const ap = a0 + soff[0];
          a_cur[0] = *reinterpret_cast<const f32x4*>(ap);
          a_cur[1] = *reinterpret_cast<const f32x4*>(ap + 16);
          x_cur = xin ? ap[32 - 3 * fg] : 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
          f32x4 a_nxt[2] = {a_cur[0], a_cur[1]};
          float x_nxt = x_cur;
          if (tap + 1 < NTAP) {
            const float* const ap = a0 + soff[(tap + 1) / KH] + ((tap + 1) % KH) * q.rowf;
            a_nxt[0] = *reinterpret_cast<const f32x4*>(ap);
            a_nxt[1] = *reinterpret_cast<const f32x4*>(ap + 16);
            x_nxt = xin ? ap[32 - 3 * fg] : 0.f;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0][s], wreg[tap][0][s], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1][s], wreg[tap][1][s], acc1, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x_cur, wx[tap], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          a_cur[0] = a_nxt[0];
          a_cur[1] = a_nxt[1];
          x_cur = x_nxt;
        }
        acc += acc1;
        // C layout: column fr = (h, co), rows 4 fg + r = pair within the block
        if (wco < d.Cout) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int w = 2 * (blk * 16 + 4 * fg + r) + (fr >> 3);
            if (w < d.Wo) {
              float v = acc[r] * sc + bi;
              if (relu) v = fminf(fmaxf(v, 0.f), hi);
              q.out[(orow0 + w) * d.out_cs + d.out_coff + wco] = v;
            }
          }
        }
      }
    }
  }
}

}  // namespace

// Returns SF_OK when the launch was made, 1 when the problem is not this kernel's (the caller takes the implicit-GEMM
// path), or an error.  SF_CONV_STEM=0 switches it off (A/B).
int sf_conv_stem_fwd_try(const sf_conv_desc* d, const float* in, const float* w, const float* scale,
                         const float* bias, const float* res, float* out, hipStream_t stream) {
  static const bool off = [] {
    const char* e = getenv("SF_CONV_STEM");
    return e && e[0] == '0';
  }();
  if (off || res || d->transposed || d->out_cmul != 1 || d->os_T > 1 || d->os_H > 1 || d->os_W > 1) return 1;
  if (d->kT != KT || d->kH != KH || d->kW != 1 || d->sT != 1 || d->sW != 1 || d->pH != 0 || d->pW != 0 || d->dT != 1 ||
      d->dH != 1 || d->dW != 1 || d->cin_pad != 32 || d->Cin > 32 || d->Cout > 16 || d->in_coff != 0 ||
      (d->in_cs % 4) != 0 || !sf_aligned16(in) || d->To != d->Ti + 2 * d->pT - KT + 1)
    return 1;
  if (d->act != SF_ACT_NONE && d->act != SF_ACT_RELU && d->act != SF_ACT_RELU6) return 1;
  StemFwdArgs q;
  q.d = *d; q.in = in; q.w = w; q.scale = scale; q.bias = bias; q.out = out;
  q.rowf = d->Wi * d->in_cs;
  q.ps = d->sW * d->in_cs;
  if ((long)(d->Wo - 1) * q.ps + 32 > q.rowf || (long)(d->Ho - 1) * d->sH + KH > d->Hi) return 1;
  static const bool nopair = [] {
    const char* e = getenv("SF_STEM_PAIR");
    return e && e[0] == '0';
  }();
  const bool pair = !nopair && d->Cout <= 8 && q.ps == 8 && d->Cin <= 28;
  q.nblk = pair ? ((d->Wo + 1) / 2 + 15) / 16 : (d->Wo + 15) / 16;
  if (q.nblk > MAXBLK || KH * q.rowf > 8 * 256 * 4) return 1;
  const size_t lds6 = (size_t)RING6 * KH * q.rowf * sizeof(float);
  static const bool dealt = [] {  // SF_STEM_PAIR=1: the pair form with the taps dealt to the wavefronts (A/B)
    const char* e = getenv("SF_STEM_PAIR");
    return e && e[0] == '1';
  }();
  const bool all_taps = pair && !dealt && lds6 <= 160 * 1024 - 512;
  const size_t lds = all_taps ? lds6 : ((size_t)KT * KH * q.rowf + (size_t)4 * q.nblk * 16 * 16) * sizeof(float);
  if (lds > 160 * 1024 - 512) return 1;
  int best = 1;
  double best_fill = 0.0;
  for (int tp = 1; tp <= 4 && d->To / tp >= 4; ++tp) {
    const long u = (long)d->N * d->Ho * tp;
    const double fill = (double)u / (double)((u + 255) / 256 * 256) - 0.03 * (tp - 1);  // each part re-primes 4 slabs
    if (fill > best_fill) { best_fill = fill; best = tp; }
  }
  q.tparts = best;
  q.units = d->N * d->Ho * best;
  static SfLdsAttr at0, at1, at2;
  if (!sf_ensure_dyn_lds(at0, reinterpret_cast<const void*>(conv_stem_fwd_kernel<false>), 160 * 1024 - 512) ||
      !sf_ensure_dyn_lds(at1, reinterpret_cast<const void*>(conv_stem_fwd_kernel<true>), 160 * 1024 - 512) ||
      !sf_ensure_dyn_lds(at2, reinterpret_cast<const void*>(conv_stem_pair_kernel), 160 * 1024 - 512))
    return SF_ELAUNCH;
  if (all_taps)
    hipLaunchKernelGGL(conv_stem_pair_kernel, dim3(q.units < 256 ? q.units : 256), dim3(256), lds, stream, q);
  else if (pair)
    hipLaunchKernelGGL(conv_stem_fwd_kernel<true>, dim3(q.units < 256 ? q.units : 256), dim3(256), lds, stream, q);
  else
    hipLaunchKernelGGL(conv_stem_fwd_kernel<false>, dim3(q.units < 256 ? q.units : 256), dim3(256), lds, stream, q);
  SF_CHECK_LAUNCH();
  return SF_OK;
}
