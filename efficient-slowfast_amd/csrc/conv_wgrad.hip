// conv_wgrad.hip — weight gradient of a dense 3-D convolution on the fp32 matrix cores (gfx950).
//
//   dW[co, tap, ci] = sum_m dz[m, co] * x[row(m, tap), ci]          (m over the B*To*Ho*Wo output positions)
//
// Per tap this is a GEMM whose REDUCTION dimension is the position index m (up to 8e5 rows) and whose
// output is a small [Cout x Cin] matrix, so the positions are split over `S` workgroups per output tile;
// each writes an fp32 partial tile to a workspace [S][Cout][taps][cin_pad] that the caller sums in a
// fixed order (bit-reproducible — no float atomics).
//
// Both operands are consumed "K-major" exactly as they sit in HBM (NDHWC rows = positions, channels
// contiguous): tiles [32 positions][64 channels] are staged with 16-byte loads into LDS (80-dword pitch)
// and each v_mfma_f32_16x16x4_f32 takes A = dz^T (lane (co, g) reads dz[4s+g][co]) and B = x (lane (ci, g)
// reads x[4s+g][ci]) as conflict-free ds_read_b32.  Block tile 64 co x 64 ci, 4 wavefronts of 32 x 32.
#include "common.h"
#include <stdlib.h>

namespace {

struct WgradArgs {
  sf_conv_desc d;      // the FORWARD conv's descriptor (in = x, out dims = dz dims)
  const float* x;
  const float* dz;     // [M][dz_cs] + dz_coff
  float* part;         // [S][Cout][ntaps][cin_pad]
  int dz_cs, dz_coff;
  int M, ntaps, S, nb_co, nb_ci;
  long chunk;          // positions per split (multiple of 32)
  // grouped convolution (sf_conv_wgrad_grouped): blockIdx.z = group, d.Cin / d.Cout are ONE group's widths; the group
  // index moves both channel windows and the row block of the partial [S][G * Cout][ntaps][cin_pad].  Dense: 1, 0, 0.
  int groups = 1, g_ci = 0, g_co = 0;
};

constexpr int BM = 16;    // positions per stage

template <int BCO, int BCI, int VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
  constexpr int LPZ = BCO + 16;   // LDS row pitches (dwords): consecutive rows land 16 banks apart
  constexpr int LPX = BCI + 16;
  constexpr int TM = BCO / 32, TN = BCI / 32;        // 16x16 MFMA tiles per wave (2 x 2 waves)
  constexpr int ZF = BCO / 4, XF = BCI / 4;          // float4 per tile row
  constexpr int ZIT = (BM * ZF + 255) / 256, XIT = (BM * XF + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * (LPZ + LPX)];
  float* const Zs = smem;                  // [2][BM][LPZ]  dz tile
  float* const Xs = smem + 2 * BM * LPZ;   // [2][BM][LPX]  x tile (gathered for this tap)

  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x;
  int tile = blockIdx.x;
  const int tap = tile % p.ntaps;
  tile /= p.ntaps;
  const int tci = tile % p.nb_ci;
  const int tco = tile / p.nb_ci;
  const int co0 = tco * BCO, ci0 = tci * BCI;
  const int split = blockIdx.y;
  const int grp = (int)blockIdx.z;   // 0 for a dense conv
  const int g_dz_coff = p.dz_coff + grp * p.g_co, g_in_coff = d.in_coff + grp * p.g_ci;
  const long m_begin = (long)split * p.chunk;
  const long m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;

  const int kw = tap % d.kW;
  const int kh = (tap / d.kW) % d.kH;
  const int kt = tap / (d.kW * d.kH);

  // (n, to, ho, wo) of the x rows this thread stages, decoded once and advanced incrementally
  int pos_n[XIT], pos_t[XIT], pos_h[XIT], pos_w[XIT];
#pragma unroll
  for (int i = 0; i < XIT; ++i) {
    const int f = tid + i * 256;
    const long m = m_begin + f / XF;
    pos_w[i] = (int)(m % d.Wo);
    const long t1 = m / d.Wo;
    pos_h[i] = (int)(t1 % d.Ho);
    const long t2 = t1 / d.Ho;
    pos_t[i] = (int)(t2 % d.To);
    pos_n[i] = (int)(t2 / d.To);
  }
  f32x4 rz[ZIT], rx[XIT];
  auto load_stage = [&](long mb) {
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / ZF, lc = (f - lr * ZF) * 4;
      const long m = mb + lr;
      f32x4 vz = {0.f, 0.f, 0.f, 0.f};
      if (f < BM * ZF && m < m_end) {
        const float* zp = p.dz + m * p.dz_cs + g_dz_coff + co0 + lc;
        if (VEC == 4) {
          if (co0 + lc < d.Cout) vz = *reinterpret_cast<const f32x4*>(zp);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co0 + lc + e < d.Cout) vz[e] = zp[e];
        }
      }
      rz[i] = vz;
    }
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / XF, lc = (f - lr * XF) * 4;
      const long m = mb + lr;
      f32x4 vx = {0.f, 0.f, 0.f, 0.f};
      if (f < BM * XF && m < m_end) {
        const int n = pos_n[i];
        const int ti = pos_t[i] * d.sT - d.pT + kt * d.dT;
        const int hi = pos_h[i] * d.sH - d.pH + kh * d.dH;
        const int wi = pos_w[i] * d.sW - d.pW + kw * d.dW;
        if ((unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi) {
          const float* xp = p.x + ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.in_cs + g_in_coff + ci0 + lc;
          if (VEC == 4) {
            if (ci0 + lc < d.Cin) vx = *reinterpret_cast<const f32x4*>(xp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ci0 + lc + e < d.Cin) vx[e] = xp[e];
          }
        }
      }
      rx[i] = vx;
      // next stage: this row moves BM positions on (stages are visited in order, so no division per load)
      pos_w[i] += BM;
      while (pos_w[i] >= d.Wo) {
        pos_w[i] -= d.Wo;
        if (++pos_h[i] == d.Ho) {
          pos_h[i] = 0;
          if (++pos_t[i] == d.To) {
            pos_t[i] = 0;
            ++pos_n[i];
          }
        }
      }
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / ZF, lc = (f - lr * ZF) * 4;
      if (f < BM * ZF) *reinterpret_cast<f32x4*>(Zs + (buf * BM + lr) * LPZ + lc) = rz[i];
    }
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / XF, lc = (f - lr * XF) * 4;
      if (f < BM * XF) *reinterpret_cast<f32x4*>(Xs + (buf * BM + lr) * LPX + lc) = rx[i];
    }
  };

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wco = (wave >> 1) * (BCO / 2), wci = (wave & 1) * (BCI / 2);
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long nst = (m_end > m_begin) ? (m_end - m_begin + BM - 1) / BM : 0;
  if (nst > 0) {
    load_stage(m_begin);
    store_stage(0);
  }
  __syncthreads();
  for (long st = 0; st < nst; ++st) {
    const int buf = (int)(st & 1);
    const bool more = (st + 1) < nst;
    if (more) load_stage(m_begin + (st + 1) * BM);
    const float* zs = Zs + (buf * BM + fg) * LPZ + wco + fr;
    const float* xs = Xs + (buf * BM + fg) * LPX + wci + fr;
#pragma unroll
    for (int s = 0; s < BM / 4; ++s) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = zs[(4 * s) * LPZ + 16 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = xs[(4 * s) * LPX + 16 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }

  // partial tile: rows co = .. + 4*fg + r, cols ci = .. + fr
  float* const base = p.part + ((long)split * p.groups + grp) * d.Cout * p.ntaps * d.cin_pad;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ci = ci0 + wci + 16 * j + fr;
      if (ci >= d.cin_pad) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wco + 16 * i + 4 * fg + r;
        if (co < d.Cout) base[((long)co * p.ntaps + tap) * d.cin_pad + ci] = (ci < d.Cin) ? acc[i][j][r] : 0.f;
      }
    }
}


// Small-channel variant (Cout <= 32 or Cin <= 32: the Fast pathway, lateral / q|k|v projections, the stems):
// a 128-wide tile would be > 90 % padding there.  Tile BCO x BCI in {16, 32}^2; the four wavefronts split
// the POSITIONS of each 64-row stage (16 each) and their accumulators are summed through LDS at the end.
constexpr int BMS = 64;
template <int BCO, int BCI, int VEC>
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(const WgradArgs p) {
  constexpr int LPZ = (BCO == 16) ? 16 : 48;   // pitches = 16 (mod 32): rows g, g+1 hit disjoint bank halves
  constexpr int LPX = (BCI == 16) ? 16 : 48;
  constexpr int TM = BCO / 16, TN = BCI / 16;
  constexpr int ZF = BCO / 4, XF = BCI / 4;
  constexpr int ZIT = (BMS * ZF + 255) / 256, XIT = (BMS * XF + 255) / 256;
  constexpr int STAGE = BMS * (LPZ + LPX);
  constexpr int SMEM = (2 * STAGE > 4 * BCO * BCI) ? 2 * STAGE : 4 * BCO * BCI;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* const Zs = smem;
  float* const Xs = smem + 2 * BMS * LPZ;

  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x;
  int tile = blockIdx.x;
  const int tap = tile % p.ntaps;
  tile /= p.ntaps;
  const int tci = tile % p.nb_ci;
  const int tco = tile / p.nb_ci;
  const int co0 = tco * BCO, ci0 = tci * BCI;
  const int split = blockIdx.y;
  const int grp = (int)blockIdx.z;   // 0 for a dense conv
  const int g_dz_coff = p.dz_coff + grp * p.g_co, g_in_coff = d.in_coff + grp * p.g_ci;
  const long m_begin = (long)split * p.chunk;
  const long m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
  const int kw = tap % d.kW;
  const int kh = (tap / d.kW) % d.kH;
  const int kt = tap / (d.kW * d.kH);

  // (n, to, ho, wo) of the x rows this thread stages, decoded once and advanced incrementally
  int pos_n[XIT], pos_t[XIT], pos_h[XIT], pos_w[XIT];
#pragma unroll
  for (int i = 0; i < XIT; ++i) {
    const int f = tid + i * 256;
    const long m = m_begin + f / XF;
    pos_w[i] = (int)(m % d.Wo);
    const long t1 = m / d.Wo;
    pos_h[i] = (int)(t1 % d.Ho);
    const long t2 = t1 / d.Ho;
    pos_t[i] = (int)(t2 % d.To);
    pos_n[i] = (int)(t2 / d.To);
  }
  f32x4 rz[ZIT], rx[XIT];
  auto load_stage = [&](long mb) {
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / ZF, lc = (f - lr * ZF) * 4;
      const long m = mb + lr;
      f32x4 vz = {0.f, 0.f, 0.f, 0.f};
      if (f < BMS * ZF && m < m_end) {
        const float* zp = p.dz + m * p.dz_cs + g_dz_coff + co0 + lc;
        if (VEC == 4) {
          if (co0 + lc < d.Cout) vz = *reinterpret_cast<const f32x4*>(zp);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co0 + lc + e < d.Cout) vz[e] = zp[e];
        }
      }
      rz[i] = vz;
    }
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / XF, lc = (f - lr * XF) * 4;
      const long m = mb + lr;
      f32x4 vx = {0.f, 0.f, 0.f, 0.f};
      if (f < BMS * XF && m < m_end) {
        const int n = pos_n[i];
        const int ti = pos_t[i] * d.sT - d.pT + kt * d.dT;
        const int hi = pos_h[i] * d.sH - d.pH + kh * d.dH;
        const int wi = pos_w[i] * d.sW - d.pW + kw * d.dW;
        if ((unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi) {
          const float* xp = p.x + ((((long)n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) * d.in_cs + g_in_coff + ci0 + lc;
          if (VEC == 4) {
            if (ci0 + lc < d.Cin) vx = *reinterpret_cast<const f32x4*>(xp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ci0 + lc + e < d.Cin) vx[e] = xp[e];
          }
        }
      }
      rx[i] = vx;
      // next stage: this row moves BMS positions on (stages are visited in order, so no division per load)
      pos_w[i] += BMS;
      while (pos_w[i] >= d.Wo) {
        pos_w[i] -= d.Wo;
        if (++pos_h[i] == d.Ho) {
          pos_h[i] = 0;
          if (++pos_t[i] == d.To) {
            pos_t[i] = 0;
            ++pos_n[i];
          }
        }
      }
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < ZIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / ZF, lc = (f - lr * ZF) * 4;
      if (f < BMS * ZF) *reinterpret_cast<f32x4*>(Zs + (buf * BMS + lr) * LPZ + lc) = rz[i];
    }
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int f = tid + i * 256;
      const int lr = f / XF, lc = (f - lr * XF) * 4;
      if (f < BMS * XF) *reinterpret_cast<f32x4*>(Xs + (buf * BMS + lr) * LPX + lc) = rx[i];
    }
  };

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long nst = (m_end > m_begin) ? (m_end - m_begin + BMS - 1) / BMS : 0;
  if (nst > 0) {
    load_stage(m_begin);
    store_stage(0);
  }
  __syncthreads();
  for (long st = 0; st < nst; ++st) {
    const int buf = (int)(st & 1);
    const bool more = (st + 1) < nst;
    if (more) load_stage(m_begin + (st + 1) * BMS);
    const float* zs = Zs + (buf * BMS + wave * 16 + fg) * LPZ + fr;
    const float* xs = Xs + (buf * BMS + wave * 16 + fg) * LPX + fr;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = zs[(4 * s) * LPZ + 16 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = xs[(4 * s) * LPX + 16 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
  // cross-wave sum through LDS (the staging buffers are free after the last barrier)
  float* const red = smem;  // [4][BCO][BCI]
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * BCO + 16 * i + 4 * fg + r) * BCI + 16 * j + fr] = acc[i][j][r];
  __syncthreads();
  float* const base = p.part + ((long)split * p.groups + grp) * d.Cout * p.ntaps * d.cin_pad;
  for (int e = tid; e < BCO * BCI; e += 256) {
    const int co = co0 + e / BCI, ci = ci0 + e % BCI;
    if (co < d.Cout && ci < d.cin_pad) {
      const float v = red[e] + red[BCO * BCI + e] + red[2 * BCO * BCI + e] + red[3 * BCO * BCI + e];
      base[((long)co * p.ntaps + tap) * d.cin_pad + ci] = (ci < d.Cin) ? v : 0.f;
    }
  }
}


// ---------------------------------------------------------------------------------------------- Fast-pathway stem
// Weight gradient of the "stem trick" convolution (engine.stem_conv_bn_relu: a 5x7x7 / stride (1,2,2) conv over the
// border-padded NDHWC4 clip seen as a 5x7x1 conv over pixels of 8 floats that reads 28 contiguous floats per tap;
// Cout = 8, 3.2 M output positions).  The per-tap kernel above re-reads every input row once per tap — 35 taps x 28
// floats per position = 12.6 GB through L2, 2.8 ms.  Here a workgroup owns an output row h of a clip and walks t:
// the 5 frames x 7 input rows the row needs sit in an LDS ring (129 KB), every step of t brings in ONE new slab
// (7 contiguous rows) and retires the oldest, and all 35 taps x 32 packed channels are accumulated from LDS:
// input traffic drops from 35x to 3.5x re-reads (an input row serves 3.5 output rows).
//   A[m = packed channel j][k = position]  = ring[(t + kt) % 5][kh][w * ps + j]        (ds_read_b32, conflict free)
//   B[k = position][n = co]                = dz row in LDS (lanes n >= Cout hold 0)
// 70 tiles of 16 packed channels (35 taps x 2) are dealt to the 4 wavefronts (18 accumulators each); the 16
// positions of a block are assigned to (MFMA s, lane quarter g) as w0 + 2g + (s & 1) + 8 (s >> 1), which puts the
// quarters 16 banks apart for a pixel stride of 8 floats.  Units (clip, row, t range) are dealt round-robin to a
// persistent grid of one workgroup per CU; each workgroup writes ONE partial (summed by the usual finish path).
constexpr int STEM_KT = 5, STEM_KH = 7, STEM_CO = 8;
constexpr int STEM_NI = (2 * STEM_KT * STEM_KH + 3) / 4;  // accumulator tiles per wavefront

struct StemArgs {
  WgradArgs w;
  int rowf;      // floats per input row = Wi * in_cs
  int ps;        // floats between neighbouring output positions = sW * in_cs
  int nblk;      // 16-position blocks per output row
  int tparts;    // the t range of a (clip, row) is cut into this many units
  int units;     // N * Ho * tparts
  int pair;      // 1: conv_wgrad_stem_pair_kernel (nblk = K steps of 4 position pairs)
};

__global__ __launch_bounds__(256) void conv_wgrad_stem_kernel(const StemArgs q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WgradArgs& p = q.w;
  const sf_conv_desc& d = p.d;
  const int slab = STEM_KH * q.rowf;              // one frame's 7 rows (contiguous in HBM and here)
  float* const ring = lds;                        // [STEM_KT][slab]
  float* const dzs = lds + STEM_KT * slab;        // [nblk * 16][STEM_CO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int slab4 = slab >> 2;                    // float4 per slab
  constexpr int LD4 = 8;                          // float4 per thread per slab (<= 2048 float4 = 32 KB)

  f32x4 acc[STEM_NI];
#pragma unroll
  for (int i = 0; i < STEM_NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // this wavefront's tiles: mt = wave + 4 i -> tap = mt >> 1 = (kt, kh), packed channels 16 (mt & 1) ..
  int t_kt[STEM_NI], t_off[STEM_NI];
#pragma unroll
  for (int i = 0; i < STEM_NI; ++i) {
    const int mt = wave + 4 * i, tap = mt >> 1;
    t_kt[i] = tap / STEM_KH;
    t_off[i] = (tap % STEM_KH) * q.rowf + (mt & 1) * 16 + fr;
  }

  f32x4 rs[LD4];
  float rz[4];
  auto load_slab = [&](int n, int ti, int h) {  // frame ti of clip n, input rows h*sH .. +6 (zeros outside the clip)
    const bool ok = (unsigned)ti < (unsigned)d.Ti;
    const f32x4* src = reinterpret_cast<const f32x4*>(p.x + (((long)n * d.Ti + (ok ? ti : 0)) * d.Hi + (long)h * d.sH) * q.rowf);
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
  auto load_dz = [&](int n, int t, int h) {
    const long m0 = (((long)n * d.To + t) * d.Ho + h) * d.Wo;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * 256, w = e >> 3, co = e & 7;
      rz[u] = (w < d.Wo && co < d.Cout) ? p.dz[(m0 + w) * p.dz_cs + p.dz_coff + co] : 0.f;
    }
  };
  auto store_dz = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * 256;
      if (e < q.nblk * 16 * STEM_CO) dzs[e] = rz[u];
    }
  };

  for (int unit = blockIdx.x; unit < q.units; unit += gridDim.x) {
    const int tz = unit % q.tparts;
    const int nh = unit / q.tparts;
    const int h = nh % d.Ho, n = nh / d.Ho;
    const int tper = (d.To + q.tparts - 1) / q.tparts;
    const int t0 = tz * tper, t1 = min(d.To, t0 + tper);
    if (t0 >= t1) continue;
    __syncthreads();  // the previous unit's last step may still read the ring
    // prime the ring with the frames of step t0 except the newest one: input frame of (t, kt) is t - pT + kt
    for (int kt = 0; kt < STEM_KT - 1; ++kt) {
      const int ti = t0 - d.pT + kt;
      load_slab(n, ti, h);
      store_slab(((ti % STEM_KT) + STEM_KT) % STEM_KT);
    }
    load_slab(n, t0 - d.pT + STEM_KT - 1, h);
    load_dz(n, t0, h);
    for (int t = t0; t < t1; ++t) {
      const int tnew = t - d.pT + STEM_KT - 1;
      store_slab(((tnew % STEM_KT) + STEM_KT) % STEM_KT);
      store_dz();
      __syncthreads();
      if (t + 1 < t1) {  // next step's slab and dz row travel while this step computes
        load_slab(n, tnew + 1, h);
        load_dz(n, t + 1, h);
      }
      const int base = t - d.pT;  // frame of kt = 0
      const float* arow[STEM_NI];
#pragma unroll
      for (int i = 0; i < STEM_NI; ++i) {
        const int ti = base + t_kt[i];
        arow[i] = ring + (((ti % STEM_KT) + STEM_KT) % STEM_KT) * slab + t_off[i];
      }
      for (int blk = 0; blk < q.nblk; ++blk) {
        int wpos[4];
        float b[4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const int w = blk * 16 + 2 * fg + (sidx & 1) + 8 * (sidx >> 1);
          b[sidx] = (fr < STEM_CO) ? dzs[w * STEM_CO + fr] : 0.f;          // rows w >= Wo of dzs hold zeros
          wpos[sidx] = min(w, d.Wo - 1) * q.ps;                               // keep the A read inside the row
        }
#pragma unroll
        for (int i = 0; i < STEM_NI; ++i) {
          if (wave + 4 * i >= 2 * STEM_KT * STEM_KH) break;  // wave-uniform: the last tile slot of waves 2, 3
#pragma unroll
          for (int sidx = 0; sidx < 4; ++sidx)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[i][wpos[sidx]], b[sidx], acc[i], 0, 0, 0);
        }
      }
      __syncthreads();  // everyone is done with the oldest slab and the dz row before they are overwritten
    }
  }
  // this workgroup's partial: tile mt -> tap, packed channels 16 (mt & 1) + 4 fg + r, co = fr
  float* const out = p.part + (long)blockIdx.x * d.Cout * p.ntaps * d.cin_pad;
  if (fr < d.Cout) {
#pragma unroll
    for (int i = 0; i < STEM_NI; ++i) {
      const int mt = wave + 4 * i;
      if (mt >= 2 * STEM_KT * STEM_KH) break;
      const int tap = mt >> 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = (mt & 1) * 16 + 4 * fg + r;
        out[((long)fr * p.ntaps + tap) * d.cin_pad + ci] = (ci < d.Cin) ? acc[i][r] : 0.f;
      }
    }
  }
}

// PAIR form of the stem weight gradient (Cout <= 8, pixel stride 8 floats — the Fast stem itself).  Cout = 8 fills
// half of the 16 MFMA columns, so a K step is a PAIR of neighbouring output positions and the columns are (h, co) =
// position 2p + h, channel co (conv_stem.hip has the forward of the same idea).  A pair reads ONE window of 28 + 8 =
// 36 floats per tap; the 35 x 36 = 1260 window rows are packed densely into 79 tiles of 16 (a tile may straddle two
// taps: every lane carries its own row address) instead of 35 x 2 tiles per SINGLE position:
//   D[R = (tap, j')][(h, co)] = sum_pairs x[ring row(tap)][16 pair + j'] * dz[2 pair + h][co]
//   dW[co][tap][j] = D[(tap, j)][(0, co)] + D[(tap, j + 8)][(1, co)]          (position h sees the window shifted by 8)
// 79 x 14 MFMAs per step of t instead of 70 x 28; the A fragments of the next K step are read from LDS while this
// step's MFMAs run.  The two halves are added when the workgroup writes its partial (through LDS, the ring is free).
constexpr int STEM_WIN = 36;
constexpr int STEM_ROWS = STEM_KT * STEM_KH * STEM_WIN;   // 1260
constexpr int STEM_MT = (STEM_ROWS + 15) / 16;            // 79

// NW wavefronts per workgroup (4 or 8) share the ring; with 8, two wavefronts per SIMD cover each other's LDS
// latencies and instruction issue (the f32 MFMA shares the vector ALUs with everything else a wavefront does).
template <int NW>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_stem_pair_kernel(const StemArgs q) {
  constexpr int STEM_NI2 = (STEM_MT + NW - 1) / NW;  // accumulator tiles per wavefront (20 / 10)
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WgradArgs& p = q.w;
  const sf_conv_desc& d = p.d;
  const int slab = STEM_KH * q.rowf;
  float* const ring = lds;                        // [STEM_KT][slab]
  float* const dzs = lds + STEM_KT * slab;        // [nk * 4 pairs][2 positions][STEM_CO]; rows >= Wo hold zeros
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int slab4 = slab >> 2;
  constexpr int LD4 = 2048 / NT;  // float4s of a slab per thread (capacity 8 * 256, checked by stem_plan)
  constexpr int LDZ = 1024 / NT;  // dz floats per thread
  const int npair = (d.Wo + 1) >> 1, nk = q.nblk;  // nblk = K steps of 4 pairs

  f32x4 acc[STEM_NI2];
#pragma unroll
  for (int i = 0; i < STEM_NI2; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // tile mt = wave + NW i, this lane's window row R = 16 mt + fr -> (tap, j'); rows past 1260 repeat the last one
  int r_kt[STEM_NI2], r_off[STEM_NI2];
#pragma unroll
  for (int i = 0; i < STEM_NI2; ++i) {
    const int R = min(16 * (wave + NW * i) + fr, STEM_ROWS - 1);
    const int tap = R / STEM_WIN, jw = R - tap * STEM_WIN;
    r_kt[i] = tap / STEM_KH;
    r_off[i] = (tap % STEM_KH) * q.rowf + jw;
  }

  f32x4 rs[LD4];
  float rz[LDZ];
  auto load_slab = [&](int n, int ti, int h) {
    const bool ok = (unsigned)ti < (unsigned)d.Ti;
    const f32x4* src = reinterpret_cast<const f32x4*>(p.x + (((long)n * d.Ti + (ok ? ti : 0)) * d.Hi + (long)h * d.sH) * q.rowf);
#pragma unroll
    for (int u = 0; u < LD4; ++u) {
      const int f = tid + u * NT;
      rs[u] = (ok && f < slab4) ? src[f] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_slab = [&](int slot) {
    f32x4* dst = reinterpret_cast<f32x4*>(ring + slot * slab);
#pragma unroll
    for (int u = 0; u < LD4; ++u) {
      const int f = tid + u * NT;
      if (f < slab4) dst[f] = rs[u];
    }
  };
  auto load_dz = [&](int n, int t, int h) {
    const long m0 = (((long)n * d.To + t) * d.Ho + h) * d.Wo;
#pragma unroll
    for (int u = 0; u < LDZ; ++u) {
      const int e = tid + u * NT, w = e >> 3, co = e & 7;
      rz[u] = (w < d.Wo && co < d.Cout) ? p.dz[(m0 + w) * p.dz_cs + p.dz_coff + co] : 0.f;
    }
  };
  auto store_dz = [&]() {
#pragma unroll
    for (int u = 0; u < LDZ; ++u) {
      const int e = tid + u * NT;
      if (e < nk * 64) dzs[e] = rz[u];
    }
  };

  for (int unit = blockIdx.x; unit < q.units; unit += gridDim.x) {
    const int tz = unit % q.tparts;
    const int nh = unit / q.tparts;
    const int h = nh % d.Ho, n = nh / d.Ho;
    const int tper = (d.To + q.tparts - 1) / q.tparts;
    const int t0 = tz * tper, t1 = min(d.To, t0 + tper);
    if (t0 >= t1) continue;
    __syncthreads();
    for (int kt = 0; kt < STEM_KT - 1; ++kt) {
      const int ti = t0 - d.pT + kt;
      load_slab(n, ti, h);
      store_slab(((ti % STEM_KT) + STEM_KT) % STEM_KT);
    }
    load_slab(n, t0 - d.pT + STEM_KT - 1, h);
    load_dz(n, t0, h);
    for (int t = t0; t < t1; ++t) {
      const int tnew = t - d.pT + STEM_KT - 1;
      store_slab(((tnew % STEM_KT) + STEM_KT) % STEM_KT);
      store_dz();
      __syncthreads();
      if (t + 1 < t1) {
        load_slab(n, tnew + 1, h);
        load_dz(n, t + 1, h);
      }
      const int b5 = (((t - d.pT) % STEM_KT) + STEM_KT) % STEM_KT;  // ring slot of kt = 0
      int addr[STEM_NI2];
#pragma unroll
      for (int i = 0; i < STEM_NI2; ++i) {
        int sl = b5 + r_kt[i];
        sl = sl >= STEM_KT ? sl - STEM_KT : sl;
        addr[i] = sl * slab + r_off[i];
      }
      // K step ks: lane quarter fg holds pair 4 ks + fg (A: the pair's window float, B: its two dz rows)
      float a_cur[STEM_NI2], b_cur;
      {
        const int po = min(fg, npair - 1) * 16;
#pragma unroll
        for (int i = 0; i < STEM_NI2; ++i) a_cur[i] = ring[addr[i] + po];
        b_cur = dzs[fg * 16 + fr];
      }
      for (int ks = 0; ks < nk; ++ks) {
        float a_nxt[STEM_NI2], b_nxt;
        {
          const int pn = 4 * (ks + 1) + fg;                       // past the last step: re-reads in-range data, unused
          const int po = min(pn, npair - 1) * 16;
#pragma unroll
          for (int i = 0; i < STEM_NI2; ++i) a_nxt[i] = ring[addr[i] + po];
          b_nxt = dzs[min(pn, 4 * nk - 1) * 16 + fr];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEM_NI2; ++i) {
          if (wave + NW * i >= STEM_MT) break;  // wave-uniform: the last tile slot of the last wavefronts
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], b_cur, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEM_NI2; ++i) a_cur[i] = a_nxt[i];
        b_cur = b_nxt;
      }
      __syncthreads();  // everyone is done with the oldest slab and the dz rows before they are overwritten
    }
  }
  // ---- this workgroup's partial: D through LDS (the ring is free), the two position halves added on the way out
  __syncthreads();
  float* const Dl = lds;  // [STEM_MT * 16][16]
#pragma unroll
  for (int i = 0; i < STEM_NI2; ++i) {
    const int mt = wave + NW * i;
    if (mt >= STEM_MT) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) Dl[(16 * mt + 4 * fg + r) * 16 + fr] = acc[i][r];
  }
  __syncthreads();
  float* const out = p.part + (long)blockIdx.x * d.Cout * p.ntaps * d.cin_pad;
  const int per_co = p.ntaps * d.cin_pad;
  for (int e = tid; e < d.Cout * per_co; e += NT) {
    const int co = e / per_co, rem = e - co * per_co;
    const int tap = rem / d.cin_pad, j = rem - tap * d.cin_pad;
    float v = 0.f;
    if (j < d.Cin) v = Dl[(tap * STEM_WIN + j) * 16 + co] + Dl[(tap * STEM_WIN + j + q.ps) * 16 + 8 + co];
    out[e] = v;
  }
}

// Does this problem take the stem kernel, and with how many workgroups (= partials)?
static bool stem_plan(const sf_conv_desc* d, int dz_cs, StemArgs* q) {
  static const bool off = [] { const char* e = getenv("SF_WGRAD_STEM"); return e && e[0] == '0'; }();
  if (off) return false;
  if (d->kT != STEM_KT || d->kH != STEM_KH || d->kW != 1 || d->sT != 1 || d->sW != 1 || d->pH != 0 || d->pW != 0 ||
      d->dT != 1 || d->dH != 1 || d->dW != 1 || d->cin_pad != 32 || d->Cin > 32 || d->Cout > STEM_CO ||
      d->in_coff != 0 || (d->in_cs % 4) != 0 || d->To != d->Ti + 2 * d->pT - STEM_KT + 1)
    return false;
  const int rowf = d->Wi * d->in_cs, ps = d->sW * d->in_cs;
  if ((long)(d->Wo - 1) * ps + 32 > rowf) return false;                       // a tap reads 32 floats from its pixel
  if ((long)(d->Ho - 1) * d->sH + STEM_KH > d->Hi) return false;
  const int nblk = (d->Wo + 15) / 16;
  if (nblk * 16 * STEM_CO > 1024 || STEM_KH * rowf > 8 * 256 * 4) return false; // staging capacity per thread
  const long lds = ((long)STEM_KT * STEM_KH * rowf + (long)nblk * 16 * STEM_CO) * 4;
  if (lds > 160 * 1024 - 512) return false;
  static const bool nopair = [] { const char* e = getenv("SF_STEM_PAIR"); return e && e[0] == '0'; }();
  // pair form: the window of the last pair (28 + 8 floats from its first position) stays inside the staged slab
  const int nk = ((d->Wo + 1) / 2 + 3) / 4;
  // (pair p starts at float 16 p of the row and reads 36 floats: 16 (npair - 1) + 36 <= rowf, else the last pair of
  // an odd-Wo row would run into the next kh row of the slab — fall back to the one-position form)
  const int npair = (d->Wo + 1) / 2;
  const bool pair = !nopair && ps == 8 && d->Cin <= 28 && d->Cout <= STEM_CO && nk * 64 <= 1024 &&
                    16L * (npair - 1) + 36 <= rowf;
  if (q) {
    q->rowf = rowf; q->ps = ps; q->nblk = pair ? nk : nblk;
    q->pair = pair ? 1 : 0;
    // units: (clip, output row) x t parts; cut t so that the units deal evenly onto 256 workgroups
    int best = 1;
    double best_fill = 0.0;
    for (int tp = 1; tp <= 4 && d->To / tp >= 4; ++tp) {
      const long u = (long)d->N * d->Ho * tp;
      const double fill = (double)u / (double)((u + 255) / 256 * 256) - 0.03 * (tp - 1);  // each part re-primes 4 slabs
      if (fill > best_fill) { best_fill = fill; best = tp; }
    }
    q->tparts = best;
    q->units = d->N * d->Ho * best;
  }
  (void)dz_cs;
  return true;
}

static int stem_workgroups(const StemArgs& q) { return q.units < 256 ? q.units : 256; }

// tile shape for a problem: 128 on a side only when that side has >= 128 channels
static inline void wgrad_tile(const sf_conv_desc* d, int* bco, int* bci) {
  if (d->Cout <= 32 || d->Cin <= 32) {  // small-channel kernel
    *bco = d->Cout <= 16 ? 16 : 32;
    *bci = d->Cin <= 16 ? 16 : 32;
    return;
  }
  *bco = d->Cout >= 128 ? 128 : 64;
  *bci = d->Cin >= 128 ? 128 : 64;
}

template <int BCO, int BCI>
static int launch_wgrad_small(WgradArgs a, bool vec4, hipStream_t s) {
  a.nb_co = sf_cdiv(a.d.Cout, BCO);
  a.nb_ci = sf_cdiv(a.d.cin_pad, BCI);   // cover the zero-padded packed width too
  dim3 grid(a.nb_co * a.nb_ci * a.ntaps, a.S, a.groups);
  if (vec4)
    hipLaunchKernelGGL((conv_wgrad_small_kernel<BCO, BCI, 4>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_small_kernel<BCO, BCI, 1>), grid, dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int BCO, int BCI>
static int launch_wgrad(WgradArgs a, bool vec4, hipStream_t s) {
  a.nb_co = sf_cdiv(a.d.Cout, BCO);
  a.nb_ci = sf_cdiv(a.d.Cin, BCI);
  dim3 grid(a.nb_co * a.nb_ci * a.ntaps, a.S, a.groups);
  if (vec4)
    hipLaunchKernelGGL((conv_wgrad_kernel<BCO, BCI, 4>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<BCO, BCI, 1>), grid, dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// dst[co][ci][tap] (+)= sum_s part[s][co][tap'][c']: the split partials summed in a fixed order AND un-packed to the
// nn.Conv3d weight layout [Cout, Cin, kT, kH, kW] in one pass.  One workgroup per (co, 64-channel tile): partial
// rows are read coalesced along c, transposed through LDS ([c][tap] — which IS the destination order), and the
// ptaps*64 contiguous destination floats are written / accumulated coalesced.
constexpr int FIN_C = 64;
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ part, int S, int Cout, int ptaps,
                                                           int cin_pad, int Cin, float* __restrict__ dst,
                                                           int accumulate) {
  extern __shared__ float tile[];  // [FIN_C][ptaps] + [4][FIN_C] partial sums
  float* const red = tile + FIN_C * ptaps;
  const int co = blockIdx.x, c0 = blockIdx.y * FIN_C;
  const int c = threadIdx.x & (FIN_C - 1), q = threadIdx.x / FIN_C;
  const long stride = (long)Cout * ptaps * cin_pad;
  // the four thread groups take the splits s = q, q + 4, ... of every tap (the splits number up to ~100 for small
  // weights over many positions); the four sub-sums are combined in group order — a fixed summation order
  for (int tap = 0; tap < ptaps; ++tap) {
    float v = 0.f;
    if (c0 + c < cin_pad) {
      const float* src = part + ((long)co * ptaps + tap) * cin_pad + c0 + c;
      int s = q;
      for (; s + 12 < S; s += 16) {  // four of this group's splits per trip, loads issued before the adds
        const float t0 = src[s * stride], t1 = src[(s + 4) * stride], t2 = src[(s + 8) * stride],
                    t3 = src[(s + 12) * stride];
        v += t0;
        v += t1;
        v += t2;
        v += t3;
      }
      for (; s < S; s += 256 / FIN_C) v += src[s * stride];
    }
    red[q * FIN_C + c] = v;
    __syncthreads();
    if (q == 0) tile[c * ptaps + tap] = (red[c] + red[FIN_C + c]) + (red[2 * FIN_C + c] + red[3 * FIN_C + c]);
    __syncthreads();
  }
  const int ncols = (Cin - c0) < FIN_C ? (Cin - c0) : FIN_C;  // real channels in this tile
  if (ncols <= 0) return;
  float* out = dst + ((long)co * Cin + c0) * ptaps;
  for (int e = threadIdx.x; e < ncols * ptaps; e += 256) out[e] = accumulate ? out[e] + tile[e] : tile[e];
}

// Small weights (the Fast pathway, the lateral / q|k|v projections: <= 64 K partial elements, up to ~1000 splits): the
// kernel above would run Cout x 1 workgroups that walk taps x S/4 strided loads one barrier pair per tap (10-14 us for
// a 2 KB result).  Here a workgroup owns EPW consecutive elements of the partial layout and 256 / EPW groups of splits
// (EPW = 16 for long split lists): thread (e, q) adds splits q, q + SG, ... in order, eight loads in flight, the SG
// sub-sums are combined in group order — a fixed summation order — and thread (e, 0) writes its element un-packed.
template <int EPW>
__global__ __launch_bounds__(256) void wgrad_finish_flat_kernel(const float* __restrict__ part, int S, int total,
                                                                int ptaps, int cin_pad, int Cin,
                                                                float* __restrict__ dst, int accumulate) {
  constexpr int SG = 256 / EPW;
  __shared__ float red[256];
  const int e = threadIdx.x % EPW, q = threadIdx.x / EPW;
  const int idx = blockIdx.x * EPW + e;
  float v = 0.f;
  if (idx < total) {
    const float* src = part + idx;
    int s = q;
    for (; s + 7 * SG < S; s += 8 * SG) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[(long)(s + u * SG) * total];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < S; s += SG) v += src[(long)s * total];
  }
  red[threadIdx.x] = v;
  __syncthreads();
  if (q == 0 && idx < total) {
    float t = red[e];
#pragma unroll
    for (int k = 1; k < SG; ++k) t += red[k * EPW + e];
    const int c = idx % cin_pad;
    const int r = idx / cin_pad;
    const int tap = r % ptaps, co = r / ptaps;
    if (c < Cin) {
      float* o = dst + ((long)co * Cin + c) * ptaps + tap;
      *o = accumulate ? *o + t : t;
    }
  }
}

// stem layout: a packed "channel" c' is (kw, ci) = (c' / 4, c' % 4) and a packed tap is (kt, kh); tiny tensors.
__global__ void wgrad_finish_stem_kernel(const float* __restrict__ part, int S, int Cout, int ptaps, int cin_pad,
                                         int Cin, int fold_kw, float* __restrict__ dst, int accumulate, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % cin_pad);
  const long r = idx / cin_pad;
  const int tap = (int)(r % ptaps);
  const int co = (int)(r / ptaps);
  const int kw = c >> 2, ci = c & 3;
  if (kw >= fold_kw || ci >= Cin) return;
  const long o = (((long)co * Cin + ci) * ptaps + tap) * fold_kw + kw;
  const long stride = (long)Cout * ptaps * cin_pad;
  float v = 0.f;
  for (int s = 0; s < S; ++s) v += part[s * stride + idx];
  dst[o] = accumulate ? dst[o] + v : v;
}

}  // namespace

extern "C" int sf_conv_wgrad_finish(const float* partial, int S, int Cout, int packed_taps, int cin_pad, int Cin,
                                    int fold_kw, float* dst, int accumulate, void* stream) {
  if (!partial || !dst || S <= 0 || Cout <= 0 || packed_taps <= 0 || cin_pad <= 0 || Cin <= 0 || fold_kw < 0)
    return SF_EINVAL;
  if (fold_kw > 0 ? (Cin > 4 || fold_kw * 4 > cin_pad) : Cin > cin_pad) return SF_EINVAL;
  if (fold_kw > 0) {
    const long total = (long)Cout * packed_taps * cin_pad;
    hipLaunchKernelGGL(wgrad_finish_stem_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, partial,
                       S, Cout, packed_taps, cin_pad, Cin, fold_kw, dst, accumulate, total);
  } else if ((long)Cout * packed_taps * cin_pad <= 65536L) {
    const int total = Cout * packed_taps * cin_pad;
    if (S > 32)
      hipLaunchKernelGGL(wgrad_finish_flat_kernel<16>, dim3(sf_cdiv(total, 16)), dim3(256), 0, (hipStream_t)stream,
                         partial, S, total, packed_taps, cin_pad, Cin, dst, accumulate);
    else
      hipLaunchKernelGGL(wgrad_finish_flat_kernel<64>, dim3(sf_cdiv(total, 64)), dim3(256), 0, (hipStream_t)stream,
                         partial, S, total, packed_taps, cin_pad, Cin, dst, accumulate);
  } else {
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(Cout, sf_cdiv(cin_pad, FIN_C)), dim3(256),
                       (size_t)FIN_C * (packed_taps + 4) * sizeof(float), (hipStream_t)stream, partial, S, Cout, packed_taps,
                       cin_pad, Cin, dst, accumulate);
  }
  SF_CHECK_LAUNCH();
  return SF_OK;
}

int sf_wgrad_rows_splits(const sf_conv_desc* d);  // conv_wgrad_rows.hip
int sf_wgrad_rows_try(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, float* partial,
                      hipStream_t stream);
int sf_wgrad_wave_splits(const sf_conv_desc* d);  // conv_wgrad_wave.hip
int sf_wgrad_wave_try(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, float* partial,
                      hipStream_t stream);

// Position splits of the LDS-tiled kernels above for a problem of `groups` block-diagonal GEMMs of d's widths each.
static int tiled_wgrad_splits(const sf_conv_desc* d, int groups) {
  int bco, bci;
  wgrad_tile(d, &bco, &bci);
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  const long tiles = (long)sf_cdiv(d->Cout, bco) * sf_cdiv(bco <= 32 ? d->cin_pad : d->Cin, bci) * d->kT * d->kH *
                     d->kW * groups;
  static const long target = [] {
    const char* e = getenv("SF_WGRAD_WGS");  // tuning aid
    return e ? atol(e) : 768L;
  }();
  // aim at ~768 workgroups (3 per CU): every workgroup writes its whole partial tile, so the partial traffic (and
  // the sum over S afterwards) grows with the split count — 1536 measured 75.5 ms per train step, 768 75.1, 256 76.0
  long S = (target + tiles - 1) / tiles;
  const long maxS = (M + 255) / 256;              // at least 256 positions per split
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  if (S > 1024) S = 1024;
  // Equal workgroups run in rounds of 256 (one per CU) and the launch ends with its busiest CU: 9 tiles x 171 splits
  // = 1539 workgroups is 6.01 rounds -> 7.  Among the split counts within 25 % of the target take the one that
  // wastes the least of its last round (ties: the larger, i.e. shorter workgroups).
  long best = S;
  double best_fill = 0.0;
  for (long c = S - S / 4; c <= S + S / 4 && c <= maxS && c <= 1024; ++c) {
    if (c < 1) continue;
    const long wg = tiles * c;
    const double fill = (double)wg / (double)((wg + 255) / 256 * 256);
    if (fill >= best_fill - 1e-12) { best_fill = fill; best = c; }
  }
  return (int)best;
}

// Number of position splits the kernel will use for this problem (the caller sizes the workspace with it).
extern "C" int sf_conv_wgrad_splits(const sf_conv_desc* d) {
  if (!d) return 0;
  StemArgs sq;
  if (stem_plan(d, 0, &sq)) return stem_workgroups(sq);
  {
    const int s = sf_wgrad_rows_splits(d);  // small-channel stride-1 "same" layers: the rows kernel's plan
    if (s > 0) return s;
  }
  {
    const int s = sf_wgrad_wave_splits(d);  // >= 64 channels on both sides: the per-wavefront kernel's own plan
    if (s > 0) return s;
  }
  return tiled_wgrad_splits(d, 1);
}

static bool grouped_desc(const sf_conv_desc* d, int groups, sf_conv_desc* g) {
  if (!d || groups < 1 || groups > 65535 || d->Cin <= 0 || d->Cout <= 0 || d->Cin % groups || d->Cout % groups)
    return false;
  *g = *d;
  g->Cin = d->Cin / groups;
  g->Cout = d->Cout / groups;
  return g->cin_pad >= g->Cin;
}

// Weight gradient of nn.Conv3d(groups = G) as ONE launch: group = grid z.  `d` is the forward descriptor of the WHOLE
// layer (cin_pad = the packed width of one group's Cin / G channels); partial [S][Cout][taps][cin_pad] holds group g's
// rows at [g * Cout / G, (g + 1) * Cout / G), i.e. sf_conv_wgrad_finish(partial, S, Cout, taps, cin_pad, Cin / G, ...)
// stores the gradient in nn.Conv3d's own grouped layout [Cout][Cin / G][kT][kH][kW].
extern "C" int sf_conv_wgrad_grouped_splits(const sf_conv_desc* d, int groups) {
  sf_conv_desc g;
  if (!grouped_desc(d, groups, &g)) return 0;
  return groups == 1 ? sf_conv_wgrad_splits(d) : tiled_wgrad_splits(&g, groups);
}

extern "C" int sf_conv_wgrad_grouped(const sf_conv_desc* d, int groups, const float* x, const float* dz, int dz_cs,
                                     int dz_coff, float* partial, void* stream) {
  if (groups == 1) return sf_conv_wgrad(d, x, dz, dz_cs, dz_coff, partial, stream);
  sf_conv_desc g;
  if (!x || !dz || !partial || !grouped_desc(d, groups, &g)) return SF_EINVAL;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M <= 0 || M > 0x7fffffffL) return SF_EINVAL;
  WgradArgs a;
  a.d = g;
  a.x = x; a.dz = dz; a.part = partial; a.dz_cs = dz_cs; a.dz_coff = dz_coff;
  a.M = (int)M;
  a.ntaps = d->kT * d->kH * d->kW;
  a.S = tiled_wgrad_splits(&g, groups);
  a.nb_co = a.nb_ci = 0;
  a.groups = groups;
  a.g_ci = g.Cin;
  a.g_co = g.Cout;
  a.chunk = ((M + a.S - 1) / a.S + BMS - 1) / BMS * BMS;
  const bool vec4 = (g.Cin % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && sf_aligned16(x) &&
                    (g.Cout % 4 == 0) && (dz_cs % 4 == 0) && (dz_coff % 4 == 0) && sf_aligned16(dz);
  int bco, bci;
  wgrad_tile(&g, &bco, &bci);
  hipStream_t s = (hipStream_t)stream;
  if (bco <= 32) {
    if (bco == 16 && bci == 16) return launch_wgrad_small<16, 16>(a, vec4, s);
    if (bco == 16) return launch_wgrad_small<16, 32>(a, vec4, s);
    if (bci == 16) return launch_wgrad_small<32, 16>(a, vec4, s);
    return launch_wgrad_small<32, 32>(a, vec4, s);
  }
  if (bco == 128 && bci == 128) return launch_wgrad<128, 128>(a, vec4, s);
  if (bco == 128) return launch_wgrad<128, 64>(a, vec4, s);
  if (bci == 128) return launch_wgrad<64, 128>(a, vec4, s);
  return launch_wgrad<64, 64>(a, vec4, s);
}

extern "C" int sf_conv_wgrad(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff,
                             float* partial, void* stream) {
  if (!d || !x || !dz || !partial) return SF_EINVAL;
  if (d->Cin <= 0 || d->Cout <= 0 || d->cin_pad < d->Cin) return SF_EINVAL;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M <= 0 || M > 0x7fffffffL) return SF_EINVAL;
  WgradArgs a;
  a.d = *d;
  a.x = x; a.dz = dz; a.part = partial; a.dz_cs = dz_cs; a.dz_coff = dz_coff;
  a.M = (int)M;
  a.ntaps = d->kT * d->kH * d->kW;
  a.S = sf_conv_wgrad_splits(d);
  a.nb_co = a.nb_ci = 0;
  a.chunk = 0;
  StemArgs sq;
  if (sf_aligned16(x) && stem_plan(d, dz_cs, &sq)) {
    sq.w = a;
    size_t lds = ((size_t)STEM_KT * STEM_KH * sq.rowf + (size_t)sq.nblk * (sq.pair ? 64 : 16 * STEM_CO)) * sizeof(float);
    if (sq.pair && lds < (size_t)STEM_MT * 16 * 16 * sizeof(float)) lds = (size_t)STEM_MT * 16 * 16 * sizeof(float);
    static SfLdsAttr at0, at1;  // raise the dynamic-LDS cap once per device (129 KB of the CU's 160 KB)
    if (!sf_ensure_dyn_lds(at0, reinterpret_cast<const void*>(conv_wgrad_stem_kernel), 160 * 1024 - 512) ||
        !sf_ensure_dyn_lds(at1, reinterpret_cast<const void*>(conv_wgrad_stem_pair_kernel<4>), 160 * 1024 - 512))
      return SF_ELAUNCH;
    static SfLdsAttr at2;
    if (!sf_ensure_dyn_lds(at2, reinterpret_cast<const void*>(conv_wgrad_stem_pair_kernel<8>), 160 * 1024 - 512))
      return SF_ELAUNCH;
    static const int stem_nw = [] { const char* e = getenv("SF_STEM_NW"); return e ? atoi(e) : 8; }();
    if (sq.pair && stem_nw == 8)
      hipLaunchKernelGGL(conv_wgrad_stem_pair_kernel<8>, dim3(stem_workgroups(sq)), dim3(512), lds,
                         (hipStream_t)stream, sq);
    else if (sq.pair)
      hipLaunchKernelGGL(conv_wgrad_stem_pair_kernel<4>, dim3(stem_workgroups(sq)), dim3(256), lds, (hipStream_t)stream,
                         sq);
    else
      hipLaunchKernelGGL(conv_wgrad_stem_kernel, dim3(stem_workgroups(sq)), dim3(256), lds, (hipStream_t)stream, sq);
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
  {
    const int rc = sf_wgrad_rows_try(d, x, dz, dz_cs, dz_coff, partial, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  {
    const int rc = sf_wgrad_wave_try(d, x, dz, dz_cs, dz_coff, partial, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  a.chunk = ((M + a.S - 1) / a.S + BMS - 1) / BMS * BMS;
  const bool vec4 = (d->Cin % 4 == 0) && (d->in_cs % 4 == 0) && (d->in_coff % 4 == 0) && sf_aligned16(x) &&
                    (d->Cout % 4 == 0) && (dz_cs % 4 == 0) && (dz_coff % 4 == 0) && sf_aligned16(dz);
  int bco, bci;
  wgrad_tile(d, &bco, &bci);
  hipStream_t s = (hipStream_t)stream;
  if (bco <= 32) {
    if (bco == 16 && bci == 16) return launch_wgrad_small<16, 16>(a, vec4, s);
    if (bco == 16) return launch_wgrad_small<16, 32>(a, vec4, s);
    if (bci == 16) return launch_wgrad_small<32, 16>(a, vec4, s);
    return launch_wgrad_small<32, 32>(a, vec4, s);
  }
  if (bco == 128 && bci == 128) return launch_wgrad<128, 128>(a, vec4, s);
  if (bco == 128) return launch_wgrad<128, 64>(a, vec4, s);
  if (bci == 128) return launch_wgrad<64, 128>(a, vec4, s);
  return launch_wgrad<64, 64>(a, vec4, s);
}
