// conv_wgrad_wave.hip — weight gradient of a dense 3-D convolution as a per-WAVEFRONT GEMM on the fp32 matrix cores.
//
//   dW[co, tap, ci] = sum_m dz[m, co] * x[row(m, tap), ci]          (m over the B*To*Ho*Wo output positions)
//
// The reduction runs over positions, and both operands lie in HBM position-major with their channels contiguous
// (NDHWC).  One 16-byte load per lane therefore feeds FOUR 16x16x4 MFMA operands at once: lane (r, g) of the A side
// loads dz[m0 + g][co0 + 4r .. 4r+3], and register e of that float4 is the A operand (row r, k = g) of the output row
// set {co0 + 4r + e}; lane (c, g) of the B side loads x[row(m0 + g, tap)][ci0 + 4c .. 4c+3] likewise.  A wave-load is
// then 4 positions x 256 contiguous bytes, and one A + one B load per lane drive 16 MFMAs (a 64 x 64 tile whose rows
// and columns are interleaved mod 4 — undone for free in the store: a lane ends up holding 4 CONSECUTIVE ci of one
// co).  Nothing is staged through LDS and the loop has no barrier; fragments travel three K steps ahead in a register
// ring.  A wavefront owns NA 64-channel blocks of Cout x NB 64-wide blocks of the packed (tap, Cin) axis; the four
// wavefronts of a workgroup split the workgroup's position range and are summed in wavefront order through LDS; the
// position range itself is split over S workgroups whose partial tiles [S][Cout][taps][cin_pad] the finish kernel
// sums in split order — no float atomics anywhere, bit-reproducible.
//
// Replaces the reference's autograd weight gradients of every dense conv with >= 64 channels on both sides
// (resnet_helper.py:182-223, :326-335, video_model_builder.py:128-135, stem_helper.py:157-164 slow pathway);
// conv_wgrad.hip keeps the small-channel and stem variants and the finish kernels.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr unsigned OOB = 0x80000000u;

struct WgWaveArgs {
  sf_conv_desc d;      // the FORWARD conv's descriptor (in = x, out dims = dz dims)
  const float* x;
  const float* dz;
  float* part;         // [S][Cout][ntaps][cin_pad]
  int dz_cs, dz_coff;
  int M, ntaps, S;
  int nci;             // 64-wide blocks per tap = ceil(cin_pad / 64)
  int ncb;             // column blocks = ntaps * nci
  int tiles_b;         // ceil(ncb / NB)
  int tiles;           // ceil(nco / NA) * tiles_b
  long chunk;          // positions per split (multiple of 16)
  unsigned x_bytes, dz_bytes;
  unsigned wo_mul, wo_sh, ho_mul, ho_sh, to_mul, to_sh;
  int plain;           // 1: 1x1x1, stride 1, no padding -> input row == output position
};

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned mul, unsigned sh) {
  return mul ? (__umulhi(n, mul) >> sh) : n;
}

constexpr int DEPTH = 3;  // K steps (of 4 positions) in flight

template <int NA, int NB>
__global__ __launch_bounds__(256) void conv_wgrad_wave_kernel(const WgWaveArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[4 * 16 * 256];  // [wave][16 tiles][64 lanes][4]: one (a, b) block
  const sf_conv_desc& d = p.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int tile = blockIdx.x;
  const int split = blockIdx.y;
  const int tile_a = tile / p.tiles_b, tile_b = tile - tile_a * p.tiles_b;

  // ---- this wavefront's position range: a quarter of the workgroup's chunk, in whole K steps of 4 positions
  const long wg_begin = (long)split * p.chunk;
  const long wg_end = (wg_begin + p.chunk < p.M) ? wg_begin + p.chunk : p.M;
  const long quarter = ((wg_end > wg_begin ? wg_end - wg_begin : 0) + 15) / 16 * 4;  // multiple of 4
  const long m_begin = wg_begin + quarter * wave;
  const long m_end = (m_begin + quarter < wg_end) ? m_begin + quarter : wg_end;
  const int nsteps = m_end > m_begin ? (int)((m_end - m_begin + 3) / 4) : 0;

  // ---- operand descriptors
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, (int)p.dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  unsigned z_col[NA];   // byte offset of this lane's 4 output channels inside a dz row, OOB past Cout
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int co = (tile_a * NA + a) * 64 + fr * 4;
    z_col[a] = co < d.Cout ? (unsigned)(p.dz_coff + co) * 4u : OOB;
  }
  unsigned x_col[NB];   // byte offset of this lane's 4 input channels inside an x row, OOB past Cin / dead blocks
  unsigned x_tap[NB];   // byte offset of the block's tap relative to the window origin
  int x_kt[NB], x_kh[NB], x_kw[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int cb = tile_b * NB + b;
    const int tap = cb / p.nci;
    const int ci = (cb - tap * p.nci) * 64 + fr * 4;
    const bool ok = cb < p.ncb && ci < d.Cin;
    x_col[b] = ok ? (unsigned)(d.in_coff + ci) * 4u : OOB;
    x_kw[b] = tap % d.kW;
    x_kh[b] = (tap / d.kW) % d.kH;
    x_kt[b] = tap / (d.kW * d.kH);
    x_tap[b] = (unsigned)(((x_kt[b] * d.dT * d.Hi + x_kh[b] * d.dH) * d.Wi + x_kw[b] * d.dW) * d.in_cs) * 4u;
  }

  // ---- this lane's position walk.  load() is called for steps 0, 1, 2, .. in order, so the position (lane quarter
  // fg of step s is m_begin + 4 s + fg) is DECODED ONCE and then advanced by 4 per call with carries — fp32 MFMAs run
  // on the vector ALUs (tools/microbench/mfma_coexec.hip), and the three magic divisions + 32-bit multiplies of a
  // per-step decode were a third of this loop's issue time (14 v_mul_lo_u32 + 6 v_mul_hi_u32 per 64 MFMAs).
  int left = (int)(m_end - (m_begin + fg));          // > 0: the position exists
  unsigned zrow, org;                                // byte offsets: dz row, input-window origin (wraps when "negative")
  int wo, ho, to;
  {
    const unsigned mm = left > 0 ? (unsigned)(m_begin + fg) : 0u;
    zrow = mm * (unsigned)p.dz_cs * 4u;
    const unsigned q1 = fast_div(mm, p.wo_mul, p.wo_sh);
    wo = (int)(mm - q1 * (unsigned)d.Wo);
    const unsigned q2 = fast_div(q1, p.ho_mul, p.ho_sh);
    ho = (int)(q1 - q2 * (unsigned)d.Ho);
    const unsigned q3 = fast_div(q2, p.to_mul, p.to_sh);
    to = (int)(q2 - q3 * (unsigned)d.To);
    org = p.plain ? mm * (unsigned)d.in_cs * 4u
                  : (unsigned)(((((int)q3 * d.Ti + to * d.sT - d.pT) * d.Hi + ho * d.sH - d.pH) * d.Wi + wo * d.sW - d.pW) *
                               d.in_cs) * 4u;
  }
  // uniform steps of the origin: 4 positions along w, and the corrections when w / h / t wrap into the next row / frame / clip
  const unsigned z_step = (unsigned)p.dz_cs * 16u;
  const unsigned o_step = p.plain ? (unsigned)d.in_cs * 16u : (unsigned)(d.sW * d.in_cs) * 16u;
  const unsigned o_w = (unsigned)((d.sH * d.Wi - d.Wo * d.sW) * d.in_cs) * 4u;
  const unsigned o_h = (unsigned)(((d.sT * d.Hi - d.Ho * d.sH) * d.Wi) * d.in_cs) * 4u;
  const unsigned o_t = (unsigned)((((d.Ti - d.To * d.sT) * d.Hi) * d.Wi) * d.in_cs) * 4u;

  struct Frag { f32x4 z[NA]; f32x4 x[NB]; };
  auto load = [&](Frag& f) {
    const bool live = left > 0;
#pragma unroll
    for (int a = 0; a < NA; ++a)
      f.z[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rs, live ? zrow + z_col[a] : OOB, 0, 0));
    if (p.plain) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
        f.x[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, live ? org + x_col[b] : OOB, 0, 0));
    } else {
      const int t0 = __mul24(to, d.sT) - d.pT, h0 = __mul24(ho, d.sH) - d.pH, w0 = __mul24(wo, d.sW) - d.pW;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const bool v = live & ((unsigned)(t0 + x_kt[b] * d.dT) < (unsigned)d.Ti) &
                       ((unsigned)(h0 + x_kh[b] * d.dH) < (unsigned)d.Hi) & ((unsigned)(w0 + x_kw[b] * d.dW) < (unsigned)d.Wi);
        f.x[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, v ? org + x_tap[b] + x_col[b] : OOB, 0, 0));
      }
    }
    // advance by one K step (4 positions along w; Wo >= 4, so at most one wrap per level)
    left -= 4;
    zrow += z_step;
    org += o_step;
    if (!p.plain) {
      wo += 4;  // branch-free carries: m = all ones where the coordinate wrapped
      const int m1 = -(int)(wo >= d.Wo);
      wo -= d.Wo & m1;
      org += o_w & (unsigned)m1;
      ho -= m1;
      const int m2 = -(int)(ho >= d.Ho);
      ho -= d.Ho & m2;
      org += o_h & (unsigned)m2;
      to -= m2;
      const int m3 = -(int)(to >= d.To);
      to -= d.To & m3;
      org += o_t & (unsigned)m3;
    }
  };

  f32x4 acc[NA][NB][4][4];  // [a][b][ea][eb]: rows co0 + 64a + 4i + ea, columns ci0 + 4j + eb of the (a, b) block
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int ea = 0; ea < 4; ++ea)
#pragma unroll
        for (int eb = 0; eb < 4; ++eb) acc[a][b][ea][eb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- main loop: a ring of DEPTH fragment sets; each set is reloaded (for DEPTH steps later) right behind its MFMAs
  {
    Frag ring[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) load(ring[u]);
    for (int step = 0; step < nsteps; step += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int ea = 0; ea < 4; ++ea)
#pragma unroll
              for (int eb = 0; eb < 4; ++eb)
                acc[a][b][ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u].z[a][ea], ring[u].x[b][eb],
                                                                         acc[a][b][ea][eb], 0, 0, 0);
        load(ring[u]);   // the next step in order; positions past m_end read as zero
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- sum of the four wavefronts (in wavefront order) and the partial-tile store.  Wavefront w finishes the tiles
  // with ea == w of each (a, b) block: for a fixed (ea, register r) a lane holds the four CONSECUTIVE input channels
  // ci0 + 4c + eb, eb = 0..3 of output channel co0 + 64a + 4(4g + r) + ea -> one 16-byte store.
  float* const part = p.part + (long)split * d.Cout * p.ntaps * d.cin_pad;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (a + b > 0) __syncthreads();
#pragma unroll
      for (int ea = 0; ea < 4; ++ea)
#pragma unroll
        for (int eb = 0; eb < 4; ++eb)
          *reinterpret_cast<f32x4*>(smem + ((wave * 16 + ea * 4 + eb) * 64 + lane) * 4) = acc[a][b][ea][eb];
      __syncthreads();
      f32x4 sum[4];
#pragma unroll
      for (int eb = 0; eb < 4; ++eb) {
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + ((0 * 16 + wave * 4 + eb) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(smem + ((w * 16 + wave * 4 + eb) * 64 + lane) * 4);
        sum[eb] = v;
      }
      const int cb = tile_b * NB + b;
      const int tap = cb / p.nci;
      const int ci = (cb - tap * p.nci) * 64 + fr * 4;
      if (cb < p.ncb && ci < d.cin_pad) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = (tile_a * NA + a) * 64 + 4 * (4 * fg + r) + wave;
          if (co < d.Cout) {
            const f32x4 v = {sum[0][r], sum[1][r], sum[2][r], sum[3][r]};
            *reinterpret_cast<f32x4*>(part + ((long)co * p.ntaps + tap) * d.cin_pad + ci) = v;
          }
        }
      }
    }
}

static void magic(unsigned dv, unsigned* mul, unsigned* sh) {
  if (dv <= 1) { *mul = 0; *sh = 0; return; }
  unsigned l = 0;
  while ((1u << l) < dv) ++l;
  const unsigned long long num = 1ull << (31 + l);
  *mul = (unsigned)((num + dv - 1) / dv);
  *sh = l - 1;
}

int g_wg_enable = 1;   // sf_conv_tune(10, e)
int g_wg_force = -1;   // sf_conv_tune(11, c): 0 = 1x1, 1 = 2x1, 2 = 1x2, 3 = 2x2 blocks per wavefront
int g_wg_target = 0;   // sf_conv_tune(12, n): workgroups to aim at (0: default)

struct WgPlan { int na, nb, S; long chunk; int tiles_b, tiles; };

static bool wg_enabled() {
  static const int env_on = [] {
    const char* e = getenv("SF_WGRAD_WAVE");
    return e ? atoi(e) : 1;
  }();
  return env_on && g_wg_enable;
}

// Blocks per wavefront and position splits.  A (2, 1) / (1, 2) wavefront runs 32 MFMAs per 3 fragment loads; the
// workgroup count aims at two rounds of 256 with the last round as full as the split count allows.
static bool wg_plan(const sf_conv_desc* d, WgPlan* pl) {
  if (!wg_enabled()) return false;
  if (d->Cout < 64 || d->Cin < 64) return false;
  if ((d->Cin % 4) || (d->in_cs % 4) || (d->in_coff % 4) || (d->Cout % 4) || (d->cin_pad % 16)) return false;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M < 1024 || d->Wo < 4) return false;  // the position walk advances 4 along w with at most one wrap per level
  if ((long)d->N * d->Ti * d->Hi * d->Wi * d->in_cs * 4 > 0x7ffffff0L) return false;
  const int ntaps = d->kT * d->kH * d->kW;
  const int nco = sf_cdiv(d->Cout, 64), nci = sf_cdiv(d->cin_pad, 64), ncb = ntaps * nci;
  int na = nco >= 2 ? 2 : 1, nb = na == 2 ? 1 : (ncb >= 2 ? 2 : 1);
  // 1x1x1 layers with a short reduction side (Cin <= 256): one block per wavefront — twice the tiles, half the
  // registers; measured 4-10 % faster than 2x1 / 1x2 on every such layer of cfg #3 (r02b_wgrad_wave_ab.txt)
  static const bool rule11 = getenv("SF_WGRAD_11") == nullptr;  // SF_WGRAD_11=0: A/B
  if (rule11 && ntaps == 1 && d->Cin <= 256) na = nb = 1;
  if (g_wg_force >= 0) { na = (g_wg_force & 1) + 1; nb = (g_wg_force >> 1) + 1; }
  pl->na = na; pl->nb = nb;
  pl->tiles_b = sf_cdiv(ncb, nb);
  pl->tiles = sf_cdiv(nco, na) * pl->tiles_b;
  const long target = g_wg_target > 0 ? g_wg_target : 512;
  long S = (target + pl->tiles - 1) / pl->tiles;
  const long maxS = (M + 511) / 512;   // at least 512 positions per workgroup (128 per wavefront)
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  long best = S;
  double best_fill = 0.0;
  for (long c = S - S / 4; c <= S + S / 4 && c <= maxS; ++c) {
    if (c < 1) continue;
    const long wg = pl->tiles * c;
    const double fill = (double)wg / (double)((wg + 255) / 256 * 256);
    if (fill >= best_fill - 1e-12) { best_fill = fill; best = c; }
  }
  pl->S = (int)best;
  pl->chunk = ((M + best - 1) / best + 15) / 16 * 16;
  return true;
}

template <int NA, int NB>
static int launch(const WgWaveArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((conv_wgrad_wave_kernel<NA, NB>), dim3(a.tiles, a.S), dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

}  // namespace

int sf_wgrad_wave_tune(int knob, int value) {
  if (knob == 10) g_wg_enable = value;
  else if (knob == 11) g_wg_force = value;
  else if (knob == 12) g_wg_target = value;
  else return SF_EINVAL;
  return SF_OK;
}

// Position splits when this kernel takes the problem, else 0 (conv_wgrad.hip then plans its own).
int sf_wgrad_wave_splits(const sf_conv_desc* d) {
  WgPlan pl;
  return wg_plan(d, &pl) ? pl.S : 0;
}

// Returns 1 when the problem is not taken.
int sf_wgrad_wave_try(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, float* partial,
                      hipStream_t stream) {
  WgPlan pl;
  if (!wg_plan(d, &pl)) return 1;
  if ((dz_cs % 4) || (dz_coff % 4) || !sf_aligned16(dz) || !sf_aligned16(x) || !sf_aligned16(partial)) return 1;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M * dz_cs * 4 > 0x7ffffff0L) return 1;
  WgWaveArgs a;
  a.d = *d;
  a.x = x; a.dz = dz; a.part = partial; a.dz_cs = dz_cs; a.dz_coff = dz_coff;
  a.M = (int)M;
  a.ntaps = d->kT * d->kH * d->kW;
  a.S = pl.S;
  a.nci = sf_cdiv(d->cin_pad, 64);
  a.ncb = a.ntaps * a.nci;
  a.tiles_b = pl.tiles_b;
  a.tiles = pl.tiles;
  a.chunk = pl.chunk;
  a.x_bytes = (unsigned)((long)d->N * d->Ti * d->Hi * d->Wi * d->in_cs * 4);
  a.dz_bytes = (unsigned)(M * dz_cs * 4);
  magic((unsigned)d->Wo, &a.wo_mul, &a.wo_sh);
  magic((unsigned)d->Ho, &a.ho_mul, &a.ho_sh);
  magic((unsigned)d->To, &a.to_mul, &a.to_sh);
  a.plain = (a.ntaps == 1 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 0 && d->pH == 0 && d->pW == 0 &&
             d->Ti == d->To && d->Hi == d->Ho && d->Wi == d->Wo) ? 1 : 0;
  if (pl.na == 2 && pl.nb == 2) return launch<2, 2>(a, stream);
  if (pl.na == 2) return launch<2, 1>(a, stream);
  if (pl.nb == 2) return launch<1, 2>(a, stream);
  return launch<1, 1>(a, stream);
}
