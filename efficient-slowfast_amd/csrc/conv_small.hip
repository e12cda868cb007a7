// conv_small.hip — bandwidth-class dense 3-D convolution for SMALL channel counts (gfx950): the Fast pathway's
// C <= 32 bottlenecks (resnet_helper.py:182-223 at dim_inner = 8 / 16 / 32: 3x1x1, 1x3x3 and 1x1x1 layers) and their
// data gradients.
//
// Why not the matrix cores: these layers are 1-1.2 GFLOP over 25-130 MB, i.e. HBM-bound by 2-4x, and on gfx950 the
// fp32 MFMA runs at the vector rate anyway (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD either way).  What the implicit-GEMM
// kernels of conv_wave.hip pay for them is LOAD INSTRUCTIONS: every tap of every 16-row block is its own 1 KiB
// buffer_load (half empty at Cin = 8), ~60 cycles of the CU's load path each — 450 k of them for the 8 -> 8 1x3x3 layer
// = 50 us for a layer whose bytes take 8 us (profiles/r03_conv_per_shape_train.txt: 0.11 of the HBM roof).
//
// Here a workgroup stages the input of a block of output positions ONCE, with coalesced 16-byte loads, into LDS — for
// the 1x3x3 layers a halo tile (TH + 2) x (W + 2) of one frame, for 3x1x1 the three temporal slabs, for 1x1x1 the rows
// themselves — and every thread then owns ONE output position (x CO_T output channels in registers): per tap it reads
// its CK input channels from LDS (ds_read_b128) and runs CK x CO_T fused multiply-adds whose weight operand is a
// SCALAR register: the weights of a (channel, tap) row are wave-uniform, so they arrive by s_load from the packed
// weights as they lie ([out channel][tap][in channel], the layout both the forward and the data-gradient launches
// already carry) and cost no vector loads, no LDS traffic and no vector registers.  Long input rows are cut into
// CK-channel chunks (LDS budget: two workgroups per CU).  The epilogue is the fused one of the other conv kernels
// (scale / bias / residual / ReLU) plus the training-mode BN batch statistics of sf_conv_fwd_stats (one record per
// workgroup and channel: [count, K, sum(v - K), sum((v - K)^2)], K = the tile's first row).
#include "common.h"
#include <stdlib.h>

namespace {

#define SF_CONST __attribute__((address_space(4)))

struct SmallArgs {
  const float* in;
  const float* w;      // [Cout][ntaps][cin_pad]
  const float* scale;
  const float* bias;
  const float* res;
  float* out;
  float* stats;
  int in_cs, in_coff, out_cs, out_coff, res_cs, res_coff;
  int T, H, W, HW;
  int Cin, Cout, cin_pad;
  int kT, kH, kW;
  int flip;            // 1: data gradient (taps mirrored)
  int act;
  int P, G, TH;        // positions per tile, thread groups per position (output-channel split), rows per tile (spatial)
  int spatial;         // 1: halo tile of one frame (kH or kW > 1, kT == 1); 0: flat rows (+ temporal slabs)
  int tiles_per_frame; // spatial
  int c4_shift;        // log2(CK / 4)
  int M;
  int dbg;             // microbenchmarks: 1 = no staging loads, 2 = no FMA loop, 4 = no stores
  unsigned hw_mul, hw_sh, t_mul, t_sh, w_mul, w_sh;  // n / HW, n / T, n / W as mulhi + shift (mul == 0: divisor 1)
};

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned mul, unsigned sh) { return mul ? (__umulhi(n, mul) >> sh) : n; }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dppf<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dppf<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dppf<0x141>(v);   // row_half_mirror
  v += dppf<0x140>(v);   // row_mirror
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int CO_T, int CK>
__global__ __launch_bounds__(256) void conv_small_kernel(const SmallArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int PITCH = CK + 4;                 // floats per staged position: 16-byte aligned, banks spread
  const int tid = threadIdx.x;
  const int g = __builtin_amdgcn_readfirstlane(tid / p.P);   // wave-uniform: P is a multiple of 64 or G == 1
  const int pos = tid - g * p.P;
  const int tile = blockIdx.x;
  const int ntaps = p.kT * p.kH * p.kW;

  // ---- tile geometry
  int m0, rows_in_tile, frame = 0, h0 = 0;
  if (p.spatial) {
    frame = tile / p.tiles_per_frame;
    h0 = (tile - frame * p.tiles_per_frame) * p.TH;
    const int th = min(p.TH, p.H - h0);
    m0 = frame * p.HW + h0 * p.W;
    rows_in_tile = th * p.W;
  } else {
    m0 = tile * p.P;
    rows_in_tile = min(p.P, p.M - m0);
  }
  const bool active = g < p.G && pos < rows_in_tile;
  const int m = m0 + pos;
  int pr = 0, pw = pos;                          // spatial: row / column of this thread's position inside the tile
  if (p.spatial) {
    pr = (int)fdiv((unsigned)pos, p.w_mul, p.w_sh);
    pw = pos - pr * p.W;
  }
  const int Wp = p.spatial ? p.W + p.kW - 1 : p.P;            // staged columns per slab row
  const int R = p.spatial ? p.TH + p.kH - 1 : 1;              // staged rows per slab
  const int nslab = p.spatial ? 1 : p.kT;
  const int slab_floats = R * Wp * PITCH;

  float acc[CO_T];
#pragma unroll
  for (int c = 0; c < CO_T; ++c) acc[c] = 0.f;
  const SF_CONST float* const wbase = (const SF_CONST float*)p.w + (long)(g * CO_T) * ntaps * p.cin_pad;

  for (int k0 = 0; k0 < p.Cin; k0 += CK) {
    if (k0) __syncthreads();                     // everyone is done reading the previous chunk
    // ---- stage: [slab][row][col][CK] <- input, zero outside the frame / the tensor / the channel range.  U loads per
    //      thread are issued before the first LDS store (one load -> wait -> store per iteration made the kernel wait a
    //      full memory latency per 16 bytes: 32 -> 8 3x1x1 took 148 us for 128 MB)
    {
      const int per_slab = R * Wp << p.c4_shift;               // float4 items per slab
      const int total = per_slab * nslab;
      constexpr int U = 6;
      for (int base = tid; base < total; base += 256 * U) {
        f32x4 v[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * 256;
          dst[u] = -1;
          v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (idx < total) {
            const int s = idx / per_slab;                      // nslab <= 3: cheap
            const int it = idx - s * per_slab;
            const int c4 = it & ((1 << p.c4_shift) - 1);
            const int cell = it >> p.c4_shift;                 // row * Wp + col
            long src = -1;
            if (p.spatial) {
              const int rr = cell / Wp, cc = cell - rr * Wp;
              const int hh = h0 + rr - (p.kH >> 1), ww = cc - (p.kW >> 1);
              if ((unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W) src = (long)frame * p.HW + hh * p.W + ww;
            } else {
              const int mm = m0 + cell;
              if (mm < p.M) {
                const int o = s - (p.kT >> 1);                 // temporal offset of this slab
                if (o == 0) src = mm;
                else {
                  const unsigned f = fdiv((unsigned)mm, p.hw_mul, p.hw_sh);
                  const int t = (int)(f - fdiv(f, p.t_mul, p.t_sh) * (unsigned)p.T);
                  if ((unsigned)(t + o) < (unsigned)p.T) src = (long)mm + (long)o * p.HW;
                }
              }
            }
            const int ch = k0 + c4 * 4;
            dst[u] = s * slab_floats + cell * PITCH + c4 * 4;
            if (src >= 0 && ch < p.Cin && !(p.dbg & 1)) v[u] = *reinterpret_cast<const f32x4*>(p.in + src * p.in_cs + p.in_coff + ch);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (dst[u] >= 0) *reinterpret_cast<f32x4*>(lds + dst[u]) = v[u];
      }
    }
    __syncthreads();
    // ---- compute: this thread's position x CO_T output channels over the chunk's CK input channels of every tap
    if (active && !(p.dbg & 2)) {
      for (int kt = 0; kt < p.kT; ++kt)
        for (int kh = 0; kh < p.kH; ++kh)
          for (int kw = 0; kw < p.kW; ++kw) {
            const int tap = (kt * p.kH + kh) * p.kW + kw;
            const int st = p.flip ? p.kT - 1 - kt : kt, sh = p.flip ? p.kH - 1 - kh : kh, sw = p.flip ? p.kW - 1 - kw : kw;
            const float* xp = p.spatial ? lds + ((pr + sh) * Wp + pw + sw) * PITCH : lds + st * slab_floats + pos * PITCH;
            float x[CK];
#pragma unroll
            for (int q = 0; q < CK / 4; ++q) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(xp + q * 4);
              x[q * 4] = v[0]; x[q * 4 + 1] = v[1]; x[q * 4 + 2] = v[2]; x[q * 4 + 3] = v[3];
            }
            const SF_CONST float* const wt = wbase + tap * p.cin_pad + k0;
#pragma unroll
            for (int co = 0; co < CO_T; ++co) {
              const SF_CONST float* const wr = wt + (long)co * ntaps * p.cin_pad;   // wave-uniform: scalar loads
#pragma unroll
              for (int c = 0; c < CK; ++c) acc[co] = __builtin_fmaf(x[c], wr[c], acc[co]);
            }
          }
    }
  }

  // ---- epilogue: scale / bias / residual / activation, 16-byte stores, optional BN batch statistics
  const int n0 = g * CO_T;
  const bool relu = p.act == SF_ACT_RELU || p.act == SF_ACT_RELU6;
  const float hi = p.act == SF_ACT_RELU6 ? 6.f : 3.0e38f;
  if (g < p.G) {
#pragma unroll
    for (int c = 0; c < CO_T; ++c) {
      float v = acc[c];
      if (p.scale) v *= p.scale[n0 + c];
      if (p.bias) v += p.bias[n0 + c];
      acc[c] = v;
    }
  }
  if (active && !(p.dbg & 4)) {
    float* const o = p.out + (long)m * p.out_cs + p.out_coff + n0;
    const float* const r = p.res ? p.res + (long)m * p.res_cs + p.res_coff + n0 : nullptr;
#pragma unroll
    for (int q = 0; q < CO_T / 4; ++q) {
      f32x4 v = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
      if (r) v += *reinterpret_cast<const f32x4*>(r + q * 4);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), hi);
      }
      *reinterpret_cast<f32x4*>(o + q * 4) = v;
      acc[q * 4] = v[0]; acc[q * 4 + 1] = v[1]; acc[q * 4 + 2] = v[2]; acc[q * 4 + 3] = v[3];
    }
  }
  if (p.stats) {
    // K = the tile's first row (always a valid row), per output channel: broadcast through LDS
    __syncthreads();                             // the staged input is dead
    float* const kbuf = lds;                     // [G * CO_T]
    float* const red = lds + 128;                // [4 waves][2 * CO_T]
    if (pos == 0 && g < p.G) {
#pragma unroll
      for (int c = 0; c < CO_T; ++c) kbuf[n0 + c] = acc[c];
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < CO_T; ++c) {
      const float k = g < p.G ? kbuf[n0 + c] : 0.f;
      const float dv = active ? acc[c] - k : 0.f;
      const float s1 = wave_sum(dv), s2 = wave_sum(dv * dv);
      if (lane == 0) { red[wave * 2 * CO_T + c] = s1; red[wave * 2 * CO_T + CO_T + c] = s2; }
    }
    __syncthreads();
    if (tid < p.G * CO_T) {                      // one thread per output channel: sum its group's wavefronts in order
      const int gg = tid / CO_T, c = tid - gg * CO_T;
      const int wpg = p.P >> 6 ? p.P >> 6 : 1;   // wavefronts per group (P >= 64 whenever G > 1)
      const int w0 = p.G == 1 ? 0 : gg * wpg, w1 = p.G == 1 ? 4 : w0 + wpg;
      float s1 = 0.f, s2 = 0.f;
      for (int w = w0; w < w1; ++w) { s1 += red[w * 2 * CO_T + c]; s2 += red[w * 2 * CO_T + CO_T + c]; }
      const int n = gg * CO_T + c;
      float* const o = p.stats + ((long)tile * (p.Cout >> 2) + (n >> 2)) * 16 + (n & 3);
      o[0] = (float)rows_in_tile;
      o[4] = kbuf[n];
      o[8] = s1;
      o[12] = s2;
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

int g_small_enable = 1;  // sf_conv_tune(6, e): bit 0 = enable, bit 1 = every shape the instantiations cover (as
int g_small_all = 0;     // SF_CONV_SMALL=2; tests), bits 4.. = ablation mask (microbenchmarks)
int g_small_dbg = 0;

static int small_level() {
  static const int env_on = [] {
    const char* e = getenv("SF_CONV_SMALL");
    return e ? atoi(e) : 1;
  }();
  return g_small_enable ? (g_small_all ? 2 : env_on) : 0;
}
static bool small_enabled() { return small_level() > 0; }

template <int CO_T, int CK>
static int launch_small(const SmallArgs& a, int ntile, size_t lds_bytes, hipStream_t s) {
  static SfLdsAttr lds_attr;
  if (lds_bytes > 48 * 1024 &&
      !sf_ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(conv_small_kernel<CO_T, CK>), 96 * 1024))
    return SF_ELAUNCH;
  hipLaunchKernelGGL((conv_small_kernel<CO_T, CK>), dim3(ntile), dim3(256), lds_bytes, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

}  // namespace

int sf_conv_small_tune(int value) {
  g_small_enable = value & 1;
  g_small_all = (value >> 1) & 1;
  g_small_dbg = value >> 4;
  return SF_OK;
}

long sf_conv_wave_max_parts(long M);  // conv_wave.hip: what sf_conv_stats_ws_floats sized the statistics rows for

// Shape-only decision (pointer alignment is checked at launch).  parts_out: workgroups = statistics records.
int sf_conv_small_takes(const sf_conv_desc* d, int* parts_out) {
  if (parts_out) *parts_out = 0;
  if (!small_enabled()) return 0;
  if (d->sT != 1 || d->sH != 1 || d->sW != 1 || d->dT != 1 || d->dH != 1 || d->dW != 1) return 0;
  if (d->Ti != d->To || d->Hi != d->Ho || d->Wi != d->Wo) return 0;
  if (!(d->kT & 1) || !(d->kH & 1) || !(d->kW & 1) || d->kT > 3 || d->kH > 3 || d->kW > 3) return 0;
  if (d->pT != d->kT / 2 || d->pH != d->kH / 2 || d->pW != d->kW / 2) return 0;
  const bool spatial = d->kH > 1 || d->kW > 1;
  if (spatial && d->kT != 1) return 0;
  if (d->out_cmul != 1 || d->os_T > 1 || d->os_H > 1 || d->os_W > 1) return 0;
  const int cin = d->Cin, cout = d->Cout;
  if (cin != 8 && cin != 16 && cin != 32 && cin != 64 && cin != 128) return 0;
  if (cout != 8 && cout != 16 && cout != 32 && cout != 64 && cout != 128) return 0;
  if ((cin < cout ? cin : cout) > 32) return 0;                 // the matrix-core kernels are fine from 64 x 64 on
  // Measured on MI355X (profiles/r03_conv_small_ab.txt, cfg #3 training step): only the tiniest layers win so far —
  // 8 -> 8 1x3x3 at 32x56x56: 58 -> 39-44 us forward and data gradient.  From 16 channels on the kernel is bound by the
  // latency of its scalar weight loads (a 32 x 3 x 128 table is 49 KB: past the scalar cache, ~700 cycles per wait with
  // one wavefront per SIMD) and by row-per-lane stores of wide outputs: 128 -> 32 3x1x1 25 -> 236 us.  Until those two
  // are fixed (weights through LDS with several positions per thread, LDS-transposed stores) everything else stays
  // on the matrix-core kernels; SF_CONV_SMALL=2 takes every shape the instantiations cover (A/B runs).
  if (small_level() < 2 && !(cin <= 8 && cout <= 8 && (d->kH > 1 || d->kW > 1))) return 0;
  if ((d->in_cs % 4) || (d->in_coff % 4) || (d->out_cs % 4) || (d->out_coff % 4)) return 0;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M < 4096 || M > 0x7fffffffL) return 0;
  const int G = cout > 32 ? cout / 32 : 1;
  if (spatial) {
    if (G != 1) return 0;
    if (d->Wo > 256) return 0;
    const int TH = 256 / d->Wo < d->Ho ? 256 / d->Wo : d->Ho;
    if (TH < 1) return 0;
    const int ck = cin < 32 ? cin : 32;
    if ((long)(TH + d->kH - 1) * (d->Wo + d->kW - 1) * (ck + 4) * 4 > 64 * 1024) return 0;
    if (parts_out) *parts_out = d->N * d->To * ((d->Ho + TH - 1) / TH);
  } else if (parts_out) {
    *parts_out = (int)((M + 256 / G - 1) / (256 / G));
  }
  return 1;
}

// Returns 1 when the shape is not taken (the caller falls through to the matrix-core kernels), else SF_OK / an error.
int sf_conv_small_try(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                      const float* bias, const float* res, float* out, hipStream_t stream, float* stats,
                      int* stat_parts) {
  if (stat_parts) *stat_parts = 0;
  int parts = 0;
  if (!sf_conv_small_takes(d, &parts)) return 1;
  if (!sf_aligned16(in) || !sf_aligned16(out) || !sf_aligned16(w_packed) || (d->cin_pad % 4)) return 1;
  if (res && ((d->res_cs % 4) || (d->res_coff % 4) || !sf_aligned16(res))) return 1;
  SmallArgs a;
  a.in = in; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.in_cs = d->in_cs; a.in_coff = d->in_coff; a.out_cs = d->out_cs; a.out_coff = d->out_coff;
  a.res_cs = d->res_cs; a.res_coff = d->res_coff;
  a.T = d->To; a.H = d->Ho; a.W = d->Wo; a.HW = d->Ho * d->Wo;
  a.Cin = d->Cin; a.Cout = d->Cout; a.cin_pad = d->cin_pad;
  a.kT = d->kT; a.kH = d->kH; a.kW = d->kW;
  a.flip = d->transposed ? 1 : 0;
  a.act = d->act;
  a.M = (int)((long)d->N * d->To * d->Ho * d->Wo);
  a.dbg = g_small_dbg;
  a.G = d->Cout > 32 ? d->Cout / 32 : 1;
  const int co_t = d->Cout > 32 ? 32 : d->Cout;
  a.spatial = (d->kH > 1 || d->kW > 1) ? 1 : 0;
  int ck;
  size_t lds_floats;
  if (a.spatial) {
    a.TH = 256 / d->Wo < d->Ho ? 256 / d->Wo : d->Ho;
    a.P = a.TH * d->Wo;
    a.tiles_per_frame = (d->Ho + a.TH - 1) / a.TH;
    ck = d->Cin < 32 ? d->Cin : 32;
    lds_floats = (size_t)(a.TH + d->kH - 1) * (d->Wo + d->kW - 1) * (ck + 4);
  } else {
    a.TH = 1;
    a.P = 256 / a.G;
    a.tiles_per_frame = 1;
    ck = d->kT == 1 ? (d->Cin < 32 ? d->Cin : 32) : (d->Cin < 16 ? d->Cin : 16);   // three slabs: 16-channel chunks
    lds_floats = (size_t)d->kT * a.P * (ck + 4);
  }
  if (lds_floats < 128 + 4 * 2 * 32) lds_floats = 128 + 4 * 2 * 32;  // the statistics scratch
  a.c4_shift = ck == 8 ? 1 : (ck == 16 ? 2 : 3);
  magic((unsigned)a.HW, &a.hw_mul, &a.hw_sh);
  magic((unsigned)a.T, &a.t_mul, &a.t_sh);
  magic((unsigned)a.W, &a.w_mul, &a.w_sh);
  // one statistics record per workgroup: frames of fewer than 112 positions can make more of them than the workspace
  // (sized for conv_wave's tiles) holds — no statistics then, the caller runs the separate pass
  const bool want = stats && stat_parts && !scale && !res && d->act == SF_ACT_NONE && sf_aligned16(stats) &&
                    parts <= sf_conv_wave_max_parts((long)d->N * d->To * d->Ho * d->Wo);
  a.stats = want ? stats : nullptr;
  if (want) *stat_parts = parts;
  const size_t lb = lds_floats * sizeof(float);
#define SF_SMALL_CASE(CO, CKK) if (co_t == CO && ck == CKK) return launch_small<CO, CKK>(a, parts, lb, stream)
  SF_SMALL_CASE(8, 8);   SF_SMALL_CASE(8, 16);   SF_SMALL_CASE(8, 32);
  SF_SMALL_CASE(16, 8);  SF_SMALL_CASE(16, 16);  SF_SMALL_CASE(16, 32);
  SF_SMALL_CASE(32, 8);  SF_SMALL_CASE(32, 16);  SF_SMALL_CASE(32, 32);
#undef SF_SMALL_CASE
  if (stat_parts) *stat_parts = 0;
  return 1;
}
