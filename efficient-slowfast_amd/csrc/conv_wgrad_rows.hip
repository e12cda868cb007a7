// conv_wgrad_rows.hip — weight gradient of the small-channel stride-1 "same" convolutions (Cin <= 32 or Cout <= 32:
// the Fast pathway's 1x3x3 / 3x1x1 / 1x1x1 layers, the q|k|v and lateral projections) — gfx950.
//
//   dW[co, tap, ci] = sum_m dz[m, co] * x[m + off(tap), ci]       (m over the N*T*H*W positions)
//
// conv_wgrad_small_kernel (conv_wgrad.hip) gives every TAP its own workgroups and stages 64 positions per barrier
// through registers: x and dz are re-read once per tap, a stage is 4 MFMAs per wavefront, and the launch runs at a
// sixth of what its operands cost at HBM speed whatever the split count.  For a stride-1 conv with "same" padding the
// input row of (position m, tap) is the FLAT row m + (kt - pT) H W + (kh - pH) W + (kw - pW) wherever the tap is inside
// the clip, so here a workgroup owns a run of positions and walks it in stages of L (128 / 64):
//   * one stage = L rows of dz + per kt ONE window of L + 2 halo rows of x (halo = pH W + pW), copied HBM -> LDS by
//     direct-to-LDS buffer loads (16 bytes per lane, no registers; rows outside the tensor come back as zeros from the
//     buffer bounds check), double buffered: stage s + 1 is in flight while stage s is multiplied;
//   * ALL taps are accumulated from that image: lane (c, k) of a v_mfma_f32_16x16x4_f32 reads dz[4g + k][co tile + c]
//     once per group of 4 positions and x[4g + k + kh W + kw][ci tile + c] per tap (ds_read_b32, conflict free for
//     rows of 8 / 16 / 32 floats); taps that fall outside the clip for a position (other row / frame / clip in flat
//     order) are masked on the dz operand with a per-position validity word the staging threads leave in LDS;
//   * the four wavefronts split the position groups; their accumulators are summed through LDS tap by tap and the
//     workgroup writes ONE partial [Cout block][taps][Cin block] (the usual [S][Cout][taps][cin_pad] workspace, summed
//     in split order by sf_conv_wgrad_finish: bit-reproducible, no float atomics).
// Channel blocks are 16 or 32 wide on both sides (wider layers — 32 -> 128, 128 -> 32 — take blockIdx.y blocks).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void;

struct RowsArgs {
  sf_conv_desc d;
  const float* x;
  const float* dz;
  float* part;           // [S][Cout][ntaps][cin_pad]
  int dz_cs, dz_coff;
  int M, S, chunk, L;    // positions, splits, positions per split (multiple of L), positions per stage
  int nb_ci;             // blockIdx.y = co block * nb_ci + ci block
  int zw, xw;            // floats per dz / x row in LDS: 8, 16 or 32
  int wrp;               // rows of one x window in LDS (L + 2 halo, padded to whole 1 KiB pieces)
  int halo, HW;
  unsigned x_bytes, z_bytes;
  unsigned w_mul, w_sh, h_mul, h_sh, t_mul, t_sh;
  int stage_floats;      // one stage buffer: dz rows, x windows, validity words
};

__device__ __forceinline__ unsigned mdiv(unsigned n, unsigned mul, unsigned sh) {
  return mul ? (__umulhi(n, mul) >> sh) : n;
}

constexpr unsigned ROWS_OOB = 0x80000000u;  // >= every buffer's size (checked by the launcher): reads as zeros

// ZW / XW: floats per dz / x row in LDS (8, 16, 32; compile-time so that the kw offsets are immediates).  PACK
// (8 -> 8 channels): an MFMA's rows 8..15 and columns 8..15 take a SECOND position set (the stage's upper half), so the
// two diagonal 8 x 8 blocks of the accumulator both hold wanted sums — half the MFMAs; the blocks are added at the end.
template <int NTAP, int NCO, int ZW, int NCI, int XW>
__global__ __launch_bounds__(256) void conv_wgrad_rows_kernel(const RowsArgs p) {
  constexpr int BCO = 16 * NCO, BCI = 16 * NCI;
  constexpr int KT = (NTAP == 3) ? 3 : 1;
  constexpr bool PACK = (ZW == 8 && XW == 8);
  extern __shared__ __attribute__((aligned(16))) float rows_smem[];
  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const int bco = blockIdx.y / p.nb_ci, bci = blockIdx.y - bco * p.nb_ci;
  const int co0 = bco * BCO, ci0 = bci * BCI;
  const int L = p.L;
  constexpr int zw = ZW, xw = XW;
  const int mb = split * p.chunk;
  const int me = (mb + p.chunk < p.M) ? mb + p.chunk : p.M;
  const int nst = (me > mb) ? (me - mb + L - 1) / L : 0;
  const int nzb = (L * zw) >> 8;          // 1 KiB pieces of the dz rows
  const int nxb = (p.wrp * xw) >> 8;      // ... of one x window
  const int nblk = nzb + KT * nxb;
  constexpr int zq_sh = (ZW == 8) ? 1 : (ZW == 16 ? 2 : 3);  // log2(chunks per row)
  constexpr int xq_sh = (XW == 8) ? 1 : (XW == 16 ? 2 : 3);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.z_bytes, 0x00020000);
  const int rows_total = p.M;  // "same" geometry: input positions = output positions

  auto issue = [&](int st, int buf) {
    const int m0 = mb + st * L;
    float* const sb = rows_smem + buf * p.stage_floats;
    for (int b = wave; b < nblk; b += 4) {
      unsigned vo = ROWS_OOB;
      if (b < nzb) {
        const int c = (b << 6) + lane;
        const int row = c >> zq_sh, col = (c - (row << zq_sh)) << 2;
        const int m = m0 + row;
        if (m < me && co0 + col < d.Cout) vo = ((unsigned)m * (unsigned)p.dz_cs + (unsigned)(p.dz_coff + co0 + col)) << 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rs, (lds_void*)(sb + (b << 8)), 16, vo, 0, 0, 0);
      } else {
        const int bb = b - nzb;
        const int kt = (KT == 1) ? 0 : bb / nxb;
        const int c = ((bb - kt * nxb) << 6) + lane;
        const int row = c >> xq_sh, col = (c - (row << xq_sh)) << 2;
        const int q = m0 + (kt - d.pT) * p.HW - p.halo + row;
        if (q >= 0 && q < rows_total && ci0 + col < d.Cin)
          vo = ((unsigned)q * (unsigned)d.in_cs + (unsigned)(d.in_coff + ci0 + col)) << 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(sb + (b << 8)), 16, vo, 0, 0, 0);
      }
    }
    if (NTAP > 1 && tid < L) {  // which taps of position m0 + tid lie inside its clip
      const unsigned m = (unsigned)(m0 + tid);
      const unsigned q1 = mdiv(m, p.w_mul, p.w_sh);
      const int w = (int)(m - q1 * (unsigned)d.Wo);
      const unsigned q2 = mdiv(q1, p.h_mul, p.h_sh);
      const int h = (int)(q1 - q2 * (unsigned)d.Ho);
      const unsigned q3 = mdiv(q2, p.t_mul, p.t_sh);
      const int t = (int)(q2 - q3 * (unsigned)d.To);
      unsigned bits = 0;
      if (NTAP == 3) {
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
          if ((unsigned)(t + kt - 1) < (unsigned)d.To) bits |= 1u << kt;
      } else {  // separable: bits 0..2 = row h + kh - 1 inside, bits 4..6 = column w + kw - 1 inside
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          if ((unsigned)(h + kk - 1) < (unsigned)d.Ho) bits |= 1u << kk;
          if ((unsigned)(w + kk - 1) < (unsigned)d.Wo) bits |= 16u << kk;
        }
      }
      reinterpret_cast<unsigned*>(sb + L * zw + KT * p.wrp * xw)[tid] = bits;
    }
  };

  f32x4 acc[NTAP][NCO][NCI];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
      for (int j = 0; j < NCI; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int m16 = lane & 15, k4 = lane >> 4;
  const int win = p.wrp * xw;  // floats of one x window
  if (nst > 0) issue(0, 0);
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's pieces of stage st have landed ...
    __syncthreads();                                   // ... and everybody's; buffer buf ^ 1 is free again
    if (st + 1 < nst) issue(st + 1, buf ^ 1);
    const float* const zs = rows_smem + buf * p.stage_floats;
    const float* const xs = zs + L * zw;
    const unsigned* const vm = reinterpret_cast<const unsigned*>(xs + KT * win);
    const int ngroups = PACK ? (L >> 3) : (L >> 2);
    const int half = (PACK && m16 >= 8) ? (L >> 1) : 0;  // PACK: rows / columns 8..15 work on the stage's upper half
    const int mc = PACK ? (m16 & 7) : m16;
    for (int g = wave; g < ngroups; g += 4) {
      const int pp = 4 * g + k4 + half;
      // every operand of the group first (one wait), then the MFMAs
      float a[NCO];
#pragma unroll
      for (int i = 0; i < NCO; ++i) a[i] = zs[pp * ZW + 16 * i + mc];
      const unsigned vb = (NTAP > 1) ? vm[pp] : 1u;
      const float* const xr = xs + pp * XW + mc;
      float bv[NTAP][NCI];
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int off = (NTAP == 3) ? t * win : ((NTAP == 9) ? (t / 3) * d.Wi * XW + (t % 3) * XW : 0);
#pragma unroll
        for (int j = 0; j < NCI; ++j) bv[t][j] = xr[off + 16 * j];
      }
      // validity: 3x1x1 one bit per kt; 1x3x3 separable — bits 0..2 = kh inside, bits 4..6 = kw inside
      float am[NTAP][NCO];
      if constexpr (NTAP == 1) {
#pragma unroll
        for (int i = 0; i < NCO; ++i) am[0][i] = a[i];
      } else if constexpr (NTAP == 3) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < NCO; ++i) am[t][i] = ((vb >> t) & 1u) ? a[i] : 0.f;
      } else {
#pragma unroll
        for (int i = 0; i < NCO; ++i) {
          const float w0 = (vb & 16u) ? a[i] : 0.f, w2 = (vb & 64u) ? a[i] : 0.f;  // kw = 1 is always inside
          const bool h0 = (vb & 1u) != 0, h2 = (vb & 4u) != 0;
          am[0][i] = h0 ? w0 : 0.f; am[1][i] = h0 ? a[i] : 0.f; am[2][i] = h0 ? w2 : 0.f;
          am[3][i] = w0;            am[4][i] = a[i];            am[5][i] = w2;
          am[6][i] = h2 ? w0 : 0.f; am[7][i] = h2 ? a[i] : 0.f; am[8][i] = h2 ? w2 : 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < NCO; ++i)
#pragma unroll
          for (int j = 0; j < NCI; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[t][i], bv[t][j], acc[t][i][j], 0, 0, 0);
    }
  }

  // cross-wavefront sum, tap by tap (the stage buffers are free: nothing is in flight after the last stage)
  float* const red = rows_smem;  // [4][BCO][BCI]
  float* const base = p.part + (long)split * d.Cout * NTAP * d.cin_pad;
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
      for (int j = 0; j < NCI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * BCO + 16 * i + 4 * k4 + r) * BCI + 16 * j + m16] = acc[t][i][j][r];
    __syncthreads();
    for (int e = tid; e < BCO * BCI; e += 256) {
      const int co = co0 + e / BCI, ci = ci0 + e % BCI;
      if (co < d.Cout && ci < d.cin_pad) {
        float v = (red[e] + red[BCO * BCI + e]) + (red[2 * BCO * BCI + e] + red[3 * BCO * BCI + e]);
        if constexpr (PACK) {
          if (e / BCI < 8 && e % BCI < 8) {
            const int e2 = e + 8 * BCI + 8;
            v += (red[e2] + red[BCO * BCI + e2]) + (red[2 * BCO * BCI + e2] + red[3 * BCO * BCI + e2]);
          }
        }
        base[((long)co * NTAP + t) * d.cin_pad + ci] = (ci < d.Cin) ? v : 0.f;
      }
    }
  }
}

// ---- 3x1x1 layers as a RING OVER t ---------------------------------------------------------------------------------------
// The rows kernel above stages, for a 3x1x1 layer, three windows of x per run of positions (rows m - HW, m, m + HW): every
// x row travels HBM/L2 -> LDS three times, and the launch is bound by exactly that traffic (3x1x1 32 -> 8 at 802 816
// positions: 334 MB in 77 us, profiles/r05_conv_per_shape.txt rows at 0.12 .. 0.21 of their operands' time).  Here a
// workgroup owns a COLUMN — one clip, one block of L positions of the frame — and walks a segment of t: per step it loads
// frame t + 1 + PD of x once (and dz of frame t + PD), keeps the frames t - 1, t, t + 1 in a ring of 3 + PD LDS slots, and
// multiplies dz[t] with all three — every x row is read once (+ one halo frame at either end of a segment).  Frames
// outside the clip and rows past the frame are out-of-range offsets: they land as zeros, so the loop needs NO validity
// masks.  Loads are LDS-DMA pieces, PD steps ahead, behind a counted s_waitcnt vmcnt (every load of the loop is an
// LDS-DMA and the loop stores nothing: the count is exact), raw s_barrier.
struct TringArgs {
  sf_conv_desc d;
  const float* x;
  const float* dz;
  float* part;           // [S][Cout][3][cin_pad], S = workgroups along x
  int dz_cs, dz_coff;
  int L, nhb, nseg, tseg, PD;  // positions per block, blocks per frame, t segments, frames per segment, prefetch distance
  int nb_ci, HW;
  unsigned x_bytes, z_bytes;
};

__device__ __forceinline__ void tring_vmwait(int n) {
#define TR_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    TR_VM(0) TR_VM(1) TR_VM(2) TR_VM(3) TR_VM(4) TR_VM(5) TR_VM(6) TR_VM(7) TR_VM(8) TR_VM(9) TR_VM(10) TR_VM(11)
    TR_VM(12) TR_VM(13) TR_VM(14) TR_VM(15) TR_VM(16) TR_VM(17) TR_VM(18) TR_VM(19) TR_VM(20)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef TR_VM
}

template <int NCO, int ZW, int NCI, int XW>
__global__ __launch_bounds__(256) void conv_wgrad_tring_kernel(const TringArgs p) {
  constexpr int BCO = 16 * NCO, BCI = 16 * NCI;
  constexpr int zq_sh = (ZW == 8) ? 1 : (ZW == 16 ? 2 : 3);  // log2(16-byte chunks per row)
  constexpr int xq_sh = (XW == 8) ? 1 : (XW == 16 ? 2 : 3);
  extern __shared__ __attribute__((aligned(16))) float rows_smem[];
  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int unit = blockIdx.x;
  const int seg = unit % p.nseg, col = unit / p.nseg;
  const int hb = col % p.nhb, n = col / p.nhb;
  const int bco = blockIdx.y / p.nb_ci, bci = blockIdx.y - bco * p.nb_ci;
  const int co0 = bco * BCO, ci0 = bci * BCI;
  const int L = p.L, PD = p.PD, T = d.To;
  const int t0 = seg * p.tseg, t1 = (t0 + p.tseg < T) ? t0 + p.tseg : T;
  const int NXS = 3 + PD, NZS = 1 + PD;             // ring slots
  const int xslot = L * XW, zslot = L * ZW;         // floats
  float* const xring = rows_smem;
  float* const zring = rows_smem + NXS * xslot;
  const int nxb = xslot >> 8, nzb = zslot >> 8;     // 1 KiB pieces
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.z_bytes, 0x00020000);
  const int r_base = hb * L;                        // first row of the block within its frame
  // pieces this wavefront issues per step (x frame + dz frame): wave, wave + 4, ...
  const int nper = ((nxb > wave) ? (nxb - wave + 3) / 4 : 0) + ((nzb > wave) ? (nzb - wave + 3) / 4 : 0);

  // frame f of x into ring slot (f + 1) % NXS (f >= -1); frame t of dz into slot t % NZS; f / t outside their ranges: zeros
  auto issue = [&](int fx, int xs, int tz, int zs) {
    const bool fok = (unsigned)fx < (unsigned)T && fx <= t1;  // (frames past the segment's halo are never read)
    const long frow = ((long)n * T + fx) * p.HW + r_base;
    for (int b = wave; b < nxb; b += 4) {
      const int c = (b << 6) + lane;
      const int row = c >> xq_sh, colx = (c - (row << xq_sh)) << 2;
      unsigned vo = ROWS_OOB;
      if (fok && r_base + row < p.HW && ci0 + colx < d.Cin)
        vo = ((unsigned)(frow + row) * (unsigned)d.in_cs + (unsigned)(d.in_coff + ci0 + colx)) << 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(xring + xs * xslot + (b << 8)), 16, vo, 0, 0, 0);
    }
    const bool tok = tz < t1;
    const long zrow = ((long)n * T + tz) * p.HW + r_base;
    for (int b = wave; b < nzb; b += 4) {
      const int c = (b << 6) + lane;
      const int row = c >> zq_sh, colz = (c - (row << zq_sh)) << 2;
      unsigned vo = ROWS_OOB;
      if (tok && r_base + row < p.HW && co0 + colz < d.Cout)
        vo = ((unsigned)(zrow + row) * (unsigned)p.dz_cs + (unsigned)(p.dz_coff + co0 + colz)) << 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rs, (lds_void*)(zring + zs * zslot + (b << 8)), 16, vo, 0, 0, 0);
    }
  };

  f32x4 acc[3][NCO][NCI];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
      for (int j = 0; j < NCI; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int m16 = lane & 15, k4 = lane >> 4;

  // prologue: x frames t0 - 1, t0 (alone, without dz), then per prefetch step one x frame + one dz frame, so that
  // every "step" of loads has the same number of pieces except the first two (drained by the first wait: see below)
  int xs_next = 0, zs_next = 0;                     // slots of the next frames to issue
  int fx_next = t0 - 1, tz_next = t0;
  {
    // x(t0 - 1), x(t0): two x-only issues
    for (int k = 0; k < 2; ++k) {
      const bool fok = (unsigned)fx_next < (unsigned)T;
      const long frow = ((long)n * T + fx_next) * p.HW + r_base;
      for (int b = wave; b < nxb; b += 4) {
        const int c = (b << 6) + lane;
        const int row = c >> xq_sh, colx = (c - (row << xq_sh)) << 2;
        unsigned vo = ROWS_OOB;
        if (fok && r_base + row < p.HW && ci0 + colx < d.Cin)
          vo = ((unsigned)(frow + row) * (unsigned)d.in_cs + (unsigned)(d.in_coff + ci0 + colx)) << 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(xring + xs_next * xslot + (b << 8)), 16, vo, 0, 0, 0);
      }
      ++fx_next;
      if (++xs_next == NXS) xs_next = 0;
    }
    // x(t0 + 1 .. t0 + PD) with dz(t0 .. t0 + PD - 1): the loads of steps t0 .. t0 + PD - 1
    for (int k = 0; k < PD; ++k) {
      issue(fx_next, xs_next, tz_next, zs_next);
      ++fx_next; ++tz_next;
      if (++xs_next == NXS) xs_next = 0;
      if (++zs_next == NZS) zs_next = 0;
    }
  }
  int xs_lo = 0;   // ring slot of frame t - 1
  int zs_cur = 0;  // ring slot of dz frame t
  for (int t = t0; t < t1; ++t) {
    // frames t - 1, t, t + 1 of x and frame t of dz have landed when only the loads of the PD - 1 younger steps remain
    tring_vmwait((PD - 1) * nper);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue(fx_next, xs_next, tz_next, zs_next);       // x(t + 1 + PD), dz(t + PD): the loads of step t + PD
    ++fx_next; ++tz_next;
    if (++xs_next == NXS) xs_next = 0;
    if (++zs_next == NZS) zs_next = 0;
    __builtin_amdgcn_sched_barrier(0);
    const float* const zs = zring + zs_cur * zslot;
    const float* xs[3];
    {
      int s0 = xs_lo;
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        xs[kt] = xring + s0 * xslot;
        if (++s0 == NXS) s0 = 0;
      }
    }
    const int ngroups = L >> 2;
    for (int g = wave; g < ngroups; g += 4) {
      const int pp = 4 * g + k4;
      float a[NCO];
#pragma unroll
      for (int i = 0; i < NCO; ++i) a[i] = zs[pp * ZW + 16 * i + m16];
      float bv[3][NCI];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int j = 0; j < NCI; ++j) bv[kt][j] = xs[kt][pp * XW + 16 * j + m16];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int i = 0; i < NCO; ++i)
#pragma unroll
          for (int j = 0; j < NCI; ++j)
            acc[kt][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[kt][j], acc[kt][i][j], 0, 0, 0);
    }
    if (++xs_lo == NXS) xs_lo = 0;
    if (++zs_cur == NZS) zs_cur = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's (out-of-range) loads still write zeros into the rings
  // cross-wavefront sum, tap by tap (the rings are free now)
  float* const red = rows_smem;  // [4][BCO][BCI]
  float* const base = p.part + (long)unit * d.Cout * 3 * d.cin_pad;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
      for (int j = 0; j < NCI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * BCO + 16 * i + 4 * k4 + r) * BCI + 16 * j + m16] = acc[t][i][j][r];
    __syncthreads();
    for (int e = tid; e < BCO * BCI; e += 256) {
      const int co = co0 + e / BCI, ci = ci0 + e % BCI;
      if (co < d.Cout && ci < d.cin_pad) {
        const float v = (red[e] + red[BCO * BCI + e]) + (red[2 * BCO * BCI + e] + red[3 * BCO * BCI + e]);
        base[((long)co * 3 + t) * d.cin_pad + ci] = (ci < d.Cin) ? v : 0.f;
      }
    }
  }
}

void magic(unsigned dv, unsigned* mul, unsigned* sh) {
  if (dv <= 1) { *mul = 0; *sh = 0; return; }
  unsigned l = 0;
  while ((1u << l) < dv) ++l;
  const unsigned long long num = 1ull << (31 + l);
  *mul = (unsigned)((num + dv - 1) / dv);
  *sh = l - 1;
}

int g_rows_enable = 1;  // sf_conv_tune(10, e)

bool rows_enabled() {
  static const int env_on = [] {
    const char* e = getenv("SF_WGRAD_ROWS");
    return e ? atoi(e) : 1;
  }();
  return env_on && g_rows_enable;
}

int pow2_width(int c) { return c <= 8 ? 8 : (c <= 16 ? 16 : 32); }

// Geometry + schedule for a problem this kernel covers; false = leave it to the other weight-gradient kernels.
bool rows_plan(const sf_conv_desc* d, RowsArgs* a) {
  if (!rows_enabled() || d->transposed) return false;
  if (d->sT != 1 || d->sH != 1 || d->sW != 1 || d->dT != 1 || d->dH != 1 || d->dW != 1) return false;
  if (d->To != d->Ti || d->Ho != d->Hi || d->Wo != d->Wi) return false;
  int ntap;
  if (d->kT == 1 && d->kH == 1 && d->kW == 1 && d->pT == 0 && d->pH == 0 && d->pW == 0) ntap = 1;
  else if (d->kT == 3 && d->kH == 1 && d->kW == 1 && d->pT == 1 && d->pH == 0 && d->pW == 0) ntap = 3;
  else if (d->kT == 1 && d->kH == 3 && d->kW == 3 && d->pT == 0 && d->pH == 1 && d->pW == 1) ntap = 9;
  else return false;
  if (d->Cin > 32 && d->Cout > 32) return false;  // both wide: conv_wgrad_wave / conv_bx
  if (d->Cin < 8 || d->Cout < 8) return false;
  if (d->Cin % 4 || d->in_cs % 4 || d->in_coff % 4 || d->Cout % 4) return false;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M <= 0 || M > 0x3fffffffL) return false;
  const long xb = ((M - 1) * d->in_cs + d->in_coff + d->Cin) * 4L;
  if (xb >= 0x7fffffffL) return false;
  a->d = *d;
  a->M = (int)M;
  a->x_bytes = (unsigned)xb;
  const int bco = d->Cout <= 16 ? 16 : 32, bci = d->cin_pad <= 16 ? 16 : 32;
  a->zw = pow2_width(d->Cout < bco ? d->Cout : bco);
  a->xw = pow2_width(d->Cin < bci ? d->Cin : bci);
  a->nb_ci = sf_cdiv(d->cin_pad, bci);
  a->HW = d->Hi * d->Wi;
  a->halo = d->pH * d->Wi + d->pW;
  const int KT = ntap == 3 ? 3 : 1;
  int L = 128;
  for (;; L = 64) {
    const int gran = 256 / a->xw;  // window rows per 1 KiB piece
    a->wrp = (L + 2 * a->halo + gran - 1) / gran * gran;
    a->stage_floats = L * a->zw + KT * a->wrp * a->xw + L;
    if (a->stage_floats * 8 <= 72 * 1024 || L == 64) break;
  }
  if (a->stage_floats * 8 > 150 * 1024) return false;
  if (a->stage_floats * 2 < 4 * bco * bci) return false;  // the reduction image lives in the stage buffers
  a->L = L;
  magic((unsigned)d->Wo, &a->w_mul, &a->w_sh);
  magic((unsigned)d->Ho, &a->h_mul, &a->h_sh);
  magic((unsigned)d->To, &a->t_mul, &a->t_sh);
  // splits: ~640 workgroups over the channel blocks, at least two stages each
  const int blocks = sf_cdiv(d->Cout, bco) * a->nb_ci;
  const long stages = (M + L - 1) / L;
  long S = 640 / blocks;
  if (S < 1) S = 1;
  if (S > (stages + 1) / 2) S = (stages + 1) / 2;
  if (S < 1) S = 1;
  if (S > 1024) S = 1024;
  long chunk = ((M + S - 1) / S + L - 1) / L * L;
  S = (M + chunk - 1) / chunk;
  a->S = (int)S;
  a->chunk = (int)chunk;
  return true;
}

int g_tring_enable = -1;  // sf_conv_tune(23, e): -1 = SF_WGRAD_TRING (default 1), 0 off, 1 on

bool tring_enabled() {
  static const int env_on = [] {
    const char* e = getenv("SF_WGRAD_TRING");
    return e ? atoi(e) : 1;
  }();
  return g_tring_enable < 0 ? env_on != 0 : g_tring_enable != 0;
}

// Schedule of the ring-over-t form for a 3x1x1 problem rows_plan accepted (a holds its widths); false = not taken.
bool tring_plan(const sf_conv_desc* d, const RowsArgs& a, TringArgs* t, size_t* lds) {
  if (!tring_enabled()) return false;
  if (!(d->kT == 3 && d->kH == 1 && d->kW == 1)) return false;
  const int HW = d->Hi * d->Wi, T = d->To, N = d->N;
  if (T < 2 || HW < 32) return false;
  const int bco = d->Cout <= 16 ? 16 : 32, bci = d->cin_pad <= 16 ? 16 : 32;
  const int blocks = sf_cdiv(d->Cout, bco) * a.nb_ci;
  // positions per block: 128 for large frames, 64 otherwise (fewer padded rows in the last block of a frame)
  static const int l_env = [] { const char* e = getenv("SF_WGRAD_TRING_L"); return e ? atoi(e) : 0; }();
  // positions per block: 64 (measured, profiles/r06_wgrad_tring_ab.txt: 128-position blocks 50.5 / 33.6 / 43.5 us where
  // 64-position ones take 45.0 / 28.8 / 31.0 — more workgroups per CU, fewer padded rows in a frame's last block)
  int L = 64;
  if (l_env == 64 || l_env == 128) L = l_env;
  if (L * a.xw * 4 < 1024 || L * a.zw * 4 < 1024) L = 128;    // whole 1 KiB pieces (8-float rows)
  if ((L * a.xw) % 256 || (L * a.zw) % 256) return false;
  const int nhb = sf_cdiv(HW, L);
  // segments of t: ~768 workgroups over the channel blocks, at least 2 frames each, at most 1024 partials
  int nseg = 1;
  while (nseg * 2 <= T / 2 && (long)N * nhb * nseg * blocks < 600 && (long)N * nhb * nseg * 2 <= 1024) nseg *= 2;
  const int tseg = sf_cdiv(T, nseg);
  nseg = sf_cdiv(T, tseg);
  const long S = (long)N * nhb * nseg;
  if (S > 1024 || S < 1) return false;
  t->d = *d;
  t->L = L; t->nhb = nhb; t->nseg = nseg; t->tseg = tseg;
  t->nb_ci = a.nb_ci; t->HW = HW;
  t->x_bytes = a.x_bytes;
  static const int pd_env = [] { const char* e = getenv("SF_WGRAD_TRING_PD"); return e ? atoi(e) : 1; }();  // (1: deeper rings measured no faster)
  int PD = pd_env < 1 ? 1 : (pd_env > 4 ? 4 : pd_env);
  size_t bytes;
  for (;; --PD) {
    bytes = ((size_t)(3 + PD) * L * a.xw + (size_t)(1 + PD) * L * a.zw) * sizeof(float);
    static const int cap_kb = [] { const char* e = getenv("SF_WGRAD_TRING_KB"); return e ? atoi(e) : 76; }();
    if (bytes <= (size_t)cap_kb * 1024 || PD == 1) break;
  }
  if (bytes > 150 * 1024) return false;
  const size_t red = (size_t)4 * bco * bci * sizeof(float);
  t->PD = PD;
  *lds = bytes > red ? bytes : red;
  return true;
}

template <int NCO, int ZW, int NCI, int XW>
int launch_tring(const TringArgs& t, size_t lds, hipStream_t s) {
  static SfLdsAttr attr;
  if (!sf_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_wgrad_tring_kernel<NCO, ZW, NCI, XW>), 152 * 1024))
    return SF_ELAUNCH;
  const int nb_co = sf_cdiv(t.d.Cout, 16 * NCO);
  hipLaunchKernelGGL((conv_wgrad_tring_kernel<NCO, ZW, NCI, XW>), dim3(t.d.N * t.nhb * t.nseg, nb_co * t.nb_ci),
                     dim3(256), lds, s, t);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int NCO, int ZW>
int launch_tring_x(const TringArgs& t, int xw, size_t lds, hipStream_t s) {
  if (xw == 8) return launch_tring<NCO, ZW, 1, 8>(t, lds, s);
  if (xw == 16) return launch_tring<NCO, ZW, 1, 16>(t, lds, s);
  return launch_tring<NCO, ZW, 2, 32>(t, lds, s);
}

int launch_tring_tiles(const TringArgs& t, int zw, int xw, size_t lds, hipStream_t s) {
  if (zw == 8) return launch_tring_x<1, 8>(t, xw, lds, s);
  if (zw == 16) return launch_tring_x<1, 16>(t, xw, lds, s);
  return launch_tring_x<2, 32>(t, xw, lds, s);
}

template <int NTAP, int NCO, int ZW, int NCI, int XW>
int launch_rows(const RowsArgs& a, hipStream_t s) {
  const size_t lds = (size_t)a.stage_floats * 8;
  static SfLdsAttr attr;
  if (!sf_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_wgrad_rows_kernel<NTAP, NCO, ZW, NCI, XW>),
                         152 * 1024))
    return SF_ELAUNCH;
  const int nb_co = sf_cdiv(a.d.Cout, 16 * NCO);
  hipLaunchKernelGGL((conv_wgrad_rows_kernel<NTAP, NCO, ZW, NCI, XW>), dim3(a.S, nb_co * a.nb_ci), dim3(256), lds, s,
                     a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int NTAP, int NCO, int ZW>
int launch_rows_x(const RowsArgs& a, hipStream_t s) {
  if (a.xw == 8) return launch_rows<NTAP, NCO, ZW, 1, 8>(a, s);
  if (a.xw == 16) return launch_rows<NTAP, NCO, ZW, 1, 16>(a, s);
  return launch_rows<NTAP, NCO, ZW, 2, 32>(a, s);
}

template <int NTAP>
int launch_rows_tiles(const RowsArgs& a, hipStream_t s) {
  if (a.zw == 8) return launch_rows_x<NTAP, 1, 8>(a, s);
  if (a.zw == 16) return launch_rows_x<NTAP, 1, 16>(a, s);
  return launch_rows_x<NTAP, 2, 32>(a, s);
}

}  // namespace

int sf_wgrad_rows_tune(int value) {
  g_rows_enable = value ? 1 : 0;
  return SF_OK;
}

// Split count of the rows kernel for this problem, 0 = not covered.
int sf_wgrad_rows_splits(const sf_conv_desc* d) {
  RowsArgs a;
  if (!rows_plan(d, &a)) return 0;
  TringArgs t;
  size_t lds;
  if (tring_plan(d, a, &t, &lds)) return d->N * t.nhb * t.nseg;
  return a.S;
}

int sf_wgrad_tring_tune(int value) {
  g_tring_enable = value;
  return SF_OK;
}

// 1 = not covered (the caller goes on to its other kernels), else SF_OK / an error.
int sf_wgrad_rows_try(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, float* partial,
                      hipStream_t stream) {
  RowsArgs a;
  if (dz_cs % 4 || dz_coff % 4 || !sf_aligned16(x) || !sf_aligned16(dz)) return 1;
  if (!rows_plan(d, &a)) return 1;
  const long zb = ((long)(a.M - 1) * dz_cs + dz_coff + d->Cout) * 4L;
  if (zb >= 0x7fffffffL) return 1;
  a.z_bytes = (unsigned)zb;
  a.x = x; a.dz = dz; a.part = partial; a.dz_cs = dz_cs; a.dz_coff = dz_coff;
  {
    TringArgs t;
    size_t lds;
    if (tring_plan(d, a, &t, &lds)) {  // 3x1x1: every x frame once, through a ring over t
      t.x = x; t.dz = dz; t.part = partial; t.dz_cs = dz_cs; t.dz_coff = dz_coff; t.z_bytes = a.z_bytes;
      return launch_tring_tiles(t, a.zw, a.xw, lds, stream);
    }
  }
  const int ntap = d->kT * d->kH * d->kW;
  if (ntap == 1) return launch_rows_tiles<1>(a, stream);
  if (ntap == 3) return launch_rows_tiles<3>(a, stream);
  return launch_rows_tiles<9>(a, stream);
}
