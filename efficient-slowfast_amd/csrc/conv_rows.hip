// conv_rows.hip — forward and data gradient of the small-channel stride-1 "same" convolutions (Cin <= 32 or Cout <= 32,
// both <= 256 / 128): the Fast pathway's 1x1x1 / 3x1x1 / 1x3x3 layers, the lateral and q|k|v projections — gfx950.
//
//   out[m, co] = act(scale[co] * sum_tap sum_ci x[m + off(tap), ci] * W[co, tap, ci] + bias[co] + res[m, co])
//
// Replaces for these shapes nn.Conv3d of resnet_helper.py:182-223 at dim_inner 8 .. 64 (temp_kernel_size 3 on the Fast
// pathway: custom_video_model_builder.py:155-162), wdf_attention_helper.py:21-29 (query / key / value convs) and their
// autograd data gradients (desc.transposed: the same sum with mirrored offsets over the transposed pack).
//
// conv_wave.hip's per-wavefront implicit GEMM feeds every MFMA operand straight from L2 with 16-byte loads whose 64
// lanes touch 16 different rows: with 8 .. 64-channel rows (32 .. 256 bytes) a 1 KiB wave-load is 16 partial cache
// lines, every tap re-fetches its rows, and a 16-wide output tile gets ONE MFMA per fragment load — these layers ran at
// 0.12 .. 0.45 of their operands' HBM time (profiles/r05_conv_per_shape.txt: 3.1 ms of forward + data gradient per step
// against 0.84 ms).  Here, as in conv_wgrad_rows.hip, a workgroup owns a RUN of positions and walks it in stages of 64:
//   * the stage's input rows — per kt ONE window of 64 (+ 2 halo, halo = W + 1 for 1x3x3) rows of <= 64 channels — arrive
//     in LDS by direct-to-LDS buffer loads of whole contiguous rows (1 KiB per wave-instruction, no registers; rows
//     outside the tensor read as zeros), double buffered; layers with 128 input channels take two channel blocks per
//     position stage.  A row's 16-byte chunks are XOR-swizzled on the SOURCE address so that the 16 rows a quarter
//     wavefront reads with ds_read_b128 land on 16 different bank groups;
//   * the packed weights of the workgroup's <= 64 output channels sit in LDS for the whole launch (row pitch = 4 mod 64
//     floats: conflict-free fragment reads);
//   * operands SWAPPED (v_mfma_f32_16x16x4_f32, A = W[16 co][k], B = x^T[k][16 positions]): a lane's four accumulator
//     registers are four consecutive output channels of one position = one 16-byte NDHWC store; the K index of the
//     four MFMAs behind one pair of fragment reads is permuted (lane group g takes channels 16 q + 4 g + s at step s),
//     so one ds_read_b128 per operand feeds four MFMAs;
//   * taps that leave the clip in flat order (other row / frame / clip) are masked on the x operand by per-position
//     validity bits the staging threads leave in LDS;
//   * epilogue in registers: scale, bias, residual (which may BE the output: accumulating data gradients), activation,
//     and — for training-mode forward convs — the BN batch statistics of what was stored, one [count, K, S1, S2] record
//     per wavefront and channel quad in conv_wave.hip's format (K = the wavefront's first output row).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned CR_OOB = 0x80000000u;
constexpr int CR_L = 64;  // positions per stage: one 16-position MFMA column tile per wavefront

struct CRowsArgs {
  sf_conv_desc d;
  const float* x;
  const float* w;        // [Cout][ntaps][cin_pad]
  const float* scale;
  const float* bias;
  const float* res;
  float* out;
  float* stats;          // [4 * S][Cout / 4][4][4] or NULL
  int M, S, chunk;       // positions, workgroups along the positions, positions per workgroup (multiple of 64)
  int KB;                // channel blocks of XW floats per position stage
  int NB;                // stage buffers of the ring (NB - 1 steps in flight)
  int nhb, nseg, tseg;   // ring over t (3x1x1): 64-position blocks per frame, t segments, frames per segment
  int dbg;               // SF_CONV_ROWS_DBG timing ablations (results invalid): 1 no loads, 2 no MFMA loop, 4 no stores
  int rows_needed;       // rows of a window that are read (64 + 2 halo); the rest of wrp is padding
  int HW, halo, wrp;     // frame size, halo rows of a 1x3x3 window, rows of one x window in LDS
  int wstride;           // floats per weight row in LDS
  int sg;                // +1 forward, -1 data gradient (mirrored taps)
  unsigned x_bytes, res_bytes;
  unsigned w_mul, w_sh, h_mul, h_sh, t_mul, t_sh;
};

__device__ __forceinline__ unsigned cr_mdiv(unsigned n, unsigned mul, unsigned sh) {
  return mul ? (__umulhi(n, mul) >> sh) : n;
}

// Swizzle of a row's 16-byte chunks: 16 consecutive rows read at the same chunk index must land on 16 different
// 16-byte bank groups (the LDS serves 256 bytes per cycle).  XW floats per row = XW / 4 chunks.
template <int XW>
__device__ __forceinline__ int cr_swz(int row) {
  if (XW == 8) return (row >> 3) & 1;
  if (XW == 16) return (row >> 2) & 3;
  if (XW == 32) return (row >> 1) & 7;
  return row & 15;  // 64 floats: one row = one sweep of the banks
}

// ds_read_b128 as an instruction the compiler does not look into: its waitcnt pass drains vmcnt in front of every LDS
// access that may read what an outstanding LDS-DMA writes, which would serialise the next step's loads with this step's
// MFMAs.  The caller owns lgkmcnt: the data is there after cr_wait<N>() with N = LDS reads issued behind this one.
__device__ __forceinline__ f32x4 cr_lds128(unsigned lds_byte_addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(lds_byte_addr));
  return r;
}

// One (tap, 16-channel chunk) of operands: the x fragment and the NCT weight fragments behind it.
template <int NCT> struct CrFrag { f32x4 b; f32x4 a[NCT]; };

// Wait until at most N LDS reads are outstanding; every register of `f` is an operand, so no MFMA on them can be
// scheduled in front of the wait.
template <int N, int NCT>
__device__ __forceinline__ void cr_wait(CrFrag<NCT>& f) {
  if constexpr (NCT == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.b), "+v"(f.a[0]) : "n"(N));
  else if constexpr (NCT == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.b), "+v"(f.a[0]), "+v"(f.a[1]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f.b), "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]) : "n"(N));
}

// s_waitcnt vmcnt(n) for a wavefront-uniform runtime n (the instruction takes an immediate).
__device__ __forceinline__ void cr_vmwait(int n) {
#define CR_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    CR_VM(0) CR_VM(1) CR_VM(2) CR_VM(3) CR_VM(4) CR_VM(5) CR_VM(6) CR_VM(7) CR_VM(8) CR_VM(9) CR_VM(10) CR_VM(11)
    CR_VM(12) CR_VM(13) CR_VM(14) CR_VM(15) CR_VM(16) CR_VM(17) CR_VM(18) CR_VM(19) CR_VM(20) CR_VM(21) CR_VM(22)
    CR_VM(23) CR_VM(24) CR_VM(25) CR_VM(26) CR_VM(27) CR_VM(28) CR_VM(29) CR_VM(30) CR_VM(31) CR_VM(32) CR_VM(33)
    CR_VM(34) CR_VM(35) CR_VM(36) CR_VM(37) CR_VM(38) CR_VM(39) CR_VM(40) CR_VM(41) CR_VM(42) CR_VM(43) CR_VM(44)
    CR_VM(45) CR_VM(46) CR_VM(47) CR_VM(48)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef CR_VM
}

// Where a step sits: position stage, window group, channel block, ring buffer, validity slot — advanced by increments
// (a runtime integer division costs ~30 vector instructions even for uniform values; the first version spent more issue
// cycles on st / SPP, st % NB ... than on its MFMAs).
struct CrCursor {
  int st, ps, wg0, cb, buf, vslot;
  __device__ __forceinline__ void next(int KB, int NWG, int NB) {
    ++st;
    if (++buf == NB) buf = 0;
    if (++cb == KB) {
      cb = 0;
      if (++wg0 == NWG) {
        wg0 = 0;
        ++ps;
        if (++vslot == NB) vslot = 0;
      }
    }
  }
};

// WPS: windows (kt taps) of a 3x1x1 layer staged per step — 3 for rows of <= 32 floats, 1 for 64-float rows.
// RES: a residual operand exists (eval-mode shortcuts, accumulating data gradients); its loads are raw buffer loads
// behind a counted wait, issued in front of the step's LDS-DMA loads.
//
// Pipeline: a ring of NB = D + 1 stage buffers, the loads of steps st + 1 .. st + D in flight while step st is
// multiplied (a step's LDS-DMA round trip is ~2.7 us under load against 0.2 .. 0.8 us of MFMAs: with ONE step in flight
// — round 6's first version — every workgroup ran at a DMA latency per step).  Every wavefront issues the same number
// `nper` of LDS-DMA instructions per step (the window is padded to whole groups of four 1 KiB pieces; steps past the
// end are issued with out-of-range offsets), so "step st has landed" is s_waitcnt vmcnt((D - 1) nper [+ residual
// loads]): the count of YOUNGER LOADS only — stores complete in any order against loads, so they are not counted, and a
// store that lingers only makes the wait longer.  No compiler-visible LDS access or global load sits inside the loop
// (hipcc would put s_waitcnt vmcnt(0) in front of it: MI355X_MICROARCH.md), the barrier is the raw s_barrier.
template <int NTAP, int XW, int NCT, int WPS, bool RES>
__global__ __launch_bounds__(256) void conv_rows_kernel(const CRowsArgs p) {
  constexpr int BCO = 16 * NCT;
  constexpr int NQ = XW >= 16 ? XW / 16 : 1;       // 16-channel chunks per row
  constexpr int XSH = (XW == 8) ? 1 : (XW == 16 ? 2 : (XW == 32 ? 3 : 4));  // log2(chunks per row)
  constexpr int TPS = (NTAP == 9) ? 9 : WPS;       // taps multiplied per step
  constexpr int NI = TPS * NQ;                     // operand items per step
  extern __shared__ __attribute__((aligned(16))) float cr_smem[];
  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int co0 = blockIdx.y * BCO;
  const int mb = blockIdx.x * p.chunk;
  const int me = (mb + p.chunk < p.M) ? mb + p.chunk : p.M;
  const int nps = (me > mb) ? (me - mb + CR_L - 1) / CR_L : 0;
  const int KB = p.KB;
  constexpr int NWG = (NTAP == 3) ? 3 / WPS : 1;     // window groups per position stage
  const int SPP = NWG * KB;                        // steps per position stage: (window group, channel block)
  const int krow = NTAP * KB * XW;                 // floats of a weight row
  const int win = p.wrp * XW;                      // floats of one x window
  const int stage_floats = WPS * win;
  const int NB = p.NB, D = NB - 1;
  float* const Wl = cr_smem;
  float* const stg = Wl + BCO * p.wstride;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)cr_smem;
  const unsigned stg_base = lds_base + (unsigned)((BCO * p.wstride) << 2);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int nxb = win >> 8;                        // 1 KiB pieces of one window
  const int nper = (WPS * nxb) >> 2;               // LDS-DMA instructions per wavefront and step (launcher: exact)
  const int nst = nps * SPP;

  // Per-lane constants of this wavefront's LDS-DMA pieces (piece b = wave + 4 i of every step): byte offset of the lane's
  // 16-byte chunk within the step's source rows, or out of range for the padding rows of a window.  A step then costs
  // one add per piece: source offset = wavefront-uniform base of the step's window + this constant.  (Offsets past the
  // tensor — rows behind the last position, rows in front of the first: negative bases wrap to > 2^31 — read as zeros
  // through the buffer's bounds check; the launcher keeps the tensor below 2^30 bytes so that base + 2^31 stays out.)
  constexpr int MAXP = 6;
  unsigned lc[MAXP];
  int chlim[MAXP];  // channels of the row left at this lane's chunk (a channel block reads while cb * XW < chlim)
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int b = wave + 4 * i;
    const int wi = (WPS == 1) ? 0 : (b >= 2 * nxb ? 2 : (b >= nxb ? 1 : 0));
    const int f = ((b - wi * nxb) << 6) + lane;  // 16-byte chunk index within the window
    const int row = f >> XSH, slot = f - (row << XSH);
    const int c4 = 4 * (slot ^ cr_swz<XW>(row));
    lc[i] = (row < p.rows_needed && i < nper) ? (unsigned)((row * d.in_cs + c4) << 2) : CR_OOB;
    chlim[i] = d.Cin - c4;
  }
  auto issue = [&](const CrCursor& c) {
    const int m0 = mb + c.ps * CR_L;
    const bool live = c.st < nst && !(p.dbg & 1);
    const unsigned lds0 = stg_base + (unsigned)((c.buf * stage_floats) << 2);
    const int cbx = c.cb * XW;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (i < nper) {
        const int b = wave + 4 * i;
        const int wi = (WPS == 1) ? 0 : (b >= 2 * nxb ? 2 : (b >= nxb ? 1 : 0));  // window within the step
        const int kt = (NTAP == 3) ? c.wg0 * WPS + wi : 1;
        // wavefront-uniform base of the window: row (m0 - halo + sg (kt - 1) HW), channel block cb
        const unsigned base = live ? (unsigned)(((m0 - p.halo + p.sg * (kt - 1) * p.HW) * d.in_cs + d.in_coff + cbx) << 2)
                                   : CR_OOB;
        const unsigned vo = (cbx < chlim[i]) ? base + lc[i] : CR_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            x_rs, (lds_void*)(__attribute__((address_space(3))) char*)(uintptr_t)(lds0 + (unsigned)(b << 10)), 16, vo, 0, 0, 0);
      }
    }
  };

  // ---- the workgroup's weights: rows co0 .. co0 + BCO - 1, [tap][KB * XW channels] each, zero where the layer ends
  //      (the pad behind a row is zero-filled too: with 8-channel rows lane groups 2 and 3 read past a tap's channels,
  //      and what they read is multiplied by zeros — it must be finite)
  {
    const int r4 = p.wstride >> 2;
    for (int idx = tid; idx < BCO * r4; idx += 256) {
      const int r = idx / r4, e = (idx - r * r4) << 2;
      const int tap = e / (KB * XW), c = e - tap * (KB * XW);
      const int co = co0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < krow && co < d.Cout && c + 3 < d.cin_pad)
        v = *reinterpret_cast<const f32x4*>(p.w + ((long)co * NTAP + tap) * d.cin_pad + c);
      *reinterpret_cast<f32x4*>(Wl + r * p.wstride + e) = v;
    }
  }

  f32x4 acc[NCT], acc2[NCT];  // two chains per tile (K steps s = 0, 2 / 1, 3): no MFMA waits for the one in front of it
  // statistics (per lane: the four channels co0 + 16 ct + 4 g .. + 3, its positions j of every stage)
  f32x4 sk[NCT], s1[NCT], s2[NCT];
  float cnt = 0.f;
  bool have_k = false;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    sk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s1[ct] = sk[ct];
    s2[ct] = sk[ct];
    acc[ct] = sk[ct];
    acc2[ct] = sk[ct];
  }
  // per-channel epilogue vectors once, in front of the loop
  f32x4 esc[NCT], ebi[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = co0 + 16 * ct + 4 * g;
    esc[ct] = (f32x4){1.f, 1.f, 1.f, 1.f};
    ebi[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (n < d.Cout) {
      if (p.scale) esc[ct] = *reinterpret_cast<const f32x4*>(p.scale + n);
      if (p.bias) ebi[ct] = *reinterpret_cast<const f32x4*>(p.bias + n);
    }
  }
  const unsigned a_lane = lds_base + (unsigned)((j * p.wstride + 4 * g) << 2);  // this lane's weight row / K group
  const unsigned wrow16 = (unsigned)((16 * p.wstride) << 2);                    // bytes between column tiles
  __amdgpu_buffer_rsrc_t r_rs = x_rs;
  if constexpr (RES) r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res_bytes, 0x00020000);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // weights, scale, bias: nothing of them is counted below

  CrCursor ic = {0, 0, 0, 0, 0, 0}, cc = {0, 0, 0, 0, 0, 0};  // issue / consume positions
  unsigned vb = 0u;
  for (int i = 0; i < D; ++i) {
    issue(ic);
    ic.next(KB, NWG, NB);
  }
  for (int st = 0; st < nst; ++st, cc.next(KB, NWG, NB)) {
    const int ps = cc.ps, wg0 = cc.wg0, cb = cc.cb;
    const bool last_of_stage = (wg0 == NWG - 1) && (cb == KB - 1);
    // step st has landed when at most the loads issued behind it are outstanding: D - 1 steps of LDS-DMA (+ residual
    // loads of those steps' epilogues, issued in front of their DMA)
    cr_vmwait((D - 1) * (nper + (RES ? NCT : 0)));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int m = mb + ps * CR_L + 16 * wave + j;
    const bool mok = m < me;
    f32x4 rv[NCT];
    if constexpr (RES) {
      if (last_of_stage) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int n = co0 + 16 * ct + 4 * g;
          const unsigned ro = (mok && n < d.Cout) ? ((unsigned)m * (unsigned)d.res_cs + (unsigned)(d.res_coff + n)) << 2 : CR_OOB;
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rv[ct]) : "v"(ro), "s"(r_rs) : "memory");
        }
      }
    }
    issue(ic);
    ic.next(KB, NWG, NB);
    __builtin_amdgcn_sched_barrier(0);
    if (NTAP > 1 && wg0 == 0 && cb == 0) {  // which taps of this lane's position lie inside its clip (once per stage)
      const unsigned mm = (unsigned)(mb + ps * CR_L + 16 * wave + j);
      const unsigned q1 = cr_mdiv(mm, p.w_mul, p.w_sh);
      const int w = (int)(mm - q1 * (unsigned)d.Wo);
      const unsigned q2 = cr_mdiv(q1, p.h_mul, p.h_sh);
      const int h = (int)(q1 - q2 * (unsigned)d.Ho);
      const unsigned q3 = cr_mdiv(q2, p.t_mul, p.t_sh);
      const int t = (int)(q2 - q3 * (unsigned)d.To);
      vb = 0u;
      if (NTAP == 3) {
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
          if ((unsigned)(t + p.sg * (kt - 1)) < (unsigned)d.To) vb |= 1u << kt;
      } else {  // separable: bits 0..2 = row h + sg (kh - 1) inside, bits 4..6 = column w + sg (kw - 1) inside
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          if ((unsigned)(h + p.sg * (kk - 1)) < (unsigned)d.Ho) vb |= 1u << kk;
          if ((unsigned)(w + p.sg * (kk - 1)) < (unsigned)d.Wo) vb |= 16u << kk;
        }
      }
    }
    const unsigned xs = stg_base + (unsigned)((cc.buf * stage_floats) << 2);
    // operand item it = (tap of the step, 16-channel chunk): reads issued one item ahead of the MFMAs
    auto fetch = [&](int it, CrFrag<NCT>& f, bool& ok) {
      const int tl = it / NQ, q = it - tl * NQ;   // tap within the step
      int R, tap;
      unsigned xw;
      ok = true;
      if (NTAP == 9) {
        const int kh = tl / 3, kw = tl - 3 * kh;
        tap = tl;
        R = p.halo + 16 * wave + j + p.sg * ((kh - 1) * d.Wi + (kw - 1));
        ok = ((vb >> kh) & 1u) && ((vb >> (4 + kw)) & 1u);
        xw = xs;
      } else if (NTAP == 3) {
        tap = wg0 * WPS + tl;
        R = 16 * wave + j;
        ok = (vb >> tap) & 1u;
        xw = xs + (unsigned)((tl * win) << 2);
      } else {
        tap = 0;
        R = 16 * wave + j;
        xw = xs;
      }
      const int c = 4 * q + g;
      f.b = cr_lds128(xw + (unsigned)((R * XW + ((c ^ cr_swz<XW>(R)) << 2)) << 2));
      const unsigned ao = a_lane + (unsigned)((tap * (KB * XW) + cb * XW + 16 * q) << 2);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) f.a[ct] = cr_lds128(ao + ct * wrow16);
    };
    CrFrag<NCT> fr[2];
    bool okv[2];
    if (!(p.dbg & 2)) {
    fetch(0, fr[0], okv[0]);
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      CrFrag<NCT>& f = fr[it & 1];
      if (it + 1 < NI) {
        fetch(it + 1, fr[(it + 1) & 1], okv[(it + 1) & 1]);
        cr_wait<1 + NCT, NCT>(f);
      } else {
        cr_wait<0, NCT>(f);
      }
      f32x4 b = f.b;
      if (!okv[it & 1] || (XW < 16 && g >= XW / 4)) b = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][0], b[0], acc[ct], 0, 0, 0);
        acc2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][1], b[1], acc2[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][2], b[2], acc[ct], 0, 0, 0);
        acc2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][3], b[3], acc2[ct], 0, 0, 0);
      }
    }
    }
    if (!last_of_stage) continue;
    // ---- epilogue of position stage ps: lane = position m, channels co0 + 16 ct + 4 g .. + 3
    const int m_first = mb + ps * CR_L + 16 * wave;  // this wavefront's first position of the stage (lane j = 0)
    if constexpr (RES) {  // the residual loads sit in front of this step's nper LDS-DMA loads
      cr_vmwait(nper);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) asm volatile("" : "+v"(rv[ct]));
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int n = co0 + 16 * ct + 4 * g;
      const bool nok = n < d.Cout;
      f32x4 v = (acc[ct] + acc2[ct]) * esc[ct] + ebi[ct];
      acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[ct] = acc[ct];
      if constexpr (RES) v += rv[ct];
      if (mok && nok && !(p.dbg & 4)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sf_act(v[r], d.act);
        *reinterpret_cast<f32x4*>(p.out + (long)m * d.out_cs + d.out_coff + n) = v;
      }
      if (p.stats) {
        if (!have_k && m_first < me) {  // K = the wavefront's first output row (a sample of the channel): lane 16 g
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float k0 = __builtin_amdgcn_readlane(v[r], 0), k1 = __builtin_amdgcn_readlane(v[r], 16);
            const float k2 = __builtin_amdgcn_readlane(v[r], 32), k3 = __builtin_amdgcn_readlane(v[r], 48);
            sk[ct][r] = g == 0 ? k0 : (g == 1 ? k1 : (g == 2 ? k2 : k3));
          }
        }
        if (mok) {
          const f32x4 dv = v - sk[ct];
          s1[ct] += dv;
          s2[ct] += dv * dv;
        }
      }
    }
    if (p.stats) {
      if (m_first < me) have_k = true;
      if (mok) cnt += 1.f;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the out-of-range tail loads still write (zeros) into LDS

  if (p.stats) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      cnt += __shfl_xor(cnt, off, 64);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[ct][r] += __shfl_xor(s1[ct][r], off, 64);
          s2[ct][r] += __shfl_xor(s2[ct][r], off, 64);
        }
    }
    if (j == 0) {
      const long part = (long)blockIdx.x * 4 + wave;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int n = co0 + 16 * ct + 4 * g;
        if (n < d.Cout) {
          float* const o = p.stats + (part * (d.Cout >> 2) + (n >> 2)) * 16;
          *reinterpret_cast<f32x4*>(o) = (f32x4){cnt, cnt, cnt, cnt};
          *reinterpret_cast<f32x4*>(o + 4) = sk[ct];
          *reinterpret_cast<f32x4*>(o + 8) = s1[ct];
          *reinterpret_cast<f32x4*>(o + 12) = s2[ct];
        }
      }
    }
  }
}

// ---- 3x1x1 layers as a RING OVER t ---------------------------------------------------------------------------------------
// (as conv_wgrad_rows.hip's conv_wgrad_tring_kernel) A workgroup owns a COLUMN — one clip, one block of 64 positions of
// the frame — and walks a segment of t: per step it loads frame t + 2 of x ONCE into a ring of four LDS slots and
// multiplies the frames t - 1, t, t + 1 with the three taps' weights — a third of the staged bytes of the window form
// above, a third of its LDS (two .. four workgroups per CU), and NO validity bits: frames outside the clip and rows past
// the frame are out-of-range offsets that land as zeros.  Everything else (swizzled rows, weights in LDS, swapped-operand
// MFMAs, raw LDS reads behind counted waits, epilogue with scale / bias / residual / activation / statistics) is the
// kernel above.  One step in flight (deeper rings measured no faster for the weight gradient's ring).
template <int XW, int NCT, bool RES>
__global__ __launch_bounds__(256) void conv_rows_tring_kernel(const CRowsArgs p) {
  constexpr int BCO = 16 * NCT;
  constexpr int NQ = XW >= 16 ? XW / 16 : 1;
  constexpr int XSH = (XW == 8) ? 1 : (XW == 16 ? 2 : (XW == 32 ? 3 : 4));
  constexpr int NXB = CR_L * XW / 256;             // 1 KiB pieces of a frame's block
  constexpr int MAXP = (NXB + 3) / 4;
  constexpr int SLOT = CR_L * XW;                  // floats
  extern __shared__ __attribute__((aligned(16))) float cr_smem[];
  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int co0 = blockIdx.y * BCO;
  const int unit = blockIdx.x;
  const int seg = unit % p.nseg, col = unit / p.nseg;
  const int hb = col % p.nhb, n = col / p.nhb;
  const int T = d.To;
  const int t0 = seg * p.tseg, t1 = (t0 + p.tseg < T) ? t0 + p.tseg : T;
  const int r_base = hb * CR_L;
  const int krow = 3 * XW;
  float* const Wl = cr_smem;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)cr_smem;
  const unsigned ring_base = lds_base + (unsigned)((BCO * p.wstride) << 2);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int nper = (NXB > wave) ? (NXB - wave + 3) / 4 : 0;   // pieces this wavefront issues per frame

  unsigned lc[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int b = wave + 4 * i;
    const int f = (b << 6) + lane;
    const int row = f >> XSH, slot = f - (row << XSH);
    const int c4 = 4 * (slot ^ cr_swz<XW>(row));
    lc[i] = (b < NXB && r_base + row < p.HW && c4 < d.Cin) ? (unsigned)((row * d.in_cs + c4) << 2) : CR_OOB;
  }
  auto issue = [&](int fx, int slot) {
    const bool fok = (unsigned)fx < (unsigned)T && fx >= t0 - 1 && fx <= t1 && !(p.dbg & 1);
    const unsigned base = fok ? (unsigned)(((((long)n * T + fx) * p.HW + r_base) * d.in_cs + d.in_coff) << 2) : CR_OOB;
    const unsigned lds0 = ring_base + (unsigned)((slot * SLOT) << 2);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int b = wave + 4 * i;
      const unsigned vo = base + lc[i];  // (named: with the sum written inline in the call, hipcc's HOST pass emitted no
                                         //  stub for this kernel — no diagnostic, an undefined symbol at dlopen)
      if (b < NXB) __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(__attribute__((address_space(3))) char*)(uintptr_t)(lds0 + (unsigned)(b << 10)), 16, vo, 0, 0, 0);
    }
  };

  {  // weights: rows co0 .. co0 + BCO - 1, [tap][XW channels], zero where the layer ends (and in the pad)
    const int r4 = p.wstride >> 2;
    for (int idx = tid; idx < BCO * r4; idx += 256) {
      const int r = idx / r4, e = (idx - r * r4) << 2;
      const int tap = e / XW, c = e - tap * XW;
      const int co = co0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < krow && co < d.Cout && c + 3 < d.cin_pad)
        v = *reinterpret_cast<const f32x4*>(p.w + ((long)co * 3 + tap) * d.cin_pad + c);
      *reinterpret_cast<f32x4*>(Wl + r * p.wstride + e) = v;
    }
  }
  f32x4 sk[NCT], s1[NCT], s2[NCT], esc[NCT], ebi[NCT];
  float cnt = 0.f;
  bool have_k = false;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    sk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s1[ct] = sk[ct];
    s2[ct] = sk[ct];
    const int nn = co0 + 16 * ct + 4 * g;
    esc[ct] = (f32x4){1.f, 1.f, 1.f, 1.f};
    ebi[ct] = sk[ct];
    if (nn < d.Cout) {
      if (p.scale) esc[ct] = *reinterpret_cast<const f32x4*>(p.scale + nn);
      if (p.bias) ebi[ct] = *reinterpret_cast<const f32x4*>(p.bias + nn);
    }
  }
  const unsigned a_lane = lds_base + (unsigned)((j * p.wstride + 4 * g) << 2);
  const unsigned wrow16 = (unsigned)((16 * p.wstride) << 2);
  const int R = 16 * wave + j;                                  // this lane's row within the block
  __amdgpu_buffer_rsrc_t r_rs = x_rs;
  if constexpr (RES) r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res_bytes, 0x00020000);
  const bool rok = r_base + R < p.HW;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  issue(t0 - 1, 0);
  issue(t0, 1);
  issue(t0 + 1, 2);
  int s_lo = 0;  // ring slot of frame t - 1; frame t + 2 goes to (s_lo + 3) & 3
  for (int t = t0; t < t1; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // frame t + 1 (issued a step ago) has landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const long m = ((long)n * T + t) * p.HW + r_base + R;
    f32x4 rv[NCT];
    if constexpr (RES) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int nn = co0 + 16 * ct + 4 * g;
        const unsigned ro = (rok && nn < d.Cout) ? ((unsigned)m * (unsigned)d.res_cs + (unsigned)(d.res_coff + nn)) << 2 : CR_OOB;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rv[ct]) : "v"(ro), "s"(r_rs) : "memory");
      }
    }
    issue(t + 2, (s_lo + 3) & 3);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NCT], acc2[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[ct] = acc[ct];
    }
    if (!(p.dbg & 2)) {
      auto fetch = [&](int it, CrFrag<NCT>& f) {
        const int tap = it / NQ, q = it - tap * NQ;
        const int slot = (s_lo + 1 + p.sg * (tap - 1)) & 3;   // frame t + sg (tap - 1)
        const int c = 4 * q + g;
        f.b = cr_lds128(ring_base + (unsigned)((slot * SLOT + R * XW + ((c ^ cr_swz<XW>(R)) << 2)) << 2));
        const unsigned ao = a_lane + (unsigned)((tap * XW + 16 * q) << 2);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) f.a[ct] = cr_lds128(ao + ct * wrow16);
      };
      constexpr int NI = 3 * NQ;
      CrFrag<NCT> fr[2];
      fetch(0, fr[0]);
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        CrFrag<NCT>& f = fr[it & 1];
        if (it + 1 < NI) {
          fetch(it + 1, fr[(it + 1) & 1]);
          cr_wait<1 + NCT, NCT>(f);
        } else {
          cr_wait<0, NCT>(f);
        }
        f32x4 b = f.b;
        if (XW < 16 && g >= XW / 4) b = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][0], b[0], acc[ct], 0, 0, 0);
          acc2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][1], b[1], acc2[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][2], b[2], acc[ct], 0, 0, 0);
          acc2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ct][3], b[3], acc2[ct], 0, 0, 0);
        }
      }
    }
    // ---- epilogue of frame t: lane = position m, channels co0 + 16 ct + 4 g .. + 3
    const bool first_ok = r_base + 16 * wave < p.HW;  // this wavefront's first row of the block exists
    if constexpr (RES) {
      cr_vmwait(nper);  // the residual loads sit in front of this step's LDS-DMA pieces
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) asm volatile("" : "+v"(rv[ct]));
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int nn = co0 + 16 * ct + 4 * g;
      const bool nok = nn < d.Cout;
      f32x4 v = (acc[ct] + acc2[ct]) * esc[ct] + ebi[ct];
      if constexpr (RES) v += rv[ct];
      if (rok && nok && !(p.dbg & 4)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sf_act(v[r], d.act);
        *reinterpret_cast<f32x4*>(p.out + m * d.out_cs + d.out_coff + nn) = v;
      }
      if (p.stats) {
        if (!have_k && first_ok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float k0 = __builtin_amdgcn_readlane(v[r], 0), k1 = __builtin_amdgcn_readlane(v[r], 16);
            const float k2 = __builtin_amdgcn_readlane(v[r], 32), k3 = __builtin_amdgcn_readlane(v[r], 48);
            sk[ct][r] = g == 0 ? k0 : (g == 1 ? k1 : (g == 2 ? k2 : k3));
          }
        }
        if (rok) {
          const f32x4 dv = v - sk[ct];
          s1[ct] += dv;
          s2[ct] += dv * dv;
        }
      }
    }
    if (p.stats) {
      if (first_ok) have_k = true;
      if (rok) cnt += 1.f;
    }
    s_lo = (s_lo + 1) & 3;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.stats) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      cnt += __shfl_xor(cnt, off, 64);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[ct][r] += __shfl_xor(s1[ct][r], off, 64);
          s2[ct][r] += __shfl_xor(s2[ct][r], off, 64);
        }
    }
    if (j == 0) {
      const long part = (long)unit * 4 + wave;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int nn = co0 + 16 * ct + 4 * g;
        if (nn < d.Cout) {
          float* const o = p.stats + (part * (d.Cout >> 2) + (nn >> 2)) * 16;
          *reinterpret_cast<f32x4*>(o) = (f32x4){cnt, cnt, cnt, cnt};
          *reinterpret_cast<f32x4*>(o + 4) = sk[ct];
          *reinterpret_cast<f32x4*>(o + 8) = s1[ct];
          *reinterpret_cast<f32x4*>(o + 12) = s2[ct];
        }
      }
    }
  }
}

void cr_magic(unsigned dv, unsigned* mul, unsigned* sh) {
  if (dv <= 1) { *mul = 0; *sh = 0; return; }
  unsigned l = 0;
  while ((1u << l) < dv) ++l;
  const unsigned long long num = 1ull << (31 + l);
  *mul = (unsigned)((num + dv - 1) / dv);
  *sh = l - 1;
}

// sf_conv_tune(22, e): 0 off, 1 = the environment's level (SF_CONV_ROWS, default 1), 2 every shape the kernels cover.
// Level 1 takes ONLY the 3x1x1 layers with <= 16 output channels over <= 64 input channels, on the ring over t
// (conv_rows_tring_kernel): the one form that beats what it replaces (profiles/r06_conv_rows_ab.txt: 32 -> 8 65 -> 51 us,
// 16 -> 8 40 -> 31, 64 -> 16 127 -> 82 / 41.5 -> 34; wider outputs — the data gradients 16 -> 64, 32 -> 128 — lose 10-25 %).
// The window form (conv_rows_kernel: every other shape) stays OFF — measured on MI355X (profiles/r06_conv_rows_ab.txt, tools/microbench/conv_rows_bench.py, cold operands,
// launch counts of cfg #3 at 8 clips): 3.72 ms per step against 2.52 ms for the kernels it would replace.  What the
// ablations say (SF_CONV_ROWS_DBG, true kernel durations by rocprofv3): with loads, MFMAs and stores all switched off a
// launch keeps 40 .. 50 % of its time, and neither the ring depth (2 .. 8 steps in flight) nor the number of workgroups
// moves it — it is the step SKELETON: ~100 .. 400 instructions per wavefront and step (piece addressing, the validity
// bits' three divisions by multiplication, cursor, counted wait, barrier, epilogue) around 4 .. 48 MFMAs, issued by ONE
// wavefront per SIMD wherever the windows + weights need > 80 KB of LDS, so every instruction's latency is exposed.
// It is NOT the LDS-DMA path: tools/microbench/lds_delivery.hip measures 7.2 TB/s (HBM) .. 14 TB/s (L2) of contiguous
// 1 KiB direct-to-LDS pieces, the same as register loads (profiles/r06_lds_delivery.txt).  What would change it: 256
// positions per stage (four column tiles per wavefront: a quarter of the skeleton per position, and 1.45x instead of
// 2.8x halo rows for 1x3x3), a ring over t for 3x1x1 (one window per frame instead of three: a third of the LDS, two
// workgroups per CU), validity bits advanced incrementally.  Kept: correct for every shape it covers
// (tests/test_conv_rows_gpu.py), selectable for A/B runs.
int g_crows_enable = 1;

int crows_level() {
  static const int env = [] {
    const char* e = getenv("SF_CONV_ROWS");
    return e ? atoi(e) : 1;
  }();
  return g_crows_enable == 1 ? env : g_crows_enable;
}

struct CRowsPlan { int ntap, xw, nct, nby; size_t lds; bool ring; };

// Geometry + schedule for a problem this kernel covers; false = leave it to the other dense kernels.
bool crows_plan(const sf_conv_desc* d, CRowsArgs* a, CRowsPlan* pl) {
  const int level = crows_level();
  if (level <= 0) return false;
  if (d->sT != 1 || d->sH != 1 || d->sW != 1 || d->dT != 1 || d->dH != 1 || d->dW != 1) return false;
  if (d->To != d->Ti || d->Ho != d->Hi || d->Wo != d->Wi) return false;
  if (d->out_cmul != 1 || d->os_T > 1 || d->os_H > 1 || d->os_W > 1) return false;
  int ntap;
  if (d->kT == 1 && d->kH == 1 && d->kW == 1 && d->pT == 0 && d->pH == 0 && d->pW == 0) ntap = 1;
  else if (d->kT == 3 && d->kH == 1 && d->kW == 1 && d->pT == 1 && d->pH == 0 && d->pW == 0) ntap = 3;
  else if (d->kT == 1 && d->kH == 3 && d->kW == 3 && d->pT == 0 && d->pH == 1 && d->pW == 1) ntap = 9;
  else return false;
  if (d->Cin > 32 && d->Cout > 32) return false;       // both wide: conv_wave / conv_bx / conv_pw_bx
  if (d->Cin < 8 || d->Cout < 8 || d->Cin > 128 || d->Cout > 256) return false;
  if (ntap == 9 && d->Cin > 32) return false;           // (a 1x3x3 window of wide rows would not fit beside the weights)
  if (d->Cin % 4 || d->in_cs % 4 || d->in_coff % 4 || d->Cout % 4 || d->out_cs % 4 || d->out_coff % 4) return false;
  if (d->cin_pad % 4) return false;
  if (d->act != SF_ACT_NONE && d->act != SF_ACT_RELU && d->act != SF_ACT_RELU6) return false;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M < 1024 || M > 0x3fffffffL) return false;
  const long xb = ((M - 1) * d->in_cs + d->in_coff + d->Cin) * 4L;
  if (xb >= 0x3fffffffL) return false;  // (see the kernel's offset arithmetic)
  // 8 -> 8 spatial layers: conv_small.hip (vector FMAs from an LDS halo tile) unless forced
  if (level < 2 && d->Cin <= 8 && d->Cout <= 8 && ntap == 9) return false;
  int xw = 8;
  while (xw < d->Cin && xw < 64) xw <<= 1;
  a->d = *d;
  a->M = (int)M;
  a->x_bytes = (unsigned)xb;
  a->KB = sf_cdiv(d->Cin, xw);
  a->HW = d->Hi * d->Wi;
  a->halo = (ntap == 9) ? d->Wi + 1 : 0;
  const int wps = (ntap == 3 && xw <= 32) ? 3 : 1;  // windows staged per step
  // window rows: whole 1 KiB pieces, and a step's pieces a multiple of four (every wavefront issues the same number)
  const int gran = 256 / xw * (wps == 3 ? 4 : 4);
  a->wrp = (CR_L + 2 * a->halo + gran - 1) / gran * gran;
  a->rows_needed = CR_L + 2 * a->halo;
  const int nct = d->Cout <= 16 ? 1 : (d->Cout <= 32 ? 2 : 4);
  if (xw > 32 && nct > 2) return false;
  const int krow = ntap * a->KB * xw;
  a->wstride = (krow + 63) / 64 * 64 + 4;  // = 4 (mod 64) floats: 16 consecutive rows on 16 different bank groups
  a->sg = d->transposed ? -1 : 1;
  // ring depth: as many stage buffers as fit beside the weights in ~72 KB (two workgroups per CU), 3 .. 8
  const size_t wbytes = (size_t)16 * nct * a->wstride * 4, sbytes = (size_t)wps * a->wrp * xw * 4;
  static const int nb_env = [] { const char* e = getenv("SF_CONV_ROWS_NB"); return e ? atoi(e) : 0; }();
  int nb = nb_env > 0 ? nb_env : (int)((72 * 1024 - (long)wbytes) / (long)sbytes);
  if (nb < 3) nb = 3;
  if (nb > 8) nb = 8;
  while (nb > 2 && wbytes + nb * sbytes > 150 * 1024) --nb;
  a->NB = nb;
  static const int dbg_env = [] { const char* e = getenv("SF_CONV_ROWS_DBG"); return e ? atoi(e) : 0; }();
  a->dbg = dbg_env;
  pl->lds = wbytes + nb * sbytes;
  if (pl->lds > 150 * 1024) return false;
  if (((wps * a->wrp * xw) >> 10) > 24) return false;                   // at most 6 pieces per wavefront and step
  if ((nb - 1) * (((wps * a->wrp * xw) >> 12) + 4) > 48) return false;  // the counted wait's immediate
  pl->ntap = ntap; pl->xw = xw; pl->nct = nct; pl->nby = sf_cdiv(d->Cout, 16 * nct);
  pl->ring = false;
  a->nhb = a->nseg = a->tseg = 0;
  static const int ring_env = [] { const char* e = getenv("SF_CONV_ROWS_RING"); return e ? atoi(e) : 1; }();
  if (ntap == 3 && a->KB == 1 && ring_env && d->To >= 2 && a->HW >= 32) {  // 3x1x1 over <= 64 channels: the ring over t
    const int nhb = sf_cdiv(a->HW, CR_L);
    int nseg = 1;
    while (nseg * 2 <= d->To / 2 && (long)d->N * nhb * nseg * pl->nby < 600 && (long)d->N * nhb * nseg * 2 <= 1024) nseg *= 2;
    const int tseg = sf_cdiv(d->To, nseg);
    nseg = sf_cdiv(d->To, tseg);
    const long units = (long)d->N * nhb * nseg;
    const size_t ring_lds = wbytes + (size_t)4 * CR_L * xw * sizeof(float);
    if (units <= 1024 && ring_lds <= 150 * 1024) {
      pl->ring = true;
      pl->lds = ring_lds;
      a->nhb = nhb; a->nseg = nseg; a->tseg = tseg;
      a->S = (int)units;
    }
  }
  if (level < 2 && !(pl->ring && d->Cout <= 16)) return false;  // level 1: the ring over t where it measured faster
  cr_magic((unsigned)d->Wo, &a->w_mul, &a->w_sh);
  cr_magic((unsigned)d->Ho, &a->h_mul, &a->h_sh);
  cr_magic((unsigned)d->To, &a->t_mul, &a->t_sh);
  // workgroups along the positions: ~1024 over the channel blocks, at least two stages each, at most 1024 (the
  // statistics workspace holds >= 4096 records per channel: four per workgroup)
  const long stages = (M + CR_L - 1) / CR_L;
  static const int wgs = [] { const char* e = getenv("SF_CONV_ROWS_WGS"); return e ? atoi(e) : 768; }();
  long S = wgs / pl->nby;
  if (S > 1024) S = 1024;
  if (S > (stages + 1) / 2) S = (stages + 1) / 2;
  if (S < 1) S = 1;
  long chunk = ((M + S - 1) / S + CR_L - 1) / CR_L * CR_L;
  S = (M + chunk - 1) / chunk;
  if (!pl->ring) a->S = (int)S;  // (the ring's grid is its columns x t segments)
  a->chunk = (int)chunk;
  return true;
}

template <int NTAP, int XW, int NCT>
int launch_crows(const CRowsArgs& a, const CRowsPlan& pl, hipStream_t s) {
  constexpr int WPS = (NTAP == 3 && XW <= 32) ? 3 : 1;
  static SfLdsAttr attr, attr_res;
  if (a.res) {
    if (!sf_ensure_dyn_lds(attr_res, reinterpret_cast<const void*>(conv_rows_kernel<NTAP, XW, NCT, WPS, true>), 152 * 1024))
      return SF_ELAUNCH;
    hipLaunchKernelGGL((conv_rows_kernel<NTAP, XW, NCT, WPS, true>), dim3(a.S, pl.nby), dim3(256), pl.lds, s, a);
    SF_CHECK_LAUNCH();
    return SF_OK;
  }
  if (!sf_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_rows_kernel<NTAP, XW, NCT, WPS, false>), 152 * 1024))
    return SF_ELAUNCH;
  hipLaunchKernelGGL((conv_rows_kernel<NTAP, XW, NCT, WPS, false>), dim3(a.S, pl.nby), dim3(256), pl.lds, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int NTAP, int XW>
int launch_crows_ct(const CRowsArgs& a, const CRowsPlan& pl, hipStream_t s) {
  if (pl.nct == 1) return launch_crows<NTAP, XW, 1>(a, pl, s);
  if (pl.nct == 2) return launch_crows<NTAP, XW, 2>(a, pl, s);
  if constexpr (XW <= 32) return launch_crows<NTAP, XW, 4>(a, pl, s);
  return 1;
}

template <int NTAP>
int launch_crows_xw(const CRowsArgs& a, const CRowsPlan& pl, hipStream_t s) {
  if (pl.xw == 8) return launch_crows_ct<NTAP, 8>(a, pl, s);
  if (pl.xw == 16) return launch_crows_ct<NTAP, 16>(a, pl, s);
  if (pl.xw == 32) return launch_crows_ct<NTAP, 32>(a, pl, s);
  if constexpr (NTAP != 9) return launch_crows_ct<NTAP, 64>(a, pl, s);
  return 1;
}

template <int XW, int NCT>
int launch_cring(const CRowsArgs& a, const CRowsPlan& pl, hipStream_t s) {
  static SfLdsAttr attr, attr_res;
  if (a.res) {
    if (!sf_ensure_dyn_lds(attr_res, reinterpret_cast<const void*>(conv_rows_tring_kernel<XW, NCT, true>), 152 * 1024))
      return SF_ELAUNCH;
    hipLaunchKernelGGL((conv_rows_tring_kernel<XW, NCT, true>), dim3(a.S, pl.nby), dim3(256), pl.lds, s, a);
  } else {
    if (!sf_ensure_dyn_lds(attr, reinterpret_cast<const void*>(conv_rows_tring_kernel<XW, NCT, false>), 152 * 1024))
      return SF_ELAUNCH;
    hipLaunchKernelGGL((conv_rows_tring_kernel<XW, NCT, false>), dim3(a.S, pl.nby), dim3(256), pl.lds, s, a);
  }
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int XW>
int launch_cring_ct(const CRowsArgs& a, const CRowsPlan& pl, hipStream_t s) {
  if (pl.nct == 1) return launch_cring<XW, 1>(a, pl, s);
  if (pl.nct == 2) return launch_cring<XW, 2>(a, pl, s);
  if constexpr (XW <= 32) return launch_cring<XW, 4>(a, pl, s);
  return 1;
}

int launch_cring_xw(const CRowsArgs& a, const CRowsPlan& pl, hipStream_t s) {
  if (pl.xw == 8) return launch_cring_ct<8>(a, pl, s);
  if (pl.xw == 16) return launch_cring_ct<16>(a, pl, s);
  if (pl.xw == 32) return launch_cring_ct<32>(a, pl, s);
  return launch_cring_ct<64>(a, pl, s);
}

}  // namespace

int sf_conv_rows_tune(int value) {
  if (value < 0 || value > 2) return SF_EINVAL;
  g_crows_enable = value;
  return SF_OK;
}

// Shape-only decision; *parts = statistics records per channel the launch leaves (4 per workgroup).
int sf_conv_rows_takes(const sf_conv_desc* d, int* parts) {
  CRowsArgs a;
  CRowsPlan pl;
  if (parts) *parts = 0;
  if (!crows_plan(d, &a, &pl)) return 0;
  if (parts) *parts = 4 * a.S;
  return 1;
}

extern "C" int sf_conv_rows_parts(const sf_conv_desc* d) {
  int parts = 0;
  return (d && sf_conv_rows_takes(d, &parts)) ? parts : 0;
}

long sf_conv_wave_max_parts(long M);  // conv_wave.hip: what sf_conv_stats_ws_floats sized the statistics rows for

// Returns 1 when the shape is not taken (the caller goes on to its other kernels), else SF_OK / an error code.
int sf_conv_rows_try(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                     const float* bias, const float* res, float* out, hipStream_t stream, float* stats,
                     int* stat_parts) {
  if (stat_parts) *stat_parts = 0;
  CRowsArgs a;
  CRowsPlan pl;
  if (!crows_plan(d, &a, &pl)) return 1;
  if (!sf_aligned16(in) || !sf_aligned16(out) || !sf_aligned16(w_packed)) return 1;
  if ((scale && !sf_aligned16(scale)) || (bias && !sf_aligned16(bias))) return 1;
  if (res && ((d->res_cs % 4) || (d->res_coff % 4) || !sf_aligned16(res))) return 1;
  a.x = in; a.w = w_packed; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.res_bytes = 0;
  if (res) {
    const long rb = ((long)(a.M - 1) * d->res_cs + d->res_coff + d->Cout) * 4L;
    if (rb >= 0x7fffffffL) return 1;
    a.res_bytes = (unsigned)rb;
  }
  const bool want = stats && stat_parts && !scale && !res && d->act == SF_ACT_NONE && !d->transposed &&
                    sf_aligned16(stats) && 4L * a.S <= sf_conv_wave_max_parts(a.M);
  a.stats = want ? stats : nullptr;
  if (want) *stat_parts = 4 * a.S;
  int rc;
  if (pl.ring) rc = launch_cring_xw(a, pl, stream);
  else if (pl.ntap == 1) rc = launch_crows_xw<1>(a, pl, stream);
  else if (pl.ntap == 3) rc = launch_crows_xw<3>(a, pl, stream);
  else rc = launch_crows_xw<9>(a, pl, stream);
  if (rc != SF_OK && stat_parts) *stat_parts = 0;
  return rc;
}
