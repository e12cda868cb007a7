// conv_bx.hip — long-reduction dense convolutions as an implicit GEMM on the BF16 matrix pipe with fp32-exact
// operands (gfx950).
//
// gfx950's f32-input MFMA runs at the vector rate; its bf16 MFMA is 16x that.  Every fp32 operand is the exact sum of
// three bf16 pieces and six v_mfma_f32_32x32x16_bf16 per product give the fp32 product to fp32 rounding level (bx.h),
// at 6/16 of the f32 MFMA's cycles.  The long-K layers of the Slow pathway (1x3x3 and 3x1x1 over 128..2048 channels,
// resnet_helper.py:182-223) are MFMA-bound at ~0.7 of the f32 peak in conv_wave.hip — that instruction is the ceiling —
// so they run here instead:
//
//   * operands as PLANES, written once per call: the activation rows as planes[3][rows + 1][Cin] bf16 (row `rows` is
//     zeros: every padded tap and every ragged tile row points there, so the loop has no predicate) and the packed
//     weights as planes[3][Cout + 1][taps * Cin] (sf_bx_split_rows: one pass, 4 B read + 6 B written per element);
//   * workgroup = 8 wavefronts, tile 256 x BN (BN = 256 | 128) outputs, K step 16: per step and plane the tile's rows
//     arrive by direct-to-LDS buffer loads (buffer_load_dwordx4 ... lds, 1 KiB = 32 rows x 32 B per wave-instruction,
//     no staging VGPRs), three stages deep with a counted vmcnt and ONE raw barrier per step.  A row's two 16-byte
//     chunks are swapped in rows 16..31 of each 32-row block (on the SOURCE address — the LDS image of an LDS-DMA is
//     lane-linear), which makes every ds_read_b128 fragment read conflict-free;
//   * wavefront tile 64 x BN/2 = 2 x (BN/64) MFMA tiles, 6 MFMAs per tile and step on three A and three B pieces;
//   * short-M layers (res4 / res5: 49 / 26 tiles) share a tile's K steps between S workgroups; every workgroup stores a
//     raw fp32 partial tile to ws[S][M][Cout] and the split-K finish kernel of conv_igemm.hip sums them in split order
//     and applies the epilogue (scale, bias, residual, activation, scattered stores) — no float atomics.
//
// Replaces (same descriptors, same results to fp32 rounding): sf_conv_fwd_ws for the shapes sf_conv_bx_takes accepts,
// forward and (desc.transposed) data gradient.
#include "bx.h"
#include <stdlib.h>

namespace {

constexpr int BXC_BM = 256;      // rows per tile
constexpr int BXC_BK = 16;       // channels per K step (one 32x32x16 MFMA k extent)
constexpr int BXC_STAGES = 3;
constexpr int BXC_ROWB = BXC_BK * 2;            // bytes of a row per plane and step
constexpr int BXC_A_STAGE = 3 * BXC_BM * BXC_ROWB;  // 24 KiB

struct BxArgs {
  sf_conv_desc d;
  const unsigned short* ap;  // activation planes [3][a_rows + 1][Cin]
  const unsigned short* bp;  // weight planes [3][Cout + 1][K]
  float* ws;                 // [S][M][Cout] (S > 1, or an epilogue the kernel does not do itself)
  const float* scale;        // DIRECT epilogue: out = act(scale * acc + bias + res)
  const float* bias;
  const float* res;
  float* out;
  int M, ntaps, cpk, nk, S, nk_per, nb_n, tiles;
  int tap_major;             // K order: 1 = tap outer / channel chunk inner (strided layers), 0 = chunk outer / tap inner
  unsigned a_rows, a_bytes, b_bytes;      // rows of a plane (the zero row's index), bytes of all three planes
  unsigned a_plane, b_plane;              // bytes per plane
  unsigned wo_mul, wo_sh, ho_mul, ho_sh, to_mul, to_sh;
  const float* xf;           // AREG (pointwise layers): the fp32 activation rows themselves, split in registers
  unsigned xf_bytes;
  float* stats;              // AREG + DIRECT: per-tile channel statistics of the stored outputs (conv_wave's records)
};

__device__ __forceinline__ unsigned mdiv(unsigned n, unsigned mul, unsigned sh) {
  return mul ? (__umulhi(n, mul) >> sh) : n;
}

typedef __attribute__((address_space(3))) void lds_void;

template <int BN, bool DIRECT, int VAR>
__global__ __launch_bounds__(512, 2) void conv_bx_kernel(const BxArgs p) {
  // VAR & 32 (AF32): the A operand arrives as the fp32 rows themselves (16 KB per step instead of 24 KB of pieces, and
  // no activation planes) and is split into its three pieces after the fragment read — conv_pw_bx_kernel's A path.
  constexpr bool AF32 = (VAR & 32) != 0;
  constexpr int A_ST = AF32 ? BXC_BM * 64 : BXC_A_STAGE;
  constexpr int B_STAGE = 3 * BN * BXC_ROWB;
  constexpr int STAGE = A_ST + B_STAGE;
  constexpr int NT = BN / 64;  // 32-column MFMA tiles per wavefront
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  // ---- which tile and which share of the K steps
  const int bid = xcd_remap(blockIdx.x, p.tiles * p.S);
  const int split = bid / p.tiles, tile = bid - split * p.tiles;
  const int tile_m = tile / p.nb_n, tile_n = tile - tile_m * p.nb_n;
  const int m0 = tile_m * BXC_BM, n0 = tile_n * BN;
  const int ks0 = split * p.nk_per;
  const int ks1 = ks0 + p.nk_per < p.nk ? ks0 + p.nk_per : p.nk;
  const int nsteps = ks1 - ks0;

  // ---- loader role.  A: wave w brings rows 32w .. 32w+31 of the tile (all three planes); lane = (row, chunk).
  const int lrow = lane >> 1;
  const unsigned lchunk = (unsigned)((lane & 1) ^ ((lrow >> 4) & 1)) * 16u;  // source chunk (swizzle, see header)
  const int am = m0 + 32 * wave + lrow;
  const bool a_ok = am < p.M;
  int t0, h0, w0, nimg;
  {
    const unsigned mm = a_ok ? (unsigned)am : 0u;
    const unsigned q1 = mdiv(mm, p.wo_mul, p.wo_sh);
    const unsigned wo = mm - q1 * (unsigned)d.Wo;
    const unsigned q2 = mdiv(q1, p.ho_mul, p.ho_sh);
    const unsigned ho = q1 - q2 * (unsigned)d.Ho;
    const unsigned q3 = mdiv(q2, p.to_mul, p.to_sh);
    const unsigned to = q2 - q3 * (unsigned)d.To;
    nimg = (int)q3;
    t0 = d.transposed ? (int)to + d.pT : (int)to * d.sT - d.pT;
    h0 = d.transposed ? (int)ho + d.pH : (int)ho * d.sH - d.pH;
    w0 = d.transposed ? (int)wo + d.pW : (int)wo * d.sW - d.pW;
  }
  // AF32 loader: wave w, piece u brings rows 32 w + 16 u + (lane >> 2), 16-byte slot lane & 3 = source chunk
  // slot ^ ((row >> 2) & 3) (conflict-free fragment reads, see conv_pw_bx_kernel)
  // (plain structs, not arrays indexed by u: hipcc kept the arrays in SCRATCH — 40 bytes per lane, re-read by every K
  // step's address computation with a memory latency each and counted in the same vmcnt as the LDS-DMA pieces)
  struct FRow { int t0, h0, w0, n; bool ok; unsigned chunk; };
  FRow fr0 = {0, 0, 0, 0, false, 0u}, fr1 = fr0;
  if constexpr (AF32) {
    auto frow = [&](int u) -> FRow {
      FRow f;
      const int r = 32 * wave + 16 * u + (lane >> 2);
      const int fm = m0 + r;
      f.ok = fm < p.M;
      f.chunk = (unsigned)((lane & 3) ^ ((r >> 2) & 3)) * 16u;
      const unsigned mm = f.ok ? (unsigned)fm : 0u;
      const unsigned q1 = mdiv(mm, p.wo_mul, p.wo_sh);
      const unsigned wo = mm - q1 * (unsigned)d.Wo;
      const unsigned q2 = mdiv(q1, p.ho_mul, p.ho_sh);
      const unsigned ho = q1 - q2 * (unsigned)d.Ho;
      const unsigned q3 = mdiv(q2, p.to_mul, p.to_sh);
      const unsigned to = q2 - q3 * (unsigned)d.To;
      f.n = (int)q3;
      f.t0 = d.transposed ? (int)to + d.pT : (int)to * d.sT - d.pT;
      f.h0 = d.transposed ? (int)ho + d.pH : (int)ho * d.sH - d.pH;
      f.w0 = d.transposed ? (int)wo + d.pW : (int)wo * d.sW - d.pW;
      return f;
    };
    fr0 = frow(0);
    fr1 = frow(1);
  }
  auto f_voff = [&](const FRow f, int kt, int kh, int kw) -> unsigned {  // byte offset of the row's chunk in the fp32 tensor
    const int ti = d.transposed ? f.t0 - kt * d.dT : f.t0 + kt * d.dT;
    const int hi = d.transposed ? f.h0 - kh * d.dH : f.h0 + kh * d.dH;
    const int wi = d.transposed ? f.w0 - kw * d.dW : f.w0 + kw * d.dW;
    const bool ok = f.ok && (unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
    if (!ok) return 0x80000000u;  // past the buffer: the DMA writes zeros
    const unsigned r = (unsigned)(((f.n * d.Ti + ti) * d.Hi + hi) * d.Wi + wi);
    return (r * (unsigned)d.in_cs + (unsigned)d.in_coff) * 4u + f.chunk;
  };
  const unsigned row_bytes = (unsigned)d.Cin * 2u;
  auto a_voff = [&](int kt, int kh, int kw) -> unsigned {  // byte offset of this lane's row for a tap (or the zero row)
    const int ti = d.transposed ? t0 - kt * d.dT : t0 + kt * d.dT;
    const int hi = d.transposed ? h0 - kh * d.dH : h0 + kh * d.dH;
    const int wi = d.transposed ? w0 - kw * d.dW : w0 + kw * d.dW;
    const bool ok = a_ok && (unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
    const unsigned r = ok ? (unsigned)(((nimg * d.Ti + ti) * d.Hi + hi) * d.Wi + wi) : p.a_rows;
    return r * row_bytes + lchunk;
  };
  // B: BN = 256: wave w brings rows 32w .. 32w+31; BN = 128: waves 0..3 only
  const bool b_loader = BN == 256 || wave < BN / 32;
  const int bn = n0 + 32 * wave + lrow;
  const unsigned kbytes = (unsigned)p.nk * (unsigned)BXC_ROWB;  // bytes of a weight row per plane
  const unsigned b_voff = (unsigned)((b_loader && bn < d.Cout) ? bn : d.Cout) * kbytes + lchunk;

  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.ap, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.bp, 0, p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xf, 0, AF32 ? p.xf_bytes : 0u, 0x00020000);

  // ---- K iteration state of the loader (runs two steps ahead of the MFMAs).  K order: 16-channel chunk OUTER, tap
  // INNER — consecutive steps read the same input rows shifted by one tap, so a workgroup's share of the K steps
  // touches (rows + halo) x (its few channel chunks) and the taps' re-reads hit L1 / L2; tap-major order streamed
  // whole rows of all channels once per tap.  Strided layers (whose taps' rows do not overlap) keep the tap-major order:
  // measured 171 vs 135 TFLOP/s on the 1x3x3 stride-2 layer; the stride-1 layers are within noise either way.
  // The loader's K position is a VALUE (struct LS, advanced by a pure function), wave-uniform and said so
  // (readfirstlane: the integer divisions run on the vector ALU).  As five int locals advanced inside [&] lambdas by
  // `if (++x == n) { x = 0; ++y; }` ladders it stayed in 40 bytes of SCRATCH: hipcc merged the ladders' `++l_kt` /
  // `++l_c` into one increment through a pointer to either, which defeated scalar replacement for all ten loader
  // variables — ~15 scratch loads / stores per K step, each a memory latency in front of the next step's DMA addresses
  // and each counted by the same vmcnt as the LDS-DMA pieces, and on every tap wrap a flat_load behind
  // s_waitcnt vmcnt(0) lgkmcnt(0): a full drain of the three-stage pipeline (round 5; hipcc -S:
  // private_segment_fixed_size 40 -> 0).
  struct LS { int c, tap, kw, kh, kt; };
  LS ls;
  ls.c = __builtin_amdgcn_readfirstlane(p.tap_major ? ks0 % p.cpk : ks0 / p.ntaps);          // 16-channel chunk
  ls.tap = __builtin_amdgcn_readfirstlane(p.tap_major ? ks0 / p.cpk : ks0 - ls.c * p.ntaps);
  ls.kw = __builtin_amdgcn_readfirstlane(ls.tap % d.kW);
  ls.kh = __builtin_amdgcn_readfirstlane((ls.tap / d.kW) % d.kH);
  ls.kt = __builtin_amdgcn_readfirstlane(ls.tap / (d.kW * d.kH));
  const int tap_major = p.tap_major, cpk = p.cpk, kW = d.kW, kH = d.kH, kT = d.kT;
  auto next = [=](const LS o) -> LS {
    LS n;
    if (tap_major) {
      const int c1 = o.c + 1;
      const bool wc = c1 == cpk;
      n.c = wc ? 0 : c1;
      n.tap = o.tap + (wc ? 1 : 0);
      const int w1 = o.kw + (wc ? 1 : 0);
      const bool ww = w1 == kW;
      n.kw = ww ? 0 : w1;
      const int h1 = o.kh + (ww ? 1 : 0);
      const bool wh = h1 == kH;
      n.kh = wh ? 0 : h1;
      n.kt = o.kt + (wh ? 1 : 0);
      return n;
    }
    const int w1 = o.kw + 1;
    const bool ww = w1 == kW;
    n.kw = ww ? 0 : w1;
    const int h1 = o.kh + (ww ? 1 : 0);
    const bool wh = h1 == kH;
    n.kh = wh ? 0 : h1;
    const int t1 = o.kt + (wh ? 1 : 0);
    const bool wt = t1 == kT;  // next channel chunk
    n.kt = wt ? 0 : t1;
    n.tap = wt ? 0 : o.tap + 1;
    n.c = o.c + (wt ? 1 : 0);
    return n;
  };
  // one step's loads = 6 LDS-DMA pieces per wave (3 A planes, 3 B planes; fp32 rows: 2 + 3)
  auto issue = [&](int stage) {
    const LS l = ls;
    unsigned cur_voff = 0, cur_aso, cur_f0 = 0, cur_f1 = 0;
    if constexpr (AF32) {
      cur_f0 = f_voff(fr0, l.kt, l.kh, l.kw);
      cur_f1 = f_voff(fr1, l.kt, l.kh, l.kw);
      cur_aso = (unsigned)l.c * 64u;
    } else {
      cur_voff = a_voff(l.kt, l.kh, l.kw);
      cur_aso = (unsigned)l.c * (unsigned)BXC_ROWB;
    }
    const unsigned cur_bso = (unsigned)(l.tap * p.cpk + l.c) * (unsigned)BXC_ROWB;
    if constexpr (AF32) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(smem + stage * STAGE + wave * 2048), 16, cur_f0, cur_aso,
                                               0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(smem + stage * STAGE + wave * 2048 + 1024), 16, cur_f1,
                                               cur_aso, 0, 0);
    } else {
      char* const sa = smem + stage * STAGE + wave * 1024;
#pragma unroll
      for (int g = 0; g < 3; ++g)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_void*)(sa + g * (BXC_BM * BXC_ROWB)), 16, cur_voff,
                                                 cur_aso + (unsigned)g * p.a_plane, 0, 0);
    }
    if (b_loader) {
      char* const sb = smem + stage * STAGE + A_ST + wave * 1024;
#pragma unroll
      for (int g = 0; g < 3; ++g)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_void*)(sb + g * (BN * BXC_ROWB)), 16, b_voff,
                                                 cur_bso + (unsigned)g * p.b_plane, 0, 0);
    }
    ls = next(l);
  };

  // ---- fragment addresses: lane (r = lane & 31, h = lane >> 5) reads 16 bytes of row r, k = 8h .. 8h+7
  const unsigned frag_lane = (unsigned)(lane & 31) * BXC_ROWB + (unsigned)(((lane >> 5) ^ ((lane >> 4) & 1)) * 16);
  const unsigned a_frag = (unsigned)(wm * 64) * BXC_ROWB + frag_lane;
  const unsigned b_frag = A_ST + (unsigned)(wn * (BN / 2)) * BXC_ROWB + frag_lane;
  unsigned f_frag[2][2] = {{0, 0}, {0, 0}};
  if constexpr (AF32) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 2; ++q) f_frag[i][q] = (unsigned)r * 64u + (unsigned)(((2 * (lane >> 5) + q) ^ ((r >> 2) & 3)) * 16);
    }
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);
  for (int it = 0; it < nsteps; ++it) {
    // this wave's loads of step `it` have landed once at most the loads of step it + 1 are outstanding
    if (it + 1 < nsteps) {
      if constexpr (AF32) {
        if (b_loader) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        if (b_loader) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // ... and every other wave's: the stage is complete, and stage it - 1 is free
    if (!(VAR & 16) && it + 2 < nsteps) issue((it + 2) % BXC_STAGES);
    const char* const st = smem + (it % BXC_STAGES) * STAGE;
    u32x4 af[2][3], bf[NT][3];
    if constexpr (AF32) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(st + f_frag[i][0]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(st + f_frag[i][1]);
        split_pair(lo[0], lo[1], af[i], 0);
        split_pair(lo[2], lo[3], af[i], 1);
        split_pair(hi[0], hi[1], af[i], 2);
        split_pair(hi[2], hi[3], af[i], 3);
      }
    }
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      if constexpr (!AF32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i][pc] = *reinterpret_cast<const u32x4*>(st + a_frag + pc * (BXC_BM * BXC_ROWB) + i * (32 * BXC_ROWB));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bf[j][pc] = *reinterpret_cast<const u32x4*>(st + b_frag + pc * (BN * BXC_ROWB) + j * (32 * BXC_ROWB));
    }
    if (VAR & 8) {  // ablation: no MFMAs (the fragments stay live through one xor chain)
      unsigned x = 0;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
        for (int i = 0; i < 2; ++i) x ^= af[i][pc][0] ^ af[i][pc][1] ^ af[i][pc][2] ^ af[i][pc][3];
#pragma unroll
        for (int j = 0; j < NT; ++j) x ^= bf[j][pc][0] ^ bf[j][pc][1] ^ bf[j][pc][2] ^ bf[j][pc][3];
      }
      acc[0][0][0] += __builtin_bit_cast(float, x & 0x007fffffu);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_split(af[i], bf[j], acc[i][j]);
    }
  }

  // ---- an accumulator register holds column lane & 31 of rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): 128
  // contiguous bytes per half-wave and row.  DIRECT (one workgroup per tile, dense stores): the conv epilogue here;
  // otherwise the raw partial tile -> ws[split][m][n] for the finish kernel
  const int col = lane & 31, rsub = 4 * (lane >> 5);
  if ((VAR & 4) && p.M > 0) {  // ablation: no stores (the accumulators stay live: the sum goes out when M <= 0, i.e. never)
    return;
  }
  if (DIRECT) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + col;
      const bool n_ok = n < d.Cout;
      const float sc = (p.scale && n_ok) ? p.scale[n] : 1.f;
      const float bi = (p.bias && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // the residual may BE the output buffer (accumulating data gradients): all 16 loads of a tile are issued
        // before its first store, so they are in flight together instead of one round trip per element
        float r[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + rsub;
          r[e] = (p.res && m < p.M && n_ok) ? p.res[(long)m * d.res_cs + d.res_coff + n] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + rsub;
          if (m < p.M && n_ok) p.out[(long)m * d.out_cs + d.out_coff + n] = sf_act(acc[i][j][e] * sc + bi + r[e], d.act);
        }
      }
    }
    return;
  }
  float* const wsp = p.ws + (long)split * p.M * d.Cout;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + col;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + rsub;
        if (m < p.M && n < d.Cout) wsp[(long)m * d.Cout + n] = acc[i][j][e];
      }
    }
}

// ---- pointwise layers ------------------------------------------------------------------------------------------------
// 1x1x1 stride-1 convs (and their data gradients: the same GEMM on the transposed packed weight) with 64..2048 channels
// on both sides — conv_a / conv_c of every bottleneck, resnet_helper.py:182-223.  Here the activation operand is used
// ONCE per output column block, so writing it out as piece planes first (4 B read + 6 B written per element, then 6 B
// read) costs more than the f32 MFMA it replaces.  Instead the step's fp32 rows themselves ([256][16 channels], 16 KB)
// arrive in LDS by LDS-DMA next to the weights' piece planes (three stages, one barrier per step, rows past M read as
// zeros), a lane reads the 32 bytes of its MFMA A fragment (channels 16 c + 8 h .. + 7 of row r) with two
// ds_read_b128 and splits them into the three bf16 pieces IN REGISTERS (bx.h split_pair, 11 vector instructions per
// pair, in the shadow of the bf16 MFMAs).  (Register loads for the rows instead — no LDS for A — were slower: mixed
// with LDS-DMA loads the compiler waits for vmcnt(0) at every use, and the prefetch is lost.)  Tile 256 x BN, 8 wavefronts of 64 x BN/2, direct epilogue (scale, bias,
// residual, activation) and, for training-mode forward convs, the BN batch statistics of what was stored, in
// conv_wave.hip's record format [part = M tile][C / 4][count, K, S1, S2][4] (K = the tile's first row).
// ST = 3: three LDS stages, one workgroup per CU (two wavefronts per SIMD).  ST = 2 (BN = 128 only, <= 128 registers):
// two stages = 56 KB, so TWO workgroups share a CU — four wavefronts per SIMD, and while one workgroup sits in its
// barrier / load-issue / fragment-read phase the other one's MFMAs keep the matrix pipe busy.
template <int BN, int ST>
__global__ __launch_bounds__(512, ST == 2 ? 4 : 2) void conv_pw_bx_kernel(const BxArgs p) {
  constexpr int A_STAGE = BXC_BM * 64;            // fp32 rows of a step: 256 x 16 channels x 4 B
  constexpr int B_STAGE = 3 * BN * BXC_ROWB;
  constexpr int STAGE = A_STAGE + B_STAGE;
  constexpr int NT = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int bid = xcd_remap(blockIdx.x, p.tiles);
  const int tile_m = bid / p.nb_n, tile_n = bid - tile_m * p.nb_n;
  const int m0 = tile_m * BXC_BM, n0 = tile_n * BN;
  const int nsteps = p.nk;

  // B loader (as conv_bx_kernel): BN = 256: wave w brings rows 32w .. 32w+31; BN = 128: waves 0..3 only
  const int lrow = lane >> 1;
  const unsigned lchunk = (unsigned)((lane & 1) ^ ((lrow >> 4) & 1)) * 16u;
  const bool b_loader = BN == 256 || wave < BN / 32;
  const int bn = n0 + 32 * wave + lrow;
  const unsigned kbytes = (unsigned)p.nk * (unsigned)BXC_ROWB;
  const unsigned b_voff = (unsigned)((b_loader && bn < d.Cout) ? bn : d.Cout) * kbytes + lchunk;
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.bp, 0, p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xf, 0, p.xf_bytes, 0x00020000);
  // A loader: the step's fp32 rows [256][16 channels] as they lie in memory, by LDS-DMA (every load of this kernel is
  // an LDS-DMA: mixed with register loads the compiler drains the whole queue at every use).  Wave w, piece u brings
  // rows 32 w + 16 u .. + 15: lane = (row, 16-byte slot); slot s of row r holds source chunk s ^ ((r >> 2) & 3), which
  // puts the 16 rows a quarter-wave reads (64 B apart) on 16 different 16-byte bank groups.
  unsigned a_lvoff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = 32 * wave + 16 * u + (lane >> 2);
    const int m = m0 + r;
    const unsigned chunk = (unsigned)((lane & 3) ^ ((r >> 2) & 3));
    a_lvoff[u] = m < p.M ? ((unsigned)m * (unsigned)d.in_cs + (unsigned)d.in_coff) * 4u + chunk * 16u : 0x80000000u;
  }
  auto issue = [&](int step) {
    char* const st = smem + (step % ST) * STAGE;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(st + wave * 2048 + u * 1024), 16, a_lvoff[u],
                                               (unsigned)step * 64u, 0, 0);
    if (b_loader) {
      char* const sb = st + A_STAGE + wave * 1024;
#pragma unroll
      for (int g = 0; g < 3; ++g)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_void*)(sb + g * (BN * BXC_ROWB)), 16, b_voff,
                                                 (unsigned)step * (unsigned)BXC_ROWB + (unsigned)g * p.b_plane, 0, 0);
    }
  };
  // fragment addresses.  A: lane (r = lane & 31, h = lane >> 5) of tile i reads channels 8 h .. 8 h + 7 of row
  // 64 wm + 32 i + r: source chunks 2 h and 2 h + 1, each at slot chunk ^ ((row >> 2) & 3)
  unsigned a_frag[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      a_frag[i][q] = (unsigned)r * 64u + (unsigned)(((2 * (lane >> 5) + q) ^ ((r >> 2) & 3)) * 16);
  }
  const unsigned frag_lane = (unsigned)(lane & 31) * BXC_ROWB + (unsigned)(((lane >> 5) ^ ((lane >> 4) & 1)) * 16);
  const unsigned b_frag = A_STAGE + (unsigned)(wn * (BN / 2)) * BXC_ROWB + frag_lane;

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nsteps > 0) issue(0);
  if (ST == 3 && nsteps > 1) issue(1);
  for (int it = 0; it < nsteps; ++it) {
    if (ST == 3 && it + 1 < nsteps) {  // this wave's loads of step `it` have landed once at most those of step it + 1 are outstanding
      if (b_loader) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (it + ST - 1 < nsteps) issue(it + ST - 1);
    const char* const st = smem + (it % ST) * STAGE;
    u32x4 af[2][3], bf[NT][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(st + a_frag[i][0]);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(st + a_frag[i][1]);
      split_pair(lo[0], lo[1], af[i], 0);
      split_pair(lo[2], lo[3], af[i], 1);
      split_pair(hi[0], hi[1], af[i], 2);
      split_pair(hi[2], hi[3], af[i], 3);
    }
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bf[j][pc] = *reinterpret_cast<const u32x4*>(st + b_frag + pc * (BN * BXC_ROWB) + j * (32 * BXC_ROWB));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = mfma_split(af[i], bf[j], acc[i][j]);
  }

  // ---- epilogue: register e of a lane = column lane & 31, rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of its tiles
  const int col = lane & 31, rsub = 4 * (lane >> 5);
  const bool want_stats = p.stats != nullptr;
  float* const kred = reinterpret_cast<float*>(smem);  // [BN] shift values, then [4][BN][2] sums (the B stages are free)
  if (want_stats) __syncthreads();
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int cl = wn * (BN / 2) + j * 32 + col;  // column within the tile
    const int n = n0 + cl;
    const bool n_ok = n < d.Cout;
    const float sc = (p.scale && n_ok) ? p.scale[n] : 1.f;
    const float bi = (p.bias && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float r[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + rsub;
        r[e] = (p.res && m < p.M && n_ok) ? p.res[(long)m * d.res_cs + d.res_coff + n] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + rsub;
        const float v = sf_act(acc[i][j][e] * sc + bi + r[e], d.act);
        acc[i][j][e] = v;
        if (m < p.M && n_ok) p.out[(long)m * d.out_cs + d.out_coff + n] = v;
      }
    }
    if (want_stats && wm == 0 && rsub == 0) kred[cl] = acc[0][j][0];  // the tile's first row (always a valid row)
  }
  if (!want_stats) return;
  __syncthreads();
  float* const sred = kred + BN;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int cl = wn * (BN / 2) + j * 32 + col;
    const float k0 = kred[cl];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + rsub;
        if (m < p.M) {
          const float dv = acc[i][j][e] - k0;
          s1 += dv;
          s2 += dv * dv;
        }
      }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (rsub == 0) {
      sred[(wm * BN + cl) * 2] = s1;
      sred[(wm * BN + cl) * 2 + 1] = s2;
    }
  }
  __syncthreads();
  if (tid < BN) {
    const int n = n0 + tid;
    if (n < d.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s1 += sred[(w * BN + tid) * 2];
        s2 += sred[(w * BN + tid) * 2 + 1];
      }
      const int left = p.M - m0;
      const float cnt = (float)(left < BXC_BM ? left : BXC_BM);
      float* const o = p.stats + ((long)tile_m * (d.Cout >> 2) + (n >> 2)) * 16 + (n & 3);
      o[0] = cnt;
      o[4] = kred[tid];
      o[8] = s1;
      o[12] = s2;
    }
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------
// dW[co][tap][ci] = sum_m dz[m][co] * x[row(m, tap)][ci]: the reduction runs over the POSITIONS and both operands lie
// position-major with channels contiguous, so a K step is 16 positions: the dz rows [16][BCO channels] and, per tap of
// the column tile, the gathered x rows [16][channels] arrive by LDS-DMA as they lie in memory, and the MFMA fragments
// (8 consecutive positions of one channel per lane) are read with ds_read_b64_tr_b16, the transposing LDS read: per 16
// lanes a block of 4 positions x 16 channels comes back channel-major.  A row's 16-byte chunks are XOR-swizzled by
// 4 * (position & 3) (on the DMA's source address) so that the four position rows of a block — 512 bytes apart, the
// same banks — land on four different 16-bank windows: conflict-free.
// Tile: BCO output channels x 256 (tap, ci) columns, 8 wavefronts, wavefront tile 64 x (128 | 64); the positions are
// shared by S workgroups per tile, each storing a raw partial tile part[S][Cout][taps * Cin] — exactly what
// sf_conv_wgrad_finish sums and un-packs.
struct BxwArgs {
  sf_conv_desc d;
  const unsigned short* xp;  // input planes [3][x_rows + 1][Cin]
  const unsigned short* zp;  // dz planes [3][M + 1][Cout]
  float* part;               // [S][Cout][Kc]
  int M, Kc, S, chunk, nb_col, tiles;
  unsigned x_rows, x_plane, x_bytes, z_plane, z_bytes;
  unsigned wo_mul, wo_sh, ho_mul, ho_sh, to_mul, to_sh, ci_mul, ci_sh;
};

__device__ __forceinline__ u32x2 lds_tr(const char* p) {
  return lds_read_tr(reinterpret_cast<const unsigned short*>(p));
}
// The same read as an instruction the compiler does not look into: its waitcnt pass puts s_waitcnt vmcnt(0) in front of
// the ds_read_tr intrinsic whenever an LDS-DMA load is outstanding (no memory operand: it may read what the DMA
// writes), which turns a three-stage LDS-DMA ring into one round trip per step.  The caller owns both counters: LDS
// address = byte offset in the workgroup's LDS, and the data is there after bxw_lgkm_drain().
__device__ __forceinline__ u32x2 lds_tr_raw(unsigned lds_byte_addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr));
  return r;
}

template <int BCO>
__global__ __launch_bounds__(512, 2) void conv_bx_wgrad_kernel(const BxwArgs p) {
  constexpr int BCOL = 256;
  constexpr int NWC = BCO == 256 ? 2 : 4;     // wavefronts along the columns
  constexpr int NT = BCOL / NWC / 32;         // 32-column MFMA tiles per wavefront
  constexpr int ROW_A = BCO * 2, ROW_B = BCOL * 2;              // bytes of a position row per plane
  constexpr int PL_A = 16 * ROW_A, PL_B = 16 * ROW_B;           // bytes of a plane per stage
  constexpr int A_STAGE = 3 * PL_A, STAGE = A_STAGE + 3 * PL_B;
  constexpr int NA = PL_A / 1024;             // A loader wavefronts (one 1 KiB piece per plane each)
  constexpr int CPR_A = BCO / 8;              // 16-byte chunks per position row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const sf_conv_desc& d = p.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / NWC, wc = wave - wr * NWC;
  const int bid = xcd_remap(blockIdx.x, p.tiles * p.S);
  const int split = bid / p.tiles, tile = bid - split * p.tiles;
  const int tile_co = tile / p.nb_col, tile_col = tile - tile_co * p.nb_col;
  const int co0 = tile_co * BCO, col0 = tile_col * BCOL;
  const int mb = split * p.chunk;
  const int me = mb + p.chunk < p.M ? mb + p.chunk : p.M;
  const int nsteps = me > mb ? (me - mb + 15) / 16 : 0;

  // ---- B loader (every wave): position 2 * wave + (lane >> 5) of the step, LDS chunk lane & 31 holds source chunk cs
  const int b_pos = 2 * wave + (lane >> 5);
  const int b_cs = (lane & 31) ^ ((b_pos & 3) << 2);
  const int b_col = col0 + b_cs * 8;
  const bool b_colok = b_col < p.Kc;
  const unsigned b_tap = b_colok ? mdiv((unsigned)b_col, p.ci_mul, p.ci_sh) : 0u;
  const unsigned b_ci = b_colok ? (unsigned)b_col - b_tap * (unsigned)d.Cin : 0u;
  const int b_kw = (int)(b_tap % (unsigned)d.kW), b_kh = (int)((b_tap / (unsigned)d.kW) % (unsigned)d.kH),
            b_kt = (int)(b_tap / (unsigned)(d.kW * d.kH));
  const int b_dt = b_kt * d.dT - d.pT, b_dh = b_kh * d.dH - d.pH, b_dw = b_kw * d.dW - d.pW;
  const unsigned x_rowb = (unsigned)d.Cin * 2u;
  auto b_voff = [&](int m) -> unsigned {
    unsigned r = p.x_rows;
    if (m < me && b_colok) {
      const unsigned q1 = mdiv((unsigned)m, p.wo_mul, p.wo_sh);
      const int wo = (int)((unsigned)m - q1 * (unsigned)d.Wo);
      const unsigned q2 = mdiv(q1, p.ho_mul, p.ho_sh);
      const int ho = (int)(q1 - q2 * (unsigned)d.Ho);
      const unsigned q3 = mdiv(q2, p.to_mul, p.to_sh);
      const int to = (int)(q2 - q3 * (unsigned)d.To);
      const int ti = to * d.sT + b_dt, hi = ho * d.sH + b_dh, wi = wo * d.sW + b_dw;
      if ((unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi)
        r = (unsigned)((((int)q3 * d.Ti + ti) * d.Hi + hi) * d.Wi + wi);
    }
    return r * x_rowb + b_ci * 2u;
  };
  // ---- A loader (waves 0 .. NA-1): dz rows, no gather
  const bool a_loader = wave < NA;
  const int a_pos = wave * (64 / CPR_A) + lane / CPR_A;
  const int a_cs = (lane % CPR_A) ^ ((a_pos & 3) << 2);
  const int a_co = co0 + a_cs * 8;
  const unsigned z_rowb = (unsigned)d.Cout * 2u;
  auto a_voff = [&](int m) -> unsigned {
    const unsigned r = (m < me && a_co < d.Cout) ? (unsigned)m : (unsigned)p.M;
    return r * z_rowb + (unsigned)(a_co < d.Cout ? a_co : 0) * 2u;
  };
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.xp, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.zp, 0, p.z_bytes, 0x00020000);
  int l_m = mb;  // first position of the step the loader issues next
  auto issue = [&](int stage) {
    char* const st = smem + stage * STAGE;
    if (a_loader) {
      const unsigned vo = a_voff(l_m + a_pos);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(z_rs, (lds_void*)(st + pc * PL_A + wave * 1024), 16, vo,
                                                 (unsigned)pc * p.z_plane, 0, 0);
    }
    const unsigned vb = b_voff(l_m + b_pos);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void*)(st + A_STAGE + pc * PL_B + wave * 1024), 16, vb,
                                               (unsigned)pc * p.x_plane, 0, 0);
    l_m += 16;
  };

  // ---- transposing fragment reads: lane = 16 g + 4 q + pq, k half h = g >> 1; read u brings positions 8h + 4u + {0..3}
  // of channels cb + 16 (g & 1) + {0..15}; this lane addresses position row q, channels 4 pq .. 4 pq + 3 of the block
  const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3, h = g >> 1;
  unsigned a_addr[2], b_addr[NT];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = wr * 64 + i * 32 + 16 * (g & 1) + 4 * pq;
    a_addr[i] = (unsigned)((8 * h + q) * ROW_A + (((ch >> 3) ^ (q << 2)) << 4) + (ch & 7) * 2);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch = wc * (BCOL / NWC) + j * 32 + 16 * (g & 1) + 4 * pq;
    b_addr[j] = (unsigned)(A_STAGE + (8 * h + q) * ROW_B + (((ch >> 3) ^ (q << 2)) << 4) + (ch & 7) * 2);
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);
  for (int it = 0; it < nsteps; ++it) {
    if (it + 1 < nsteps) {
      if (a_loader) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // This step's fragments by raw transposing reads (see lds_tr_raw: the intrinsic form made hipcc drain vmcnt in front
    // of them every step — round 4's kernel ran one DMA round trip per K step, not three stages in flight), the loads
    // of step it+2 behind them, one explicit LDS drain in front of the MFMAs.
    const unsigned st = lds_base + (unsigned)((it % BXC_STAGES) * STAGE);
    u32x4 af[2][3], bf[NT][3];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x2 lo = lds_tr_raw(st + a_addr[i] + pc * PL_A), hi = lds_tr_raw(st + a_addr[i] + pc * PL_A + 4 * ROW_A);
        af[i][pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const u32x2 lo = lds_tr_raw(st + b_addr[j] + pc * PL_B), hi = lds_tr_raw(st + b_addr[j] + pc * PL_B + 4 * ROW_B);
        bf[j][pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (it + 2 < nsteps) issue((it + 2) % BXC_STAGES);
    __builtin_amdgcn_sched_barrier(0);
    // every fragment register is an operand of the drain, so no MFMA can be scheduled in front of it
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      if constexpr (NT == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(af[0][pc]), "+v"(af[1][pc]), "+v"(bf[0][pc]), "+v"(bf[1][pc]), "+v"(bf[2][pc]), "+v"(bf[3][pc]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][pc]), "+v"(af[1][pc]), "+v"(bf[0][pc]), "+v"(bf[1][pc]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = mfma_split(af[i], bf[j], acc[i][j]);
  }

  // ---- raw partial tile -> part[split][co][col]: register e of a lane = column lane & 31, rows (e & 3) + 8 (e >> 2) + 4h
  float* const out = p.part + (long)split * d.Cout * p.Kc;
  const int ccol = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = col0 + wc * (BCOL / NWC) + j * 32 + ccol;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + rsub;
        if (co < d.Cout && col < p.Kc) out[(long)co * p.Kc + col] = acc[i][j][e];
      }
    }
}

// fp32 rows (pitch cs, C channels from coff) -> three bf16 piece planes [3][rows + 1][C]; row `rows` of every plane is
// written as zeros.  Thread = 8 channels of a row (32 B read, 3 x 16 B written).
__global__ __launch_bounds__(256) void bx_split_rows_kernel(const float* __restrict__ x, int cs, int coff, long rows,
                                                            int C, unsigned short* __restrict__ planes) {
  const int c8 = C >> 3;
  const long total = (rows + 1) * c8;
  const long plane = (rows + 1) * (long)C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / c8;
    const int c = (int)(idx - r * c8) * 8;
    u32x4 w[3];
    if (r < rows) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * cs + coff + c);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + r * cs + coff + c + 4);
      split_pair(v0[0], v0[1], w, 0);
      split_pair(v0[2], v0[3], w, 1);
      split_pair(v1[0], v1[1], w, 2);
      split_pair(v1[2], v1[3], w, 3);
    } else {
      w[0] = w[1] = w[2] = (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(planes + pc * plane + r * C + c) = w[pc];
  }
}

// The same split for MANY dense tensors in one launch (the packed weights of every bf16-piece layer after an optimizer
// step: ~50 launches of ~5 us otherwise).  items[i] = one tensor [rows][C] (pitch C); blk0[i] = its first workgroup.
struct BxSplitItem {
  const float* x;
  unsigned short* planes;
  long rows;
  int C, pad_;
};
__global__ __launch_bounds__(256) void bx_split_batched_kernel(const BxSplitItem* __restrict__ items,
                                                               const int* __restrict__ blk0, int n) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const BxSplitItem it = items[lo];
  const int c8 = it.C >> 3;
  const long idx = (long)((int)blockIdx.x - blk0[lo]) * 256 + threadIdx.x;
  if (idx >= (it.rows + 1) * c8) return;
  const long plane = (it.rows + 1) * (long)it.C;
  const long r = idx / c8;
  const int c = (int)(idx - r * c8) * 8;
  u32x4 w[3];
  if (r < it.rows) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(it.x + r * it.C + c);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(it.x + r * it.C + c + 4);
    split_pair(v0[0], v0[1], w, 0);
    split_pair(v0[2], v0[3], w, 1);
    split_pair(v1[0], v1[1], w, 2);
    split_pair(v1[2], v1[3], w, 3);
  } else {
    w[0] = w[1] = w[2] = (u32x4){0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(it.planes + pc * plane + r * it.C + c) = w[pc];
}

void magic(unsigned dv, unsigned* mul, unsigned* sh) {
  if (dv <= 1) { *mul = 0; *sh = 0; return; }
  unsigned l = 0;
  while ((1u << l) < dv) ++l;
  const unsigned long long num = 1ull << (31 + l);
  *mul = (unsigned)((num + dv - 1) / dv);
  *sh = l - 1;
}

int g_bx_enable = 1;  // sf_conv_tune(7, e)
int g_bx_dbg = 0;     // sf_conv_tune(8, mask): timing ablations (microbenchmarks): 1 = skip the operand splits (planes of
                      // the previous call), 2 = skip the finish kernel, 4 = skip the GEMM

bool bx_enabled() {
  static const int env_on = [] {
    const char* e = getenv("SF_CONV_BX");
    return e ? atoi(e) : 1;
  }();
  return env_on && g_bx_enable;
}

struct BxPlan { int bn, S, nk, nk_per, nb_n, tiles, direct; long a_rows; };

// Which shapes run here: long reductions (>= 32 K steps = 512 channel-taps) into >= 128 output channels over >= 2048
// positions — the 1x3x3 / 3x1x1 layers of res3..res5 and the 1x1x1 layers over >= 512 channels.  `gate`: also apply
// the measured win / lose rule (tools/microbench/conv_bx_bench.py); the shape rules alone decide what the kernel CAN run.
bool bx_af32(const sf_conv_desc* d);

bool bx_plan(const sf_conv_desc* d, BxPlan* pl, bool gate = true) {
  if (!bx_enabled()) return false;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  const int ntaps = d->kT * d->kH * d->kW;
  if ((d->Cin % 16) || d->cin_pad != d->Cin || (d->in_cs % 4) || (d->in_coff % 4)) return false;
  if (d->Cout < 128 || (d->Cout % 4) || d->out_cmul != 1) return false;
  const int nk = ntaps * (d->Cin / BXC_BK);
  if (nk < 32 || M < 2048 || M > 0x7fffffffL) return false;
  if (d->transposed && (d->sT != 1 || d->sH != 1 || d->sW != 1)) return false;
  const long a_rows = (long)d->N * d->Ti * d->Hi * d->Wi;
  if (3 * (a_rows + 1) * d->Cin * 2 > 0xfffffff0L) return false;
  if (3L * (d->Cout + 1) * nk * BXC_ROWB > 0xfffffff0L) return false;
  pl->bn = d->Cout >= 256 ? 256 : 128;
  pl->nb_n = sf_cdiv(d->Cout, pl->bn);
  pl->tiles = sf_cdiv(M, BXC_BM) * pl->nb_n;
  pl->nk = nk;
  pl->a_rows = a_rows;
  // Time model in microseconds, fitted on MI355X with COLD operands (conv_bx_bench.py flushes the Infinity Cache in
  // front of every timed call, as inside a training step; back-to-back calls read 20-30 % faster): a workgroup needs
  // ~2.05 us per K step of a 256-wide tile and ~1.6 of a 128-wide one — the loop is bound by the arrival of the
  // operand pieces in LDS, so the narrow tile is not half the wide one; the launch ends with its busiest CU; S > 1
  // writes S partial tiles and the finish kernel reads them back (~4 TB/s each way) — one workgroup per tile stores the
  // finished outputs itself.
  const bool scatter = d->os_T > 1 || d->os_H > 1 || d->os_W > 1;
  const double step_us = pl->bn == 256 ? 2.05 : 1.6;  // round 5 (loader state out of scratch): 2.4 / 1.85 before
  const double out_mb = (double)M * d->Cout * 4e-6;
  int best = 1;
  double best_t = 1e30;
  for (int S = 1; S <= 16 && nk / S >= 8; ++S) {
    const long wg = (long)pl->tiles * S;
    if (wg > 2048) break;
    const long rounds = sf_cdiv(wg, 256);
    double t = 6.0 + (double)rounds * sf_cdiv(nk, S) * step_us;
    if (S > 1 || scatter) t += 3.0 + out_mb * (2.0 * S + 1.0) / 4.0;  // partials out + back in, outputs out; MB / (4 TB/s) = 0.25 us
    else t += out_mb / 4.0;
    if (t < best_t - 1e-9) { best_t = t; best = S; }
  }
  static const int forced_s = [] { const char* e = getenv("SF_CONV_BX_S"); return e ? atoi(e) : 0; }();
  if (forced_s > 0 && nk / forced_s >= 1) best = forced_s;
  pl->S = best;
  pl->nk_per = sf_cdiv(nk, best);
  pl->direct = (best == 1 && !scatter) ? 1 : 0;
  if (gate && g_bx_enable < 2) {
    // against conv_wave.hip at ~100 TFLOP/s (x 0.9 for the short reductions) plus this path's activation split
    const double flop = 2.0 * M * (double)nk * BXC_BK * d->Cout;
    const double wave_us = flop / (nk >= 64 ? 100e6 : 90e6);
    // (plane-fed mode only: fp32-row launches have no activation split)
    const double split_us = bx_af32(d) ? 0.0 : 4.0 + (double)a_rows * d->Cin * 10e-6 / 3.0;  // 10 B per element at ~3 TB/s
    // column blocks that are partly padding cost as much as full ones (576 outputs = 2.25 blocks of 256: the 3x1x1
    // data gradient 256 -> 576 ran 0.46 ms here against 0.35 on conv_wave)
    const double pad = (double)(pl->nb_n * pl->bn) / (double)d->Cout;
    if (best_t * pad + split_us > 0.93 * wave_us) return false;
  }
  return true;
}


// ---- pointwise plan ---------------------------------------------------------------------------------------------------
int g_pw_enable = 1;  // sf_conv_tune(21, e): 0 off, 1 where the time model says it wins, 2 every shape it covers

struct PwPlan { int bn, nb_n, tiles, nk; };

bool pw_plan(const sf_conv_desc* d, PwPlan* pl, bool gate = true) {
  static const int env_on = [] { const char* e = getenv("SF_CONV_PW"); return e ? atoi(e) : 1; }();
  if (!bx_enabled() || !env_on || !g_pw_enable) return false;
  if (d->kT != 1 || d->kH != 1 || d->kW != 1 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT || d->pH || d->pW)
    return false;
  if (d->To != d->Ti || d->Ho != d->Hi || d->Wo != d->Wi) return false;
  if (d->os_T > 1 || d->os_H > 1 || d->os_W > 1 || d->out_cmul != 1) return false;
  if ((d->Cin % 16) || d->cin_pad != d->Cin || (d->in_cs % 4) || (d->in_coff % 4)) return false;
  if (d->Cin < 64 || d->Cout < 64 || (d->Cout % 4)) return false;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (M < 2048 || M > 0x3fffffffL) return false;
  if (((M - 1) * d->in_cs + d->in_coff + d->Cin) * 4L >= 0x7fffffffL) return false;
  pl->nk = d->Cin / BXC_BK;
  if (3L * (d->Cout + 1) * pl->nk * BXC_ROWB > 0xfffffff0L) return false;
  // 256 x 128 tiles on two LDS stages, two workgroups per CU (SF_PW_WIDE=1: 256 x 256 on three stages, one per CU)
  static const int wide = [] { const char* e = getenv("SF_PW_WIDE"); return e ? atoi(e) : 0; }();
  // 64-wide outputs (res2's 256 -> 64 / 72 -> 64 projections and the data gradients of its 64 -> 256 ones): 256 x 64
  // tiles — the rows of a tile are read ONCE (one column block), 44 KB of LDS on two stages: SF_PW_64=0 keeps them on the
  // f32 kernels
  static const int n64 = [] { const char* e = getenv("SF_PW_64"); return e ? atoi(e) : 1; }();
  pl->bn = (d->Cout >= 256 && wide) ? 256 : (d->Cout <= 64 ? 64 : 128);
  if (pl->bn == 64 && !n64) return false;
  pl->nb_n = sf_cdiv(d->Cout, pl->bn);
  pl->tiles = sf_cdiv(M, BXC_BM) * pl->nb_n;
  if (gate && g_pw_enable < 2 && g_bx_enable < 2) {
    // Measured (tools/microbench/conv_pw_bench.py, hot and cold alike): every workgroup-step moves 16 KB of rows + 24
    // (12) KB of weight pieces into LDS and the chip delivers ~2.3-3 TB/s of that whatever the shape, so a bottleneck
    // layer (6.6 GFLOP) takes 55-60 us against 70-85 on the f32 kernels — where the f32 kernels are not HBM-bound
    // themselves (res2: 200 704 rows), the reduction is not too short to amortise the pipeline (Cin >= 128) and the
    // tiles fill the chip (res5: 13 row tiles).
    // Time model (us) fitted on the training step's per-shape table (tools/prof_convs.py, two workgroups per CU): a
    // resident set of up to 256 tiles advances one K step in ~1.5 us (a workgroup's step is a DMA round trip on two
    // stages — there is no split-K here, so few tiles with a long reduction lose: 1024 -> 256 at 12 544 rows, 98
    // tiles x 64 steps, 66 -> 88 us), never faster than its operands at ~4.5 TB/s; the f32 kernels run these layers
    // at ~95 TFLOP/s with half of their HBM time beside it.  64-wide outputs fill half a column block: left out.
    const double mb = (double)M * (d->Cin + d->Cout) * 4e-6;
    if (pl->bn == 64) {
      // one column block: every row tile is read once; the layer is its operands' HBM time or its tiles' K steps
      // (measured: tools/microbench/conv_pw_bench.py, profiles/r06_conv_pw64_ab.txt)
      if (d->Cin < 128) return false;
      return true;
    }
    if (d->Cout < 128) return false;
    double t_pw = 6.0 + (double)sf_cdiv(pl->tiles, 256) * pl->nk * 1.5;
    if (t_pw < mb / 4.5) t_pw = mb / 4.5;
    const double t_f32 = 6.0 + 2.0 * M * (double)d->Cin * d->Cout / 95e6 + 0.5 * mb / 4.5;
    if (t_pw > 0.85 * t_f32) return false;
  }
  return true;
}

template <int BN, int ST>
int launch_pw(const BxArgs& a, int grid, hipStream_t stream) {
  constexpr int lds = ST * (BXC_BM * 64 + 3 * BN * BXC_ROWB);
  static_assert(lds >= (BN + 4 * BN * 2) * 4, "the statistics scratch lives in the stages");
  static SfLdsAttr at;
  if (!sf_ensure_dyn_lds(at, reinterpret_cast<const void*>(conv_pw_bx_kernel<BN, ST>), lds)) return SF_ELAUNCH;
  hipLaunchKernelGGL((conv_pw_bx_kernel<BN, ST>), dim3(grid), dim3(512), lds, stream, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// SF_CONV_BX_AF32 (default 1): forward / data-gradient launches take their activation operand as fp32 rows (no planes)
bool bx_af32(const sf_conv_desc* d) {
  static const int on = [] { const char* e = getenv("SF_CONV_BX_AF32"); return e ? atoi(e) : 1; }();
  if (!on) return false;
  const long rows = (long)d->N * d->Ti * d->Hi * d->Wi;
  return ((rows - 1) * d->in_cs + d->in_coff + d->Cin) * 4L < 0x7fffffffL;
}

long align4(long floats) { return (floats + 3) & ~3L; }
long a_plane_floats(const sf_conv_desc* d, long a_rows) { return align4((3 * (a_rows + 1) * d->Cin + 1) / 2); }
long b_plane_floats(const sf_conv_desc* d, int nk) { return align4((3L * (d->Cout + 1) * nk * BXC_BK + 1) / 2); }

template <int BN, bool DIRECT, int VAR>
int launch_bx_v(const BxArgs& a, int grid, hipStream_t stream) {
  constexpr int lds = BXC_STAGES * (((VAR & 32) ? BXC_BM * 64 : BXC_A_STAGE) + 3 * BN * BXC_ROWB);
  static SfLdsAttr at;
  if (!sf_ensure_dyn_lds(at, reinterpret_cast<const void*>(conv_bx_kernel<BN, DIRECT, VAR>), lds)) return SF_ELAUNCH;
  hipLaunchKernelGGL((conv_bx_kernel<BN, DIRECT, VAR>), dim3(grid), dim3(512), lds, stream, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int BN, bool DIRECT>
int launch_bx(const BxArgs& a, int grid, hipStream_t stream) {
  // SF_CONV_BX_VAR: timing ablations (results invalid) — 4 no stores, 8 no MFMAs, 16 no loads inside the loop.
  // Measured with them on the res4 1x3x3 layer (85 us): fragment reads + barriers alone 25 us, + the LDS-DMA loads
  // 53 us, MFMAs + fragment reads without the loads 63 us: the loop is bound by the ~48 GB/s per CU at which the
  // operand pieces arrive in LDS, not by the matrix pipe.  Tried without gain: the loads behind the fragment reads,
  // s_setprio around the MFMAs, one LDS-DMA piece per MFMA group instead of a burst behind the barrier, tap-major K order.
  static const int var = [] { const char* e = getenv("SF_CONV_BX_VAR"); return e ? atoi(e) : 0; }();
  if (a.xf) return launch_bx_v<BN, DIRECT, 32>(a, grid, stream);  // fp32 rows, split after the fragment read
  switch (var) {
    case 4: return launch_bx_v<BN, DIRECT, 4>(a, grid, stream);
    case 12: return launch_bx_v<BN, DIRECT, 12>(a, grid, stream);
    case 20: return launch_bx_v<BN, DIRECT, 20>(a, grid, stream);
    case 28: return launch_bx_v<BN, DIRECT, 28>(a, grid, stream);
    default: return launch_bx_v<BN, DIRECT, 0>(a, grid, stream);
  }
}

}  // namespace

int sf_conv_bx_tune(int value) { g_bx_enable = value; return SF_OK; }  // 0 off, 1 on where it wins, 2 every shape it covers
int sf_conv_bx_dbg(int value) { g_bx_dbg = value; return SF_OK; }

int sf_conv_bx_takes(const sf_conv_desc* d) {
  BxPlan pl;
  return bx_plan(d, &pl) ? 1 : 0;
}

// bf16 elements of the planes [3][rows + 1][C] of a [rows][C] fp32 operand
extern "C" long sf_bx_planes_elems(long rows, int C) { return 3 * (rows + 1) * (long)C; }

int sf_bx_split_rows(const float* x, int cs, int coff, long rows, int C, unsigned short* planes, hipStream_t s) {
  if ((C & 7) || (cs & 3) || (coff & 3) || !sf_aligned16(x) || !sf_aligned16(planes)) return SF_EALIGN;
  const long total = (rows + 1) * (C >> 3);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(bx_split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, cs, coff, rows, C, planes);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// x [rows][cs] fp32 (C channels from coff) -> planes: every fp32 value as the exact sum of three bf16 pieces
extern "C" int sf_bx_split(const float* x, int cs, int coff, long rows, int C, unsigned short* planes, void* stream) {
  if (!x || !planes || rows <= 0 || C <= 0) return SF_EINVAL;
  return sf_bx_split_rows(x, cs, coff, rows, C, planes, (hipStream_t)stream);
}

// sf_bx_split of n dense tensors in one launch: items = n records {const float* x; uint16* planes; int64 rows; int32 C;
// int32 0} (x [rows][C], C % 8 == 0, both 16-byte aligned), blk0[i] = first workgroup of item i (256 threads x 8
// channels each over (rows + 1) * C / 8 elements), blk0[n] = nblocks.
extern "C" int sf_bx_split_batched(const void* items, const int* blk0, int n, int nblocks, void* stream) {
  if (!items || !blk0 || n <= 0 || nblocks <= 0) return SF_EINVAL;
  hipLaunchKernelGGL(bx_split_batched_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const BxSplitItem*>(items), blk0, n);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

int sf_conv_pw_tune(int value) { g_pw_enable = value; return SF_OK; }

int sf_conv_pw_takes(const sf_conv_desc* d) {
  PwPlan pl;
  return pw_plan(d, &pl) ? 1 : 0;
}

// Workspace floats of sf_conv_fwd_pw (0: shape not served): the weight planes unless handed in (+ a non-zero token).
extern "C" long sf_conv_pw_ws_floats(const sf_conv_desc* d, int have_w_planes) {
  PwPlan pl;
  if (!d || !pw_plan(d, &pl)) return 0;
  return 4 + (have_w_planes ? 0 : b_plane_floats(d, pl.nk));
}

// Floats of the statistics scratch of sf_conv_fwd_pw (one record row per 256-position tile and 4 channels), 0: the shape
// is not served or leaves no statistics (an epilogue with scale / residual / activation).
extern "C" long sf_conv_pw_stats_floats(const sf_conv_desc* d) {
  PwPlan pl;
  if (!d || d->transposed || d->act != SF_ACT_NONE || !pw_plan(d, &pl)) return 0;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  return (long)sf_cdiv(M, BXC_BM) * 4 * d->Cout;
}

// 1 = not taken.  stats != NULL (with parts): BN batch statistics of the stored outputs, *parts = record rows per channel.
int sf_conv_pw_try(const sf_conv_desc* d, const float* in, const float* w_packed, const unsigned short* w_planes,
                   const float* scale, const float* bias, const float* res, float* out, float* ws, float* stats,
                   int* parts, hipStream_t stream) {
  PwPlan pl;
  if (parts) *parts = 0;
  if (!ws || !pw_plan(d, &pl)) return 1;
  if (!sf_aligned16(in) || !sf_aligned16(w_packed) || !sf_aligned16(ws)) return 1;
  if (w_planes && !sf_aligned16(w_planes)) return SF_EALIGN;
  if (stats && (scale || res || d->act != SF_ACT_NONE || !parts || !sf_aligned16(stats))) return 1;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  if (!w_planes) {
    unsigned short* const bp = reinterpret_cast<unsigned short*>(ws + 4);
    const int rc = sf_bx_split_rows(w_packed, pl.nk * BXC_BK, 0, d->Cout, pl.nk * BXC_BK, bp, stream);
    if (rc != SF_OK) return rc;
    w_planes = bp;
  }
  BxArgs a = {};
  a.d = *d;
  a.bp = w_planes;
  a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.M = (int)M;
  a.ntaps = 1; a.cpk = pl.nk; a.nk = pl.nk; a.S = 1; a.nk_per = pl.nk; a.nb_n = pl.nb_n; a.tiles = pl.tiles;
  a.b_plane = (unsigned)((long)(d->Cout + 1) * pl.nk * BXC_ROWB);
  a.b_bytes = 3u * a.b_plane;
  a.xf = in;
  a.xf_bytes = (unsigned)(((M - 1) * d->in_cs + d->in_coff + d->Cin) * 4L);
  a.stats = stats;
  if (stats) *parts = (int)sf_cdiv(M, BXC_BM);
  if (pl.bn == 256) return launch_pw<256, 3>(a, pl.tiles, stream);
  if (pl.bn == 64) return launch_pw<64, 2>(a, pl.tiles, stream);
  return launch_pw<128, 2>(a, pl.tiles, stream);
}

// sf_conv_fwd_ws for the pointwise shapes sf_conv_pw_ws_floats accepts, with the weight planes handed in (or NULL: made
// in ws) and, optionally, the training-mode BN statistics of the output (stats / parts as sf_conv_fwd_stats).
extern "C" int sf_conv_fwd_pw(const sf_conv_desc* d, const float* in, const float* w_packed,
                              const unsigned short* w_planes, const float* scale, const float* bias, const float* res,
                              float* out, float* ws, float* stats, int* parts, void* stream) {
  if (!d || !in || !w_packed || !out || !ws) return SF_EINVAL;
  const int rc = sf_conv_pw_try(d, in, w_packed, w_planes, scale, bias, res, out, ws, stats, parts, (hipStream_t)stream);
  return rc == 1 ? SF_ENOTTAKEN : rc;
}

// workspace floats of sf_conv_fwd_bx: the planes the call has to make itself + the S partial tiles (0: shape not served)
extern "C" long sf_conv_bx_ws_floats(const sf_conv_desc* d, int have_in_planes, int have_w_planes) {
  BxPlan pl;
  if (!d || !bx_plan(d, &pl)) return 0;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  return 4 + ((have_in_planes || bx_af32(d)) ? 0 : a_plane_floats(d, pl.a_rows)) +
         (have_w_planes ? 0 : b_plane_floats(d, pl.nk)) + (pl.direct ? 0 : (long)pl.S * M * d->Cout);
}

int sf_conv_splitk_finish(const sf_conv_desc* d, const float* ws, int S, const float* scale, const float* bias,
                          const float* res, float* out, hipStream_t s);  // conv_igemm.hip

// Returns 1 when the shape is not taken, else SF_OK / an error code.  in_planes / w_planes: sf_bx_split of the input
// view ([rows of the input][Cin]) and of the packed weights ([Cout][taps * Cin]), or NULL (made here, in ws).
int sf_conv_bx_try(const sf_conv_desc* d, const float* in, const unsigned short* in_planes, const float* w_packed,
                   const unsigned short* w_planes, const float* scale, const float* bias, const float* res, float* out,
                   float* ws, hipStream_t stream) {
  BxPlan pl;
  if (!ws || !bx_plan(d, &pl)) return 1;
  if (!sf_aligned16(in) || !sf_aligned16(w_packed) || !sf_aligned16(ws)) return 1;
  if ((in_planes && !sf_aligned16(in_planes)) || (w_planes && !sf_aligned16(w_planes))) return SF_EALIGN;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  float* cur = ws;
  int rc = SF_OK;
  const bool af32 = bx_af32(d);
  if (!in_planes && !af32) {
    unsigned short* const ap = reinterpret_cast<unsigned short*>(cur);
    cur += a_plane_floats(d, pl.a_rows);
    if (!(g_bx_dbg & 1)) rc = sf_bx_split_rows(in, d->in_cs, d->in_coff, pl.a_rows, d->Cin, ap, stream);
    if (rc != SF_OK) return rc;
    in_planes = ap;
  }
  if (!w_planes) {
    unsigned short* const bp = reinterpret_cast<unsigned short*>(cur);
    cur += b_plane_floats(d, pl.nk);
    if (!(g_bx_dbg & 1)) rc = sf_bx_split_rows(w_packed, pl.nk * BXC_BK, 0, d->Cout, pl.nk * BXC_BK, bp, stream);
    if (rc != SF_OK) return rc;
    w_planes = bp;
  }
  BxArgs a = {};
  a.d = *d;
  a.ap = in_planes; a.bp = w_planes; a.ws = cur;
  if (af32) {
    a.xf = in;
    a.xf_bytes = (unsigned)(((pl.a_rows - 1) * d->in_cs + d->in_coff + d->Cin) * 4L);
  }
  a.scale = scale; a.bias = bias; a.res = res; a.out = out;
  a.M = (int)M;
  a.ntaps = d->kT * d->kH * d->kW;
  a.cpk = d->Cin / BXC_BK;
  a.nk = pl.nk; a.S = pl.S; a.nk_per = pl.nk_per; a.nb_n = pl.nb_n; a.tiles = pl.tiles;
  a.tap_major = (d->sT > 1 || d->sH > 1 || d->sW > 1) ? 1 : 0;
  a.a_rows = (unsigned)pl.a_rows;
  a.a_plane = (unsigned)((pl.a_rows + 1) * d->Cin * 2);
  a.a_bytes = 3u * a.a_plane;
  a.b_plane = (unsigned)((long)(d->Cout + 1) * pl.nk * BXC_ROWB);
  a.b_bytes = 3u * a.b_plane;
  magic((unsigned)d->Wo, &a.wo_mul, &a.wo_sh);
  magic((unsigned)d->Ho, &a.ho_mul, &a.ho_sh);
  magic((unsigned)d->To, &a.to_mul, &a.to_sh);
  const int grid = pl.tiles * pl.S;
  if (!(g_bx_dbg & 4)) {
    if (pl.bn == 256) rc = pl.direct ? launch_bx<256, true>(a, grid, stream) : launch_bx<256, false>(a, grid, stream);
    else rc = pl.direct ? launch_bx<128, true>(a, grid, stream) : launch_bx<128, false>(a, grid, stream);
    if (rc != SF_OK) return rc;
  }
  if (pl.direct || (g_bx_dbg & 2)) return SF_OK;
  return sf_conv_splitk_finish(d, cur, pl.S, scale, bias, res, out, stream);
}

// sf_conv_fwd_ws with the operands' bf16 piece planes handed in (sf_bx_split of the input view and / or of the packed
// weights; either may be NULL): a weight's planes are made once per optimizer step, an activation's once for its
// forward conv and its weight gradient.  SF_EINVAL when the shape is not served (sf_conv_bx_ws_floats(d, ., .) == 0).
extern "C" int sf_conv_fwd_bx(const sf_conv_desc* d, const float* in, const unsigned short* in_planes,
                              const float* w_packed, const unsigned short* w_planes, const float* scale,
                              const float* bias, const float* res, float* out, float* ws, void* stream) {
  if (!d || !in || !w_packed || !out || !ws) return SF_EINVAL;
  const int rc = sf_conv_bx_try(d, in, in_planes, w_packed, w_planes, scale, bias, res, out, ws, (hipStream_t)stream);
  return rc == 1 ? SF_ENOTTAKEN : rc;
}

// ---- weight gradient on the bf16 pipe ---------------------------------------------------------------------------------
namespace {

struct BxwPlan { int bco, S, nb_col, tiles; long chunk, x_rows; };

int g_bxw_enable = 1;  // sf_conv_tune(9, e): 0 off, 1 where it wins, 2 every shape it covers

bool bxw_plan(const sf_conv_desc* d, BxwPlan* pl) {
  static const int env_on = [] { const char* e = getenv("SF_WGRAD_BX"); return e ? atoi(e) : 1; }();
  if (!env_on || !g_bxw_enable) return false;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  const int ntaps = d->kT * d->kH * d->kW;
  if ((d->Cin % 8) || d->cin_pad != d->Cin || (d->in_cs % 4) || (d->in_coff % 4)) return false;
  if (d->Cout < 128 || (d->Cout % 8)) return false;
  const long Kc = (long)ntaps * d->Cin;
  if (Kc < 512 || M < 2048 || M > 0x7ffffff0L) return false;
  const long x_rows = (long)d->N * d->Ti * d->Hi * d->Wi;
  if (3 * (x_rows + 1) * d->Cin * 2 > 0xfffffff0L || 3 * (M + 1) * d->Cout * 2 > 0xfffffff0L) return false;
  pl->bco = d->Cout >= 256 ? 256 : 128;
  pl->nb_col = sf_cdiv(Kc, 256);
  pl->tiles = sf_cdiv(d->Cout, pl->bco) * pl->nb_col;
  pl->x_rows = x_rows;
  // positions are shared by S workgroups per tile: aim at whole rounds of 256 workgroups, >= 256 positions each
  const double step_us = pl->bco == 256 ? 2.0 : 1.1;
  const double out_mb = (double)d->Cout * Kc * 4e-6;
  int best = 1;
  double best_t = 1e30;
  const int maxS = (int)(M / 256 > 512 ? 512 : M / 256);
  for (int S = 1; S <= maxS; ++S) {
    const long wg = (long)pl->tiles * S;
    if (wg > 1536) break;
    const long rounds = sf_cdiv(wg, 256);
    const long steps = sf_cdiv(sf_cdiv(M, S), 16);
    const double t = 6.0 + (double)rounds * steps * step_us + out_mb * (2.0 * S + 1.0) / 4.0;
    if (t < best_t - 1e-9) { best_t = t; best = S; }
  }
  static const int forced_s = [] { const char* e = getenv("SF_WGRAD_BX_S"); return e ? atoi(e) : 0; }();
  if (forced_s > 0) best = forced_s;
  pl->S = best;
  pl->chunk = (sf_cdiv(M, best) + 15) / 16 * 16;
  if (g_bxw_enable < 2) {
    const double flop = 2.0 * M * (double)Kc * d->Cout;
    const double wave_us = flop / 95e6;
    const double split_us = 8.0 + ((double)x_rows * d->Cin + (double)M * d->Cout) * 10e-6 / 3.0;
    if (best_t + split_us > 0.93 * wave_us) return false;
  }
  return true;
}

long x_plane_floats(const sf_conv_desc* d, long x_rows) { return align4((3 * (x_rows + 1) * d->Cin + 1) / 2); }
long z_plane_floats(const sf_conv_desc* d, long M) { return align4((3 * (M + 1) * d->Cout + 1) / 2); }

template <int BCO>
int launch_bxw(const BxwArgs& a, int grid, hipStream_t stream) {
  constexpr int lds = BXC_STAGES * (3 * 16 * BCO * 2 + 3 * 16 * 256 * 2);
  static SfLdsAttr at;
  if (!sf_ensure_dyn_lds(at, reinterpret_cast<const void*>(conv_bx_wgrad_kernel<BCO>), lds)) return SF_ELAUNCH;
  hipLaunchKernelGGL((conv_bx_wgrad_kernel<BCO>), dim3(grid), dim3(512), lds, stream, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

}  // namespace

int sf_conv_bxw_tune(int value) { g_bxw_enable = value; return SF_OK; }

// Position splits S of the bf16-piece weight-gradient kernel (0: the shape is not served — sf_conv_wgrad_splits /
// sf_conv_wgrad plan their own); the partial buffer is [S][Cout][taps][Cin] as for sf_conv_wgrad.
extern "C" int sf_conv_wgrad_bx_splits(const sf_conv_desc* d) {
  BxwPlan pl;
  return (d && bxw_plan(d, &pl)) ? pl.S : 0;
}

// workspace floats: the operand planes the call has to make itself (x: [rows of the input][Cin], dz: [M][Cout])
extern "C" long sf_conv_wgrad_bx_ws_floats(const sf_conv_desc* d, int have_x_planes, int have_dz_planes) {
  BxwPlan pl;
  if (!d || !bxw_plan(d, &pl)) return 0;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  return 4 + (have_x_planes ? 0 : x_plane_floats(d, pl.x_rows)) + (have_dz_planes ? 0 : z_plane_floats(d, M));
}

// sf_conv_wgrad on the bf16 matrix pipe: partial[S][Cout][taps][Cin] with S = sf_conv_wgrad_bx_splits(d);
// x_planes / dz_planes = sf_bx_split of the input view / of dz ([M][Cout]), or NULL (made in ws).
extern "C" int sf_conv_wgrad_bx(const sf_conv_desc* d, const float* x, const unsigned short* x_planes, const float* dz,
                                int dz_cs, int dz_coff, const unsigned short* dz_planes, float* partial, float* ws,
                                void* stream) {
  BxwPlan pl;
  if (!d || !x || !dz || !partial || !ws) return SF_EINVAL;
  if (!bxw_plan(d, &pl)) return SF_ENOTTAKEN;
  if (!sf_aligned16(x) || !sf_aligned16(dz) || !sf_aligned16(ws) || !sf_aligned16(partial) || (dz_cs % 4) || (dz_coff % 4))
    return SF_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long M = (long)d->N * d->To * d->Ho * d->Wo;
  float* cur = ws;
  int rc = SF_OK;
  if (!x_planes) {
    unsigned short* const xp = reinterpret_cast<unsigned short*>(cur);
    cur += x_plane_floats(d, pl.x_rows);
    if (!(g_bx_dbg & 1)) rc = sf_bx_split_rows(x, d->in_cs, d->in_coff, pl.x_rows, d->Cin, xp, s);
    if (rc != SF_OK) return rc;
    x_planes = xp;
  }
  if (!dz_planes) {
    unsigned short* const zp = reinterpret_cast<unsigned short*>(cur);
    cur += z_plane_floats(d, M);
    if (!(g_bx_dbg & 1)) rc = sf_bx_split_rows(dz, dz_cs, dz_coff, M, d->Cout, zp, s);
    if (rc != SF_OK) return rc;
    dz_planes = zp;
  }
  BxwArgs a;
  a.d = *d;
  a.xp = x_planes; a.zp = dz_planes; a.part = partial;
  a.M = (int)M;
  a.Kc = d->kT * d->kH * d->kW * d->Cin;
  a.S = pl.S; a.chunk = (int)pl.chunk; a.nb_col = pl.nb_col; a.tiles = pl.tiles;
  a.x_rows = (unsigned)pl.x_rows;
  a.x_plane = (unsigned)((pl.x_rows + 1) * d->Cin * 2);
  a.x_bytes = 3u * a.x_plane;
  a.z_plane = (unsigned)((M + 1) * d->Cout * 2);
  a.z_bytes = 3u * a.z_plane;
  magic((unsigned)d->Wo, &a.wo_mul, &a.wo_sh);
  magic((unsigned)d->Ho, &a.ho_mul, &a.ho_sh);
  magic((unsigned)d->To, &a.to_mul, &a.to_sh);
  magic((unsigned)d->Cin, &a.ci_mul, &a.ci_sh);
  if (g_bx_dbg & 4) return SF_OK;
  const int grid = pl.tiles * pl.S;
  return pl.bco == 256 ? launch_bxw<256>(a, grid, s) : launch_bxw<128>(a, grid, s);
}
