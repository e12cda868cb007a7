// attn_flash.hip — CMDA SpatialAttention as a streaming (flash-style) fp32 kernel for gfx950.
//
// Reference semantics (wdf_attention_helper.py:41-54, then custom_video_model_builder.py:143-146):
//     S[i,j] = sum_c q[i,c] k[j,c]            (NO 1/sqrt(d) scaling)
//     P      = softmax_j(S)                    (over all N = T*H*W keys)
//     o[i,c] = sum_j P[i,j] v[j,c]
//     y      = gamma * o + x ;  z = relu(bn_s2f(y)) ;  z repeated alpha times along T ; written into
//                                                      the fast tensor's first C channels (torch.cat)
// The reference materialises S and P (2 x N^2 fp32 = 5 GB per clip at N = 25088); here neither ever
// leaves registers: online softmax with running max / sum per query row.
//
// Mapping (one wavefront = 32 query rows, 4 wavefronts per workgroup, K/V tiles shared through LDS):
//   * "swapped" first product  S^T = K * Q^T  on v_mfma_f32_32x32x2_f32: the accumulator then has the
//     QUERY on the lane (col = lane&31) and 16 KEYS in registers (row = (r&3) + 8*(r>>2) + 4*(lane>>5)),
//     so the row max / row sum of the softmax are in-register reductions plus ONE exchange between the
//     two lane halves — no cross-lane shuffles per element.
//   * second product  O^T = V^T * P^T : P^T's registers are already the B operand (k = key index on
//     the register, col = query on the lane) — the key order kappa(s,h) = (s&3) + 8*(s>>2) + 4h is
//     simply used for the V^T A-operand reads as well (ds_read_b32, conflict free: 32 consecutive
//     channels per half-wave).  O^T keeps the query on the lane, so the per-row rescale is per-lane.
//   * K rows are read for the first product as 16-byte fragments: lane half h takes channels
//     8q+4h..8q+4h+3 for A (K) and B (Q) alike (K order permuted identically on both operands).
//   * K/V tiles: global -> registers (issued before the tile's MFMAs) -> LDS (after them), double
//     buffered, one barrier per tile.
// fp32 throughout (the reference never leaves fp32; gfx950 has no reduced-precision f32 MFMA path).
#include "attn_args.h"
#include "attn_bx.h"
#include <stdlib.h>
#include <type_traits>

namespace {

using AttnArgs = SfAttnArgs;

constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;

// v_pk_add_f32: two fp32 adds in one vector-ALU pass (the compiler's SLP vectoriser leaves most of them scalar).
// The s_nop covers the wait state a vector instruction needs after a transcendental one (v_exp_f32) wrote its
// source: the compiler tracks that hazard for its own instructions, not for inline assembly.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// What a wavefront does with its finished O^T tile (query on the lane, channels in registers): either park the key
// part's unnormalised (O^T, m, l) for attn_fwd_merge_kernel, or normalise and run the attention epilogue
// y = gamma * o / l + x ; z = act(scale*y + bias) ; nearest-upsample x alpha along T.
template <int CP, int VEC>
__device__ __forceinline__ void attn_fwd_finish(const AttnArgs& p, const f32x16 (&o)[(CP + 31) / 32], float m_run,
                                                float l_tot, int b, int bz, int qrow, int lh) {
  constexpr int CT = (CP + 31) / 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  if (qrow >= N) return;
  if (p.zs > 1) {  // this key part's (O^T, m, l): merged and finished by attn_fwd_merge_kernel
    const long prow = (long)bz * N + qrow;
    if (lh == 0) *reinterpret_cast<float2*>(p.part_ml + prow * 2) = make_float2(m_run, l_tot);
    float* po = p.part_o + prow * CP;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = ct * 32 + 8 * g + 4 * lh;
        if (c0 < CP)
          *reinterpret_cast<f32x4*>(po + c0) =
              (f32x4){o[ct][4 * g], o[ct][4 * g + 1], o[ct][4 * g + 2], o[ct][4 * g + 3]};
      }
    return;
  }
  const float inv_l = 1.0f / l_tot;
  const float gamma = p.gamma ? p.gamma[0] : 1.0f;
  const int HW = p.H * p.W;
  const int tq = qrow / HW;
  const int hw = qrow - tq * HW;
  const float* xp = p.x + (brow + qrow) * p.x_cs;
  const long orow0 = ((long)b * p.T * p.alpha + (long)tq * p.alpha) * HW + hw;
  if (p.lse_save && lh == 0) p.lse_save[brow + qrow] = m_run + __log2f(l_tot);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = ct * 32 + 8 * g + 4 * lh;
      if (c0 >= C) continue;
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        const bool okc = c < C;
        if (p.o_save && okc) p.o_save[(brow + qrow) * C + c] = o[ct][4 * g + e] * inv_l;
        float v = gamma * (o[ct][4 * g + e] * inv_l) + (okc ? xp[c] : 0.f);
        if (p.scale && okc) v = v * p.scale[c] + p.bias[c];
        if (p.act == SF_ACT_RELU) v = fmaxf(v, 0.f);
        y[e] = v;
      }
      for (int r = 0; r < p.alpha; ++r) {
        float* op = p.out + (orow0 + (long)r * HW) * p.out_cs + p.out_coff + c0;
        if (VEC == 4) {
          *reinterpret_cast<f32x4*>(op) = (f32x4){y[0], y[1], y[2], y[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((c0 + e) < C) op[e] = y[e];
        }
      }
    }
  }
}

template <int CP, int VEC, bool STALE>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
  constexpr int KT = (CP >= 128) ? 32 : 64;   // keys per LDS tile
  constexpr int NSUB = KT / 32;
  constexpr int KS = CP + 4;                  // K row pitch (dwords): ds_read_b128 conflict-free
  constexpr int VS = CP;                      // V row pitch
  constexpr int CT = (CP + 31) / 32;          // 32-channel output tiles
  constexpr int QS = CP / 8;                  // 8-channel groups in the first product
  constexpr int F4 = CP / 4;                  // float4 per row
  constexpr int NF = (KT * F4 + 255) / 256;   // float4 per thread per tile (K and V each)

  __shared__ __attribute__((aligned(16))) float smem[2 * KT * (KS + VS)];
  float* const Ks = smem;
  float* const Vs = smem + 2 * KT * KS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int bz = blockIdx.x / p.nqt;  // workgroup -> (clip b, key part z, query tile)
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int q0 = (blockIdx.x - bz * p.nqt) * 128 + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;

  // ---- Q fragments (B operand of S^T = K Q^T): lane (i, h) holds Q[i][8q + 4h + e]
  float qf[QS * 4];
  {
    const int qrow = q0 + li;
    const bool ok = qrow < N;
    const float* qp = p.q + (brow + (ok ? qrow : 0)) * p.q_cs;
#pragma unroll
    for (int s = 0; s < QS; ++s) {
      const int c = s * 8 + lh * 4;
      if (VEC == 4) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (ok && c < C) t = *reinterpret_cast<const f32x4*>(qp + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) qf[s * 4 + e] = t[e] * LOG2E;  // exp2 domain: p = 2^(s' - m')
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) qf[s * 4 + e] = (ok && (c + e) < C) ? qp[c + e] * LOG2E : 0.f;
      }
    }
  }

  f32x16 o[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
  float m_run = NEG_BIG;            // the reference maximum m_ref of this lane's query (see the S^T tile below)
  f32x2 lacc = {0.f, 0.f};          // packed partial row sums of 2^(s - m_ref)
  f32x16 negm;                      // -m_ref on all 16 registers: the C operand of a tile's first MFMA
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = -NEG_BIG;

  f32x4 rk[NF], rv[NF];
  auto load_tile = [&](int j0) {
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + u * 256;
      const int row = f / F4;
      const int c = (f - row * F4) * 4;
      const int j = j0 + row;
      f32x4 tk = {0.f, 0.f, 0.f, 0.f}, tv = {0.f, 0.f, 0.f, 0.f};
      if (f < KT * F4 && j < N) {
        const float* kp = p.k + (brow + j) * p.k_cs + c;
        const float* vp = p.v + (brow + j) * p.v_cs + c;
        if (VEC == 4) {
          if (c < C) {
            tk = *reinterpret_cast<const f32x4*>(kp);
            tv = *reinterpret_cast<const f32x4*>(vp);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((c + e) < C) {
              tk[e] = kp[e];
              tv[e] = vp[e];
            }
        }
      }
      rk[u] = tk;
      rv[u] = tv;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + u * 256;
      if (f < KT * F4) {
        const int row = f / F4;
        const int c = (f - row * F4) * 4;
        *reinterpret_cast<f32x4*>(Ks + (buf * KT + row) * KS + c) = rk[u];
        *reinterpret_cast<f32x4*>(Vs + (buf * KT + row) * VS + c) = rv[u];
      }
    }
  };

  const int tz = ((N + KT - 1) / KT + p.zs - 1) / p.zs;  // key tiles per part
  const int t0 = z * tz;
  const int ntiles = min((N + KT - 1) / KT, t0 + tz);
  load_tile(t0 * KT);
  store_tile(0);
  __syncthreads();
  for (int t = t0; t < ntiles; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = (t + 1) < ntiles;
    if (more) load_tile((t + 1) * KT);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      const int jbase = t * KT + sub * 32;
      if (jbase >= N) break;  // wave-uniform
      f32x16 s;
      const float* krow;
      const bool ragged = jbase + 32 > N;  // wave-uniform
      if constexpr (STALE) {
      // ---- S^T tile: rows = keys (registers), cols = queries (lanes).  fp32 MFMAs run on the SIMD's vector ALUs —
      // a v_sub / v_add issued beside them is NOT hidden (tools/microbench/mfma_coexec.hip: +4.5 cycles each, v_exp
      // +7.5) — so the softmax keeps its vector work minimal: the tile is accumulated ON TOP of -m_ref (the first
      // MFMA's C operand is a register block holding -m_ref of this lane's query), with m_ref a STALE running
      // maximum that is only refreshed when a tile exceeds it by more than 2^soft_t (then S is recomputed from 0 and
      // O, l are rescaled — the classic online-softmax step, taken a handful of times per sweep): no subtraction,
      // no per-tile rescale factor, and the row sums are kept as packed partial sums.
      krow = Ks + (buf * KT + sub * 32 + li) * KS + lh * 4;
      s = negm;
#pragma unroll
      for (int g = 0; g < QS; ++g) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + g * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g * 4 + e], s, 0, 0, 0);
      }
      if (ragged) {                        // mask keys >= N
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = jbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (j >= N) s[r] = NEG_BIG;
        }
      }
      float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
      mx = fmaxf(mx, s[15]);
      if (__any(mx > p.soft_t)) {  // refresh the reference (always on the first tile: -m_ref = +BIG there)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < QS; ++g) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + g * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g * 4 + e], s, 0, 0, 0);
        }
        if (ragged) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = jbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (j >= N) s[r] = NEG_BIG;
          }
        }
        float mloc = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - mnew);  // 0 on the first tile
        lacc *= alpha;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        m_run = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          negm[r] = -mnew;
          s[r] -= mnew;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
#pragma unroll
      for (int r = 0; r < 16; r += 2) lacc = pk_add(lacc, (f32x2){s[r], s[r + 1]});
      } else {
      // ---- S^T tile: rows = keys (registers), cols = queries (lanes)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      krow = Ks + (buf * KT + sub * 32 + li) * KS + lh * 4;
#pragma unroll
      for (int g = 0; g < QS; ++g) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + g * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g * 4 + e], s, 0, 0, 0);
      }
      if (jbase + 32 > N) {  // ragged last tile: mask keys >= N (wave-uniform branch)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = jbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (j >= N) s[r] = NEG_BIG;
        }
      }
      // ---- online softmax, one query row per lane (both halves hold half of the keys)
      float mloc = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - mnew);
      float lsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - mnew);
        lsum += s[r];
      }
      lacc[0] = lacc[0] * alpha + lsum;
      m_run = mnew;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
      }
      }
      // ---- O^T += V^T P^T : A = V[key kappa(r,h)][channel lane], B = P^T register r
      const float* vrow = Vs + (buf * KT + sub * 32 + 4 * lh) * VS + (CP >= 32 ? li : (li & (CP - 1)));
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float vf = vrow[((r & 3) + 8 * (r >> 2)) * VS + ct * 32];
          o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], o[ct], 0, 0, 0);
        }
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: y = gamma * o / l + x ; z = act(scale*y + bias) ; nearest-upsample x alpha along T
  const float l_run = lacc[0] + lacc[1];
  attn_fwd_finish<CP, VEC>(p, o, m_run, l_run + __shfl_xor(l_run, 32, 64), b, bz, q0 + li, lh);
}

// ---- split planes (attn_bx.h) -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bx_split_kernel(const float* __restrict__ src, int cs,
                                                            const float* __restrict__ mul, int N, int C,
                                                            unsigned short* rows, unsigned short* cols, int n64) {
  __shared__ float vt[BX_KT][33];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / n64, tile = blockIdx.x - b * n64;
  const long brow = (long)b * N;
  const long plane = (long)n64 * BX_KT * 32;  // elements of one piece of one clip
  const float m = mul ? mul[0] : 1.0f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {  // coalesced rows into LDS
    const int f = tid + u * 256;
    const int row = f >> 3, c0 = (f & 7) * 4;
    const int j = tile * BX_KT + row;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (j < N && c0 < C) t = *reinterpret_cast<const f32x4*>(src + (brow + j) * cs + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) vt[row][c0 + e] = (c0 + e < C) ? t[e] * m : 0.f;
  }
  __syncthreads();
  if (rows) {  // thread = (row, 8 channels)
    const int row = tid >> 2, c0 = (tid & 3) * 8;
    u32x4 w[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(vt[row][c0 + 2 * e], vt[row][c0 + 2 * e + 1], w, e);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      *reinterpret_cast<u32x4*>(rows + ((long)b * 3 + pc) * plane + ((long)tile * BX_KT + row) * 32 + c0) = w[pc];
  }
  if (cols) {  // thread = (channel, 8 row positions)
    const int c = tid >> 3, pos0 = (tid & 7) * 8;
    u32x4 w[3];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      split_pair(vt[bx_key_of_pos(pos0 + 2 * e)][c], vt[bx_key_of_pos(pos0 + 2 * e + 1)][c], w, e);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      *reinterpret_cast<u32x4*>(cols + ((long)b * 3 + pc) * plane + ((long)tile * 32 + c) * BX_KT + pos0) = w[pc];
  }
}

// Packed planes (attn_bx.h), C <= 8.
__global__ __launch_bounds__(256) void attn_bx_split_packed_kernel(const float* __restrict__ src, int cs,
                                                                   const float* __restrict__ mul, int N, int C,
                                                                   unsigned short* rows, unsigned short* cols, int n64) {
  __shared__ float vt[BX_KT][9];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / n64, tile = blockIdx.x - b * n64;
  const long brow = (long)b * N;
  const float m = mul ? mul[0] : 1.0f;
  if (tid < 128) {  // coalesced rows into LDS: thread = (row, 4 channels)
    const int row = tid >> 1, c0 = (tid & 1) * 4;
    const int j = tile * BX_KT + row;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (j < N && c0 < C) t = *reinterpret_cast<const f32x4*>(src + (brow + j) * cs + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) vt[row][c0 + e] = (c0 + e < C) ? t[e] * m : 0.f;
  }
  __syncthreads();
  if (rows && tid < BX_KT) {  // thread = row: [p1 | p2 | p3 | 0]
    u32x4 w[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair(vt[tid][2 * e], vt[tid][2 * e + 1], w, e);
    unsigned short* o = rows + ((long)b * n64 * BX_KT + (long)tile * BX_KT + tid) * 32;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(o + 8 * pc) = w[pc];
    *reinterpret_cast<u32x4*>(o + 24) = (u32x4){0u, 0u, 0u, 0u};
  }
  if (cols) {  // thread = (packed channel, 8 row positions)
    const int pr = tid >> 3, pos0 = (tid & 7) * 8;
    u32x4 w[3];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      split_pair(vt[bx_key_of_pos(pos0 + 2 * e)][pr & 7], vt[bx_key_of_pos(pos0 + 2 * e + 1)][pr & 7], w, e);
    const int pc = pr >> 3;
    const u32x4 v = pc == 0 ? w[0] : (pc == 1 ? w[1] : (pc == 2 ? w[2] : (u32x4){0u, 0u, 0u, 0u}));
    *reinterpret_cast<u32x4*>(cols + (((long)b * n64 + tile) * 32 + pr) * BX_KT + pos0) = v;
  }
}

// The streaming kernel above on split operands (d = 32, 16-byte aligned rows).  Same mapping — S^T = K Q^T with the
// query on the lane, O^T = V^T P^T with P^T's accumulator registers as the B operand — on v_mfma_f32_32x32x16_bf16.
// K / V^T tiles come pre-split from the planes above (six 16-byte loads and LDS stores per thread and tile, no vector
// work); Q's pieces live in registers for the whole sweep; P's are made from the exp2 results — the only per-element
// split in the loop.  Stale-reference softmax as above (the tile accumulates on top of -m_ref).
//
// A bf16 MFMA is 32 cycles against 64 for the f32 one, and a wavefront's exp2 / split instructions only hide in the
// 24 issue cycles each MFMA leaves free if they stand BETWEEN MFMAs in program order.  So the loop is software
// pipelined over 32-key blocks: one straight-line step issues the first product of block g+1 (independent of
// everything else in the step) around the softmax of block g, then the second product of block g around the
// reference check of block g+1.  K is therefore needed one block earlier than V: three K buffers (tile t+2 is staged
// while t is swept), two V buffers, one barrier per 64-key tile.
// DBG: 1 = P^T is not split (its leading piece three times), 4 = one product instead of six (timing ablations,
// results invalid); 32 = the compiler's instruction order instead of the placed one (valid results, 4 % slower).
template <int DBG, class V>
__device__ __forceinline__ void split_pair_dbg(float a, float b, V (&dst)[3], int idx) {
  if constexpr (DBG != 0) {
    dst[0][idx] = dst[1][idx] = dst[2][idx] = cvt_pk_bf16(a, b);
  } else {
    split_pair(a, b, dst, idx);
  }
}
template <int DBG>
__device__ __forceinline__ f32x16 mfma_split_dbg(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
  if constexpr (DBG & 4) return mfma_bf(a[0], b[0], c);
  return mfma_split(a, b, c);
}

constexpr int BX_VP = BX_KT + 8;         // V^T plane row pitch in LDS (144 B)
constexpr int BX_KPL = BX_KT * BX_KP, BX_VPL = 32 * BX_VP;
constexpr int BX_LDS_BYTES = (3 * 3 * BX_KPL + 2 * 3 * BX_VPL) * 2;  // 72 KB: two workgroups per CU

template <int CP, int DBG = 0>
__global__ __launch_bounds__(256, 2) void attn_fwd_bx_kernel(const AttnArgs p, const unsigned short* kb,
                                                             const unsigned short* vb, int n64) {
  static_assert(CP == 32, "plane layout and staging are laid out for d = 32");
  constexpr int KT = BX_KT, KP = BX_KP, VP = BX_VP, KPL = BX_KPL, VPL = BX_VPL;
  constexpr int NC = CP / 16;             // k-steps of the first product
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* const Ks = smem;                 // [3 buffers][piece][key][KP]
  unsigned short* const Vs = smem + 3 * 3 * KPL;   // [2 buffers][piece][channel][VP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int bz = blockIdx.x / p.nqt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int q0 = (blockIdx.x - bz * p.nqt) * 128 + wave * 32;
  const int N = p.N;
  const long brow = (long)b * N;

  // ---- Q pieces (B operand of S^T = K Q^T): lane (i, h), k-step c holds Q[i][16c + 8h + j] * log2(e)
  u32x4 qf[NC][3];
  {
    const int qrow = q0 + li;
    const float* qp = p.q + (brow + (qrow < N ? qrow : 0)) * p.q_cs + 8 * lh;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
      if (qrow < N) {
        if (16 * c + 8 * lh < p.C) t0 = *reinterpret_cast<const f32x4*>(qp + 16 * c);
        if (16 * c + 8 * lh + 4 < p.C) t1 = *reinterpret_cast<const f32x4*>(qp + 16 * c + 4);
      }
      t0 *= LOG2E;
      t1 *= LOG2E;
      split_pair(t0[0], t0[1], qf[c], 0);
      split_pair(t0[2], t0[3], qf[c], 1);
      split_pair(t1[0], t1[1], qf[c], 2);
      split_pair(t1[2], t1[3], qf[c], 3);
    }
  }

  f32x16 o[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = 0.f;
  float m_run = NEG_BIG;
  f32x2 lacc = {0.f, 0.f};
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = -NEG_BIG;

  // ---- staging: a piece of a tile is 4 KB in either plane = one 16-byte element per thread
  // (buffer loads: per-thread byte offset in a vector register once, the tile's offset as the scalar operand — no
  // 64-bit address arithmetic per load; a clip's three planes are < 2^31 bytes)
  const long plane = (long)n64 * KT * 32;
  const __amdgpu_buffer_rsrc_t k_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(kb + (long)b * 3 * plane), 0, (int)(3 * plane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(vb + (long)b * 3 * plane), 0, (int)(3 * plane * 2), 0x00020000);
  const int ks_off = (tid >> 2) * KP + (tid & 3) * 8;
  const int vs_off = (tid >> 3) * VP + (tid & 7) * 8;
  u32x4 rk[3], rv[3];
  auto load_k = [&](int t) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      rk[pc] = __builtin_amdgcn_raw_buffer_load_b128(k_rs, tid * 16 + pc * (int)(plane * 2), t * (KT * 32 * 2), 0);
  };
  auto load_v = [&](int t) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      rv[pc] = __builtin_amdgcn_raw_buffer_load_b128(v_rs, tid * 16 + pc * (int)(plane * 2), t * (KT * 32 * 2), 0);
  };
  auto store_k = [&](int buf) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(Ks + (buf * 3 + pc) * KPL + ks_off) = rk[pc];
  };
  auto store_v = [&](int buf) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(Vs + (buf * 3 + pc) * VPL + vs_off) = rv[pc];
  };

  const int tz = (n64 + p.zs - 1) / p.zs;
  const int t0 = z * tz;
  const int nt = min(n64, t0 + tz) - t0;  // tiles of this key part (<= 0: an empty part)

  // first product of block (tile buffer kbuf, half sub): S^T on top of -m_ref
  f32x16 s_next;
  u32x4 kf[NC][3];
  auto qk = [&](int kbuf, int sub) {
    const unsigned short* krow = Ks + kbuf * 3 * KPL + (sub * 32 + li) * KP + 8 * lh;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) kf[c][pc] = *reinterpret_cast<const u32x4*>(krow + pc * KPL + 16 * c);
    s_next = negm;
#pragma unroll
    for (int c = 0; c < NC; ++c) s_next = mfma_split_dbg<DBG>(kf[c], qf[c], s_next);
  };
  // Slow path, a handful of times per sweep: the block exceeds the stale reference by more than 2^soft_t somewhere
  // (always on the first block, where -m_ref = +BIG), or it reaches past N (the planes are zero there, the scores must
  // be -BIG).  Recompute it from 0, mask, move the reference, rescale O and l.
  auto refresh = [&](int jbase) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) s = mfma_split_dbg<DBG>(kf[c], qf[c], s);
    if (jbase + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (jbase + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) s[r] = NEG_BIG;
    }
    float mloc = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - mnew);  // 0 on the first block
    lacc *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] *= alpha;
    m_run = mnew;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      negm[r] = -mnew;
      s_next[r] = s[r] - mnew;
    }
  };
  auto check = [&](int jbase) {
    float mx = fmaxf(fmaxf(s_next[0], s_next[1]), s_next[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s_next[r]), s_next[r + 1]);
    mx = fmaxf(mx, s_next[15]);
    if (__any(mx > p.soft_t) || jbase + 32 > N) refresh(jbase);
  };
  // one pipeline step: softmax + second product of the block in s_next, first product of the block after it
  auto step = [&](auto HAS_NEXT, int vbuf, int sub, int kbuf_n, int sub_n) {
    f32x16 s = s_next;
    if constexpr (decltype(HAS_NEXT)::value) qk(kbuf_n, sub_n);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      lacc[0] += s[r];
      lacc[1] += s[r + 1];
    }
    u32x4 pf[2][3];  // P^T pieces: registers 8m .. 8m+7 are the fragment of k-step m
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) split_pair_dbg<(DBG & 1)>(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], pf[m], e);
    const unsigned short* vrow = Vs + vbuf * 3 * VPL + li * VP + sub * 32 + 8 * lh;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      u32x4 vf[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) vf[pc] = *reinterpret_cast<const u32x4*>(vrow + pc * VPL + 16 * m);
      o[0] = mfma_split_dbg<DBG>(vf, pf[m], o[0]);
    }
  };
  // The same step with its vector work PLACED: one MFMA, then the fillers that fit the 24 issue cycles it leaves
  // (v_exp 8, the others ~4.3 each), pinned by sched_barrier — the compiler's own order front-loads the softmax and
  // leaves the second product's MFMAs bare.  Worth 4 % (3.61 -> 3.48 ms at N = 25 088, B = 8): with two wavefronts per
  // SIMD the issue port is shared, and under this load the part holds ~1.6 GHz (20 ns per 32x32x16 MFMA).
  float mx_next = 0.f;  // max of s_next over this lane's 16 keys, formed in the step's last MFMA gaps
  auto check_placed = [&](int jbase) {
    if (__any(mx_next > p.soft_t) || jbase + 32 > N) refresh(jbase);
  };
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0};  // DBG & 64: cycles per phase (s_memtime), printed by one wavefront
  auto step_placed = [&](int vbuf, int sub, int kbuf_n, int sub_n) {
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
    if constexpr (DBG & 64) { __builtin_amdgcn_sched_barrier(0); tq0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    constexpr int OA[6] = {0, 2, 1, 0, 1, 0}, OB[6] = {2, 0, 1, 1, 0, 0};  // mfma_split's order, small terms first
    f32x16 s = s_next;
    const unsigned short* krow = Ks + kbuf_n * 3 * KPL + (sub_n * 32 + li) * KP + 8 * lh;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) kf[c][pc] = *reinterpret_cast<const u32x4*>(krow + pc * KPL + 16 * c);
    const unsigned short* vrow = Vs + vbuf * 3 * VPL + li * VP + sub * 32 + 8 * lh;
    u32x4 vf[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) vf[m][pc] = *reinterpret_cast<const u32x4*>(vrow + pc * VPL + 16 * m);
    u32x4 pf[2][3];
    float xa[8], xb[8];
    auto E = [&](int r) { s[r] = __builtin_amdgcn_exp2f(s[r]); };
    auto SA = [&](int m, int e) {
      const float a = s[8 * m + 2 * e], b = s[8 * m + 2 * e + 1];
      const unsigned p1 = cvt_pk_bf16(a, b);
      pf[m][0][e] = p1;
      xa[4 * m + e] = bx_res_lo(a, p1);
      xb[4 * m + e] = bx_res_hi(b, p1);
    };
    auto SB = [&](int m, int e) {
      const unsigned p2 = cvt_pk_bf16(xa[4 * m + e], xb[4 * m + e]);
      pf[m][1][e] = p2;
      xa[4 * m + e] = bx_res_lo(xa[4 * m + e], p2);
      xb[4 * m + e] = bx_res_hi(xb[4 * m + e], p2);
      pf[m][2][e] = cvt_pk_bf16(xa[4 * m + e], xb[4 * m + e]);
    };
    auto QK = [&](int i) {  // MFMA i of the first product of the next block
      const int c = i / 6, t = i % 6;
      s_next = mfma_bf(kf[c][OA[t]], qf[c][OB[t]], i == 0 ? negm : s_next);
    };
    auto PV = [&](int i) {
      const int m = i / 6, t = i % 6;
      o[0] = mfma_bf(vf[m][OA[t]], pf[m][OB[t]], o[0]);
    };
#define SF_GAP() __builtin_amdgcn_sched_barrier(0)
    QK(0); E(0); E(1); E(2); SF_GAP();
    QK(1); E(3); E(4); E(5); SF_GAP();
    QK(2); E(6); E(7); E(8); SF_GAP();
    QK(3); SA(0, 0); SF_GAP();
    QK(4); SB(0, 0); SF_GAP();
    QK(5); SA(0, 1); SF_GAP();
    QK(6); SB(0, 1); SF_GAP();
    QK(7); SA(0, 2); SF_GAP();
    QK(8); SB(0, 2); SF_GAP();
    QK(9); SA(0, 3); SF_GAP();
    QK(10); SB(0, 3); SF_GAP();
    QK(11); E(9); E(10); E(11); SF_GAP();
    if constexpr (DBG & 64) { tq1 = __builtin_amdgcn_s_memtime(); SF_GAP(); }
    PV(0); E(12); E(13); E(14); SF_GAP();
    PV(1); E(15); SA(1, 0); SF_GAP();
    PV(2); SB(1, 0); SF_GAP();
    PV(3); SA(1, 1); SF_GAP();
    PV(4); SB(1, 1); SF_GAP();
    PV(5); SA(1, 2); SF_GAP();
    SB(1, 2); SA(1, 3); SB(1, 3); SF_GAP();
    if constexpr (DBG & 64) { tq2 = __builtin_amdgcn_s_memtime(); SF_GAP(); }
    PV(6); lacc[0] += s[0]; lacc[1] += s[1]; lacc[0] += s[2]; lacc[1] += s[3]; lacc[0] += s[4]; SF_GAP();
    PV(7); lacc[1] += s[5]; lacc[0] += s[6]; lacc[1] += s[7]; lacc[0] += s[8]; lacc[1] += s[9]; SF_GAP();
    PV(8); lacc[0] += s[10]; lacc[1] += s[11]; lacc[0] += s[12]; lacc[1] += s[13]; lacc[0] += s[14]; SF_GAP();
    PV(9); lacc[1] += s[15]; mx_next = fmaxf(fmaxf(s_next[0], s_next[1]), s_next[2]);
    mx_next = fmaxf(fmaxf(mx_next, s_next[3]), s_next[4]); mx_next = fmaxf(fmaxf(mx_next, s_next[5]), s_next[6]); SF_GAP();
    PV(10); mx_next = fmaxf(fmaxf(mx_next, s_next[7]), s_next[8]); mx_next = fmaxf(fmaxf(mx_next, s_next[9]), s_next[10]);
    mx_next = fmaxf(fmaxf(mx_next, s_next[11]), s_next[12]); SF_GAP();
    PV(11); mx_next = fmaxf(fmaxf(mx_next, s_next[13]), s_next[14]); mx_next = fmaxf(mx_next, s_next[15]); SF_GAP();
    if constexpr (DBG & 64) {
      tq3 = __builtin_amdgcn_s_memtime();
      SF_GAP();
      st_acc[0] += tq1 - tq0;
      st_acc[1] += tq2 - tq1;
      st_acc[2] += tq3 - tq2;
      st_acc[3] += 1;
    }
#undef SF_GAP
  };
  using T = std::true_type;
  using F = std::false_type;

  unsigned long long tl0 = 0, st_tile[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (DBG & 64) tl0 = __builtin_amdgcn_s_memtime();
  if (nt > 0) {
    load_k(t0);
    load_v(t0);
    store_k(0);
    store_v(0);
    load_k(min(t0 + 1, n64 - 1));
    store_k(1);
    __syncthreads();
    qk(0, 0);
    check(t0 * KT);
    int kcur = 0;  // K buffer of the tile being swept
    for (int r = 0; r < nt; ++r) {
      const int t = t0 + r;
      const int knext = kcur == 2 ? 0 : kcur + 1, kafter = knext == 2 ? 0 : knext + 1;
      unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0;
      if constexpr (DBG & 64) u0 = __builtin_amdgcn_s_memtime();
      load_k(min(t + 2, n64 - 1));  // past the part's end: a tile nobody reads
      load_v(min(t + 1, n64 - 1));
      if constexpr (DBG & 64) { __builtin_amdgcn_sched_barrier(0); u1 = __builtin_amdgcn_s_memtime(); }
      if constexpr (DBG & 32) step(T{}, r & 1, 0, kcur, 1); else step_placed(r & 1, 0, kcur, 1);
      if constexpr (DBG & 64) { __builtin_amdgcn_sched_barrier(0); u2 = __builtin_amdgcn_s_memtime(); }
      if constexpr (DBG & 32) check(t * KT + 32); else check_placed(t * KT + 32);
      if constexpr (DBG & 64) { __builtin_amdgcn_sched_barrier(0); u3 = __builtin_amdgcn_s_memtime(); }
      if (r + 1 < nt) {
        if constexpr (DBG & 32) step(T{}, r & 1, 1, knext, 0); else step_placed(r & 1, 1, knext, 0);
        if constexpr (DBG & 32) check(t * KT + 64); else check_placed(t * KT + 64);
      } else {
        step(F{}, r & 1, 1, 0, 0);
      }
      if constexpr (DBG & 64) { __builtin_amdgcn_sched_barrier(0); u4 = __builtin_amdgcn_s_memtime(); }
      store_k(kafter);
      store_v((r + 1) & 1);
      __syncthreads();
      if constexpr (DBG & 64) {
        u5 = __builtin_amdgcn_s_memtime();
        st_tile[0] += u1 - u0;  // loads issue
        st_tile[1] += u2 - u1;  // step a
        st_tile[2] += u3 - u2;  // check a
        st_tile[3] += u4 - u3;  // step b + check b
        st_tile[4] += u5 - u4;  // stores + barrier
        st_tile[5] += 1;
      }
      kcur = knext;
    }
  }

  if constexpr (DBG & 64) {
    st_acc[4] = __builtin_amdgcn_s_memtime() - tl0;
    if (blockIdx.x == 7 && tid == 0)
      printf("fwd_bx stamps (wave 0 of workgroup 7): steps %llu  per step: qk-phase %llu  pv0-phase %llu  pv1-phase %llu  | sweep total %llu = %llu per step\n",
             st_acc[3], st_acc[0] / st_acc[3], st_acc[1] / st_acc[3], st_acc[2] / st_acc[3], st_acc[4], st_acc[4] / st_acc[3]);
    if (blockIdx.x == 7 && tid == 0)
      printf("fwd_bx per tile: loads %llu  step a %llu  check a %llu  step b + check %llu  stores + barrier %llu\n",
             st_tile[0] / st_tile[5], st_tile[1] / st_tile[5], st_tile[2] / st_tile[5], st_tile[3] / st_tile[5],
             st_tile[4] / st_tile[5]);
  }
  const float l_run = lacc[0] + lacc[1];
  attn_fwd_finish<CP, 4>(p, o, m_run, l_run + __shfl_xor(l_run, 32, 64), b, bz, q0 + li, lh);
}

// ---- C <= 8 on packed planes (attn_bx.h) ----------------------------------------------------------------------------
// The kernel above with one plane per operand: S^T = K Q^T is three MFMAs ([k1|k1][q1|q2], [k2|k2][q1|q2],
// [k1|k3][q3|q1]), O^T = V^T P^T three per k-step (the packed V^T rows times P^T's pieces) with
// O[c] = rows c + (8 + c) + (16 + c) at the end.  9 MFMAs per 32-key block instead of 24; the exp2 / split work per
// element is unchanged, so this form is bound by the vector pipe — which the f32 MFMAs of attn_small.hip share with
// it and these do not.
constexpr int BXP_LDS_BYTES = (3 * BX_KPL + 2 * BX_VPL) * 2;
__global__ __launch_bounds__(256, 2) void attn_fwd_bxp_kernel(const AttnArgs p, const unsigned short* kb,
                                                              const unsigned short* vb, int n64) {
  constexpr int KT = BX_KT, KP = BX_KP, VP = BX_VP, KPL = BX_KPL, VPL = BX_VPL;
  __shared__ __attribute__((aligned(16))) unsigned short smem[3 * KPL + 2 * VPL];
  unsigned short* const Ks = smem;             // [3 buffers][key][KP]
  unsigned short* const Vs = smem + 3 * KPL;   // [2 buffers][packed channel][VP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int bz = blockIdx.x / p.nqt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int q0 = (blockIdx.x - bz * p.nqt) * 128 + wave * 32;
  const int N = p.N;
  const long brow = (long)b * N;

  // ---- Q as B operands [q1|q2] and [q3|q1] (lane half h holds the left / right piece of its query's 8 channels)
  u32x4 qa, qc;
  {
    const int qrow = q0 + li;
    const float* qp = p.q + (brow + (qrow < N ? qrow : 0)) * p.q_cs;
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (qrow < N) {
      t0 = *reinterpret_cast<const f32x4*>(qp);
      if (p.C > 4) t1 = *reinterpret_cast<const f32x4*>(qp + 4);
    }
    t0 *= LOG2E;
    t1 *= LOG2E;
    u32x4 w[3];
    split_pair(t0[0], t0[1], w, 0);
    split_pair(t0[2], t0[3], w, 1);
    split_pair(t1[0], t1[1], w, 2);
    split_pair(t1[2], t1[3], w, 3);
    qa = lh ? w[1] : w[0];
    qc = lh ? w[0] : w[2];
  }

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = NEG_BIG;
  f32x2 lacc = {0.f, 0.f};
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = -NEG_BIG;

  // ---- staging: a tile is 4 KB in either plane = one 16-byte element per thread
  const unsigned short* kg = kb + (long)b * n64 * (KT * 32) + tid * 8;
  const unsigned short* vg = vb + (long)b * n64 * (KT * 32) + tid * 8;
  const int ks_off = (tid >> 2) * KP + (tid & 3) * 8;
  const int vs_off = (tid >> 3) * VP + (tid & 7) * 8;
  u32x4 rk, rv;
  auto load_k = [&](int t) { rk = *reinterpret_cast<const u32x4*>(kg + (long)t * (KT * 32)); };
  auto load_v = [&](int t) { rv = *reinterpret_cast<const u32x4*>(vg + (long)t * (KT * 32)); };
  auto store_k = [&](int buf) { *reinterpret_cast<u32x4*>(Ks + buf * KPL + ks_off) = rk; };
  auto store_v = [&](int buf) { *reinterpret_cast<u32x4*>(Vs + buf * VPL + vs_off) = rv; };

  const int tz = (n64 + p.zs - 1) / p.zs;
  const int t0 = z * tz;
  const int nt = min(n64, t0 + tz) - t0;

  f32x16 s_next;
  u32x4 kf[3];  // [k1|k1], [k2|k2], [k1|k3] of the block in s_next
  auto qk3 = [&](f32x16 c) {
    c = mfma_bf(kf[2], qc, c);   // k1 q3 + k3 q1
    c = mfma_bf(kf[1], qa, c);   // k2 q1 + k2 q2
    return mfma_bf(kf[0], qa, c);  // k1 q1 + k1 q2
  };
  auto qk = [&](int kbuf, int sub) {
    const unsigned short* krow = Ks + kbuf * KPL + (sub * 32 + li) * KP;
    kf[0] = *reinterpret_cast<const u32x4*>(krow);
    kf[1] = *reinterpret_cast<const u32x4*>(krow + 8);
    kf[2] = *reinterpret_cast<const u32x4*>(krow + (lh ? 16 : 0));
    s_next = qk3(negm);
  };
  auto refresh = [&](int jbase) {  // see attn_fwd_bx_kernel
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    s = qk3(s);
    if (jbase + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (jbase + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) s[r] = NEG_BIG;
    }
    float mloc = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - mnew);
    lacc *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
    m_run = mnew;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      negm[r] = -mnew;
      s_next[r] = s[r] - mnew;
    }
  };
  auto check = [&](int jbase) {
    float mx = fmaxf(fmaxf(s_next[0], s_next[1]), s_next[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s_next[r]), s_next[r + 1]);
    mx = fmaxf(mx, s_next[15]);
    if (__any(mx > p.soft_t) || jbase + 32 > N) refresh(jbase);
  };
  auto step = [&](auto HAS_NEXT, int vbuf, int sub, int kbuf_n, int sub_n) {
    f32x16 s = s_next;
    if constexpr (decltype(HAS_NEXT)::value) qk(kbuf_n, sub_n);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      lacc[0] += s[r];
      lacc[1] += s[r + 1];
    }
    u32x4 pf[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) split_pair(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], pf[m], e);
    const unsigned short* vrow = Vs + vbuf * VPL + li * VP + sub * 32 + 8 * lh;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const u32x4 vf = *reinterpret_cast<const u32x4*>(vrow + 16 * m);
      o = mfma_bf(vf, pf[m][2], o);
      o = mfma_bf(vf, pf[m][1], o);
      o = mfma_bf(vf, pf[m][0], o);
    }
  };
  using T = std::true_type;
  using F = std::false_type;

  if (nt > 0) {
    load_k(t0);
    load_v(t0);
    store_k(0);
    store_v(0);
    load_k(min(t0 + 1, n64 - 1));
    store_k(1);
    __syncthreads();
    qk(0, 0);
    check(t0 * KT);
    int kcur = 0;
    for (int r = 0; r < nt; ++r) {
      const int t = t0 + r;
      const int knext = kcur == 2 ? 0 : kcur + 1, kafter = knext == 2 ? 0 : knext + 1;
      load_k(min(t + 2, n64 - 1));
      load_v(min(t + 1, n64 - 1));
      step(T{}, r & 1, 0, kcur, 1);
      check(t * KT + 32);
      if (r + 1 < nt) {
        step(T{}, r & 1, 1, knext, 0);
        check(t * KT + 64);
      } else {
        step(F{}, r & 1, 1, 0, 0);
      }
      store_k(kafter);
      store_v((r + 1) & 1);
      __syncthreads();
      kcur = knext;
    }
  }

  // O[c] for c = (r & 3) + 4h: packed row groups 0-7 / 8-15 / 16-23 are registers r, r + 4, r + 8
  f32x16 of[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) of[0][r] = r < 4 ? o[r] + o[r + 4] + o[r + 8] : 0.f;
  const float l_run = lacc[0] + lacc[1];
  attn_fwd_finish<8, 4>(p, of, m_run, l_run + __shfl_xor(l_run, 32, 64), b, bz, q0 + li, lh);
}

// ---- 33 <= C <= 64: two 32-channel blocks -----------------------------------------------------------------------------
// attn_fwd_bx_kernel with the channels as two 32-wide blocks, each with its own K / V^T plane set (the split launches
// run once per block), Q pieces and O^T accumulator tile; S^T sums over both blocks' k-steps.  48 MFMAs per 32-key
// block.  The operands take ~290 registers and the planes 147 KB of LDS: one workgroup per CU, one wavefront per SIMD;
// the same software pipeline over 32-key blocks, in the compiler's instruction order.
constexpr int BX2_LDS_BYTES = 2 * (3 * 3 * BX_KPL + 2 * 3 * BX_VPL) * 2;
__global__ __launch_bounds__(256, 1) void attn_fwd_bx2_kernel(const AttnArgs p, const unsigned short* kb,
                                                              const unsigned short* vb, int n64, long blk_elems) {
  constexpr int KT = BX_KT, KP = BX_KP, VP = BX_VP, KPL = BX_KPL, VPL = BX_VPL;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* const Ks = smem;                     // [3 buffers][block][piece][key][KP]
  unsigned short* const Vs = smem + 3 * 2 * 3 * KPL;   // [2 buffers][block][piece][channel][VP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int bz = blockIdx.x / p.nqt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int q0 = (blockIdx.x - bz * p.nqt) * 128 + wave * 32;
  const int N = p.N;
  const long brow = (long)b * N;

  u32x4 qf[2][2][3];  // [block][k-step][piece]
  {
    const int qrow = q0 + li;
    const float* qp = p.q + (brow + (qrow < N ? qrow : 0)) * p.q_cs + 8 * lh;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ch = 32 * blk + 16 * c + 8 * lh;
        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
        if (qrow < N) {
          if (ch < p.C) t0 = *reinterpret_cast<const f32x4*>(qp + 32 * blk + 16 * c);
          if (ch + 4 < p.C) t1 = *reinterpret_cast<const f32x4*>(qp + 32 * blk + 16 * c + 4);
        }
        t0 *= LOG2E;
        t1 *= LOG2E;
        split_pair(t0[0], t0[1], qf[blk][c], 0);
        split_pair(t0[2], t0[3], qf[blk][c], 1);
        split_pair(t1[0], t1[1], qf[blk][c], 2);
        split_pair(t1[2], t1[3], qf[blk][c], 3);
      }
  }

  f32x16 o[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[blk][r] = 0.f;
  float m_run = NEG_BIG;
  f32x2 lacc = {0.f, 0.f};
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = -NEG_BIG;

  const long plane = (long)n64 * KT * 32;
  const unsigned short* kg = kb + (long)b * 3 * plane + tid * 8;
  const unsigned short* vg = vb + (long)b * 3 * plane + tid * 8;
  const int ks_off = (tid >> 2) * KP + (tid & 3) * 8;
  const int vs_off = (tid >> 3) * VP + (tid & 7) * 8;
  u32x4 rk[2][3], rv[2][3];
  auto load_k = [&](int t) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        rk[blk][pc] = *reinterpret_cast<const u32x4*>(kg + blk * blk_elems + pc * plane + (long)t * (KT * 32));
  };
  auto load_v = [&](int t) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        rv[blk][pc] = *reinterpret_cast<const u32x4*>(vg + blk * blk_elems + pc * plane + (long)t * (KT * 32));
  };
  auto store_k = [&](int buf) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(Ks + ((buf * 2 + blk) * 3 + pc) * KPL + ks_off) = rk[blk][pc];
  };
  auto store_v = [&](int buf) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(Vs + ((buf * 2 + blk) * 3 + pc) * VPL + vs_off) = rv[blk][pc];
  };

  const int tz = (n64 + p.zs - 1) / p.zs;
  const int t0 = z * tz;
  const int nt = min(n64, t0 + tz) - t0;

  f32x16 s_next;
  u32x4 kf[2][2][3];
  auto qk_all = [&](f32x16 c) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) c = mfma_split(kf[blk][cc], qf[blk][cc], c);
    return c;
  };
  auto qk = [&](int kbuf, int sub) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const unsigned short* krow = Ks + (kbuf * 2 + blk) * 3 * KPL + (sub * 32 + li) * KP + 8 * lh;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) kf[blk][cc][pc] = *reinterpret_cast<const u32x4*>(krow + pc * KPL + 16 * cc);
    }
    s_next = qk_all(negm);
  };
  auto refresh = [&](int jbase) {  // see attn_fwd_bx_kernel
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    s = qk_all(s);
    if (jbase + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (jbase + (r & 3) + 8 * (r >> 2) + 4 * lh >= N) s[r] = NEG_BIG;
    }
    float mloc = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - mnew);
    lacc *= alpha;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[blk][r] *= alpha;
    m_run = mnew;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      negm[r] = -mnew;
      s_next[r] = s[r] - mnew;
    }
  };
  auto check = [&](int jbase) {
    float mx = fmaxf(fmaxf(s_next[0], s_next[1]), s_next[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s_next[r]), s_next[r + 1]);
    mx = fmaxf(mx, s_next[15]);
    if (__any(mx > p.soft_t) || jbase + 32 > N) refresh(jbase);
  };
  auto step = [&](auto HAS_NEXT, int vbuf, int sub, int kbuf_n, int sub_n) {
    f32x16 s = s_next;
    if constexpr (decltype(HAS_NEXT)::value) qk(kbuf_n, sub_n);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      lacc[0] += s[r];
      lacc[1] += s[r + 1];
    }
    u32x4 pf[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) split_pair(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], pf[m], e);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const unsigned short* vrow = Vs + (vbuf * 2 + blk) * 3 * VPL + li * VP + sub * 32 + 8 * lh;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        u32x4 vf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) vf[pc] = *reinterpret_cast<const u32x4*>(vrow + pc * VPL + 16 * m);
        o[blk] = mfma_split(vf, pf[m], o[blk]);
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;

  if (nt > 0) {
    load_k(t0);
    load_v(t0);
    store_k(0);
    store_v(0);
    load_k(min(t0 + 1, n64 - 1));
    store_k(1);
    __syncthreads();
    qk(0, 0);
    check(t0 * KT);
    int kcur = 0;
    for (int r = 0; r < nt; ++r) {
      const int t = t0 + r;
      const int knext = kcur == 2 ? 0 : kcur + 1, kafter = knext == 2 ? 0 : knext + 1;
      load_k(min(t + 2, n64 - 1));
      load_v(min(t + 1, n64 - 1));
      step(T{}, r & 1, 0, kcur, 1);
      check(t * KT + 32);
      if (r + 1 < nt) {
        step(T{}, r & 1, 1, knext, 0);
        check(t * KT + 64);
      } else {
        step(F{}, r & 1, 1, 0, 0);
      }
      store_k(kafter);
      store_v((r + 1) & 1);
      __syncthreads();
      kcur = knext;
    }
  }
  const float l_run = lacc[0] + lacc[1];
  attn_fwd_finish<64, 4>(p, o, m_run, l_run + __shfl_xor(l_run, 32, 64), b, bz, q0 + li, lh);
}

// One thread per (query row, 4 channels): O = sum_z o_z 2^(m_z - m) / sum_z l_z 2^(m_z - m), m = max_z m_z, then the
// same epilogue as above.  An empty part carries (m, l, o) = (-BIG, 0, 0) and drops out with weight 0.
__global__ __launch_bounds__(256) void attn_fwd_merge_kernel(const AttnArgs p, int cp) {
  const int f4 = cp >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.B * p.N * f4;
  if (idx >= total) return;
  const int c0 = (int)(idx % f4) * 4;
  const long row = idx / f4;  // b*N + i
  const int N = p.N, C = p.C, zs = p.zs;
  const int b = (int)(row / N), qrow = (int)(row - (long)b * N);
  if (c0 >= C) return;
  const float* ml = p.part_ml + ((long)b * zs * N + qrow) * 2;
  float m = NEG_BIG;
  for (int z = 0; z < zs; ++z) m = fmaxf(m, ml[(long)z * N * 2]);
  float l_tot = 0.f;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  const float* po = p.part_o + ((long)b * zs * N + qrow) * cp + c0;
  for (int z = 0; z < zs; ++z) {
    const float w = __builtin_amdgcn_exp2f(ml[(long)z * N * 2] - m);
    l_tot += ml[(long)z * N * 2 + 1] * w;
    o += *reinterpret_cast<const f32x4*>(po + (long)z * N * cp) * w;
  }
  const float inv_l = 1.0f / l_tot;
  const float gamma = p.gamma ? p.gamma[0] : 1.0f;
  const int HW = p.H * p.W;
  const int tq = qrow / HW;
  const int hw = qrow - tq * HW;
  const float* xp = p.x + row * p.x_cs;
  const long orow0 = ((long)b * p.T * p.alpha + (long)tq * p.alpha) * HW + hw;
  if (p.lse_save && c0 == 0) p.lse_save[row] = m + __log2f(l_tot);
  float y[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    const bool okc = c < C;
    if (p.o_save && okc) p.o_save[row * C + c] = o[e] * inv_l;
    float v = gamma * (o[e] * inv_l) + (okc ? xp[c] : 0.f);
    if (p.scale && okc) v = v * p.scale[c] + p.bias[c];
    if (p.act == SF_ACT_RELU) v = fmaxf(v, 0.f);
    y[e] = v;
  }
  for (int r = 0; r < p.alpha; ++r) {
    float* op = p.out + (orow0 + (long)r * HW) * p.out_cs + p.out_coff + c0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((c0 + e) < C) op[e] = y[e];
  }
}

template <int CP>
int launch(AttnArgs a, bool vec4, hipStream_t s) {
  const int grid = a.B * a.zs * a.nqt;
  // SF_ATTN_STALE=0: the per-tile online softmax (running maximum refreshed on every tile); SF_ATTN_SOFT_T: how far
  // (log2) a tile may exceed the stale reference before it is refreshed
  static const bool stale_on = [] { const char* e = getenv("SF_ATTN_STALE"); return !(e && e[0] == '0'); }();
  a.soft_t = sf_attn_soft_t();
  const bool stale = stale_on && CP <= 64;  // d = 128: the 16 extra registers cost the second wavefront per SIMD
  if constexpr (CP == 64) {
    static const bool bx64 = [] { const char* e = getenv("SF_ATTN_BX64"); return !(e && e[0] == '0'); }();
    if (vec4 && a.bx_planes && bx64) {
      const int n64 = sf_cdiv(a.N, BX_KT);
      const long blk_elems = sf_attn_bx_plane_elems(a.B, a.N);
      unsigned short* kb = reinterpret_cast<unsigned short*>(a.bx_planes);
      unsigned short* vb = kb + 2 * blk_elems;
      for (int blk = 0; blk < 2; ++blk) {
        const int cb = a.C - 32 * blk > 32 ? 32 : a.C - 32 * blk;
        if (sf_attn_bx_split(a.k + 32 * blk, a.k_cs, nullptr, a.B, a.N, cb, kb + blk * blk_elems, nullptr, s) != SF_OK ||
            sf_attn_bx_split(a.v + 32 * blk, a.v_cs, nullptr, a.B, a.N, cb, nullptr, vb + blk * blk_elems, s) != SF_OK)
          return SF_ELAUNCH;
      }
      static SfLdsAttr lds_attr2;
      if (!sf_ensure_dyn_lds(lds_attr2, reinterpret_cast<const void*>(attn_fwd_bx2_kernel), BX2_LDS_BYTES)) return SF_ELAUNCH;
      hipLaunchKernelGGL(attn_fwd_bx2_kernel, dim3(grid), dim3(256), BX2_LDS_BYTES, s, a, kb, vb, n64, blk_elems);
      SF_CHECK_LAUNCH();
      return a.zs > 1 ? sf_attn_fwd_merge(a, CP, s) : SF_OK;
    }
  }
  if constexpr (CP == 32) {
    if (vec4 && a.bx_planes) {
      const int n64 = sf_cdiv(a.N, BX_KT);
      unsigned short* kb = reinterpret_cast<unsigned short*>(a.bx_planes);
      unsigned short* vb = kb + (long)a.B * 3 * n64 * BX_KT * 32;
      if (sf_attn_bx_split(a.k, a.k_cs, nullptr, a.B, a.N, a.C, kb, nullptr, s) != SF_OK ||
          sf_attn_bx_split(a.v, a.v_cs, nullptr, a.B, a.N, a.C, nullptr, vb, s) != SF_OK)
        return SF_ELAUNCH;
      using Kern = void (*)(const AttnArgs, const unsigned short*, const unsigned short*, int);
      static const Kern kern = [] {  // SF_ATTN_BX_DBG: timing ablations (see the kernel)
        const char* e = getenv("SF_ATTN_BX_DBG");
        const int dbg = e ? atoi(e) : 0;
        return dbg == 1 ? (Kern)attn_fwd_bx_kernel<32, 1> : dbg == 32 ? (Kern)attn_fwd_bx_kernel<32, 32> : dbg == 64 ? (Kern)attn_fwd_bx_kernel<32, 64> : dbg == 4 ? (Kern)attn_fwd_bx_kernel<32, 4>
                                                                      : (Kern)attn_fwd_bx_kernel<32, 0>;
      }();
      static const int pad = [] { const char* e = getenv("SF_ATTN_BX_PADLDS"); return e ? atoi(e) : 0; }();  // occupancy probe
      static SfLdsAttr lds_attr;  // 72 KB of dynamic LDS
      if (!sf_ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(kern), BX_LDS_BYTES + pad)) return SF_ELAUNCH;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), BX_LDS_BYTES + pad, s, a, kb, vb, n64);
      SF_CHECK_LAUNCH();
      return a.zs > 1 ? sf_attn_fwd_merge(a, CP, s) : SF_OK;
    }
  }
  if (vec4) {
    if (stale) hipLaunchKernelGGL((attn_fwd_kernel<CP, 4, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<CP, 4, false>), dim3(grid), dim3(256), 0, s, a);
  } else {
    if (stale) hipLaunchKernelGGL((attn_fwd_kernel<CP, 1, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<CP, 1, false>), dim3(grid), dim3(256), 0, s, a);
  }
  SF_CHECK_LAUNCH();
  return a.zs > 1 ? sf_attn_fwd_merge(a, CP, s) : SF_OK;
}

}  // namespace

int sf_attn_bx_split(const float* src, int cs, const float* mul, int B, int N, int C, unsigned short* rows,
                     unsigned short* cols, hipStream_t s) {
  if (!src || C > 32 || (cs & 3) || !sf_aligned16(src)) return SF_EINVAL;
  const int n64 = sf_cdiv(N, BX_KT);
  hipLaunchKernelGGL(attn_bx_split_kernel, dim3(B * n64), dim3(256), 0, s, src, cs, mul, N, C, rows, cols, n64);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

int sf_attn_bx_split_packed(const float* src, int cs, const float* mul, int B, int N, int C, unsigned short* rows,
                            unsigned short* cols, hipStream_t s) {
  if (!src || C > 8 || (cs & 3) || (C & 3) || !sf_aligned16(src)) return SF_EINVAL;
  const int n64 = sf_cdiv(N, BX_KT);
  hipLaunchKernelGGL(attn_bx_split_packed_kernel, dim3(B * n64), dim3(256), 0, s, src, cs, mul, N, C, rows, cols, n64);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

int sf_attn_fwd_merge(const SfAttnArgs& a, int cp, hipStream_t s) {
  const long total = (long)a.B * a.N * (cp / 4);
  hipLaunchKernelGGL(attn_fwd_merge_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, a, cp);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

int sf_attn_small_dispatch(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                           const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                           int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                           int alpha, float* o_save, float* lse_save, bool vec4, float* ws,
                           hipStream_t stream);  // attn_small.hip

static int attn_fwd_impl(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                         const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                         int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                         int alpha, float* o_save, float* lse_save, float* ws, void* stream) {
  if (!q || !k || !v || !x || !out) return SF_EINVAL;
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 128 || alpha <= 0) return SF_EINVAL;
  if ((scale == nullptr) != (bias == nullptr)) return SF_EINVAL;
  if (act != SF_ACT_NONE && act != SF_ACT_RELU) return SF_EINVAL;
  if (ws && !sf_aligned16(ws)) return SF_EINVAL;
  const long N = (long)T * H * W;
  if (N * B > 0x7fffffffL) return SF_EINVAL;
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.x = x; a.gamma = gamma; a.scale = scale; a.bias = bias; a.out = out;
  a.o_save = o_save; a.lse_save = lse_save;
  a.q_cs = q_cs; a.k_cs = k_cs; a.v_cs = v_cs; a.x_cs = x_cs; a.out_cs = out_cs; a.out_coff = out_coff;
  a.B = B; a.T = T; a.H = H; a.W = W; a.C = C; a.N = (int)N; a.alpha = alpha; a.act = act;
  a.nqt = sf_cdiv(N, 128);
  a.zs = 1; a.part_o = nullptr; a.part_ml = nullptr;
  const bool vec4 = (C % 4 == 0) && (q_cs % 4 == 0) && (k_cs % 4 == 0) && (v_cs % 4 == 0) && (x_cs % 4 == 0) &&
                    (out_cs % 4 == 0) && (out_coff % 4 == 0) && sf_aligned16(q) && sf_aligned16(k) &&
                    sf_aligned16(v) && sf_aligned16(x) && sf_aligned16(out);
  hipStream_t s = (hipStream_t)stream;
  // C <= 16: 16-query wavefronts on 16x16x4 tiles (no padded rows in the second product).  The same kernel also
  // instantiates for 16 < C <= 32 (SF_ATTN_FWD32=small) — a probe of whether the 16x16x4 shape, which sustains a
  // higher clock than 32x32x2 in a bare MFMA loop on this part, pays at d = 32: it does not in this form (one
  // ds_read_b32 per MFMA for K^T and V, twice the softmax rows per FLOP): 5.6 -> 6.1 ms at N = 25 088, eval forward
  // 19.31 -> 19.78 ms.  The 32x32x2 kernel stays the default.
  static const bool small32 = [] {
    const char* e = getenv("SF_ATTN_FWD32");
    return e && e[0] == 's';
  }();
  if (C > 4 && C <= 8 && C % 4 == 0 && vec4 && ws && sf_attn_bx_level() >= 1) {  // packed planes on the bf16 pipe
    a.nqt = sf_cdiv(N, 128);
    a.zs = sf_sweep_parts((long)B * a.nqt, sf_cdiv(N, 64));
    sf_attn_place_parts(a, 8, ws);
    a.soft_t = sf_attn_soft_t();
    unsigned short* kb = reinterpret_cast<unsigned short*>(ws + (long)B * SF_SWEEP_PARTS_MAX * N * (8 + 2));
    unsigned short* vb = kb + sf_attn_bx_packed_elems(B, (int)N);
    if (sf_attn_bx_split_packed(k, k_cs, nullptr, B, (int)N, C, kb, nullptr, s) != SF_OK ||
        sf_attn_bx_split_packed(v, v_cs, nullptr, B, (int)N, C, nullptr, vb, s) != SF_OK)
      return SF_ELAUNCH;
    hipLaunchKernelGGL(attn_fwd_bxp_kernel, dim3(B * a.zs * a.nqt), dim3(256), 0, s, a, kb, vb, sf_cdiv(N, BX_KT));
    SF_CHECK_LAUNCH();
    return a.zs > 1 ? sf_attn_fwd_merge(a, 8, s) : SF_OK;
  }
  if (C <= 16 || (C <= 32 && small32))
    return sf_attn_small_dispatch(q, q_cs, k, k_cs, v, v_cs, x, x_cs, gamma, scale, bias, act, out, out_cs,
                                  out_coff, B, T, H, W, C, alpha, o_save, lse_save, vec4, ws, s);
  const int cp = C <= 32 ? 32 : (C <= 64 ? 64 : 128);
  a.bx_planes = nullptr;
  if (ws) {
    a.zs = sf_sweep_parts((long)B * a.nqt, sf_cdiv(N, cp >= 128 ? 32 : 64));
    sf_attn_place_parts(a, cp, ws);
    if ((cp == 32 || (cp == 64 && C > 32)) && sf_attn_bx_level() >= 1)  // the split K / V^T planes live behind the part buffers
      a.bx_planes = ws + (long)B * SF_SWEEP_PARTS_MAX * N * (cp + 2);
  }
  if (C <= 32) return launch<32>(a, vec4, s);
  if (C <= 64) return launch<64>(a, vec4, s);
  return launch<128>(a, vec4, s);
}

extern "C" int sf_attn_fwd(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                           const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                           int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                           int alpha, float* o_save, float* lse_save, void* stream) {
  return attn_fwd_impl(q, q_cs, k, k_cs, v, v_cs, x, x_cs, gamma, scale, bias, act, out, out_cs, out_coff, B, T, H,
                       W, C, alpha, o_save, lse_save, nullptr, stream);
}

extern "C" int sf_attn_products_per_fp32(int C) {
  if (sf_attn_bx_level() < 1) return 0;
  static const bool bx64 = [] { const char* e = getenv("SF_ATTN_BX64"); return !(e && e[0] == '0'); }();
  return ((C > 16 && C <= 32) || (C > 32 && C <= 64 && bx64) || (C > 4 && C <= 8 && C % 4 == 0)) ? 6 : 0;
}

// Room for the (O^T, m, l) of up to SF_SWEEP_PARTS_MAX key parts per query row.
extern "C" long sf_attn_fwd_ws_floats(int B, int N, int C) {
  if (B <= 0 || N <= 0 || C <= 0 || C > 128) return 0;
  const int cp = C <= 4 ? 4 : (C <= 8 ? 8 : (C <= 16 ? 16 : (C <= 32 ? 32 : (C <= 64 ? 64 : 128))));
  long n = (long)B * SF_SWEEP_PARTS_MAX * N * (cp + 2);
  if (cp == 32) n += (long)B * sf_cdiv(N, 64) * 64 * 32 * 3;  // K and V^T as three bf16 planes each (12 B per element)
  if (cp == 8) n += (long)B * sf_cdiv(N, 64) * 64 * 32;       // K and V^T as one packed plane each (2 x 64 B per row)
  if (cp == 64) n += (long)B * sf_cdiv(N, 64) * 64 * 32 * 6;  // two 32-channel blocks of three-plane K and V^T sets
  return n;
}

extern "C" int sf_attn_fwd_ws(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                              const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                              int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                              int alpha, float* o_save, float* lse_save, float* ws, void* stream) {
  if (!ws) return SF_EINVAL;
  return attn_fwd_impl(q, q_cs, k, k_cs, v, v_cs, x, x_cs, gamma, scale, bias, act, out, out_cs, out_coff, B, T, H,
                       W, C, alpha, o_save, lse_save, ws, stream);
}
