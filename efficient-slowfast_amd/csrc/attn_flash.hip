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
#include <stdlib.h>

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
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const int qrow = q0 + li;
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
  if (C <= 16 || (C <= 32 && small32))
    return sf_attn_small_dispatch(q, q_cs, k, k_cs, v, v_cs, x, x_cs, gamma, scale, bias, act, out, out_cs,
                                  out_coff, B, T, H, W, C, alpha, o_save, lse_save, vec4, ws, s);
  const int cp = C <= 32 ? 32 : (C <= 64 ? 64 : 128);
  if (ws) {
    a.zs = sf_sweep_parts((long)B * a.nqt, sf_cdiv(N, cp >= 128 ? 32 : 64));
    sf_attn_place_parts(a, cp, ws);
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

// Room for the (O^T, m, l) of up to SF_SWEEP_PARTS_MAX key parts per query row.
extern "C" long sf_attn_fwd_ws_floats(int B, int N, int C) {
  if (B <= 0 || N <= 0 || C <= 0 || C > 128) return 0;
  const int cp = C <= 4 ? 4 : (C <= 8 ? 8 : (C <= 16 ? 16 : (C <= 32 ? 32 : (C <= 64 ? 64 : 128))));
  return (long)B * SF_SWEEP_PARTS_MAX * N * (cp + 2);
}

extern "C" int sf_attn_fwd_ws(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                              const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                              int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                              int alpha, float* o_save, float* lse_save, float* ws, void* stream) {
  if (!ws) return SF_EINVAL;
  return attn_fwd_impl(q, q_cs, k, k_cs, v, v_cs, x, x_cs, gamma, scale, bias, act, out, out_cs, out_coff, B, T, H,
                       W, C, alpha, o_save, lse_save, ws, stream);
}
