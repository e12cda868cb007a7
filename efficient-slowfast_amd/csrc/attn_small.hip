// attn_small.hip — flash SpatialAttention for SMALL head dims (C = d <= 16) on v_mfma_f32_16x16x4_f32.
//
// Same semantics and epilogue as attn_flash.hip (wdf_attention_helper.py:41-54 + the CMDA tail), but the
// 32x32 tiles of that kernel waste 75 % of the second product's rows at C = 8 (s1_fuse of the R50 model,
// N = 25088) and 87 % at C = 4 (GhostNet s1_fuse, N = 100352).  Here one wavefront owns 16 query rows and
// both products run on 16x16x4 tiles:
//   S^T[16 keys x 16 queries] = K * Q^T : A = K^T tile from LDS (lane (key j, quarter g) -> channel 4s+g),
//                                         B = Q fragment (lane (query i, quarter g) -> channel 4s+g);
//       the accumulator leaves query i on the lane (col = lane&15) and keys 4g..4g+3 in its 4 registers.
//   O^T[16 ch x 16 queries] += V^T * P^T : B = those registers unchanged (k = key group g, col = query),
//                                         A = V[key 4g + r][channel lane&15] (ds_read_b32, conflict free).
// Four key tiles (64 keys) are scored before each online-softmax update, so the per-row max costs one
// in-register reduction over 16 values plus two lane-quarter exchanges, as in the 32x32 kernel.
#include "attn_args.h"

namespace {

using AttnArgs = SfAttnArgs;

constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;

template <int CP, int VEC>
__global__ __launch_bounds__(256) void attn_small_kernel(const AttnArgs p) {
  constexpr int KT = 64;                        // keys per LDS tile = one softmax step
  constexpr int QS = CP / 4;                    // MFMA k-steps of the first product
  constexpr int KP = KT + 16;                   // K^T row pitch: quarter g lands 16 banks further
  constexpr int VP = (CP == 32) ? 36 : ((CP == 16) ? 20 : 12);  // V row pitch: rows 4 apart land 16 banks apart
  constexpr int F4 = CP / 4;
  constexpr int NLD = KT * F4;                  // float4 per tile (K and V each)
  constexpr int LIT = (NLD + 255) / 256;        // staging passes (2 for CP = 32)
  constexpr int CTN = (CP + 15) / 16;           // 16-channel output tiles (2 for CP = 32)

  __shared__ __attribute__((aligned(16))) float smem[2 * (CP * KP + KT * VP)];
  float* const Kt = smem;
  float* const Vs = smem + 2 * CP * KP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 15;
  const int lg = lane >> 4;
  const int bz = blockIdx.x / p.nqt;  // workgroup -> (clip b, key part z, query tile)
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int q0 = (blockIdx.x - bz * p.nqt) * 64 + wave * 16;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;

  // ---- Q fragment: lane (i, g) holds Q[i][4s + g]
  float qf[QS];
  {
    const int qrow = q0 + li;
    const bool ok = qrow < N;
    const float* qp = p.q + (brow + (ok ? qrow : 0)) * p.q_cs;
#pragma unroll
    for (int s = 0; s < QS; ++s) {
      const int c = 4 * s + lg;
      qf[s] = (ok && c < C) ? qp[c] * LOG2E : 0.f;  // scores in the exp2 domain: p = 2^(s' - m')
    }
  }

  // Two accumulators break the 16-long dependent MFMA chain of the second product (40-cycle dependent
  // latency vs 32-cycle issue).  For CP < 16 the tile has spare rows: row CP of V^T is all ones (written
  // into the V tile's pad column), so O^T row CP accumulates the softmax denominator on the matrix core
  // with exactly the same rescaling as O — no per-element VALU adds.
  constexpr bool ONES_ROW = CP < 16;
  f32x4 o0[CTN], o1[CTN];
#pragma unroll
  for (int ct = 0; ct < CTN; ++ct) {
    o0[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o1[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float m_run = NEG_BIG;                                        // the stale reference maximum m_ref of this lane's query
  f32x4 negm = {-NEG_BIG, -NEG_BIG, -NEG_BIG, -NEG_BIG};        // -m_ref: the C operand of a score tile's first MFMA
  float l_run = 0.f;

  // ---- staging: float4 f = tid + 256 i (< NLD) is (row = f / F4, channels 4*(f % F4) ..).  The row pointers are
  // kept and advanced by one tile per call (no 64-bit multiplies in the loop: integer vector work is not hidden
  // behind fp32 MFMAs either)
  f32x4 rk[LIT], rv[LIT];
  const float* kptr[LIT];
  const float* vptr[LIT];
  int jrow[LIT];
  const int tz = ((N + KT - 1) / KT + p.zs - 1) / p.zs;  // key tiles per part
  const int t0 = z * tz;
  const int ntiles = min((N + KT - 1) / KT, t0 + tz);
#pragma unroll
  for (int i = 0; i < LIT; ++i) {
    const int f = tid + i * 256;
    const int srow = f / F4, sc4 = (f - srow * F4) * 4;
    jrow[i] = t0 * KT + srow;
    kptr[i] = p.k + (brow + jrow[i]) * p.k_cs + sc4;
    vptr[i] = p.v + (brow + jrow[i]) * p.v_cs + sc4;
  }
  const long kstep = (long)KT * p.k_cs, vstep = (long)KT * p.v_cs;
  auto load_tile = [&]() {  // the next tile in order
#pragma unroll
    for (int i = 0; i < LIT; ++i) {
      const int f = tid + i * 256;
      const int sc4 = (f % F4) * 4;
      rk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (f < NLD && jrow[i] < N) {
        const float* kp = kptr[i];
        const float* vp = vptr[i];
        if (VEC == 4) {
          if (sc4 < C) {
            rk[i] = *reinterpret_cast<const f32x4*>(kp);
            rv[i] = *reinterpret_cast<const f32x4*>(vp);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((sc4 + e) < C) {
              rk[i][e] = kp[e];
              rv[i][e] = vp[e];
            }
        }
      }
      jrow[i] += KT;
      kptr[i] += kstep;
      vptr[i] += vstep;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LIT; ++i) {
      const int f = tid + i * 256;
      const int srow = f / F4, sc4 = (f - srow * F4) * 4;
      if (f < NLD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) Kt[(buf * CP + sc4 + e) * KP + srow] = rk[i][e];
        *reinterpret_cast<f32x4*>(Vs + (buf * KT + srow) * VP + sc4) = rv[i];
        if (ONES_ROW && sc4 == 0) Vs[(buf * KT + srow) * VP + CP] = 1.0f;
      }
    }
  };

  load_tile();
  store_tile(0);
  __syncthreads();
  for (int t = t0; t < ntiles; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = (t + 1) < ntiles;
    if (more) load_tile();
    // ---- scores for 4 key tiles of 16.  fp32 MFMAs run on the SIMD's vector ALUs, so every v_sub / v_max beside
    // them costs its full issue time (tools/microbench/mfma_coexec.hip) and this kernel has only 24 MFMAs per 16
    // exponentials: the scores are accumulated ON TOP of -m_ref (C operand of each tile's first MFMA: all 16 scores
    // of a lane belong to ONE query, so a 4-register block serves the four tiles), m_ref being a stale running
    // maximum that is refreshed only when a tile exceeds it by more than 2^soft_t — then the scores are recomputed
    // from zero and O (with its ones-row denominator) rescaled, the classic online-softmax step.
    f32x4 s[4];
    const float* kbase = Kt + (buf * CP + lg) * KP + li;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = negm;
#pragma unroll
      for (int u = 0; u < QS; ++u)
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kbase[(4 * u) * KP + kt * 16], qf[u], s[kt], 0, 0, 0);
    }
    // the second product's A operands (V^T: 16 keys x CTN channel tiles per lane) are read NOW: their LDS latency
    // passes behind the score MFMAs and the exponentials instead of in front of every MFMA pair
    const float* vbase = Vs + (buf * KT + 4 * lg) * VP + (ONES_ROW ? (li <= CP ? li : 0) : li);
    float vfr[4][4][CTN];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) vfr[kt][r][ct] = vbase[(kt * 16 + r) * VP + 16 * ct];
    const int jbase = t * KT;
    const bool ragged = jbase + KT > N;  // last tile (wave-uniform)
    if (ragged) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (jbase + kt * 16 + 4 * lg + r >= N) s[kt][r] = NEG_BIG;
    }
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
    mx = fmaxf(fmaxf(mx, s[0][3]), s[1][0]);
    mx = fmaxf(fmaxf(mx, s[1][1]), s[1][2]);
    mx = fmaxf(fmaxf(mx, s[1][3]), s[2][0]);
    mx = fmaxf(fmaxf(mx, s[2][1]), s[2][2]);
    mx = fmaxf(fmaxf(mx, s[2][3]), s[3][0]);
    mx = fmaxf(fmaxf(mx, s[3][1]), s[3][2]);
    mx = fmaxf(mx, s[3][3]);
    if (__any(mx > p.soft_t)) {  // refresh the reference (always on the first tile: -m_ref = +BIG there)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < QS; ++u)
          s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kbase[(4 * u) * KP + kt * 16], qf[u], s[kt], 0, 0, 0);
      }
      if (ragged) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (jbase + kt * 16 + 4 * lg + r >= N) s[kt][r] = NEG_BIG;
      }
      float mloc = s[0][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, s[kt][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - mnew);  // 0 on the first tile
      if (!ONES_ROW) l_run *= alpha;
#pragma unroll
      for (int ct = 0; ct < CTN; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o0[ct][r] *= alpha;
          o1[ct][r] *= alpha;
        }
      m_run = mnew;
      negm = (f32x4){-mnew, -mnew, -mnew, -mnew};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][r] -= mnew;
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r]);
        if (!ONES_ROW) l_run += s[kt][r];
      }
    // ---- O^T += V^T P^T (lanes li > CP re-read column 0: their output rows are never stored)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) {
          o0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(vfr[kt][r][ct], s[kt][r], o0[ct], 0, 0, 0);
          o1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(vfr[kt][r + 1][ct], s[kt][r + 1], o1[ct], 0, 0, 0);
        }
      }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane (i, g) holds channels 16 ct + 4g .. +3 of query i
  f32x4 o[CTN];
#pragma unroll
  for (int ct = 0; ct < CTN; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[ct][r] = o0[ct][r] + o1[ct][r];
  float l_tot;
  if (ONES_ROW) {  // denominator = O^T row CP: lane (i, g = CP/4), register 0
    l_tot = __shfl(o[0][0], li + 16 * (CP / 4), 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
  }
  const int qrow = q0 + li;
  if (p.zs > 1) {  // this key part's (O^T, m, l): merged and finished by attn_fwd_merge_kernel (attn_flash.hip)
    if (qrow < N) {
      const long prow = (long)bz * N + qrow;
      if (lg == 0) *reinterpret_cast<float2*>(p.part_ml + prow * 2) = make_float2(m_run, l_tot);
#pragma unroll
      for (int ct = 0; ct < CTN; ++ct)
        if (16 * ct + 4 * lg < CP) *reinterpret_cast<f32x4*>(p.part_o + prow * CP + 16 * ct + 4 * lg) = o[ct];
    }
    return;
  }
  if (qrow >= N) return;
  const float inv_l = 1.0f / l_tot;
  if (p.lse_save && lg == 0) p.lse_save[brow + qrow] = m_run + __log2f(l_tot);
  const float gamma = p.gamma ? p.gamma[0] : 1.0f;
  const int HW = p.H * p.W;
  const int tq = qrow / HW;
  const int hw = qrow - tq * HW;
  const float* xp = p.x + (brow + qrow) * p.x_cs;
  const long orow0 = ((long)b * p.T * p.alpha + (long)tq * p.alpha) * HW + hw;
#pragma unroll
  for (int ct = 0; ct < CTN; ++ct) {
    const int c0 = 16 * ct + 4 * lg;
    if (c0 >= C) continue;
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      const bool okc = c < C;
      if (p.o_save && okc) p.o_save[(brow + qrow) * C + c] = o[ct][e] * inv_l;
      float v = gamma * (o[ct][e] * inv_l) + (okc ? xp[c] : 0.f);
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

template <int CP>
int launch(const AttnArgs& a, bool vec4, hipStream_t s) {
  const int grid = a.B * a.zs * a.nqt;
  if (vec4)
    hipLaunchKernelGGL((attn_small_kernel<CP, 4>), dim3(grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((attn_small_kernel<CP, 1>), dim3(grid), dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return a.zs > 1 ? sf_attn_fwd_merge(a, CP, s) : SF_OK;
}

}  // namespace

int sf_attn_lane_try(SfAttnArgs a, bool vec4, float* ws, hipStream_t stream);  // attn_lane.hip

// Called by sf_attn_fwd (attn_flash.hip) for C <= 16.
int sf_attn_small_dispatch(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                           const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                           int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                           int alpha, float* o_save, float* lse_save, bool vec4, float* ws, hipStream_t stream) {
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.x = x; a.gamma = gamma; a.scale = scale; a.bias = bias; a.out = out;
  a.o_save = o_save; a.lse_save = lse_save;
  a.q_cs = q_cs; a.k_cs = k_cs; a.v_cs = v_cs; a.x_cs = x_cs; a.out_cs = out_cs; a.out_coff = out_coff;
  a.B = B; a.T = T; a.H = H; a.W = W; a.C = C; a.N = T * H * W; a.alpha = alpha; a.act = act;
  {  // d = 4 / d = 8: one query per lane on 4x4x1 MFMA blocks (attn_lane.hip) — no padded rows in either product
    const int rc = sf_attn_lane_try(a, vec4, ws, stream);
    if (rc != 1) return rc;
  }
  a.nqt = sf_cdiv(a.N, 64);
  a.soft_t = sf_attn_soft_t();
  a.zs = 1; a.part_o = nullptr; a.part_ml = nullptr;
  if (ws) {
    a.zs = sf_sweep_parts((long)B * a.nqt, sf_cdiv(a.N, 64));
    sf_attn_place_parts(a, C <= 4 ? 4 : (C <= 8 ? 8 : (C <= 16 ? 16 : 32)), ws);
  }
  if (C <= 4) return launch<4>(a, vec4, stream);
  if (C <= 8) return launch<8>(a, vec4, stream);
  if (C <= 16) return launch<16>(a, vec4, stream);
  return launch<32>(a, vec4, stream);
}
