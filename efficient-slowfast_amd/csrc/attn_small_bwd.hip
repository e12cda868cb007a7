// attn_small_bwd.hip — flash SpatialAttention backward for SMALL head dims (C = d <= 16) on
// v_mfma_f32_16x16x4_f32 tiles (the backward counterpart of attn_small.hip; math as in attn_bwd.hip).
//
// One wavefront owns 16 queries (dQ kernel) resp. 16 keys (dK/dV kernel) on its lanes (lane&15) and sweeps
// the other index in stages of 64 (4 tiles of 16).  Products whose contraction runs over the channels take
// their A operand from a TRANSPOSED LDS tile [channel][item] (lane (item, g) reads channel 4u+g), products
// whose contraction runs over the swept index take it from the row-major tile [item][channel]
// (lane (channel, g) reads item 4g+r) and use the recomputed P / dS registers directly as B operand.
#include "bx.h"
#include <stdlib.h>

namespace {

struct BwdArgs {
  const float* q; const float* k; const float* v;
  const float* dz; const float* lse; const float* dvec; const float* gamma;
  float* dq; float* dk; float* dv;
  int q_cs, k_cs, v_cs, dz_cs, dq_cs, dk_cs, dv_cs;
  int B, C, N, nt;
  int zs;                  // fused kernel: the swept (query) range is cut into zs parts
  float* dkp; float* dvp;  // zs > 1: dK / dV partials [B][zs][N][CP], summed afterwards
};

constexpr float POS_BIG = 3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int ST = 64;        // swept items per stage
constexpr int TP = ST + 16;   // transposed-tile row pitch (quarter g lands 16 banks further)

template <int CP>
struct Geo {
  static constexpr int QS = CP / 4;
  static constexpr int RP = (CP == 16) ? 20 : 12;  // row-major tile pitch (rows 4 apart land 16 banks apart)
  static constexpr int F4 = CP / 4;
  static constexpr int NLD = ST * F4;              // float4 per tile (<= 256)
};

// ---------------------------------------------------------------------------------------------- dQ
template <int CP>
__global__ __launch_bounds__(256) void attn_small_dq_kernel(const BwdArgs p) {
  using G = Geo<CP>;
  constexpr int QS = G::QS, RP = G::RP, F4 = G::F4, NLD = G::NLD;
  constexpr int STAGE = 2 * CP * TP + ST * RP;  // Kt, Vt, Ks
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.nt;
  const int q0 = (blockIdx.x - b * p.nt) * 64 + wave * 16;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];

  const int qrow = q0 + li;
  const bool qok = qrow < N;
  float qf[QS], df[QS];
  {
    const float* qp = p.q + (brow + (qok ? qrow : 0)) * p.q_cs;
    const float* dp = p.dz + (brow + (qok ? qrow : 0)) * p.dz_cs;
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      const int c = 4 * u + lg;
      const bool ok = qok && c < C;
      qf[u] = ok ? qp[c] * LOG2E : 0.f;
      df[u] = ok ? dp[c] * gamma : 0.f;
    }
  }
  const float lse = qok ? p.lse[brow + qrow] : 0.f;
  const float dsum = qok ? p.dvec[brow + qrow] * gamma : 0.f;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};

  const int srow = tid / F4;
  const int sc4 = (tid - srow * F4) * 4;
  f32x4 rk = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
  auto load_stage = [&](int j0) {
    rk = (f32x4){0.f, 0.f, 0.f, 0.f};
    rv = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int j = j0 + srow;
    if (tid < NLD && j < N) {
      const float* kp = p.k + (brow + j) * p.k_cs + sc4;
      const float* vp = p.v + (brow + j) * p.v_cs + sc4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((sc4 + e) < C) {
          rk[e] = kp[e];
          rv[e] = vp[e];
        }
    }
  };
  auto store_stage = [&](int buf) {
    if (tid < NLD) {
      float* Kt = smem + buf * STAGE;
      float* Vt = Kt + CP * TP;
      float* Ks = Vt + CP * TP;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Kt[(sc4 + e) * TP + srow] = rk[e];
        Vt[(sc4 + e) * TP + srow] = rv[e];
      }
      *reinterpret_cast<f32x4*>(Ks + srow * RP + sc4) = rk;
    }
  };

  const int nst = (N + ST - 1) / ST;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int t = 0; t < nst; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1) < nst;
    if (more) load_stage((t + 1) * ST);
    const float* Kt = smem + buf * STAGE;
    const float* Vt = Kt + CP * TP;
    const float* Ks = Vt + CP * TP;
    f32x4 s[4], dp[4];
    const float* kb = Kt + lg * TP + li;
    const float* vb = Vt + lg * TP + li;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < QS; ++u) {
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[(4 * u) * TP + kt * 16], qf[u], s[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[(4 * u) * TP + kt * 16], df[u], dp[kt], 0, 0, 0);
      }
    }
    const int jbase = t * ST;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool jok = (jbase + kt * 16 + 4 * lg + r) < N;
        const float pr = jok ? __builtin_amdgcn_exp2f(s[kt][r] - lse) : 0.f;
        s[kt][r] = pr * (dp[kt][r] - dsum);  // dS^T
      }
    const float* kc = Ks + (4 * lg) * RP + (li & (CP - 1));
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[(kt * 16 + r) * RP], s[kt][r], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[(kt * 16 + r + 1) * RP], s[kt][r + 1], a1, 0, 0, 0);
      }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
  const int c0 = 4 * lg;
  if (!qok || c0 >= C) return;
  float* op = p.dq + (brow + qrow) * p.dq_cs + c0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c0 + e < C) op[e] = a0[e] + a1[e];
}

// ---------------------------------------------------------------------------------------------- dK, dV
template <int CP>
__global__ __launch_bounds__(256) void attn_small_dkv_kernel(const BwdArgs p) {
  using G = Geo<CP>;
  constexpr int QS = G::QS, RP = G::RP, F4 = G::F4, NLD = G::NLD;
  constexpr int STAGE = 2 * CP * TP + 2 * ST * RP + 2 * ST;  // Qt, dOt, Qs, dOs, lse, D
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.nt;
  const int j0 = (blockIdx.x - b * p.nt) * 64 + wave * 16;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];

  const int jrow = j0 + li;
  const bool jok = jrow < N;
  float kf[QS], vf[QS];
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs;
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      const int c = 4 * u + lg;
      const bool ok = jok && c < C;
      kf[u] = ok ? kp[c] * LOG2E : 0.f;
      vf[u] = ok ? vp[c] : 0.f;
    }
  }
  f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = dk0, dv0 = dk0, dv1 = dk0;

  const int srow = tid / F4;
  const int sc4 = (tid - srow * F4) * 4;
  f32x4 rq = {0.f, 0.f, 0.f, 0.f}, rd = {0.f, 0.f, 0.f, 0.f};
  float rl = 0.f, rD = 0.f;
  auto load_stage = [&](int i0) {
    rq = (f32x4){0.f, 0.f, 0.f, 0.f};
    rd = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int i = i0 + srow;
    if (tid < NLD && i < N) {
      const float* qp = p.q + (brow + i) * p.q_cs + sc4;
      const float* dp = p.dz + (brow + i) * p.dz_cs + sc4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((sc4 + e) < C) {
          rq[e] = qp[e];
          rd[e] = dp[e] * gamma;
        }
    }
    if (tid < ST) {
      const int ii = i0 + tid;
      rl = (ii < N) ? p.lse[brow + ii] : POS_BIG;
      rD = (ii < N) ? p.dvec[brow + ii] * gamma : 0.f;
    }
  };
  auto store_stage = [&](int buf) {
    float* Qt = smem + buf * STAGE;
    float* Dt = Qt + CP * TP;
    float* Qs = Dt + CP * TP;
    float* Ds = Qs + ST * RP;
    float* Ls = Ds + ST * RP;
    if (tid < NLD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Qt[(sc4 + e) * TP + srow] = rq[e];
        Dt[(sc4 + e) * TP + srow] = rd[e];
      }
      *reinterpret_cast<f32x4*>(Qs + srow * RP + sc4) = rq;
      *reinterpret_cast<f32x4*>(Ds + srow * RP + sc4) = rd;
    }
    if (tid < ST) {
      Ls[tid] = rl;
      Ls[ST + tid] = rD;
    }
  };

  const int nst = (N + ST - 1) / ST;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int t = 0; t < nst; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1) < nst;
    if (more) load_stage((t + 1) * ST);
    const float* Qt = smem + buf * STAGE;
    const float* Dt = Qt + CP * TP;
    const float* Qs = Dt + CP * TP;
    const float* Ds = Qs + ST * RP;
    const float* Ls = Ds + ST * RP;
    f32x4 s[4], dp[4];
    const float* qb = Qt + lg * TP + li;
    const float* db = Dt + lg * TP + li;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      s[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < QS; ++u) {
        s[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[(4 * u) * TP + qt * 16], kf[u], s[qt], 0, 0, 0);    // S'
        dp[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(db[(4 * u) * TP + qt * 16], vf[u], dp[qt], 0, 0, 0);  // dP
      }
    }
    f32x4 pr[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qt * 16 + 4 * lg);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ls + ST + qt * 16 + 4 * lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[qt][r] = __builtin_amdgcn_exp2f(s[qt][r] - l4[r]);
        s[qt][r] = pr[qt][r] * (dp[qt][r] - d4[r]);  // dS
      }
    }
    const float* qc = Qs + (4 * lg) * RP + (li & (CP - 1));
    const float* dc = Ds + (4 * lg) * RP + (li & (CP - 1));
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        dv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dc[(qt * 16 + r) * RP], pr[qt][r], dv0, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qc[(qt * 16 + r) * RP], s[qt][r], dk0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dc[(qt * 16 + r + 1) * RP], pr[qt][r + 1], dv1, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qc[(qt * 16 + r + 1) * RP], s[qt][r + 1], dk1, 0, 0, 0);
      }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
  const int c0 = 4 * lg;
  if (!jok || c0 >= C) return;
  float* okp = p.dk + (brow + jrow) * p.dk_cs + c0;
  float* ovp = p.dv + (brow + jrow) * p.dv_cs + c0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c0 + e < C) {
      okp[e] = dk0[e] + dk1[e];
      ovp[e] = dv0[e] + dv1[e];
    }
}

// ---------------------------------------------------------------------------------------------- fused dK, dV, dQ
// The dK/dV sweep also forms dQ (see attn_bwd_fused_kernel in attn_bwd.hip): per 16x16 dS tile the wavefront
// transposes dS through its LDS slot (4 ds_write_b32, 1 ds_read_b128) and adds 4 MFMAs  dQ(i,c) += dS(i,j) k(j,c);
// the four wavefronts' [64 x CP] partials of a stage are summed in a fixed order and stored to this 64-key block's
// plane of the workspace.
template <int CP>
__global__ __launch_bounds__(256) void attn_small_fused_kernel(const BwdArgs p, float* __restrict__ ws) {
  using G = Geo<CP>;
  constexpr int QS = G::QS, RP = G::RP, F4 = G::F4, NLD = G::NLD;
  constexpr int STAGE = 2 * CP * TP + 2 * ST * RP + 2 * ST;  // Qt, dOt, Qs, dOs, lse, D
  constexpr int SLOT = (ST * CP > 16 * 20) ? ST * CP : 16 * 20;  // dQ partial [64][CP]; also the 16x16 transposition tile
  constexpr int TT = 20;                                          // its pitch
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 4 * SLOT];
  float* const slots = smem + 2 * STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int bz = blockIdx.x / p.nt;  // workgroup -> (clip b, query part z, key block kb)
  const int kb = blockIdx.x - bz * p.nt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = kb * 64 + wave * 16;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];
  float* const myslot = slots + wave * SLOT;

  const int jrow = j0 + li;
  const bool jok = jrow < N;
  float kf[QS], vf[QS];
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs;
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      const int c = 4 * u + lg;
      const bool ok = jok && c < C;
      kf[u] = ok ? kp[c] * LOG2E : 0.f;
      vf[u] = ok ? vp[c] : 0.f;
    }
  }
  // K as B operand of dQ = dS K: lane (c = li, quarter lg) holds k[key j0 + 4*lg + s][li] for step s
  float kbr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int key = j0 + 4 * lg + s;
    kbr[s] = (key < N && li < C) ? p.k[(brow + key) * p.k_cs + li] : 0.f;
  }
  f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = dk0, dv0 = dk0, dv1 = dk0;

  const int srow = tid / F4;
  const int sc4 = (tid - srow * F4) * 4;
  f32x4 rq = {0.f, 0.f, 0.f, 0.f}, rd = {0.f, 0.f, 0.f, 0.f};
  float rl = 0.f, rD = 0.f;
  auto load_stage = [&](int i0) {
    rq = (f32x4){0.f, 0.f, 0.f, 0.f};
    rd = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int i = i0 + srow;
    if (tid < NLD && i < N) {
      const float* qp = p.q + (brow + i) * p.q_cs + sc4;
      const float* dp = p.dz + (brow + i) * p.dz_cs + sc4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((sc4 + e) < C) {
          rq[e] = qp[e];
          rd[e] = dp[e] * gamma;
        }
    }
    if (tid < ST) {
      const int ii = i0 + tid;
      rl = (ii < N) ? p.lse[brow + ii] : POS_BIG;
      rD = (ii < N) ? p.dvec[brow + ii] * gamma : 0.f;
    }
  };
  auto store_stage = [&](int buf) {
    float* Qt = smem + buf * STAGE;
    float* Dt = Qt + CP * TP;
    float* Qs = Dt + CP * TP;
    float* Ds = Qs + ST * RP;
    float* Ls = Ds + ST * RP;
    if (tid < NLD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Qt[(sc4 + e) * TP + srow] = rq[e];
        Dt[(sc4 + e) * TP + srow] = rd[e];
      }
      *reinterpret_cast<f32x4*>(Qs + srow * RP + sc4) = rq;
      *reinterpret_cast<f32x4*>(Ds + srow * RP + sc4) = rd;
    }
    if (tid < ST) {  // NEGATED: the score / dP tiles start from these as their C operand (S' - lse, dP - D)
      Ls[tid] = -rl;
      Ls[ST + tid] = -rD;
    }
  };

  float* const plane = ws + ((long)b * p.nt + kb) * N * CP;
  const int tz = ((N + ST - 1) / ST + p.zs - 1) / p.zs;  // stages per query part
  const int t0 = z * tz;
  const int nst = min((N + ST - 1) / ST, t0 + tz);
  load_stage(t0 * ST);
  store_stage(0);
  __syncthreads();
  for (int t = t0; t < nst; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = (t + 1) < nst;
    if (more) load_stage((t + 1) * ST);
    const float* Qt = smem + buf * STAGE;
    const float* Dt = Qt + CP * TP;
    const float* Qs = Dt + CP * TP;
    const float* Ds = Qs + ST * RP;
    const float* Ls = Ds + ST * RP;
    f32x4 s[4], dp[4];
    const float* qb = Qt + lg * TP + li;
    const float* db = Dt + lg * TP + li;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      // fp32 MFMAs run on the vector ALUs (a v_sub beside them is not hidden, tools/microbench/mfma_coexec.hip):
      // S' - lse and dP - D ride on the products — register r of tile qt is query 16 qt + 4 lg + r, so the
      // ds_read_b128 of -lse / -D IS the C operand of the tile's first MFMA
      s[qt] = *reinterpret_cast<const f32x4*>(Ls + qt * 16 + 4 * lg);
      dp[qt] = *reinterpret_cast<const f32x4*>(Ls + ST + qt * 16 + 4 * lg);
#pragma unroll
      for (int u = 0; u < QS; ++u) {
        s[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[(4 * u) * TP + qt * 16], kf[u], s[qt], 0, 0, 0);    // S'
        dp[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(db[(4 * u) * TP + qt * 16], vf[u], dp[qt], 0, 0, 0);  // dP
      }
    }
    f32x4 pr[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[qt][r] = __builtin_amdgcn_exp2f(s[qt][r]);
      s[qt] = pr[qt] * dp[qt];  // dS = P (dP - D)
    }
    const float* qc = Qs + (4 * lg) * RP + (li & (CP - 1));
    const float* dc = Ds + (4 * lg) * RP + (li & (CP - 1));
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        dv0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dc[(qt * 16 + r) * RP], pr[qt][r], dv0, 0, 0, 0);
        dk0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qc[(qt * 16 + r) * RP], s[qt][r], dk0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dc[(qt * 16 + r + 1) * RP], pr[qt][r + 1], dv1, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qc[(qt * 16 + r + 1) * RP], s[qt][r + 1], dk1, 0, 0, 0);
      }
    // dQ: dS tile [query 4*lg + r][key li] -> T[query][key] -> A fragment (query li, keys 4*lg .. 4*lg+3)
    f32x4 dqp[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) myslot[(4 * lg + r) * TT + li] = s[qt][r];
      const f32x4 a = *reinterpret_cast<const f32x4*>(myslot + li * TT + 4 * lg);
      dqp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        dqp[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], kbr[st], dqp[qt], 0, 0, 0);  // rows query, cols c
    }
    // partial [64][CP] (row 16*qt + 4*lg + r, channel li) into the slot
    if (li < CP) {
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) myslot[(qt * 16 + 4 * lg + r) * CP + li] = dqp[qt][r];
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
    if (tid < ST * CP / 4) {  // fixed-order sum of the four partials, one float4 per thread
      const int e0 = tid * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(slots + e0);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(slots + w * SLOT + e0);
      const int qi = t * ST + e0 / CP;
      if (qi < N) *reinterpret_cast<f32x4*>(plane + (long)qi * CP + (e0 % CP)) = v;
    }
    __syncthreads();
  }
  const int c0 = 4 * lg;
  if (p.zs > 1) {  // this query part's share; attn_dq_reduce_kernel adds the parts in order
    if (jok && c0 < CP) {
      *reinterpret_cast<f32x4*>(p.dkp + ((long)bz * N + jrow) * CP + c0) = dk0 + dk1;
      *reinterpret_cast<f32x4*>(p.dvp + ((long)bz * N + jrow) * CP + c0) = dv0 + dv1;
    }
    return;
  }
  if (!jok || c0 >= C) return;
  float* okp = p.dk + (brow + jrow) * p.dk_cs + c0;
  float* ovp = p.dv + (brow + jrow) * p.dv_cs + c0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c0 + e < C) {
      okp[e] = dk0[e] + dk1[e];
      ovp[e] = dv0[e] + dv1[e];
    }
}

// ---------------------------------------------------------------------------------------------- fused, d <= 8: 4x4x1 MFMAs
// The three products whose OUTPUT is the channels (dV = P^T dO, dK = dS^T Q, dQ = dS K) fill 4 (d = 4) or 8 of the 16
// output columns of a 16x16x4 MFMA: 48 of the fused kernel's 56 MFMAs per stage run three quarters empty, and the
// fp32 MFMA shares the vector ALUs, so the kernel is bound by exactly those cycles (1 792 MFMA cycles per 1 024
// scores; measured 1.08e12 scores/s = that).  v_mfma_f32_4x4x1_16B_f32 is the same rate (512 FLOP in 8 cycles,
// tools/microbench/mfma4x4_probe.hip) as sixteen independent 4x4 outer products: block b = lanes 4b .. 4b+3,
// D[v][lane 4b+j] += A[lane 4b+v] * B[lane 4b+j].  With the four channels of a quad as the columns j nothing is empty:
//   dV, dK: A = the P / dS register itself (block = (lane quarter lg, key quad): its four lanes hold four KEYS of one
//           query), B = dO / Q[that query][4u + (lane & 3)] (one ds_read_b32): rows v = the quad's keys.  The four
//           quarters see different queries: their sums are added once, after the sweep, through LDS.
//   dQ:     the stage's dS goes through LDS once as a [64 queries][16 keys] tile (as before per 16x16 tile) and comes
//           back with the QUERY on the lane; A = dS(query = lane, key k), B = K[key k][4u + (lane & 3)] from registers
//           (the wavefront's 16 keys are fixed): rows v = the four queries of the block, complete over the 16 keys —
//           no cross-lane sum.  The partial goes to the slot lane-linear ([u][v][lane]: conflict-free, and the summing
//           thread's float4 over the channels is contiguous).
// 8 + 48 MFMAs of 32 resp. 8 cycles = 640 cycles per stage instead of 1 792.
// BF (d = 4; an experiment, OFF by default — see the dispatch): S' and dP — the two products over the CHANNELS, whose
// operands are inputs — on the bf16 matrix pipe with
// fp32-exact operand pieces (bx.h), off the vector ALUs that the fp32 MFMAs share with exp2: the 16 k slots of one
// v_mfma_f32_16x16x16_bf16 hold four (piece, piece) pairs of the 4 channels, so [q1|q1|q2|q1][k1|k2|k1|k3] and
// [q3|q2|0|0][k1|k2|0|0] are the six kept terms in two instructions.  Q and dO are split when a stage is written to
// LDS ([query][p1|p2|p3|0][4 channels] bf16, 40-byte pitch: the lane's 8-byte fragment reads are conflict-free), K'
// and V once per wavefront into registers.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16x16(u32x2 a, u32x2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

template <int CP, bool BF = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void attn_small_fused44_kernel(const BwdArgs p, float* __restrict__ ws) {
  static_assert(CP == 4 || CP == 8, "one or two channel quads");
  static_assert(!BF || CP == 4, "the bf16-piece scores pack four channels into a k slot");
  using G = Geo<CP>;
  constexpr int QS = G::QS, RP = G::RP, F4 = G::F4, NLD = G::NLD;
  constexpr int PP = 10;                                      // piece-tile pitch in floats (40 bytes per query)
  constexpr int TSZ = BF ? ST * PP : CP * TP;                 // Q^T / dO^T tiles, or their piece tiles
  constexpr int STAGE = 2 * TSZ + 2 * ST * RP + 2 * ST;       // Qt, dOt, Qs, dOs, lse, D
  constexpr int TT = 20;                                      // pitch of the dS^T tile
  constexpr int SLOT = ST * TT;                               // [64 queries][16 keys]; then the dQ partial [QS*4][64]
  static_assert(2 * STAGE >= NW * 2 * 4 * 16 * CP, "the end-of-sweep reduction fits the stage buffers");
  // NW wavefronts = 16 NW keys per workgroup: the dQ planes (one per key block, N x CP floats each, written here and
  // read back by attn_dq_reduce_kernel) halve with NW = 8
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + NW * SLOT];
  float* const slots = smem + 2 * STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4, lc = lane & 3;
  const int bz = blockIdx.x / p.nt;  // workgroup -> (clip b, query part z, key block kb)
  const int kb = blockIdx.x - bz * p.nt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = kb * (16 * NW) + wave * 16;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];
  float* const myslot = slots + wave * SLOT;

  const int jrow = j0 + li;
  const bool jok = jrow < N;
  float kf[QS], vf[QS];
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs;
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      const int c = 4 * u + lg;
      const bool ok = jok && c < C;
      kf[u] = ok ? kp[c] * LOG2E : 0.f;
      vf[u] = ok ? vp[c] : 0.f;
    }
  }
  u32x2 kb1 = {0u, 0u}, kb2 = kb1, vb1 = kb1, vb2 = kb1;  // BF: B operands of S' / dP, slot lg of the two instructions
  if constexpr (BF) {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs;
    float kk[4], vv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      kk[c] = (jok && c < C) ? kp[c] * LOG2E : 0.f;
      vv[c] = (jok && c < C) ? vp[c] : 0.f;
    }
    u32x2 wk[3], wv[3];
    split_pair(kk[0], kk[1], wk, 0);
    split_pair(kk[2], kk[3], wk, 1);
    split_pair(vv[0], vv[1], wv, 0);
    split_pair(vv[2], vv[3], wv, 1);
    const u32x2 zero = {0u, 0u};
    kb1 = lg == 1 ? wk[1] : (lg == 3 ? wk[2] : wk[0]);   // [k1|k2|k1|k3]
    vb1 = lg == 1 ? wv[1] : (lg == 3 ? wv[2] : wv[0]);
    kb2 = lg == 0 ? wk[0] : (lg == 1 ? wk[1] : zero);     // [k1|k2|0|0]
    vb2 = lg == 0 ? wv[0] : (lg == 1 ? wv[1] : zero);
  }
  // K as B operand of dQ: every block multiplies by the SAME key k of this wavefront, lane & 3 picks the channel
  float kq[16][QS];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int u = 0; u < QS; ++u) {
      const int key = j0 + k, c = 4 * u + lc;
      kq[k][u] = (key < N && c < C) ? p.k[(brow + key) * p.k_cs + c] : 0.f;
    }
  f32x4 dkq[QS], dvq[QS];
#pragma unroll
  for (int u = 0; u < QS; ++u) dkq[u] = dvq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int srow = tid / F4;
  const int sc4 = (tid - srow * F4) * 4;
  f32x4 rq = {0.f, 0.f, 0.f, 0.f}, rd = {0.f, 0.f, 0.f, 0.f};
  float rl = 0.f, rD = 0.f;
  auto load_stage = [&](int i0) {
    rq = (f32x4){0.f, 0.f, 0.f, 0.f};
    rd = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int i = i0 + srow;
    if (tid < NLD && i < N) {
      const float* qp = p.q + (brow + i) * p.q_cs + sc4;
      const float* dp = p.dz + (brow + i) * p.dz_cs + sc4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((sc4 + e) < C) {
          rq[e] = qp[e];
          rd[e] = dp[e] * gamma;
        }
    }
    if (tid < ST) {
      const int ii = i0 + tid;
      rl = (ii < N) ? p.lse[brow + ii] : POS_BIG;
      rD = (ii < N) ? p.dvec[brow + ii] * gamma : 0.f;
    }
  };
  auto store_stage = [&](int buf) {
    float* Qt = smem + buf * STAGE;
    float* Dt = Qt + TSZ;
    float* Qs = Dt + TSZ;
    float* Ds = Qs + ST * RP;
    float* Ls = Ds + ST * RP;
    if (tid < NLD) {
      if constexpr (BF) {  // piece rows [p1|p2|p3|0] of the query's 4 channels (one float4 per query: F4 == 1)
        u32x2 wq[3], wd[3];
        split_pair(rq[0], rq[1], wq, 0);
        split_pair(rq[2], rq[3], wq, 1);
        split_pair(rd[0], rd[1], wd, 0);
        split_pair(rd[2], rd[3], wd, 1);
        u32x2* const qrow = reinterpret_cast<u32x2*>(Qt + srow * PP);
        u32x2* const drow = reinterpret_cast<u32x2*>(Dt + srow * PP);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          qrow[pc] = wq[pc];
          drow[pc] = wd[pc];
        }
        qrow[3] = drow[3] = (u32x2){0u, 0u};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Qt[(sc4 + e) * TP + srow] = rq[e];
          Dt[(sc4 + e) * TP + srow] = rd[e];
        }
      }
      *reinterpret_cast<f32x4*>(Qs + srow * RP + sc4) = rq;
      *reinterpret_cast<f32x4*>(Ds + srow * RP + sc4) = rd;
    }
    if (tid < ST) {  // NEGATED: the score / dP tiles start from these as their C operand (S' - lse, dP - D)
      Ls[tid] = -rl;
      Ls[ST + tid] = -rD;
    }
  };

  float* const plane = ws + ((long)b * p.nt + kb) * N * CP;
  const int tz = ((N + ST - 1) / ST + p.zs - 1) / p.zs;  // stages per query part
  const int t0 = z * tz;
  const int nst = min((N + ST - 1) / ST, t0 + tz);
  load_stage(t0 * ST);
  store_stage(0);
  __syncthreads();
  for (int t = t0; t < nst; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = (t + 1) < nst;
    if (more) load_stage((t + 1) * ST);
    const float* Qt = smem + buf * STAGE;
    const float* Dt = Qt + TSZ;
    const float* Qs = Dt + TSZ;
    const float* Ds = Qs + ST * RP;
    const float* Ls = Ds + ST * RP;
    f32x4 s[4], dp[4];
    const float* qb = Qt + lg * TP + li;
    const float* db = Dt + lg * TP + li;
    // BF: slot lg of the A operands [q1|q1|q2|q1] and [q3|q2|0|0] = pieces {0, 0, 1, 0} / {2, 1, 3, 3} of the row
    const int sel1 = lg == 2 ? 1 : 0, sel2 = lg == 0 ? 2 : (lg == 1 ? 1 : 3);
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      s[qt] = *reinterpret_cast<const f32x4*>(Ls + qt * 16 + 4 * lg);
      dp[qt] = *reinterpret_cast<const f32x4*>(Ls + ST + qt * 16 + 4 * lg);
      if constexpr (BF) {
        const u32x2* const qrow = reinterpret_cast<const u32x2*>(Qt + (qt * 16 + li) * PP);
        const u32x2* const drow = reinterpret_cast<const u32x2*>(Dt + (qt * 16 + li) * PP);
        s[qt] = mfma_bf16x16(qrow[sel2], kb2, s[qt]);    // the small terms first
        dp[qt] = mfma_bf16x16(drow[sel2], vb2, dp[qt]);
        s[qt] = mfma_bf16x16(qrow[sel1], kb1, s[qt]);
        dp[qt] = mfma_bf16x16(drow[sel1], vb1, dp[qt]);
      } else {
#pragma unroll
        for (int u = 0; u < QS; ++u) {
          s[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[(4 * u) * TP + qt * 16], kf[u], s[qt], 0, 0, 0);    // S'
          dp[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(db[(4 * u) * TP + qt * 16], vf[u], dp[qt], 0, 0, 0);  // dP
        }
      }
    }
    f32x4 pr[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[qt][r] = __builtin_amdgcn_exp2f(s[qt][r]);
      s[qt] = pr[qt] * dp[qt];  // dS = P (dP - D)
    }
    // dS^T tile: row = query 16 qt + 4 lg + r, column = key li (read back below with the query on the lane)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) myslot[(qt * 16 + 4 * lg + r) * TT + li] = s[qt][r];
    // dV, dK: rows = the four keys of this lane's quad, columns = the channels of quad u
    const float* dB = Ds + (4 * lg) * RP + lc;
    const float* qB = Qs + (4 * lg) * RP + lc;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < QS; ++u) {
          dvq[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(pr[qt][r], dB[(qt * 16 + r) * RP + 4 * u], dvq[u], 0, 0, 0);
          dkq[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(s[qt][r], qB[(qt * 16 + r) * RP + 4 * u], dkq[u], 0, 0, 0);
        }
    // dQ: A = dS(query = lane, key k)
    f32x4 a[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) a[k4] = *reinterpret_cast<const f32x4*>(myslot + lane * TT + 4 * k4);
    f32x4 dqq[QS];
#pragma unroll
    for (int u = 0; u < QS; ++u) dqq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int u = 0; u < QS; ++u)
        dqq[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k >> 2][k & 3], kq[k][u], dqq[u], 0, 0, 0);
    // partial of this wavefront's 16 keys: element (query 4 (lane / 4) + v, channel 4 u + (lane & 3)) at [u][v][lane]
#pragma unroll
    for (int u = 0; u < QS; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) myslot[(u * 4 + v) * 64 + lane] = dqq[u][v];
    if (more) store_stage(buf ^ 1);
    __syncthreads();
    if (tid < ST * CP / 4) {  // fixed-order sum of the four wavefronts' partials, one float4 (a channel quad) per thread
      const int e0 = tid * 4;
      const int ql = e0 / CP, u = (e0 % CP) >> 2;
      const int off = (u * 4 + (ql & 3)) * 64 + (ql >> 2) * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(slots + off);
#pragma unroll
      for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(slots + w * SLOT + off);
      const int qi = t * ST + ql;
      if (qi < N) *reinterpret_cast<f32x4*>(plane + (long)qi * CP + (e0 % CP)) = v;
    }
    __syncthreads();
  }
  // ---- the four lane quarters saw different queries: add them (quarter order) through LDS.  R[wave][lg][key][c],
  //      dK then dV; a wavefront only reads what it wrote (LDS operations of a wavefront execute in order).
  float* const R = smem + wave * (2 * 4 * 16 * CP);
#pragma unroll
  for (int u = 0; u < QS; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int o = (lg * 16 + 4 * (li >> 2) + v) * CP + 4 * u + lc;
      R[o] = dkq[u][v];
      R[4 * 16 * CP + o] = dvq[u][v];
    }
  const int c0 = 4 * lg;
  f32x4 dkf = {0.f, 0.f, 0.f, 0.f}, dvf = dkf;
  if (c0 < CP) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      dkf += *reinterpret_cast<const f32x4*>(R + (g * 16 + li) * CP + c0);
      dvf += *reinterpret_cast<const f32x4*>(R + 4 * 16 * CP + (g * 16 + li) * CP + c0);
    }
  }
  if (p.zs > 1) {  // this query part's share; attn_dq_reduce_kernel adds the parts in order
    if (jok && c0 < CP) {
      *reinterpret_cast<f32x4*>(p.dkp + ((long)bz * N + jrow) * CP + c0) = dkf;
      *reinterpret_cast<f32x4*>(p.dvp + ((long)bz * N + jrow) * CP + c0) = dvf;
    }
    return;
  }
  if (!jok || c0 >= C) return;
  float* okp = p.dk + (brow + jrow) * p.dk_cs + c0;
  float* ovp = p.dv + (brow + jrow) * p.dv_cs + c0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c0 + e < C) {
      okp[e] = dkf[e];
      ovp[e] = dvf[e];
    }
}

template <int CP>
int launch(const BwdArgs& a, int which, hipStream_t s) {
  const int grid = a.B * a.nt;
  if (which & 1) hipLaunchKernelGGL((attn_small_dq_kernel<CP>), dim3(grid), dim3(256), 0, s, a);
  if (which & 2) hipLaunchKernelGGL((attn_small_dkv_kernel<CP>), dim3(grid), dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

}  // namespace

// Called by sf_attn_bwd (attn_bwd.hip) for C <= 16.
int sf_attn_small_bwd_dispatch(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                               const float* dz, int dz_cs, const float* lse, const float* dvec, const float* gamma,
                               float* dq, int dq_cs, float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C,
                               int which, hipStream_t stream) {
  BwdArgs a;
  a.q = q; a.k = k; a.v = v; a.dz = dz; a.lse = lse; a.dvec = dvec; a.gamma = gamma;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.q_cs = q_cs; a.k_cs = k_cs; a.v_cs = v_cs; a.dz_cs = dz_cs; a.dq_cs = dq_cs; a.dk_cs = dk_cs; a.dv_cs = dv_cs;
  a.B = B; a.C = C; a.N = N; a.nt = sf_cdiv(N, 64);
  if (C <= 4) return launch<4>(a, which, stream);
  if (C <= 8) return launch<8>(a, which, stream);
  return launch<16>(a, which, stream);
}

int sf_attn_dq_reduce(const float* ws, float* dq, int dq_cs, int B, int N, int C, int CP, int nkb, hipStream_t s);

// Called by sf_attn_bwd_fused (attn_bwd.hip) for C <= 16.
int sf_attn_small_fused_dispatch(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                                 const float* dz, int dz_cs, const float* lse, const float* dvec, const float* gamma,
                                 float* dq, int dq_cs, float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C,
                                 float* ws, hipStream_t stream) {
  BwdArgs a;
  a.q = q; a.k = k; a.v = v; a.dz = dz; a.lse = lse; a.dvec = dvec; a.gamma = gamma;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.q_cs = q_cs; a.k_cs = k_cs; a.v_cs = v_cs; a.dz_cs = dz_cs; a.dq_cs = dq_cs; a.dk_cs = dk_cs; a.dv_cs = dv_cs;
  static const bool use44 = [] { const char* e = getenv("SF_ATTN_SMALL_44"); return !(e && e[0] == '0'); }();
  // 128 keys per workgroup (8 wavefronts) where that still leaves >= 2 workgroups per CU: half the dQ planes
  static const int nw_env = [] { const char* e = getenv("SF_ATTN_SMALL_NW"); return e ? atoi(e) : 0; }();
  const bool wide = use44 && C <= 4 && (nw_env == 8 || (nw_env != 4 && (long)B * sf_cdiv(N, 128) >= 2 * 256));
  a.B = B; a.C = C; a.N = N; a.nt = sf_cdiv(N, wide ? 128 : 64);
  a.zs = sf_sweep_parts((long)B * a.nt, sf_cdiv(N, ST));
  const int grid = B * a.zs * a.nt;
  int cp = C <= 4 ? 4 : (C <= 8 ? 8 : 16);
  a.dkp = ws + (long)B * a.nt * N * cp;
  a.dvp = a.dkp + (long)B * a.zs * N * cp;
  if (C <= 4) {
    cp = 4;
    // OFF by default: measured SLOWER (d = 4, N = 100 352, 2 clips: 14.95 against 12.41 ms, two alternations on one box) —
    // sixteen 16x16x16 bf16 MFMAs per stage cost more matrix-pipe time than the eight 16x16x4 fp32 ones they replace
    // free on the vector ALUs, and the sweep's waves do not hide them behind exp2.  SF_ATTN_SMALL_BF=1 selects it.
    static const bool bf = [] { const char* e = getenv("SF_ATTN_SMALL_BF"); return e && e[0] == '1'; }();
    if (wide) hipLaunchKernelGGL((attn_small_fused44_kernel<4, false, 8>), dim3(grid), dim3(512), 0, stream, a, ws);
    else if (use44 && bf) hipLaunchKernelGGL((attn_small_fused44_kernel<4, true>), dim3(grid), dim3(256), 0, stream, a, ws);
    else if (use44) hipLaunchKernelGGL((attn_small_fused44_kernel<4>), dim3(grid), dim3(256), 0, stream, a, ws);
    else hipLaunchKernelGGL((attn_small_fused_kernel<4>), dim3(grid), dim3(256), 0, stream, a, ws);
  } else if (C <= 8) {
    cp = 8;
    if (use44) hipLaunchKernelGGL((attn_small_fused44_kernel<8>), dim3(grid), dim3(256), 0, stream, a, ws);
    else hipLaunchKernelGGL((attn_small_fused_kernel<8>), dim3(grid), dim3(256), 0, stream, a, ws);
  } else {
    cp = 16;
    hipLaunchKernelGGL((attn_small_fused_kernel<16>), dim3(grid), dim3(256), 0, stream, a, ws);
  }
  SF_CHECK_LAUNCH();
  if (a.zs > 1) {
    int rc = sf_attn_dq_reduce(a.dkp, dk, dk_cs, B, N, C, cp, a.zs, stream);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dvp, dv, dv_cs, B, N, C, cp, a.zs, stream);
    if (rc != SF_OK) return rc;
  }
  return sf_attn_dq_reduce(ws, dq, dq_cs, B, N, C, cp, a.nt, stream);
}
