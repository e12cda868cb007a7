// attn_bwd.hip — backward of the flash SpatialAttention (recompute form, fp32 MFMA, gfx950).
//
// Forward (attn_flash.hip):  S = q k^T,  P = softmax_j(S),  O = P v,  z = gamma * O + x.
// Given dz = dL/dz:  dO = gamma * dz,  D_i = <dO_i, O_i>,  and with P recomputed from the saved
// log-sum-exp  P_ij = 2^(S'_ij - LSE'_i)  (S' = S * log2 e):
//     dV_j = sum_i P_ij dO_i          dP_ij = <dO_i, v_j>          dS_ij = P_ij (dP_ij - D_i)
//     dQ_i = sum_j dS_ij k_j          dK_j  = sum_i dS_ij q_i
// Two kernels, no atomics (bit-reproducible):
//   attn_bwd_dq_kernel  : one wavefront per 32 QUERIES sweeping all keys   (3 products per tile)
//   attn_bwd_dkv_kernel : one wavefront per 32 KEYS sweeping all queries   (4 products per tile)
// Both reuse the forward's register layouts: the "owned" index (query resp. key) sits on the lane, the
// swept index arrives in the accumulator's 16 registers in the order kappa(r,h) = (r&3)+8(r>>2)+4h, so
// every transposed product takes the recomputed P / dS registers directly as its B operand and reads
// its A operand from the LDS tile with conflict-free ds_read_b32; row-fragment operands are ds_read_b128.
#include "attn_args.h"
#include "attn_bx.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct BwdArgs {
  const float* q; const float* k; const float* v;   // [B, N, *] views
  const float* dz;                                   // dL/dz [B, N, C]
  const float* lse;                                  // [B, N]  (exp2 domain)
  const float* dvec;                                 // [B, N]  <dz_i, O_i>  (gamma applied in-kernel)
  const float* gamma;
  float* dq; float* dk; float* dv;                   // outputs [B, N, *] views
  int q_cs, k_cs, v_cs, dz_cs, dq_cs, dk_cs, dv_cs;
  int B, C, N, nt;
  int zs;                                            // fused kernels: the swept (query) range is cut into zs parts
  float* dkp; float* dvp;                            // zs > 1: dK / dV partials [B][zs][N][CP], summed afterwards
  float* dqp;                                        // two-kernel form with zs > 1: dQ partials [B][zs][N][CP]
  int stagger;                                       // bf16-piece sweeps: s_sleep units the OLDER wavefront of each SIMD waits per tile
};

constexpr float POS_BIG = 3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int kappa(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------- dQ
template <int CP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const BwdArgs p) {
  constexpr int KT = (CP >= 128) ? 32 : 64;
  constexpr int NSUB = KT / 32;
  constexpr int PS = CP + 4;
  constexpr int CT = (CP + 31) / 32;
  constexpr int QS = CP / 8;
  constexpr int F4 = CP / 4;
  constexpr int NF = (KT * F4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * KT * PS];
  float* const Ks = smem;
  float* const Vs = smem + 2 * KT * PS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int bz = blockIdx.x / p.nt;  // workgroup -> (clip b, key part z, query tile); zs = 1 outside the parts mode
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int q0 = (blockIdx.x - bz * p.nt) * 128 + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];

  float qf[QS * 4], df[QS * 4];
  const int qrow = q0 + li;
  const bool qok = qrow < N;
  {
    const float* qp = p.q + (brow + (qok ? qrow : 0)) * p.q_cs;
    const float* dp = p.dz + (brow + (qok ? qrow : 0)) * p.dz_cs;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = s * 8 + lh * 4 + e;
        const bool ok = qok && c < C;
        qf[s * 4 + e] = ok ? qp[c] * LOG2E : 0.f;
        df[s * 4 + e] = ok ? dp[c] * gamma : 0.f;
      }
  }
  const float lse = qok ? p.lse[brow + qrow] : 0.f;
  const float dsum = qok ? p.dvec[brow + qrow] * gamma : 0.f;

  f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

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
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((c + e) < C) {
            tk[e] = kp[e];
            tv[e] = vp[e];
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
        *reinterpret_cast<f32x4*>(Ks + (buf * KT + row) * PS + c) = rk[u];
        *reinterpret_cast<f32x4*>(Vs + (buf * KT + row) * PS + c) = rv[u];
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
      if (jbase >= N) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const float* krow = Ks + (buf * KT + sub * 32 + li) * PS + lh * 4;
      const float* vrow = Vs + (buf * KT + sub * 32 + li) * PS + lh * 4;
#pragma unroll
      for (int g = 0; g < QS; ++g) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + g * 8);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(vrow + g * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g * 4 + e], s, 0, 0, 0);     // S'^T = K Q^T
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], df[g * 4 + e], dp, 0, 0, 0);   // dP^T = V dO^T
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool jok = (jbase + kappa(r, lh)) < N;
        const float pr = jok ? __builtin_amdgcn_exp2f(s[r] - lse) : 0.f;
        s[r] = pr * (dp[r] - dsum);  // dS^T
      }
      const float* kcol = Ks + (buf * KT + sub * 32 + 4 * lh) * PS + (CP >= 32 ? li : (li & (CP - 1)));
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(kcol[((r & 3) + 8 * (r >> 2)) * PS + ct * 32], s[r],
                                                         acc[ct], 0, 0, 0);  // dQ^T += K^T dS^T
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (!qok) return;
  if (p.zs > 1) {  // this key part's share of dQ: [B][zs][N][CP], summed in part order by attn_dq_reduce_kernel
    float* pp = p.dqp + ((long)bz * N + qrow) * CP;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; r += 4)
        *reinterpret_cast<f32x4*>(pp + ct * 32 + 8 * (r >> 2) + 4 * lh) =
            (f32x4){acc[ct][r], acc[ct][r + 1], acc[ct][r + 2], acc[ct][r + 3]};
    return;
  }
  float* op = p.dq + (brow + qrow) * p.dq_cs;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = ct * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
      if (c < C) op[c] = acc[ct][r];
    }
}

// ---------------------------------------------------------------------------------------------- dK, dV
template <int CP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const BwdArgs p) {
  constexpr int QT = (CP >= 128) ? 32 : 64;   // queries per LDS tile
  constexpr int NSUB = QT / 32;
  constexpr int PS = CP + 4;
  constexpr int CT = (CP + 31) / 32;
  constexpr int QS = CP / 8;
  constexpr int F4 = CP / 4;
  constexpr int NF = (QT * F4 + 255) / 256;
  constexpr int TILE = 2 * QT * PS + 2 * QT;  // Q tile, dO tile, lse, D
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int bz = blockIdx.x / p.nt;  // workgroup -> (clip b, query part z, key block)
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = (blockIdx.x - bz * p.nt) * 128 + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];

  float kf[QS * 4], vf[QS * 4];
  const int jrow = j0 + li;
  const bool jok = jrow < N;
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = s * 8 + lh * 4 + e;
        const bool ok = jok && c < C;
        kf[s * 4 + e] = ok ? kp[c] * LOG2E : 0.f;
        vf[s * 4 + e] = ok ? vp[c] : 0.f;
      }
  }
  f32x16 dk[CT], dv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[ct][r] = 0.f; dv[ct][r] = 0.f; }

  f32x4 rq[NF], rd[NF];
  float rl = 0.f, rD = 0.f;
  auto load_tile = [&](int i0) {
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + u * 256;
      const int row = f / F4;
      const int c = (f - row * F4) * 4;
      const int i = i0 + row;
      f32x4 tq = {0.f, 0.f, 0.f, 0.f}, td = {0.f, 0.f, 0.f, 0.f};
      if (f < QT * F4 && i < N) {
        const float* qp = p.q + (brow + i) * p.q_cs + c;
        const float* dp = p.dz + (brow + i) * p.dz_cs + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((c + e) < C) {
            tq[e] = qp[e];
            td[e] = dp[e] * gamma;
          }
      }
      rq[u] = tq;
      rd[u] = td;
    }
    if (tid < QT) {
      const int i = i0 + tid;
      rl = (i < N) ? p.lse[brow + i] : POS_BIG;  // P = 2^(s - BIG) = 0 for padded queries
      rD = (i < N) ? p.dvec[brow + i] * gamma : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    float* Qs = smem + buf * TILE;
    float* Ds = Qs + QT * PS;
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + u * 256;
      if (f < QT * F4) {
        const int row = f / F4;
        const int c = (f - row * F4) * 4;
        *reinterpret_cast<f32x4*>(Qs + row * PS + c) = rq[u];
        *reinterpret_cast<f32x4*>(Ds + row * PS + c) = rd[u];
      }
    }
    if (tid < QT) {
      Qs[2 * QT * PS + tid] = rl;
      Qs[2 * QT * PS + QT + tid] = rD;
    }
  };

  const int tz = ((N + QT - 1) / QT + p.zs - 1) / p.zs;  // query tiles per part
  const int t0 = z * tz;
  const int ntiles = min((N + QT - 1) / QT, t0 + tz);
  load_tile(t0 * QT);
  store_tile(0);
  __syncthreads();
  for (int t = t0; t < ntiles; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = (t + 1) < ntiles;
    if (more) load_tile((t + 1) * QT);
    const float* Qs = smem + buf * TILE;
    const float* Ds = Qs + QT * PS;
    const float* Ls = Qs + 2 * QT * PS;
    const float* Dv = Ls + QT;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (t * QT + sub * 32 >= N) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const float* qrow = Qs + (sub * 32 + li) * PS + lh * 4;
      const float* drow = Ds + (sub * 32 + li) * PS + lh * 4;
#pragma unroll
      for (int g = 0; g < QS; ++g) {
        const f32x4 qa = *reinterpret_cast<const f32x4*>(qrow + g * 8);
        const f32x4 da = *reinterpret_cast<const f32x4*>(drow + g * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kf[g * 4 + e], s, 0, 0, 0);    // S' = Q K^T (rows i)
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da[e], vf[g * 4 + e], dp, 0, 0, 0);  // dP = dO V^T
        }
      }
      // rows of the accumulators are queries i = sub*32 + kappa(r, lh): fetch their LSE / D (LDS broadcast)
      f32x16 pr;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + sub * 32 + 8 * g4 + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(Dv + sub * 32 + 8 * g4 + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g4 + e;
          pr[r] = __builtin_amdgcn_exp2f(s[r] - l4[e]);
          s[r] = pr[r] * (dp[r] - d4[e]);  // dS
        }
      }
      const float* qcol = Qs + (sub * 32 + 4 * lh) * PS + (CP >= 32 ? li : (li & (CP - 1)));
      const float* dcol = Ds + (sub * 32 + 4 * lh) * PS + (CP >= 32 ? li : (li & (CP - 1)));
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int off = ((r & 3) + 8 * (r >> 2)) * PS + ct * 32;
          dv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(dcol[off], pr[r], dv[ct], 0, 0, 0);  // dV^T += dO^T P
          dk[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(qcol[off], s[r], dk[ct], 0, 0, 0);   // dK^T += Q^T dS
        }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (!jok) return;
  if (p.zs > 1) {
    float* okp = p.dkp + ((long)bz * N + jrow) * CP;
    float* ovp = p.dvp + ((long)bz * N + jrow) * CP;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const int c = ct * 32 + 8 * (r >> 2) + 4 * lh;
        *reinterpret_cast<f32x4*>(okp + c) = (f32x4){dk[ct][r], dk[ct][r + 1], dk[ct][r + 2], dk[ct][r + 3]};
        *reinterpret_cast<f32x4*>(ovp + c) = (f32x4){dv[ct][r], dv[ct][r + 1], dv[ct][r + 2], dv[ct][r + 3]};
      }
    return;
  }
  float* okp = p.dk + (brow + jrow) * p.dk_cs;
  float* ovp = p.dv + (brow + jrow) * p.dv_cs;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = ct * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
      if (c < C) {
        okp[c] = dk[ct][r];
        ovp[c] = dv[ct][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------- fused dK, dV, dQ
// One sweep instead of two: the dK/dV kernel above additionally forms this wavefront's share of dQ for every query
// tile (dQ_i += sum_{j in my 32 keys} dS_ij k_j), so S and dP are recomputed once — 5 products per tile in total
// instead of 3 + 4.  dS sits with the KEY on the lane, the dQ product needs the QUERY on the lane: each wavefront
// transposes its 32x32 dS tile through its own LDS slot (16 ds_write_b32 / 4 ds_read_b128, both conflict-free), K is
// held in the B-operand layout in registers for the whole sweep.  The four wavefronts' partials of a query tile are
// summed in a FIXED order through LDS and stored to a workspace plane per 128-key block; a second kernel sums the
// planes (no float atomics: results stay bit-reproducible; the workspace is 5 GB at N = 25 088, d = 32, B = 8).
template <int CP, int NW>
__global__ __launch_bounds__(64 * NW, (CP >= 64) ? 2 : 1) void attn_bwd_fused_kernel(const BwdArgs p, float* __restrict__ ws) {
  constexpr int NT = 64 * NW;                  // threads; the workgroup owns 32*NW keys
  constexpr int QT = (CP >= 64) ? 32 : 64;    // queries per LDS tile
  constexpr int NSUB = QT / 32;
  constexpr int PS = CP + 4;
  constexpr int CT = CP / 32;
  constexpr int QS = CP / 8;
  constexpr int F4 = CP / 4;
  constexpr int NF = (QT * F4 + NT - 1) / NT;
  constexpr int TILE = 2 * QT * PS + 2 * QT;  // Q tile, dO tile, lse, D
  constexpr int SLOT = QT * CP;               // one wavefront's dQ partial [QT][CP]
  constexpr int TP = 36;                      // pitch of the 32x32 transposition tile living in the slot
  static_assert(32 * TP <= SLOT, "transposition tile must fit the slot");
  static_assert(SLOT % 4 == 0, "flush uses float4");
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE + NW * SLOT];
  float* const slots = smem + 2 * TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int bz = blockIdx.x / p.nt;  // workgroup -> (clip b, query part z, key block kb)
  const int kb = blockIdx.x - bz * p.nt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = kb * (32 * NW) + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];
  float* const myslot = slots + wave * SLOT;

  float kf[QS * 4], vf[QS * 4];
  const int jrow = j0 + li;
  const bool jok = jrow < N;
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs;
#pragma unroll
    for (int s = 0; s < QS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = s * 8 + lh * 4 + e;
        const bool ok = jok && c < C;
        kf[s * 4 + e] = ok ? kp[c] * LOG2E : 0.f;
        vf[s * 4 + e] = ok ? vp[c] : 0.f;
      }
  }
  // K as the B operand of dQ = dS K: lane (c = li, half lh) holds k[key j0 + s + 16*lh][ct*32 + li] for step s
  float kbr[CT][16];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int key = j0 + s + 16 * lh, c = ct * 32 + li;
      kbr[ct][s] = (key < N && c < C) ? p.k[(brow + key) * p.k_cs + c] : 0.f;
    }
  f32x16 dk[CT], dv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[ct][r] = 0.f; dv[ct][r] = 0.f; }

  f32x4 rq[NF], rd[NF];
  float rl = 0.f, rD = 0.f;
  auto load_tile = [&](int i0) {
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + u * NT;
      const int row = f / F4;
      const int c = (f - row * F4) * 4;
      const int i = i0 + row;
      f32x4 tq = {0.f, 0.f, 0.f, 0.f}, td = {0.f, 0.f, 0.f, 0.f};
      if (f < QT * F4 && i < N) {
        const float* qp = p.q + (brow + i) * p.q_cs + c;
        const float* dp = p.dz + (brow + i) * p.dz_cs + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((c + e) < C) {
            tq[e] = qp[e];
            td[e] = dp[e] * gamma;
          }
      }
      rq[u] = tq;
      rd[u] = td;
    }
    if (tid < QT) {
      const int i = i0 + tid;
      rl = (i < N) ? p.lse[brow + i] : POS_BIG;  // P = 2^(s - BIG) = 0 for padded queries
      rD = (i < N) ? p.dvec[brow + i] * gamma : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    float* Qs = smem + buf * TILE;
    float* Ds = Qs + QT * PS;
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int f = tid + u * NT;
      if (f < QT * F4) {
        const int row = f / F4;
        const int c = (f - row * F4) * 4;
        *reinterpret_cast<f32x4*>(Qs + row * PS + c) = rq[u];
        *reinterpret_cast<f32x4*>(Ds + row * PS + c) = rd[u];
      }
    }
    if (tid < QT) {
      Qs[2 * QT * PS + tid] = rl;
      Qs[2 * QT * PS + QT + tid] = rD;
    }
  };

  // dQ partials, query-tile major: [b][query tile][key block][QT][CP] — the key blocks' partials of one query tile
  // sit in ONE contiguous run of nt * 8 KB, which is what attn_dq_reduce_tiled_kernel sums (plane-major, the 196
  // addends of an output were 3 MB apart: a TLB miss and a DRAM page per 16-byte load)
  const int nq = (N + QT - 1) / QT;
  const int tz = (nq + p.zs - 1) / p.zs;  // query tiles per part
  const int t0 = z * tz;
  const int ntiles = min((N + QT - 1) / QT, t0 + tz);
  load_tile(t0 * QT);
  store_tile(0);
  __syncthreads();
  for (int t = t0; t < ntiles; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = (t + 1) < ntiles;
    if (more) load_tile((t + 1) * QT);
    const float* Qs = smem + buf * TILE;
    const float* Ds = Qs + QT * PS;
    const float* Ls = Qs + 2 * QT * PS;
    const float* Dv = Ls + QT;
    f32x16 dqp[NSUB][CT];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqp[sub][ct][r] = 0.f;
      if (t * QT + sub * 32 >= N) continue;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      const float* qrow = Qs + (sub * 32 + li) * PS + lh * 4;
      const float* drow = Ds + (sub * 32 + li) * PS + lh * 4;
#pragma unroll
      for (int g = 0; g < QS; ++g) {
        const f32x4 qa = *reinterpret_cast<const f32x4*>(qrow + g * 8);
        const f32x4 da = *reinterpret_cast<const f32x4*>(drow + g * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kf[g * 4 + e], s, 0, 0, 0);    // S' = Q K^T (rows i)
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da[e], vf[g * 4 + e], dp, 0, 0, 0);  // dP = dO V^T
        }
      }
      f32x16 pr;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + sub * 32 + 8 * g4 + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(Dv + sub * 32 + 8 * g4 + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g4 + e;
          pr[r] = __builtin_amdgcn_exp2f(s[r] - l4[e]);
          s[r] = pr[r] * (dp[r] - d4[e]);  // dS
        }
      }
      const float* qcol = Qs + (sub * 32 + 4 * lh) * PS + li;
      const float* dcol = Ds + (sub * 32 + 4 * lh) * PS + li;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int off = ((r & 3) + 8 * (r >> 2)) * PS + ct * 32;
          dv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(dcol[off], pr[r], dv[ct], 0, 0, 0);  // dV^T += dO^T P
          dk[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(qcol[off], s[r], dk[ct], 0, 0, 0);   // dK^T += Q^T dS
        }
      // dS [query kappa(r,lh)][key li]  ->  LDS tile T[query][key]  ->  A fragments with the query on the lane
#pragma unroll
      for (int r = 0; r < 16; ++r) myslot[kappa(r, lh) * TP + li] = s[r];
      f32x4 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f32x4*>(myslot + li * TP + 16 * lh + 4 * j);
#pragma unroll
      for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          dqp[sub][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[st >> 2][st & 3], kbr[ct][st], dqp[sub][ct], 0, 0,
                                                              0);  // dQ(i, c) += dS(i, j) k(j, c)
    }
    // this wavefront's partial [QT][CP] into its slot (rows kappa(r,lh), channel on the lane)
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) myslot[(sub * 32 + kappa(r, lh)) * CP + ct * 32 + li] = dqp[sub][ct][r];
    if (more) store_tile(buf ^ 1);
    __syncthreads();
    // fixed-order sum of the NW wavefronts' partials to the plane
    for (int e0 = tid * 4; e0 < SLOT; e0 += NT * 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(slots + e0);
#pragma unroll
      for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(slots + w * SLOT + e0);
      *reinterpret_cast<f32x4*>(ws + ((((long)b * nq + t) * p.nt + kb) * (long)SLOT) + e0) = v;  // rows >= N: never read
    }
    __syncthreads();  // slots are reused as transposition tiles by the next iteration
  }
  if (!jok) return;
  if (p.zs > 1) {  // this query part's share of dK / dV; attn_dq_reduce_kernel adds the parts in order
    float* okp = p.dkp + ((long)bz * N + jrow) * CP;
    float* ovp = p.dvp + ((long)bz * N + jrow) * CP;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const int c = ct * 32 + 8 * (r >> 2) + 4 * lh;
        *reinterpret_cast<f32x4*>(okp + c) = (f32x4){dk[ct][r], dk[ct][r + 1], dk[ct][r + 2], dk[ct][r + 3]};
        *reinterpret_cast<f32x4*>(ovp + c) = (f32x4){dv[ct][r], dv[ct][r + 1], dv[ct][r + 2], dv[ct][r + 3]};
      }
    return;
  }
  float* okp = p.dk + (brow + jrow) * p.dk_cs;
  float* ovp = p.dv + (brow + jrow) * p.dv_cs;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = ct * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
      if (c < C) {
        okp[c] = dk[ct][r];
        ovp[c] = dv[ct][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------- fused, bf16 pieces
// The fused sweep above with its five products on the bf16 matrix pipe (attn_bx.h: every operand as three bf16
// pieces, six v_mfma_f32_32x32x16_bf16 per fp32 product), d <= 32, 16-byte aligned rows.  Same ownership — a
// wavefront owns 32 keys (on the lane), the workgroup sweeps 64-query tiles — and the same dQ hand-over (per-wavefront
// slots, fixed-order sum, one workspace plane per key block).  What changes:
//   * Q and dO = gamma dz arrive pre-split (sf_attn_bx_split row planes): a tile is six 4 KB pieces, staged with one
//     16-byte load + LDS store per thread and piece, no vector work.  ONE LDS image per tile serves both kinds of
//     read: rows (ds_read_b128) as the A operand of S' = Q K'^T and dP = dO V^T, columns (ds_read_b64_tr_b16, the
//     hardware transposing read) as the A operand of dV^T += dO^T P and dK^T += Q^T dS.
//   * -LSE and -D are the INITIAL accumulators of S' and dP, so P = exp2(S') and dS = P dP need no subtraction.
//   * K', V (B operands of the first two products) and K (B operand of dQ += dS K) are split once, in registers.
//   * P and dS are split in registers (the only per-element splits of the loop); dS crosses LDS once, as pieces: each
//     lane stores 4 consecutive queries of its key (8 bytes) into a [key][query] image and the dQ product reads it
//     back transposed with ds_read_b64_tr_b16.
// The tile is single-buffered (30 KB + 32 KB of slots: two workgroups per CU); the two barriers per tile are the ones
// the slot sum needs anyway.
constexpr int BXB_QT = 64;                        // queries per tile
constexpr int BXB_PL = BXB_QT * BX_KP;            // one piece of one tile in LDS (bf16 elements)
constexpr int BXB_TP = 36;                        // dS image row pitch (bf16): 72 B
// DBG (timing ablations, results invalid): 1 = P and dS are not split (leading piece three times), 2 = no dQ product
// (no image write / transposed read), 4 = one MFMA per product instead of six.
template <int DBG, class V>
__device__ __forceinline__ void bxb_split(float a, float b, V (&dst)[3], int idx) {
  if constexpr (DBG & 1) {
    dst[0][idx] = dst[1][idx] = dst[2][idx] = cvt_pk_bf16(a, b);
  } else {
    split_pair(a, b, dst, idx);
  }
}
template <int DBG>
__device__ __forceinline__ f32x16 bxb_mfma(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
  if constexpr (DBG & 4) return mfma_bf(a[0], b[0], c);
  return mfma_split(a, b, c);
}
// A workgroup barrier for LDS hand-overs only (__syncthreads() also drains vmcnt — here that would wait for the dQ plane
// store's acknowledgement once per tile): every LDS access of this wavefront has landed / been served, then the workgroup barrier; vmcnt is left alone
#define PP_BARRIER()                                                  \
  do {                                                                \
    __builtin_amdgcn_sched_barrier(0);                                \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                \
  } while (0)
constexpr int bxb_lds_bytes(int nw) { return 6 * BXB_PL * 2 + 2 * BXB_QT * 4 + nw * BXB_QT * 32 * 4; }
template <int DBG, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void attn_bwd_bx_kernel(const BwdArgs p, float* __restrict__ ws,
                                                                      const unsigned short* __restrict__ qb,
                                                                      const unsigned short* __restrict__ db, int n64) {
  constexpr int CP = 32, NT = 64 * NW, QT = BXB_QT, KP = BX_KP, PL = BXB_PL, TP = BXB_TP;
  constexpr int SLOT = QT * CP;                   // one wavefront's dQ partial [QT][CP] (floats)
  static_assert(3 * 32 * TP * 2 <= SLOT * 4, "the dS image lives in the slot");
  extern __shared__ __attribute__((aligned(16))) unsigned short bxb_smem[];
  unsigned short* const tile = bxb_smem;                                  // Q pieces 0..2, dO pieces 3..5
  float* const lsd = reinterpret_cast<float*>(bxb_smem + 6 * PL);         // -LSE, -D of the tile's queries
  float* const slots = lsd + 2 * QT;                                      // NW x SLOT

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int bz = blockIdx.x / p.nt;  // workgroup -> (clip b, query part z, key block kb)
  const int kb = blockIdx.x - bz * p.nt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = kb * (32 * NW) + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];
  float* const myslot = slots + wave * SLOT;
  unsigned short* const img = reinterpret_cast<unsigned short*>(myslot);  // [piece][key 32][TP]

  // ---- this wavefront's keys: K' = K log2(e) and V as B operands [k = channel][col = key], K as [k = key][col = channel]
  u32x4 kfb[2][3], vfb[2][3], kbr[2][3];
  const int jrow = j0 + li;
  const bool jok = jrow < N;
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs + 8 * lh;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs + 8 * lh;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
      if (jok && 16 * c + 8 * lh < C) {
        k0 = *reinterpret_cast<const f32x4*>(kp + 16 * c);
        v0 = *reinterpret_cast<const f32x4*>(vp + 16 * c);
      }
      if (jok && 16 * c + 8 * lh + 4 < C) {
        k1 = *reinterpret_cast<const f32x4*>(kp + 16 * c + 4);
        v1 = *reinterpret_cast<const f32x4*>(vp + 16 * c + 4);
      }
      k0 *= LOG2E;
      k1 *= LOG2E;
      split_pair(k0[0], k0[1], kfb[c], 0);
      split_pair(k0[2], k0[3], kfb[c], 1);
      split_pair(k1[0], k1[1], kfb[c], 2);
      split_pair(k1[2], k1[3], kfb[c], 3);
      split_pair(v0[0], v0[1], vfb[c], 0);
      split_pair(v0[2], v0[3], vfb[c], 1);
      split_pair(v1[0], v1[1], vfb[c], 2);
      split_pair(v1[2], v1[3], vfb[c], 3);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float kv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = j0 + 16 * m + 8 * lh + e;
        kv[e] = (key < N && li < C) ? p.k[(brow + key) * p.k_cs + li] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) split_pair(kv[2 * e], kv[2 * e + 1], kbr[m], e);
    }
  }
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }

  // ---- staging: a piece of a tile is 4 KB of its plane = one 16-byte element per thread
  const long plane = (long)n64 * QT * 32;
  // (8 wavefronts: the first four stage the Q pieces, the other four the dO pieces)
  const int st = tid & 255;
  const unsigned short* qg = ((NW == 8 && tid >= 256) ? db : qb) + (long)b * 3 * plane + st * 8;
  const unsigned short* dg = db + (long)b * 3 * plane + st * 8;
  const int st_off = (st >> 2) * KP + (st & 3) * 8 + ((NW == 8 && tid >= 256) ? 3 * PL : 0);
  u32x4 rq[3], rd[NW == 8 ? 1 : 3];
  float rl = 0.f, rD = 0.f;
  bool ld_ok = false;
  auto load_tile = [&](int t) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      rq[pc] = *reinterpret_cast<const u32x4*>(qg + pc * plane + (long)t * (QT * 32));
      if constexpr (NW == 4) rd[pc] = *reinterpret_cast<const u32x4*>(dg + pc * plane + (long)t * (QT * 32));
    }
    if (tid < QT) {  // raw loads only: a negation here made hipcc wait for vmcnt(0) — these loads, the tile's pieces and the
      const int i = t * QT + tid;  // previous tile's plane store — at the top of every tile (wavefront 0)
      ld_ok = i < N;
      rl = p.lse[brow + (ld_ok ? i : 0)];
      rD = p.dvec[brow + (ld_ok ? i : 0)];
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      *reinterpret_cast<u32x4*>(tile + pc * PL + st_off) = rq[pc];
      if constexpr (NW == 4) *reinterpret_cast<u32x4*>(tile + (3 + pc) * PL + st_off) = rd[pc];
    }
    if (tid < QT) {
      lsd[tid] = ld_ok ? -rl : -POS_BIG;  // P = 2^(s - BIG) = 0 for padded queries
      lsd[QT + tid] = ld_ok ? -rD * gamma : 0.f;
    }
  };
  // transposing reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the block
  const int tr_row = (lane & 15) >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nq = (N + QT - 1) / QT;
  const int tz = (nq + p.zs - 1) / p.zs;  // query tiles per part
  const int t0 = z * tz;
  const int ntiles = min(nq, t0 + tz);
  if (t0 < ntiles) {
    load_tile(t0);
    store_tile();
  }
  __syncthreads();
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0;  // DBG & 8: s_memtime per phase
  for (int t = t0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); st0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    if (more) load_tile(t + 1);
    // Both wavefronts of a SIMD leave the tile's barrier in the same phase (MFMAs of S' / dP first) and then fight for
    // the matrix pipe and the vector port in lock step; letting the older one (which wins every arbitration) start a
    // little later puts the pair in anti-phase: one in its MFMA stretch while the other splits / exponentiates.
    if (NW == 8 && wave < 4 && p.stagger > 0) {
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles each
    }
    f32x16 dqp[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dqp[sub][r] = 0.f;
      if (t * QT + sub * 32 >= N) continue;  // wave-uniform
      // ---- S' = Q K'^T - LSE,  dP = dO V^T - D   (rows = queries in registers, key on the lane)
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsd + sub * 32 + 8 * g4 + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(lsd + QT + sub * 32 + 8 * g4 + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[4 * g4 + e] = l4[e];
          dp[4 * g4 + e] = d4[e];
        }
      }
      const unsigned short* rowp = tile + (sub * 32 + li) * KP + 8 * lh;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 qa[3], da[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          qa[pc] = *reinterpret_cast<const u32x4*>(rowp + pc * PL + 16 * c);
          da[pc] = *reinterpret_cast<const u32x4*>(rowp + (3 + pc) * PL + 16 * c);
        }
        s = bxb_mfma<DBG>(qa, kfb[c], s);
        dp = bxb_mfma<DBG>(da, vfb[c], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);  // P
      // ---- dV^T += dO^T P : A = dO columns (transposed reads), B = P pieces (registers 8m..8m+7 = k-step m)
      const unsigned short* colp = tile + (sub * 32 + 4 * lh + tr_row) * KP + tr_col;
      {
        u32x4 pf[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) bxb_split<DBG>(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], pf[m], e);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          u32x4 a[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            const u32x2 lo = lds_read_tr(colp + (3 + pc) * PL + (16 * m) * KP);
            const u32x2 hi = lds_read_tr(colp + (3 + pc) * PL + (16 * m + 8) * KP);
            a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
          }
          dv = bxb_mfma<DBG>(a, pf[m], dv);
        }
      }
      // ---- dS = P dP;  dK^T += Q^T dS
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= dp[r];
      u32x4 sf[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) bxb_split<DBG>(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], sf[m], e);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        u32x4 a[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const u32x2 lo = lds_read_tr(colp + pc * PL + (16 * m) * KP);
          const u32x2 hi = lds_read_tr(colp + pc * PL + (16 * m + 8) * KP);
          a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
        dk = bxb_mfma<DBG>(a, sf[m], dk);
      }
      if constexpr (DBG & 2) continue;
      // ---- dS pieces -> [key][query] image (registers 4g..4g+3 = queries 8g + 4h .. +3 of this lane's key)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(img + (pc * 32 + li) * TP + 8 * g + 4 * lh) =
              (u32x2){sf[g >> 1][pc][2 * (g & 1)], sf[g >> 1][pc][2 * (g & 1) + 1]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- dQ(i, c) += dS(i, j) K(j, c) : A = dS with the query on the lane (transposed reads of the image)
      const unsigned short* imgp = img + (8 * lh + tr_row) * TP + tr_col;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        u32x4 a[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const u32x2 lo = lds_read_tr(imgp + (pc * 32 + 16 * m) * TP);
          const u32x2 hi = lds_read_tr(imgp + (pc * 32 + 16 * m + 4) * TP);
          a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
        dqp[sub] = bxb_mfma<DBG>(a, kbr[m], dqp[sub]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the image is rewritten by the next block
    }
    // this wavefront's partial [QT][CP] into its slot (rows kappa(r,lh), channel on the lane)
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); st1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) myslot[(sub * 32 + kappa(r, lh)) * CP + li] = dqp[sub][r];
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xc07f); st2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    PP_BARRIER();  // every wavefront is done with the tile and has its partial in place
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); st3 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    if (more) store_tile();
    for (int e0 = tid * 4; e0 < SLOT; e0 += NT * 4) {  // fixed-order sum of the NW partials to the plane
      f32x4 v = *reinterpret_cast<const f32x4*>(slots + e0);
#pragma unroll
      for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(slots + w * SLOT + e0);
      *reinterpret_cast<f32x4*>(ws + ((((long)b * nq + t) * p.nt + kb) * (long)SLOT) + e0) = v;  // rows >= N: never read
    }
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); st4 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    PP_BARRIER();
    if constexpr (DBG & 8) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long st5 = __builtin_amdgcn_s_memtime();
      st_acc[0] += st1 - st0; st_acc[1] += st2 - st1; st_acc[2] += st3 - st2; st_acc[3] += st4 - st3; st_acc[4] += st5 - st4;
    }
  }
  if constexpr (DBG & 8) {
    if ((blockIdx.x == 7 || blockIdx.x == 2000) && lane == 0 && (wave == 0 || wave == 5) && ntiles > t0) {
      const unsigned long long n = ntiles - t0;
      printf("bwd_bx stamps (workgroup %d wave %d, %llu tiles): per tile  compute %llu  slot writes %llu  barrier-1 wait %llu  "
             "stage + slot sum %llu  barrier-2 wait %llu  = %llu cycles\n", (int)blockIdx.x, wave, n, st_acc[0] / n,
             st_acc[1] / n, st_acc[2] / n, st_acc[3] / n, st_acc[4] / n,
             (st_acc[0] + st_acc[1] + st_acc[2] + st_acc[3] + st_acc[4]) / n);
    }
  }
  if (!jok) return;
  float* okp;
  float* ovp;
  if (p.zs > 1) {  // this query part's share of dK / dV; attn_dq_reduce_kernel adds the parts in order
    okp = p.dkp + ((long)bz * N + jrow) * CP;
    ovp = p.dvp + ((long)bz * N + jrow) * CP;
  } else {
    okp = p.dk + (brow + jrow) * p.dk_cs;
    ovp = p.dv + (brow + jrow) * p.dv_cs;
  }
#pragma unroll
  for (int r = 0; r < 16; r += 4) {
    const int c = 8 * (r >> 2) + 4 * lh;
    if (p.zs > 1 || c < C) {  // C % 4 == 0 on this path
      *reinterpret_cast<f32x4*>(okp + c) = (f32x4){dk[r], dk[r + 1], dk[r + 2], dk[r + 3]};
      *reinterpret_cast<f32x4*>(ovp + c) = (f32x4){dv[r], dv[r + 1], dv[r + 2], dv[r + 3]};
    }
  }
}

// ---------------------------------------------------------------------------------------------- fused, bf16 pieces, ping-pong
// attn_bwd_bx_kernel<., 8> re-scheduled (round 5).  Its stamps (SF_ATTN_BX_DBG=8) show where a 64-query tile's 12 850
// cycles go: the two wavefronts of a SIMD need 11 100 for 2 x 120 MFMAs = 7 680 cycles of the matrix pipe — every
// product waits for its fragment reads right in front of it, every exp2 / split for its product, and the older
// wavefront wins both pipes whenever it can issue — and 1 750 go to the dQ hand-over (slot stores, two barriers, the
// sum) with no MFMA in flight.  tools/microbench/mfma_pingpong.hip: a wavefront that alternates a pure MFMA segment
// with a pure vector segment, its SIMD partner one segment behind and s_setprio 1 inside MFMA segments, keeps the
// pipe 95-99 % busy.  So here the loop is cut into such segments, software-pipelined by one 32-query SUB-block k:
//   M(k):  back(k-1)  dV^T += dO^T P, dK^T += Q^T dS, dQ(k-1) = dS K   (36 MFMAs; P / dS pieces from registers, dS^T
//                     from the wavefront's LDS image), then
//          front(k)   S' = Q K'^T - LSE, dP = dO V^T - D                (24 MFMAs) — all fragment reads, no vector work;
//   V(k):  dQ(k-1) partial -> slot, P = exp2(S'), split, dS = P dP, split, dS pieces -> image; wavefronts 0-3 also sum
//          the eight slots of sub-block k-2 into the dQ plane, wavefronts 4-7 issue the LDS-DMA loads of sub-block k+2's Q / dO pieces.
// Wavefronts 4-7 (the SIMD partners of 0-3) run one segment behind; EVERY segment ends in one raw s_barrier (LDS
// counters drained, global loads / stores stay in flight), which is all the synchronisation there is: sub-block
// buffers are a ring of three (sub-block j is read in intervals 2j .. 2j+3 and refilled in 2j+4), slots alternate by
// sub-block parity, the image is private to its wavefront.  Same products, same accumulation order, same slot-sum
// order as attn_bwd_bx_kernel: the results are bit-identical to it (tests/test_attention_bx_gpu.py).
// Sub-block buffers are unpadded [piece 6][row 32][32 channels] with the 16-byte chunk index XORed by (row >> 2) & 3:
// conflict-free for the row reads (ds_read_b128), the transposing reads and the staging stores.
constexpr int PP_SUB = 32;                      // queries per sub-block
constexpr int PP_PIECE = PP_SUB * 32;           // one piece of one sub-block (bf16 elements; 2 KB)
constexpr int PP_TILE = 6 * PP_PIECE;           // Q pieces 0..2, dO pieces 3..5
constexpr int PP_IMG = 3 * 32 * BXB_TP;         // one wavefront's dS image (bf16 elements)
constexpr int PP_SLOT = PP_SUB * 32;            // one wavefront's dQ partial of a sub-block (floats)
constexpr int PP_LDS_BYTES = 3 * PP_TILE * 2 + 8 * PP_IMG * 2 + 8 * 2 * PP_SLOT * 4 + 3 * 2 * PP_SUB * 4;
static_assert(PP_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
// the barrier behind an M segment also drains vmcnt: the LDS-DMA pieces this wavefront issued at the start of its V
// segment, an interval ago, have landed before anybody reads their ring slot
#define PP_BARRIER_VM()                                                          \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
typedef __attribute__((address_space(3))) void lds_void;
template <int DBG>
__global__ __launch_bounds__(512, 1) void attn_bwd_bxpp_kernel(const BwdArgs p, float* __restrict__ ws,
                                                               const unsigned short* __restrict__ qb,
                                                               const unsigned short* __restrict__ db, int n64) {
  constexpr int CP = 32, TP = BXB_TP;
  // FOUR LDS objects, not one carved buffer: the ring is filled by LDS-DMA (buffer_load ... lds), whose completion is a
  // vmcnt event, and hipcc puts s_waitcnt vmcnt(0) in front of every LDS access that MAY touch what an outstanding
  // LDS-DMA writes — with one buffer that is every image / slot store of the segment that issued the loads (a whole
  // memory latency in the middle of it); distinct objects cannot alias.
  __shared__ __attribute__((aligned(16))) unsigned short tiles[3 * PP_TILE];     // ring of 3 sub-block buffers
  __shared__ __attribute__((aligned(16))) unsigned short imgs[8 * PP_IMG];       // [wave][piece][key 32][TP]
  __shared__ __attribute__((aligned(16))) float slots[8 * 2 * PP_SLOT];          // [wave][parity][PP_SLOT]
  __shared__ __attribute__((aligned(16))) float lsd[3 * 2 * PP_SUB];             // [buffer][-LSE 32 | -D 32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // 0: wavefronts 0-3, 1: their SIMD partners, one segment behind
  const int li = lane & 31, lh = lane >> 5;
  const int bz = blockIdx.x / p.nt;  // workgroup -> (clip b, query part z, key block kb)
  const int kb = blockIdx.x - bz * p.nt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = kb * 256 + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];
  float* const myslot = slots + wave * 2 * PP_SLOT;
  unsigned short* const img = imgs + wave * PP_IMG;

  // ---- this wavefront's keys: K' = K log2(e) and V as B operands [k = channel][col = key], K as [k = key][col = channel]
  u32x4 kfb[2][3], vfb[2][3], kbr[2][3];
  const int jrow = j0 + li;
  const bool jok = jrow < N;
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs + 8 * lh;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs + 8 * lh;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
      if (jok && 16 * c + 8 * lh < C) {
        k0 = *reinterpret_cast<const f32x4*>(kp + 16 * c);
        v0 = *reinterpret_cast<const f32x4*>(vp + 16 * c);
      }
      if (jok && 16 * c + 8 * lh + 4 < C) {
        k1 = *reinterpret_cast<const f32x4*>(kp + 16 * c + 4);
        v1 = *reinterpret_cast<const f32x4*>(vp + 16 * c + 4);
      }
      k0 *= LOG2E;
      k1 *= LOG2E;
      split_pair(k0[0], k0[1], kfb[c], 0);
      split_pair(k0[2], k0[3], kfb[c], 1);
      split_pair(k1[0], k1[1], kfb[c], 2);
      split_pair(k1[2], k1[3], kfb[c], 3);
      split_pair(v0[0], v0[1], vfb[c], 0);
      split_pair(v0[2], v0[3], vfb[c], 1);
      split_pair(v1[0], v1[1], vfb[c], 2);
      split_pair(v1[2], v1[3], vfb[c], 3);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float kv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = j0 + 16 * m + 8 * lh + e;
        kv[e] = (key < N && li < C) ? p.k[(brow + key) * p.k_cs + li] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) split_pair(kv[2 * e], kv[2 * e + 1], kbr[m], e);
    }
  }
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }

  // ---- the sub-blocks of this workgroup's query part
  const int nq = (N + 63) / 64;
  const int tz = (nq + p.zs - 1) / p.zs;  // 64-query tiles per part
  const int t0 = z * tz;
  const int ntiles = min(nq, t0 + tz);
  const int NS = t0 < ntiles ? 2 * (ntiles - t0) : 0;  // sub-blocks; sub-block j is rows (2 t0 + j) * 32 ..
  const int S0 = 2 * t0;

  // ---- staging: a sub-block is 6 pieces x 2 KB = twelve 1 KB LDS-DMA wave-instructions (buffer_load_dwordx4 ... lds:
  // no staging registers, no LDS stores), three per wavefront 4-7 at the start of V(k) for sub-block k+2: its ring slot
  // held sub-block k-1, last read in the interval before; the first reader comes two intervals later, behind the
  // vmcnt(0) barrier that ends the issuing wavefront's next M segment.  An LDS-DMA image is lane-linear, so the chunk
  // swizzle sits on the SOURCE address: lane l fills position (row l >> 2, chunk l & 3) of its 16 rows with source
  // chunk (l & 3) ^ ((row >> 2) & 3).
  const long plane = (long)n64 * 64 * 32;
  const __amdgpu_buffer_rsrc_t q_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(qb + (long)b * 3 * plane), 0, (unsigned)(3 * plane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t d_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(db + (long)b * 3 * plane), 0, (unsigned)(3 * plane * 2), 0x00020000);
  const unsigned dma_voff[2] = {  // halves: rows 0..15, 16..31
      (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4)),
      (unsigned)((16 + (lane >> 2)) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4))};
  // piece `id` (0..11: Q pieces x halves, then dO pieces x halves) of sub-block j into ring buffer buf
  auto stage_piece = [&](int j, int buf, int id) {
    const int which = id >= 6, pc = (id % 6) >> 1, half = id & 1;
    const unsigned so = (unsigned)(((long)pc * plane + (long)(S0 + j) * (PP_SUB * 32)) * 2);
    unsigned short* const dst = tiles + buf * PP_TILE + (which * 3 + pc) * PP_PIECE + half * 512;
    if (which)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rs, (lds_void*)dst, 16, dma_voff[half], so, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rs, (lds_void*)dst, 16, dma_voff[half], so, 0, 0);
  };
  // -LSE / -D of the sub-block's queries: lanes 0..31 of ONE wavefront — loaded in M (raw: nothing there may wait for
  // the loads), negated / scaled and stored in V
  float rl = 0.f, rD = 0.f;
  auto lsd_load = [&](int j) {
    const int i = (S0 + j) * PP_SUB + li;
    const long src = brow + (i < N ? i : 0);
    rl = p.lse[src];
    rD = p.dvec[src];
  };
  auto lsd_store = [&](int j, int buf) {
    if (lane < PP_SUB) {
      const bool ok = (S0 + j) * PP_SUB + li < N;
      lsd[buf * 64 + lane] = ok ? -rl : -POS_BIG;  // P = 2^(s - BIG) = 0 for padded queries
      lsd[buf * 64 + 32 + lane] = ok ? -rD * gamma : 0.f;
    }
  };
  // fragment addresses (bf16 elements inside a sub-block buffer)
  const int row_off0 = li * 32 + (((0 + lh) ^ ((li >> 2) & 3)) << 3);  // row fragment of channel chunk c = 0
  const int row_off1 = li * 32 + (((2 + lh) ^ ((li >> 2) & 3)) << 3);  //                              c = 1
  // transposing reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the block
  const int tr_row = (lane & 15) >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  // rows 16 m + 4 lh + tr_row (+ 8): (row >> 2) & 3 = lh (lh + 2)
  const int col_lo = (4 * lh + tr_row) * 32 + ((((tr_col >> 3) ^ lh) << 3) | (tr_col & 7));
  const int col_hi = (8 + 4 * lh + tr_row) * 32 + ((((tr_col >> 3) ^ (lh + 2)) << 3) | (tr_col & 7));
  const unsigned short* const imgp = img + (8 * lh + tr_row) * TP + tr_col;

  unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, sa = 0, sb = 0;  // DBG & 8: cycles in M, barrier, V, barrier; M's parts
  f32x16 s, dp, dqp;
  u32x4 pf[2][3], sf[2][3];
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;

  // ================================================================ M(k): back(k-1), then front(k); no vector work
  // Eight chains of 6 (12) MFMAs; the fragment reads of chain i+1 are issued in front of chain i's MFMAs and pinned there
  // (sched_barrier): the compiler's own order reads each chain's fragments right in front of it and waits for them
  // (~100-200 cycles of idle matrix pipe per chain — the partner wavefront is in its vector segment and cannot fill it).
#define PP_PIN() __builtin_amdgcn_sched_barrier(0)
  auto cols = [&](const unsigned short* base, int m, u32x4 (&a)[3]) {  // transposed reads: A = columns of a sub-block
    if constexpr (DBG & 128) return;  // timing ablation: no fragment reads
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      const u32x2 lo = lds_read_tr(base + pc * PP_PIECE + 512 * m + col_lo);
      const u32x2 hi = lds_read_tr(base + pc * PP_PIECE + 512 * m + col_hi);
      a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    }
  };
  auto imgf = [&](int m, u32x4 (&a)[3]) {  // transposed reads of this wavefront's dS image: A = dS, query on the lane
    if constexpr (DBG & 128) return;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      const u32x2 lo = lds_read_tr(imgp + (pc * 32 + 16 * m) * TP);
      const u32x2 hi = lds_read_tr(imgp + (pc * 32 + 16 * m + 4) * TP);
      a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    }
  };
  auto rows = [&](const unsigned short* tf, int c, u32x4 (&qa)[3], u32x4 (&da)[3]) {
    if constexpr (DBG & 128) return;
    const int ro = c ? row_off1 : row_off0;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      qa[pc] = *reinterpret_cast<const u32x4*>(tf + pc * PP_PIECE + ro);
      da[pc] = *reinterpret_cast<const u32x4*>(tf + (3 + pc) * PP_PIECE + ro);
    }
  };
  // S' and dP start from zero; -LSE and -D are added in V(k), which reads them there (eight 16-byte LDS reads that carry
  // 256 bytes of distinct data: in M they queued in front of the fragment reads the MFMAs wait for)
  auto init_acc = [&](int) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
  };
  auto mm = [&](const u32x4 (&a)[3], const u32x4 (&bb)[3], f32x16 c) -> f32x16 {
    if constexpr (DBG & 64) {  // timing ablation: no MFMAs (the fragments are still read)
      c[0] += __builtin_bit_cast(float, a[0][0] ^ a[1][1] ^ a[2][2] ^ a[0][3] ^ bb[0][0]);
      return c;
    } else {
      return mfma_split(a, bb, c);
    }
  };
  // dQ plane rows of sub-block j = fixed-order sum of the eight wavefronts' partials (256 threads, one float4 each)
  const int stid = tid & 255;
  f32x4 sum_v;
  auto sum_read = [&](int par, int w0, int w1) {
    const float* const sl = slots + par * PP_SLOT + stid * 4;
#pragma unroll
    for (int w = 0; w < 8; ++w)
      if (w >= w0 && w < w1) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(sl + w * 2 * PP_SLOT);
        sum_v = (w == 0) ? x : sum_v + x;
      }
  };
  auto sum_store = [&](int j) {  // absolute sub-block S0 + j: tile >> 1, rows (& 1) * 32 ..
    const int a = S0 + j;
    *reinterpret_cast<f32x4*>(ws + ((((long)b * nq + (a >> 1)) * p.nt + kb) * (long)(2 * PP_SLOT)) + (a & 1) * PP_SLOT +
                              stid * 4) = sum_v;  // rows >= N: never read
  };
  auto seg_m = [&](const int k, auto has_back, auto has_front) {
    constexpr bool BACK = decltype(has_back)::value, FRONT = decltype(has_front)::value;
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); sa = __builtin_amdgcn_s_memtime(); }
    if (wave == 4 && k + 2 < NS && !(DBG & 512) && !(DBG & 4096)) lsd_load(k + 2);  // raw; stored at the start of V(k)
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); st_acc[4] += __builtin_amdgcn_s_memtime() - sa; }
    if constexpr (!(DBG & 16)) __builtin_amdgcn_s_setprio(1);
    const unsigned short* const tb = tiles + ((k + 2) % 3) * PP_TILE;  // sub-block k-1
    const int bf = k % 3;
    const unsigned short* const tf = tiles + bf * PP_TILE;             // sub-block k
    u32x4 a0[3] = {}, a1[3] = {}, a2[3] = {}, qa0[3] = {}, da0[3] = {}, qa1[3] = {}, da1[3] = {};
    // fragment reads run TWO chains (384 matrix-pipe cycles) ahead of the MFMAs that take them: with the partner's vector
    // segment and the LDS stores of four other wavefronts in flight a read needs ~300 cycles, and one chain ahead left
    // every chain waiting (SF_ATTN_BX_DBG=128, no fragment reads: 9.1 -> 6.7 ms)
    // ... and they are dealt ONE per MFMA gap (sched_group_barrier: MFMA, LDS read, MFMA, LDS read ...): six reads in a
    // row in front of a chain hold the wavefront's issue past the previous MFMA's 32-cycle shadow
#define PP_MIX(n)                                              \
  do {                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {       \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       \
    }                                                          \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)
    if constexpr (BACK) {
      cols(tb + 3 * PP_PIECE, 0, a0);                                                     // dO columns, k-step 0
      cols(tb + 3 * PP_PIECE, 1, a1);                                                     //             k-step 1
      PP_PIN();
      cols(tb, 0, a2); dv = mm(a0, pf[0], dv); PP_MIX(6);                                  // dV^T += dO^T P
      cols(tb, 1, a0); dv = mm(a1, pf[1], dv); PP_MIX(6);
      imgf(0, a1);     dk = mm(a2, sf[0], dk); PP_MIX(6);                                  // dK^T += Q^T dS
      imgf(1, a2);     dk = mm(a0, sf[1], dk); PP_MIX(6);
      if constexpr (!(DBG & 1024)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dqp[r] = 0.f;
      }
      if constexpr (FRONT) rows(tf, 0, qa0, da0);
      dqp = mm(a1, kbr[0], dqp);                                                          // dQ(k-1) = dS K
      PP_MIX(6);
      if constexpr (FRONT) {
        init_acc(bf);
        rows(tf, 1, qa1, da1);
      }
      dqp = mm(a2, kbr[1], dqp);
      PP_MIX(6);
    } else {
      init_acc(bf);
      rows(tf, 0, qa0, da0);
      rows(tf, 1, qa1, da1);
      PP_PIN();
    }
    if constexpr (FRONT) {  // S' = Q K'^T,  dP = dO V^T   (queries in registers, key on the lane)
      s = mm(qa0, kfb[0], s);
      dp = mm(da0, vfb[0], dp);
      PP_PIN();
      s = mm(qa1, kfb[1], s);
      dp = mm(da1, vfb[1], dp);
    }
    if constexpr (!(DBG & 16)) __builtin_amdgcn_s_setprio(0);
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); sb = __builtin_amdgcn_s_memtime(); st_acc[0] += sb - sa; }
  };
  // ================================================================ V(k): everything that is not an MFMA
  auto seg_v = [&](const int k, auto has_back, auto has_front) {
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); sa = __builtin_amdgcn_s_memtime(); st_acc[1] += sa - sb; }
    if (grp == 1 && k + 2 < NS && !(DBG & 512) && !(DBG & 4096)) {  // sub-block k+2 into the ring (global work first: it has this whole
      const int buf = (k + 2) % 3;                  // segment and the next to land)
      if (wave == 4) lsd_store(k + 2, buf);  // BEFORE the pieces: vmcnt counts in order, and its loads are an interval old
      PP_PIN();
#pragma unroll
      for (int u = 0; u < 3; ++u) stage_piece(k + 2, buf, (wave - 4) * 3 + u);
    }
    if constexpr (decltype(has_back)::value) {  // dQ(k-1) partial [32 queries][CP] -> this wavefront's slot
      float* const sl = myslot + ((k - 1) & 1) * PP_SLOT;
#pragma unroll
      for (int r = 0; r < 16; ++r) sl[kappa(r, lh) * CP + li] = dqp[r];
    }
    if constexpr (decltype(has_front)::value && (DBG & 256)) {  // timing ablation: the probe's vector segment instead
      float f[8];
      for (int i = 0; i < 8; ++i) f[i] = s[i];
      const float bb = dp[0];
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
#pragma unroll
      for (int q = 0; q < 48; ++q) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[q & 7]) : "v"(bb));
#pragma unroll
      for (int q = 0; q < 150; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q & 7]) : "v"(bb));
      for (int i = 0; i < 8; ++i) pf[0][0][i & 3] ^= __builtin_bit_cast(unsigned, f[i]);
    }
    if constexpr (decltype(has_front)::value && !(DBG & 32) && !(DBG & 256)) {
      {
        const int bf = k % 3;
        f32x4 l4[4], d4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          l4[g4] = *reinterpret_cast<const f32x4*>(lsd + bf * 64 + 8 * g4 + 4 * lh);
          d4[g4] = *reinterpret_cast<const f32x4*>(lsd + bf * 64 + 32 + 8 * g4 + 4 * lh);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] + l4[r >> 2][r & 3]);  // P = 2^(S' - LSE)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = (dp[r] + d4[r >> 2][r & 3]) * s[r];  // dS = P (dP - D)
      }
      // Both three-way splits STAGE by stage over all 16 pairs (in place: s and dp end as the third residuals): the
      // partner wavefront is in its MFMA segment and covers nothing, so a pair's own chain (convert -> expand ->
      // subtract -> convert ...: seven dependent steps) must not be what the vector pipe waits for — the compiler's
      // order interleaved two pairs and the segment took 1 900 cycles with the matrix pipe idle beside it.
#pragma unroll
      for (int st = 0; st < 3; ++st) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pf[m][st][e] = cvt_pk_bf16(s[8 * m + 2 * e], s[8 * m + 2 * e + 1]);
            sf[m][st][e] = cvt_pk_bf16(dp[8 * m + 2 * e], dp[8 * m + 2 * e + 1]);
          }
        PP_PIN();
        if (st < 2) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s[8 * m + 2 * e] = bx_res_lo(s[8 * m + 2 * e], pf[m][st][e]);
              s[8 * m + 2 * e + 1] = bx_res_hi(s[8 * m + 2 * e + 1], pf[m][st][e]);
              dp[8 * m + 2 * e] = bx_res_lo(dp[8 * m + 2 * e], sf[m][st][e]);
              dp[8 * m + 2 * e + 1] = bx_res_hi(dp[8 * m + 2 * e + 1], sf[m][st][e]);
            }
          PP_PIN();
        }
      }
      // dS pieces -> [key][query] image (registers 4g..4g+3 = queries 8g + 4h .. +3 of this lane's key)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(img + (pc * 32 + li) * TP + 8 * g + 4 * lh) =
              (u32x2){sf[g >> 1][pc][2 * (g & 1)], sf[g >> 1][pc][2 * (g & 1) + 1]};
    }
    if (grp == 0 && k >= 2 && !(DBG & 512) && !(DBG & 2048)) {  // sub-block k-2's eight dQ partials (complete since the barrier before this
      sum_read(k & 1, 0, 8);                   // segment) -> its rows of the dQ plane
      sum_store(k - 2);
    }
    if constexpr (DBG & 8) { __builtin_amdgcn_sched_barrier(0); sb = __builtin_amdgcn_s_memtime(); st_acc[2] += sb - sa; }
  };
  auto sync = [&]() {
    PP_BARRIER();
    if constexpr (DBG & 8) { sa = __builtin_amdgcn_s_memtime(); st_acc[3] += sa - sb; sb = sa; }
  };

  if (NS > 0) {  // (every wavefront of the workgroup has the same NS: the barrier counts agree)
    {  // prologue: sub-blocks 0 and 1 (twelve pieces each, three per wavefront)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int id = wave * 3 + u;  // 0..23
        stage_piece(id / 12, id / 12, id % 12);
      }
      if (wave < 2) {
        lsd_load(wave);
        lsd_store(wave, wave);
      }
    }
    PP_BARRIER_VM();
    if (grp == 1) PP_BARRIER();  // one segment behind
    seg_m(0, no_t(), yes_t());
    PP_BARRIER_VM();
    seg_v(0, no_t(), yes_t());
    sync();
    for (int k = 1; k < NS; ++k) {
      seg_m(k, yes_t(), yes_t());
      PP_BARRIER_VM();
      seg_v(k, yes_t(), yes_t());
      sync();
    }
    seg_m(NS, yes_t(), no_t());
    PP_BARRIER_VM();
    seg_v(NS, yes_t(), no_t());
    sync();
    if (grp == 0) {
      PP_BARRIER();
      // the last sub-block's partials (written in V(NS))
      sum_read((NS - 1) & 1, 0, 8);
      sum_store(NS - 1);
    }
  }
  if constexpr (DBG & 8) {
    if ((blockIdx.x == 7 || blockIdx.x == 2000) && lane == 0 && (wave == 0 || wave == 5) && NS > 0) {
      const unsigned long long n = NS + 1;
      printf("bwd_bxpp stamps (workgroup %d wave %d, %d sub-blocks): per sub-block  M %llu (staging stores until %llu, bookkeeping until %llu, back "
             "until %llu)  barrier %llu  V %llu  barrier %llu  = %llu cycles\n", (int)blockIdx.x, wave, NS, st_acc[0] / n,
             st_acc[6] / n, st_acc[4] / n, st_acc[5] / n, st_acc[1] / n, st_acc[2] / n, st_acc[3] / n,
             (st_acc[0] + st_acc[1] + st_acc[2] + st_acc[3]) / n);
    }
  }
  if (!jok) return;
  float* okp;
  float* ovp;
  if (p.zs > 1) {  // this query part's share of dK / dV; attn_dq_reduce_kernel adds the parts in order
    okp = p.dkp + ((long)bz * N + jrow) * CP;
    ovp = p.dvp + ((long)bz * N + jrow) * CP;
  } else {
    okp = p.dk + (brow + jrow) * p.dk_cs;
    ovp = p.dv + (brow + jrow) * p.dv_cs;
  }
#pragma unroll
  for (int r = 0; r < 16; r += 4) {
    const int c = 8 * (r >> 2) + 4 * lh;
    if (p.zs > 1 || c < C) {  // C % 4 == 0 on this path
      *reinterpret_cast<f32x4*>(okp + c) = (f32x4){dk[r], dk[r + 1], dk[r + 2], dk[r + 3]};
      *reinterpret_cast<f32x4*>(ovp + c) = (f32x4){dv[r], dv[r + 1], dv[r + 2], dv[r + 3]};
    }
  }
}

// ---------------------------------------------------------------------------------------------- fused, bf16 pieces, d <= 64
// attn_bwd_bx_kernel for 33..64 channels: the channels are two 32-wide BLOCKS, each with its own plane set (the split
// launch runs once per block), K' / V / K register operands, dK^T / dV^T accumulators and dQ column tile; S' and dP
// sum over both blocks' k-steps.  24 + 24 + 24 + 24 + 24 = 120 MFMAs per 32 x 32 block of the score matrix.  The
// register operands alone are 144 VGPRs, so this form runs one wavefront per SIMD (one 256-thread workgroup per CU,
// 125 KB of LDS: two blocks' images + four 16 KB dQ slots).
constexpr int BXB2_LDS_BYTES = 2 * 6 * BXB_PL * 2 + 2 * BXB_QT * 4 + 4 * BXB_QT * 64 * 4;
__global__ __launch_bounds__(256, 1) void attn_bwd_bx2_kernel(const BwdArgs p, float* __restrict__ ws,
                                                              const unsigned short* __restrict__ qb,
                                                              const unsigned short* __restrict__ db, int n64,
                                                              long blk_elems) {
  constexpr int CP = 64, NW = 4, QT = BXB_QT, KP = BX_KP, PL = BXB_PL, TP = BXB_TP;
  constexpr int SLOT = QT * CP;
  extern __shared__ __attribute__((aligned(16))) unsigned short bxb_smem[];
  unsigned short* const tile = bxb_smem;                                  // [block][Q pieces 0..2, dO pieces 3..5][QT][KP]
  float* const lsd = reinterpret_cast<float*>(bxb_smem + 2 * 6 * PL);
  float* const slots = lsd + 2 * QT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int bz = blockIdx.x / p.nt;
  const int kb = blockIdx.x - bz * p.nt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = kb * (32 * NW) + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];
  float* const myslot = slots + wave * SLOT;
  unsigned short* const img = reinterpret_cast<unsigned short*>(myslot);  // [piece][key 32][TP]

  u32x4 kfb[2][2][3], vfb[2][2][3], kbr[2][2][3];
  const int jrow = j0 + li;
  const bool jok = jrow < N;
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs + 8 * lh;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs + 8 * lh;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ch = 32 * blk + 16 * c + 8 * lh;
        f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
        if (jok && ch < C) {
          k0 = *reinterpret_cast<const f32x4*>(kp + 32 * blk + 16 * c);
          v0 = *reinterpret_cast<const f32x4*>(vp + 32 * blk + 16 * c);
        }
        if (jok && ch + 4 < C) {
          k1 = *reinterpret_cast<const f32x4*>(kp + 32 * blk + 16 * c + 4);
          v1 = *reinterpret_cast<const f32x4*>(vp + 32 * blk + 16 * c + 4);
        }
        k0 *= LOG2E;
        k1 *= LOG2E;
        split_pair(k0[0], k0[1], kfb[blk][c], 0);
        split_pair(k0[2], k0[3], kfb[blk][c], 1);
        split_pair(k1[0], k1[1], kfb[blk][c], 2);
        split_pair(k1[2], k1[3], kfb[blk][c], 3);
        split_pair(v0[0], v0[1], vfb[blk][c], 0);
        split_pair(v0[2], v0[3], vfb[blk][c], 1);
        split_pair(v1[0], v1[1], vfb[blk][c], 2);
        split_pair(v1[2], v1[3], vfb[blk][c], 3);
      }
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float kv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int key = j0 + 16 * m + 8 * lh + e;
          kv[e] = (key < N && 32 * blk + li < C) ? p.k[(brow + key) * p.k_cs + 32 * blk + li] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) split_pair(kv[2 * e], kv[2 * e + 1], kbr[blk][m], e);
      }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[blk][r] = 0.f; dv[blk][r] = 0.f; }

  // ---- staging: per block and piece 4 KB of each plane = one 16-byte element per thread
  const long plane = (long)n64 * QT * 32;
  const unsigned short* qg = qb + (long)b * 3 * plane + tid * 8;
  const unsigned short* dg = db + (long)b * 3 * plane + tid * 8;
  const int st_off = (tid >> 2) * KP + (tid & 3) * 8;
  u32x4 rq[2][3], rd[2][3];
  float rl = 0.f, rD = 0.f;
  bool ld_ok = false;
  auto load_tile = [&](int t) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        rq[blk][pc] = *reinterpret_cast<const u32x4*>(qg + blk * blk_elems + pc * plane + (long)t * (QT * 32));
        rd[blk][pc] = *reinterpret_cast<const u32x4*>(dg + blk * blk_elems + pc * plane + (long)t * (QT * 32));
      }
    if (tid < QT) {  // raw loads only (no arithmetic on them here: see attn_bwd_bx_kernel)
      const int i = t * QT + tid;
      ld_ok = i < N;
      rl = p.lse[brow + (ld_ok ? i : 0)];
      rD = p.dvec[brow + (ld_ok ? i : 0)];
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        *reinterpret_cast<u32x4*>(tile + (blk * 6 + pc) * PL + st_off) = rq[blk][pc];
        *reinterpret_cast<u32x4*>(tile + (blk * 6 + 3 + pc) * PL + st_off) = rd[blk][pc];
      }
    if (tid < QT) {
      lsd[tid] = ld_ok ? -rl : -POS_BIG;  // P = 2^(s - BIG) = 0 for padded queries
      lsd[QT + tid] = ld_ok ? -rD * gamma : 0.f;
    }
  };
  const int tr_row = (lane & 15) >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nq = (N + QT - 1) / QT;
  const int tz = (nq + p.zs - 1) / p.zs;
  const int t0 = z * tz;
  const int ntiles = min(nq, t0 + tz);
  if (t0 < ntiles) {
    load_tile(t0);
    store_tile();
  }
  __syncthreads();
  for (int t = t0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) load_tile(t + 1);
    f32x16 dqp[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqp[sub][blk][r] = 0.f;
      if (t * QT + sub * 32 >= N) continue;  // wave-uniform
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsd + sub * 32 + 8 * g4 + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(lsd + QT + sub * 32 + 8 * g4 + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[4 * g4 + e] = l4[e];
          dp[4 * g4 + e] = d4[e];
        }
      }
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const unsigned short* rowp = tile + blk * 6 * PL + (sub * 32 + li) * KP + 8 * lh;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          u32x4 qa[3], da[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            qa[pc] = *reinterpret_cast<const u32x4*>(rowp + pc * PL + 16 * c);
            da[pc] = *reinterpret_cast<const u32x4*>(rowp + (3 + pc) * PL + 16 * c);
          }
          s = mfma_split(qa, kfb[blk][c], s);
          dp = mfma_split(da, vfb[blk][c], dp);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);  // P
      {
        u32x4 pf[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) split_pair(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], pf[m], e);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const unsigned short* colp = tile + blk * 6 * PL + (sub * 32 + 4 * lh + tr_row) * KP + tr_col;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            u32x4 a[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
              const u32x2 lo = lds_read_tr(colp + (3 + pc) * PL + (16 * m) * KP);
              const u32x2 hi = lds_read_tr(colp + (3 + pc) * PL + (16 * m + 8) * KP);
              a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            }
            dv[blk] = mfma_split(a, pf[m], dv[blk]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= dp[r];  // dS
      u32x4 sf[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) split_pair(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], sf[m], e);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const unsigned short* colp = tile + blk * 6 * PL + (sub * 32 + 4 * lh + tr_row) * KP + tr_col;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          u32x4 a[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            const u32x2 lo = lds_read_tr(colp + pc * PL + (16 * m) * KP);
            const u32x2 hi = lds_read_tr(colp + pc * PL + (16 * m + 8) * KP);
            a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
          }
          dk[blk] = mfma_split(a, sf[m], dk[blk]);
        }
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(img + (pc * 32 + li) * TP + 8 * g + 4 * lh) =
              (u32x2){sf[g >> 1][pc][2 * (g & 1)], sf[g >> 1][pc][2 * (g & 1) + 1]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned short* imgp = img + (8 * lh + tr_row) * TP + tr_col;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        u32x4 a[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const u32x2 lo = lds_read_tr(imgp + (pc * 32 + 16 * m) * TP);
          const u32x2 hi = lds_read_tr(imgp + (pc * 32 + 16 * m + 4) * TP);
          a[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) dqp[sub][blk] = mfma_split(a, kbr[blk][m], dqp[sub][blk]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) myslot[(sub * 32 + kappa(r, lh)) * CP + 32 * blk + li] = dqp[sub][blk][r];
    __syncthreads();
    if (more) store_tile();
    for (int e0 = tid * 4; e0 < SLOT; e0 += 256 * 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(slots + e0);
#pragma unroll
      for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(slots + w * SLOT + e0);
      *reinterpret_cast<f32x4*>(ws + ((((long)b * nq + t) * p.nt + kb) * (long)SLOT) + e0) = v;
    }
    __syncthreads();
  }
  if (!jok) return;
  float* okp;
  float* ovp;
  if (p.zs > 1) {
    okp = p.dkp + ((long)bz * N + jrow) * CP;
    ovp = p.dvp + ((long)bz * N + jrow) * CP;
  } else {
    okp = p.dk + (brow + jrow) * p.dk_cs;
    ovp = p.dv + (brow + jrow) * p.dv_cs;
  }
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
      const int c = 32 * blk + 8 * (r >> 2) + 4 * lh;
      if (p.zs > 1 || c < C) {
        *reinterpret_cast<f32x4*>(okp + c) = (f32x4){dk[blk][r], dk[blk][r + 1], dk[blk][r + 2], dk[blk][r + 3]};
        *reinterpret_cast<f32x4*>(ovp + c) = (f32x4){dv[blk][r], dv[blk][r + 1], dv[blk][r + 2], dv[blk][r + 3]};
      }
    }
}

// ---------------------------------------------------------------------------------------------- fused, packed planes
// The sweep above for C <= 8 on packed planes (attn_bx.h): one 32-column plane per operand holds the three pieces of
// its 8 channels.  S' and dP are three MFMAs each ([q1|q1][k1|k2], [q2|q2][k1|k2], [q1|q3][k3|k1]); dV^T, dK^T and dQ
// are three per k-step (the packed operand times each piece of P / dS; row or column groups 0-7 / 8-15 / 16-23 summed
// at the end).  24 MFMAs per 32 x 32 block instead of 60, a third of the LDS row traffic, and dQ planes of 8 columns.
constexpr int bxbp_lds_bytes(int nw) { return 2 * BXB_PL * 2 + 2 * BXB_QT * 4 + nw * BXB_QT * 32 * 4; }
// -DSF_BXP_OCC3=1: three workgroups of four wavefronts per CU (168 registers; the kernel needs 177, so 8 dwords go to
// scratch, touched once per tile).  Round 6, d = 8 N = 25 088 B = 8: backward 4.64-4.66 ms against 4.78-4.90 (-4 %);
// not the default because a hot kernel with scratch fails the build's ISA audit and 0.2 ms is not worth a waiver.
#ifndef SF_BXP_OCC3
#define SF_BXP_OCC3 0
#endif
#define BXP_WGS(nw) ((SF_BXP_OCC3 && (nw) == 4) ? 3 : 8 / (nw))
template <int NW>
__global__ __launch_bounds__(64 * NW, BXP_WGS(NW)) void attn_bwd_bxp_kernel(const BwdArgs p, float* __restrict__ ws,
                                                                       const unsigned short* __restrict__ qb,
                                                                       const unsigned short* __restrict__ db, int n64) {
  constexpr int CP = 8, NT = 64 * NW, QT = BXB_QT, KP = BX_KP, PL = BXB_PL, TP = BXB_TP;
  constexpr int SLOT = QT * 32;                   // one wavefront's dQ partial [QT][32 packed columns] (floats)
  extern __shared__ __attribute__((aligned(16))) unsigned short bxb_smem[];
  unsigned short* const tile = bxb_smem;                                  // Q packed rows, then dO packed rows
  float* const lsd = reinterpret_cast<float*>(bxb_smem + 2 * PL);         // -LSE, -D of the tile's queries
  float* const slots = lsd + 2 * QT;                                      // NW x SLOT

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int bz = blockIdx.x / p.nt;
  const int kb = blockIdx.x - bz * p.nt;
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int j0 = kb * (32 * NW) + wave * 32;
  const int N = p.N, C = p.C;
  const long brow = (long)b * N;
  const float gamma = p.gamma[0];
  float* const myslot = slots + wave * SLOT;
  unsigned short* const img = reinterpret_cast<unsigned short*>(myslot);  // [piece][key 32][TP]

  // ---- this wavefront's keys: [k1|k2], [k3|k1] of K' = K log2(e) and of V (B operands of S', dP), K packed columns
  u32x4 ka, kc, va, vc, kbr[2];
  const int jrow = j0 + li;
  const bool jok = jrow < N;
  {
    const float* kp = p.k + (brow + (jok ? jrow : 0)) * p.k_cs;
    const float* vp = p.v + (brow + (jok ? jrow : 0)) * p.v_cs;
    f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, v0 = k0, v1 = k0;
    if (jok) {
      k0 = *reinterpret_cast<const f32x4*>(kp);
      v0 = *reinterpret_cast<const f32x4*>(vp);
      if (C > 4) {
        k1 = *reinterpret_cast<const f32x4*>(kp + 4);
        v1 = *reinterpret_cast<const f32x4*>(vp + 4);
      }
    }
    k0 *= LOG2E;
    k1 *= LOG2E;
    u32x4 wk[3], wv[3];
    split_pair(k0[0], k0[1], wk, 0);
    split_pair(k0[2], k0[3], wk, 1);
    split_pair(k1[0], k1[1], wk, 2);
    split_pair(k1[2], k1[3], wk, 3);
    split_pair(v0[0], v0[1], wv, 0);
    split_pair(v0[2], v0[3], wv, 1);
    split_pair(v1[0], v1[1], wv, 2);
    split_pair(v1[2], v1[3], wv, 3);
    ka = lh ? wk[1] : wk[0];
    kc = lh ? wk[0] : wk[2];
    va = lh ? wv[1] : wv[0];
    vc = lh ? wv[0] : wv[2];
    const int pc = li >> 3, ch = li & 7;  // this lane's packed column
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float kv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = j0 + 16 * m + 8 * lh + e;
        kv[e] = (key < N && ch < C && pc < 3) ? p.k[(brow + key) * p.k_cs + ch] : 0.f;
      }
      u32x4 w[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_pair(kv[2 * e], kv[2 * e + 1], w, e);
      kbr[m] = pc == 0 ? w[0] : (pc == 1 ? w[1] : w[2]);  // pc == 3: kv = 0
    }
  }
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }

  // ---- staging: a tile is 4 KB of either plane = one 16-byte element per thread (8 wavefronts: half each)
  const int st = tid & 255;
  const bool st_d = NW == 8 && tid >= 256;
  const unsigned short* qg = (st_d ? db : qb) + (long)b * n64 * (QT * 32) + st * 8;
  const unsigned short* dg = db + (long)b * n64 * (QT * 32) + st * 8;
  const int st_off = (st >> 2) * KP + (st & 3) * 8 + (st_d ? PL : 0);
  u32x4 rq, rd;
  float rl = 0.f, rD = 0.f;
  bool ld_ok = false;
  auto load_tile = [&](int t) {
    rq = *reinterpret_cast<const u32x4*>(qg + (long)t * (QT * 32));
    if constexpr (NW == 4) rd = *reinterpret_cast<const u32x4*>(dg + (long)t * (QT * 32));
    if (tid < QT) {  // raw loads only (no arithmetic on them here: see attn_bwd_bx_kernel)
      const int i = t * QT + tid;
      ld_ok = i < N;
      rl = p.lse[brow + (ld_ok ? i : 0)];
      rD = p.dvec[brow + (ld_ok ? i : 0)];
    }
  };
  auto store_tile = [&]() {
    *reinterpret_cast<u32x4*>(tile + st_off) = rq;
    if constexpr (NW == 4) *reinterpret_cast<u32x4*>(tile + PL + st_off) = rd;
    if (tid < QT) {
      lsd[tid] = ld_ok ? -rl : -POS_BIG;  // P = 2^(s - BIG) = 0 for padded queries
      lsd[QT + tid] = ld_ok ? -rD * gamma : 0.f;
    }
  };
  const int tr_row = (lane & 15) >> 2, tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nq = (N + QT - 1) / QT;
  const int tz = (nq + p.zs - 1) / p.zs;
  const int t0 = z * tz;
  const int ntiles = min(nq, t0 + tz);
  if (t0 < ntiles) {
    load_tile(t0);
    store_tile();
  }
  __syncthreads();
  for (int t = t0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) load_tile(t + 1);
    f32x16 dqp[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dqp[sub][r] = 0.f;
      if (t * QT + sub * 32 >= N) continue;  // wave-uniform
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsd + sub * 32 + 8 * g4 + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(lsd + QT + sub * 32 + 8 * g4 + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[4 * g4 + e] = l4[e];
          dp[4 * g4 + e] = d4[e];
        }
      }
      {  // S' = Q K'^T - LSE, dP = dO V^T - D: [x1|x3][y3|y1], [x2|x2][y1|y2], [x1|x1][y1|y2]
        const unsigned short* rowp = tile + (sub * 32 + li) * KP;
        const int o3 = lh ? 16 : 0;
        const u32x4 q1 = *reinterpret_cast<const u32x4*>(rowp), q2 = *reinterpret_cast<const u32x4*>(rowp + 8),
                    q3 = *reinterpret_cast<const u32x4*>(rowp + o3);
        const u32x4 d1 = *reinterpret_cast<const u32x4*>(rowp + PL), d2 = *reinterpret_cast<const u32x4*>(rowp + PL + 8),
                    d3 = *reinterpret_cast<const u32x4*>(rowp + PL + o3);
        s = mfma_bf(q3, kc, s);
        dp = mfma_bf(d3, vc, dp);
        s = mfma_bf(q2, ka, s);
        dp = mfma_bf(d2, va, dp);
        s = mfma_bf(q1, ka, s);
        dp = mfma_bf(d1, va, dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);  // P
      const unsigned short* colp = tile + (sub * 32 + 4 * lh + tr_row) * KP + tr_col;
      {  // dV^T (packed rows) += dO^T P
        u32x4 pf[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) split_pair(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], pf[m], e);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const u32x2 lo = lds_read_tr(colp + PL + (16 * m) * KP);
          const u32x2 hi = lds_read_tr(colp + PL + (16 * m + 8) * KP);
          const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
          dv = mfma_bf(a, pf[m][2], dv);
          dv = mfma_bf(a, pf[m][1], dv);
          dv = mfma_bf(a, pf[m][0], dv);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= dp[r];  // dS
      u32x4 sf[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) split_pair(s[8 * m + 2 * e], s[8 * m + 2 * e + 1], sf[m], e);
#pragma unroll
      for (int m = 0; m < 2; ++m) {  // dK^T (packed rows) += Q^T dS
        const u32x2 lo = lds_read_tr(colp + (16 * m) * KP);
        const u32x2 hi = lds_read_tr(colp + (16 * m + 8) * KP);
        const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
        dk = mfma_bf(a, sf[m][2], dk);
        dk = mfma_bf(a, sf[m][1], dk);
        dk = mfma_bf(a, sf[m][0], dk);
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(img + (pc * 32 + li) * TP + 8 * g + 4 * lh) =
              (u32x2){sf[g >> 1][pc][2 * (g & 1)], sf[g >> 1][pc][2 * (g & 1) + 1]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned short* imgp = img + (8 * lh + tr_row) * TP + tr_col;
#pragma unroll
      for (int m = 0; m < 2; ++m)  // dQ (packed columns) += dS K
#pragma unroll
        for (int pc = 2; pc >= 0; --pc) {
          const u32x2 lo = lds_read_tr(imgp + (pc * 32 + 16 * m) * TP);
          const u32x2 hi = lds_read_tr(imgp + (pc * 32 + 16 * m + 4) * TP);
          dqp[sub] = mfma_bf((u32x4){lo[0], lo[1], hi[0], hi[1]}, kbr[m], dqp[sub]);
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) myslot[(sub * 32 + kappa(r, lh)) * 32 + li] = dqp[sub][r];
    __syncthreads();
    if (more) store_tile();
    if (tid < QT * CP / 4) {  // fixed-order sum of the NW partials and of the three column groups -> plane [QT][8]
      const int row = tid >> 1, c0 = (tid & 1) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int g = 0; g < 3; ++g) v += *reinterpret_cast<const f32x4*>(slots + w * SLOT + row * 32 + g * 8 + c0);
      *reinterpret_cast<f32x4*>(ws + ((((long)b * nq + t) * p.nt + kb) * (long)(QT * CP)) + row * CP + c0) = v;
    }
    __syncthreads();
  }
  if (!jok) return;
  float* okp;
  float* ovp;
  if (p.zs > 1) {
    okp = p.dkp + ((long)bz * N + jrow) * CP;
    ovp = p.dvp + ((long)bz * N + jrow) * CP;
  } else {
    okp = p.dk + (brow + jrow) * p.dk_cs;
    ovp = p.dv + (brow + jrow) * p.dv_cs;
  }
  const int c = 4 * lh;  // channels (r & 3) + 4h live in registers r, r + 4, r + 8 (packed row groups)
  if (p.zs > 1 || c < C) {
    *reinterpret_cast<f32x4*>(okp + c) = (f32x4){dk[0] + dk[4] + dk[8], dk[1] + dk[5] + dk[9], dk[2] + dk[6] + dk[10],
                                                 dk[3] + dk[7] + dk[11]};
    *reinterpret_cast<f32x4*>(ovp + c) = (f32x4){dv[0] + dv[4] + dv[8], dv[1] + dv[5] + dv[9], dv[2] + dv[6] + dv[10],
                                                 dv[3] + dv[7] + dv[11]};
  }
}

// dq[b, i, c] = sum over the key-block planes, in plane order.
__global__ void attn_dq_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dq, int dq_cs, int B, int N,
                                      int C, int CP, int nkb) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // float4 index over [B][N][CP/4]
  const long total = (long)B * N * (CP / 4);
  if (idx >= total) return;
  const int c = (int)(idx % (CP / 4)) * 4;
  const long row = idx / (CP / 4);  // b*N + i
  const long b = row / N, i = row - b * N;
  const float* src = ws + ((b * nkb) * N + i) * CP + c;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  // eight planes per trip, every load issued before the first add (the planes are 3 MB apart: one 16-byte load in
  // flight per thread leaves HBM at a third of its rate); the sum stays in plane order
  const long ps = (long)N * CP;
  int k = 0;
  for (; k + 8 <= nkb; k += 8) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (k + u) * ps));
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; k < nkb; ++k) v += *reinterpret_cast<const f32x4*>(src + k * ps);
  float* o = dq + row * dq_cs + c;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c + e < C) o[e] = v[e];
}

// The same sum over the query-tile-major partials of attn_bwd_fused_kernel: [B][nq][nkb][QT][CP].
__global__ void attn_dq_reduce_tiled_kernel(const float* __restrict__ ws, float* __restrict__ dq, int dq_cs, int B,
                                            int N, int C, int CP, int nkb, int QT) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // float4 index over [B][N][CP/4]
  const long total = (long)B * N * (CP / 4);
  if (idx >= total) return;
  const int c = (int)(idx % (CP / 4)) * 4;
  const long row = idx / (CP / 4);  // b*N + i
  const long b = row / N, i = row - b * N;
  const int nq = (N + QT - 1) / QT;
  const long it = i / QT, ir = i - it * QT;
  const long ps = (long)QT * CP;
  const float* src = ws + ((b * nq + it) * nkb) * ps + ir * CP + c;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 14 <= nkb; k += 14) {  // 14 planes in flight per thread (N = 25 088 at 256 keys per plane: 98 = 7 x 14)
    f32x4 t[14];
#pragma unroll
    for (int u = 0; u < 14; ++u) t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (k + u) * ps));
#pragma unroll
    for (int u = 0; u < 14; ++u) v += t[u];
  }
  for (; k + 4 <= nkb; k += 4) {
    f32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (k + u) * ps));
#pragma unroll
    for (int u = 0; u < 4; ++u) v += t[u];
  }
  for (; k < nkb; ++k) v += *reinterpret_cast<const f32x4*>(src + k * ps);
  float* o = dq + row * dq_cs + c;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c + e < C) o[e] = v[e];
}

}  // namespace

// dq = sum of the nkb key-block planes [B][nkb][N][CP] (shared with attn_small_bwd.hip)
int sf_attn_dq_reduce(const float* ws, float* dq, int dq_cs, int B, int N, int C, int CP, int nkb, hipStream_t s) {
  const long total = (long)B * N * (CP / 4);
  hipLaunchKernelGGL(attn_dq_reduce_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, ws, dq, dq_cs, B, N, C, CP, nkb);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// sf_attn_tune knobs (process-wide; the environment gives the initial values): keys per workgroup of the bf16-piece
// backward sweeps (0 = by shape, 4 = 128 keys, 8 = 256 keys) and the number of sweep parts (0 = by fill)
static int g_attn_nw = [] { const char* e = getenv("SF_ATTN_BX_NW"); return e ? atoi(e) : 0; }();
static int g_sweep_parts = [] { const char* e = getenv("SF_SWEEP_PARTS"); return e ? atoi(e) : 0; }();

// SF_ATTN_BX_PP / sf_attn_tune(2, .): the 8-wavefront bf16-piece backward as the ping-pong schedule
// (attn_bwd_bxpp_kernel) or as the free-running sweep (attn_bwd_bx_kernel<., 8>)
static int g_attn_pp = [] { const char* e = getenv("SF_ATTN_BX_PP"); return e ? atoi(e) : 0; }();

extern "C" int sf_attn_tune(int knob, int value) {
  if (knob == 0 && (value == 0 || value == 4 || value == 8)) g_attn_nw = value;
  else if (knob == 1 && value >= 0 && value <= SF_SWEEP_PARTS_MAX) g_sweep_parts = value;
  else if (knob == 2 && (value == 0 || value == 1)) g_attn_pp = value;
  else return SF_EINVAL;
  return SF_OK;
}

int sf_sweep_parts(long units, int tiles) {
  const int forced = g_sweep_parts;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  if (forced >= 1) return forced > SF_SWEEP_PARTS_MAX ? SF_SWEEP_PARTS_MAX : forced;
  int best = 1;
  double best_fill = 0.0;
  for (int z = 1; z <= SF_SWEEP_PARTS_MAX && (z == 1 || tiles / z >= 8); ++z) {
    const double per_cu = (double)units * z / cus;
    const double fill = per_cu / (double)((units * z + cus - 1) / cus) - 0.004 * z;  // each part re-loads its K / V
    if (fill > best_fill) { best_fill = fill; best = z; }
  }
  return best;
}

namespace {

template <int CP, int NW>
int launch_fused(BwdArgs a, float* ws, hipStream_t s) {
  a.nt = sf_cdiv(a.N, 32 * NW);  // key blocks (= dQ planes) per clip
  const int qt = (CP >= 64) ? 32 : 64;
  a.zs = sf_sweep_parts((long)a.B * a.nt, sf_cdiv(a.N, qt));
  const long planes = (long)a.B * a.nt * sf_cdiv(a.N, qt) * qt * CP, part = (long)a.B * a.zs * a.N * CP;
  a.dkp = ws + planes;
  a.dvp = a.dkp + part;
  hipLaunchKernelGGL((attn_bwd_fused_kernel<CP, NW>), dim3(a.B * a.zs * a.nt), dim3(64 * NW), 0, s, a, ws);
  SF_CHECK_LAUNCH();
  if (a.zs > 1) {
    int rc = sf_attn_dq_reduce(a.dkp, a.dk, a.dk_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dvp, a.dv, a.dv_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc != SF_OK) return rc;
  }
  const long total = (long)a.B * a.N * (CP / 4);
  hipLaunchKernelGGL(attn_dq_reduce_tiled_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, ws, a.dq, a.dq_cs, a.B,
                     a.N, a.C, CP, a.nt, qt);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// The bf16-piece form (d <= 32): split Q and gamma dz into row planes behind the fp32 form's workspace, then the sweep.
template <int NW>
int launch_fused_bx(BwdArgs a, float* ws, float* bx_ws, hipStream_t s) {
  constexpr int CP = 32, qt = BXB_QT;
  a.nt = sf_cdiv(a.N, 32 * NW);
  a.zs = sf_sweep_parts((long)a.B * a.nt, sf_cdiv(a.N, qt));
  const long planes = (long)a.B * a.nt * sf_cdiv(a.N, qt) * qt * CP, part = (long)a.B * a.zs * a.N * CP;
  a.dkp = ws + planes;
  a.dvp = a.dkp + part;
  unsigned short* qb = reinterpret_cast<unsigned short*>(bx_ws);
  unsigned short* db = qb + sf_attn_bx_plane_elems(a.B, a.N);
  static const int stagger = [] { const char* e = getenv("SF_ATTN_STAGGER"); return e ? atoi(e) : 0; }();
  a.stagger = stagger;
  int rc = sf_attn_bx_split(a.q, a.q_cs, nullptr, a.B, a.N, a.C, qb, nullptr, s);
  if (rc == SF_OK) rc = sf_attn_bx_split(a.dz, a.dz_cs, a.gamma, a.B, a.N, a.C, db, nullptr, s);
  if (rc != SF_OK) return rc;
  using Kern = void (*)(const BwdArgs, float*, const unsigned short*, const unsigned short*, int);
  static const Kern kern = [] {  // SF_ATTN_BX_DBG: timing ablations (see the kernel)
    const char* e = getenv("SF_ATTN_BX_DBG");
    const int dbg = e ? atoi(e) : 0;
    if (NW == 8) return dbg == 2 ? (Kern)attn_bwd_bx_kernel<2, 8> : dbg == 8 ? (Kern)attn_bwd_bx_kernel<8, 8> : (Kern)attn_bwd_bx_kernel<0, 8>;
    switch (dbg) {
      case 1: return (Kern)attn_bwd_bx_kernel<1, 4>;
      case 2: return (Kern)attn_bwd_bx_kernel<2, 4>;
      case 3: return (Kern)attn_bwd_bx_kernel<3, 4>;
      case 4: return (Kern)attn_bwd_bx_kernel<4, 4>;
      case 7: return (Kern)attn_bwd_bx_kernel<7, 4>;
      default: return (Kern)attn_bwd_bx_kernel<0, 4>;
    }
  }();
  static const int pad = [] { const char* e = getenv("SF_ATTN_BX_PADLDS"); return e ? atoi(e) : 0; }();  // occupancy probe
  if (NW == 8 && g_attn_pp) {  // the ping-pong schedule of the same sweep
    static const Kern pp = [] {
      const char* e = getenv("SF_ATTN_BX_DBG");
      switch (e ? atoi(e) : 0) {
        case 8: return (Kern)attn_bwd_bxpp_kernel<8>;
        case 16: return (Kern)attn_bwd_bxpp_kernel<16>;    // no s_setprio
        case 32: return (Kern)attn_bwd_bxpp_kernel<32>;    // V segments without exp2 / splits / image (results invalid)
        case 64: return (Kern)attn_bwd_bxpp_kernel<64>;    // M segments without MFMAs (results invalid)
        case 256: return (Kern)attn_bwd_bxpp_kernel<256>;    // (no stamps: a stamp's s_memtime drains lgkmcnt)
        case 512: return (Kern)attn_bwd_bxpp_kernel<512>;
        case 128: return (Kern)attn_bwd_bxpp_kernel<128>;
        case 640: return (Kern)attn_bwd_bxpp_kernel<640>;
        case 2048: return (Kern)attn_bwd_bxpp_kernel<2048>;  // no slot sums (results invalid)
        case 4096: return (Kern)attn_bwd_bxpp_kernel<4096>;  // no staging (results invalid)  // V segments with the probe's register-only vector work, stamps  // M segments without fragment reads (results invalid), stamps
        default: return (Kern)attn_bwd_bxpp_kernel<0>;
      }
    }();
    hipLaunchKernelGGL(pp, dim3(a.B * a.zs * a.nt), dim3(512), 0, s, a, ws, qb, db, sf_cdiv(a.N, qt));  // static LDS
  } else {
    static SfLdsAttr lds_attr;
    if (!sf_ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(kern), bxb_lds_bytes(NW) + pad)) return SF_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B * a.zs * a.nt), dim3(64 * NW), bxb_lds_bytes(NW) + pad, s, a, ws, qb, db,
                       sf_cdiv(a.N, qt));
  }
  SF_CHECK_LAUNCH();
  if (a.zs > 1) {
    rc = sf_attn_dq_reduce(a.dkp, a.dk, a.dk_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dvp, a.dv, a.dv_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc != SF_OK) return rc;
  }
  const long total = (long)a.B * a.N * (CP / 4);
  hipLaunchKernelGGL(attn_dq_reduce_tiled_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, ws, a.dq, a.dq_cs, a.B,
                     a.N, a.C, CP, a.nt, qt);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// The packed form (C <= 8): planes of Q and gamma dz behind the fp32 form's workspace, then the sweep.
template <int NW>
int launch_fused_bxp(BwdArgs a, float* ws, float* bx_ws, hipStream_t s) {
  constexpr int CP = 8, qt = BXB_QT;
  a.nt = sf_cdiv(a.N, 32 * NW);
  a.zs = sf_sweep_parts((long)a.B * a.nt, sf_cdiv(a.N, qt));
  const long planes = (long)a.B * a.nt * sf_cdiv(a.N, qt) * qt * CP, part = (long)a.B * a.zs * a.N * CP;
  a.dkp = ws + planes;
  a.dvp = a.dkp + part;
  unsigned short* qb = reinterpret_cast<unsigned short*>(bx_ws);
  unsigned short* db = qb + sf_attn_bx_packed_elems(a.B, a.N);
  int rc = sf_attn_bx_split_packed(a.q, a.q_cs, nullptr, a.B, a.N, a.C, qb, nullptr, s);
  if (rc == SF_OK) rc = sf_attn_bx_split_packed(a.dz, a.dz_cs, a.gamma, a.B, a.N, a.C, db, nullptr, s);
  if (rc != SF_OK) return rc;
  static SfLdsAttr lds_attr;
  if (!sf_ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(attn_bwd_bxp_kernel<NW>), bxbp_lds_bytes(NW)))
    return SF_ELAUNCH;
  hipLaunchKernelGGL((attn_bwd_bxp_kernel<NW>), dim3(a.B * a.zs * a.nt), dim3(64 * NW), bxbp_lds_bytes(NW), s, a, ws, qb,
                     db, sf_cdiv(a.N, qt));
  SF_CHECK_LAUNCH();
  if (a.zs > 1) {
    rc = sf_attn_dq_reduce(a.dkp, a.dk, a.dk_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dvp, a.dv, a.dv_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc != SF_OK) return rc;
  }
  const long total = (long)a.B * a.N * (CP / 4);
  hipLaunchKernelGGL(attn_dq_reduce_tiled_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, ws, a.dq, a.dq_cs, a.B,
                     a.N, a.C, CP, a.nt, qt);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

// 33 <= C <= 64 on bf16 pieces: two 32-channel blocks (see attn_bwd_bx2_kernel)
int launch_fused_bx2(BwdArgs a, float* ws, float* bx_ws, hipStream_t s) {
  constexpr int CP = 64, NW = 4, qt = BXB_QT;
  a.nt = sf_cdiv(a.N, 32 * NW);
  a.zs = sf_sweep_parts((long)a.B * a.nt, sf_cdiv(a.N, qt));
  const long planes = (long)a.B * a.nt * sf_cdiv(a.N, qt) * qt * CP, part = (long)a.B * a.zs * a.N * CP;
  a.dkp = ws + planes;
  a.dvp = a.dkp + part;
  const long blk_elems = sf_attn_bx_plane_elems(a.B, a.N);
  unsigned short* qb = reinterpret_cast<unsigned short*>(bx_ws);
  unsigned short* db = qb + 2 * blk_elems;
  int rc = SF_OK;
  for (int blk = 0; blk < 2 && rc == SF_OK; ++blk) {
    rc = sf_attn_bx_split(a.q + 32 * blk, a.q_cs, nullptr, a.B, a.N, a.C - 32 * blk > 32 ? 32 : a.C - 32 * blk,
                          qb + blk * blk_elems, nullptr, s);
    if (rc == SF_OK)
      rc = sf_attn_bx_split(a.dz + 32 * blk, a.dz_cs, a.gamma, a.B, a.N, a.C - 32 * blk > 32 ? 32 : a.C - 32 * blk,
                            db + blk * blk_elems, nullptr, s);
  }
  if (rc != SF_OK) return rc;
  static SfLdsAttr lds_attr;
  if (!sf_ensure_dyn_lds(lds_attr, reinterpret_cast<const void*>(attn_bwd_bx2_kernel), BXB2_LDS_BYTES)) return SF_ELAUNCH;
  hipLaunchKernelGGL(attn_bwd_bx2_kernel, dim3(a.B * a.zs * a.nt), dim3(256), BXB2_LDS_BYTES, s, a, ws, qb, db,
                     sf_cdiv(a.N, qt), blk_elems);
  SF_CHECK_LAUNCH();
  if (a.zs > 1) {
    rc = sf_attn_dq_reduce(a.dkp, a.dk, a.dk_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dvp, a.dv, a.dv_cs, a.B, a.N, a.C, CP, a.zs, s);
    if (rc != SF_OK) return rc;
  }
  const long total = (long)a.B * a.N * (CP / 4);
  hipLaunchKernelGGL(attn_dq_reduce_tiled_kernel, dim3(sf_cdiv(total, 256)), dim3(256), 0, s, ws, a.dq, a.dq_cs, a.B,
                     a.N, a.C, CP, a.nt, qt);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

template <int CP>
int launch(const BwdArgs& a, int which, hipStream_t s) {
  const int grid = a.B * a.zs * a.nt;
  if (which & 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<CP>), dim3(grid), dim3(256), 0, s, a);
  if (which & 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<CP>), dim3(grid), dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return SF_OK;
}

}  // namespace

int sf_attn_small_bwd_dispatch(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                               const float* dz, int dz_cs, const float* lse, const float* dvec, const float* gamma,
                               float* dq, int dq_cs, float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C,
                               int which, hipStream_t stream);  // attn_small_bwd.hip

// which: bit 0 = dQ kernel, bit 1 = dK/dV kernel (3 = both; separate launches let callers time them).
extern "C" int sf_attn_bwd(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                           const float* dz, int dz_cs, const float* lse, const float* dvec, const float* gamma,
                           float* dq, int dq_cs, float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C,
                           int which, void* stream) {
  if (!q || !k || !v || !dz || !lse || !dvec || !gamma || !dq || !dk || !dv) return SF_EINVAL;
  if (B <= 0 || N <= 0 || C <= 0 || C > 128 || (which & 3) == 0) return SF_EINVAL;
  BwdArgs a;
  a.q = q; a.k = k; a.v = v; a.dz = dz; a.lse = lse; a.dvec = dvec; a.gamma = gamma;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.q_cs = q_cs; a.k_cs = k_cs; a.v_cs = v_cs; a.dz_cs = dz_cs; a.dq_cs = dq_cs; a.dk_cs = dk_cs; a.dv_cs = dv_cs;
  a.B = B; a.C = C; a.N = N; a.nt = sf_cdiv(N, 128);
  a.zs = 1; a.dkp = a.dvp = a.dqp = nullptr; a.stagger = 0;
  hipStream_t s = (hipStream_t)stream;
  if (C <= 16)  // 16x16x4 tiles: no padded rows
    return sf_attn_small_bwd_dispatch(q, q_cs, k, k_cs, v, v_cs, dz, dz_cs, lse, dvec, gamma, dq, dq_cs, dk, dk_cs,
                                      dv, dv_cs, B, N, C, which, s);
  if (C <= 32) return launch<32>(a, which, s);
  if (C <= 64) return launch<64>(a, which, s);
  return launch<128>(a, which, s);
}

// Fused single-sweep backward (dQ, dK, dV) for C <= 64; ws: sf_attn_bwd_fused_ws_floats(B, N, C) floats
// (0 = this shape is not served by the fused kernel: call sf_attn_bwd).
constexpr int FUSED_KEYS_32 = 128;  // keys per workgroup of the d <= 32 fused kernel.  256 (8 wavefronts, half the
// planes) measured SLOWER on MI355X: 18.9 vs 16.6 ms per backward at N = 25088 (one workgroup per CU, 8-wave barriers)

int sf_attn_small_fused_dispatch(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                                 const float* dz, int dz_cs, const float* lse, const float* dvec, const float* gamma,
                                 float* dq, int dq_cs, float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C,
                                 float* ws, hipStream_t stream);  // attn_small_bwd.hip

// dQ planes (one per key block) + room for the dK / dV partials of up to SF_SWEEP_PARTS_MAX query parts
extern "C" long sf_attn_bwd_fused_ws_floats(int B, int N, int C) {
  if (B <= 0 || N <= 0 || C <= 0 || C > 128) return 0;
  if (C > 64) {  // two-kernel form with sweep parts (dQ, dK, dV parts, no planes) — only when it needs them to fill the chip
    const long units = (long)B * sf_cdiv(N, 128);
    return units < 2 * 256 ? 3L * B * SF_SWEEP_PARTS_MAX * N * 128 : 0;
  }
  const int cp = C <= 4 ? 4 : (C <= 8 ? 8 : (C <= 16 ? 16 : (C <= 32 ? 32 : 64)));
  const int keys = C <= 16 ? 64 : (C <= 32 ? FUSED_KEYS_32 : 128);
  const long nr = (long)sf_cdiv(N, 64) * 64;  // the dQ partials are stored per (whole) query tile of 32 / 64 rows
  long n = (long)B * (sf_cdiv(N, keys) * nr + 2L * SF_SWEEP_PARTS_MAX * N) * cp;
  if (cp == 32) n += sf_attn_bx_plane_elems(B, N);  // Q and gamma dz as three bf16 row planes each (2 x 2 B = 1 float)
  if (cp == 8) n += sf_attn_bx_packed_elems(B, N);  // Q and gamma dz as one packed plane each
  if (cp == 64) n += 2 * sf_attn_bx_plane_elems(B, N);  // two 32-channel blocks of Q and gamma dz planes
  return n;
}

static bool sf_attn_bx_wide(int B, int N) {
  return g_attn_nw == 8 || (g_attn_nw != 4 && (long)B * sf_cdiv(N, 256) >= 256);
}

// Which instantiation sf_attn_bwd_fused launches for (B, N, C) on 16-byte aligned views (tests assert that the variant
// the benchmark runs at 8 clips is the one they compare with fp64): 10 * family + wavefronts per workgroup, family
// 1 = f32 MFMA d <= 16, 2 = packed bf16 planes (d = 8), 3 = bf16 pieces d <= 32, 4 = bf16 pieces d <= 64 (two blocks),
// 5 / 6 = f32 MFMA d <= 32 / d <= 64, 7 = two-kernel form (d = 128); 0 = shape not served.
extern "C" int sf_attn_bwd_variant(int B, int N, int C) {
  if (sf_attn_bwd_fused_ws_floats(B, N, C) == 0) return 0;
  const bool bx = sf_attn_bx_level() >= 1;
  if (C > 4 && C <= 8 && C % 4 == 0 && bx) return 20 + (g_attn_nw == 8 ? 8 : 4);
  if (C <= 16) return 14;
  if (C > 64) return 74;
  if (C <= 32) return (C % 4 == 0 && bx) ? 30 + (sf_attn_bx_wide(B, N) ? 8 : 4) : 54;
  static const bool bx64 = [] { const char* e = getenv("SF_ATTN_BX64"); return !(e && e[0] == '0'); }();
  return (C % 4 == 0 && bx && bx64) ? 44 : 64;
}

extern "C" int sf_attn_bwd_fused(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                                 const float* dz, int dz_cs, const float* lse, const float* dvec, const float* gamma,
                                 float* dq, int dq_cs, float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C,
                                 float* ws, void* stream) {
  if (!q || !k || !v || !dz || !lse || !dvec || !gamma || !dq || !dk || !dv || !ws) return SF_EINVAL;
  if (sf_attn_bwd_fused_ws_floats(B, N, C) == 0 || !sf_aligned16(ws)) return SF_EINVAL;
  if (C > 4 && C <= 8 && C % 4 == 0 && sf_attn_bx_level() >= 1 && (q_cs % 4 == 0) && (k_cs % 4 == 0) &&
      (v_cs % 4 == 0) && (dz_cs % 4 == 0) && (dk_cs % 4 == 0) && (dv_cs % 4 == 0) && sf_aligned16(q) &&
      sf_aligned16(k) && sf_aligned16(v) && sf_aligned16(dz) && sf_aligned16(dk) && sf_aligned16(dv)) {
    BwdArgs a;  // packed planes on the bf16 pipe
    a.q = q; a.k = k; a.v = v; a.dz = dz; a.lse = lse; a.dvec = dvec; a.gamma = gamma;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.q_cs = q_cs; a.k_cs = k_cs; a.v_cs = v_cs; a.dz_cs = dz_cs; a.dq_cs = dq_cs; a.dk_cs = dk_cs; a.dv_cs = dv_cs;
    a.B = B; a.C = C; a.N = N; a.dqp = nullptr; a.stagger = 0;
    // 128 keys per workgroup: the dQ planes are a quarter of the d = 32 ones, and 8-wavefront barriers cost more than
    // halving them saves (N = 25 088, B = 8: 5.00 ms against 5.26).  SF_ATTN_BX_NW=8 forces the wide form.
    const bool wide = g_attn_nw == 8;
    float* const bx_ws = ws + (sf_attn_bwd_fused_ws_floats(B, N, C) - sf_attn_bx_packed_elems(B, N));
    return wide ? launch_fused_bxp<8>(a, ws, bx_ws, (hipStream_t)stream)
                : launch_fused_bxp<4>(a, ws, bx_ws, (hipStream_t)stream);
  }
  if (C <= 16)
    return sf_attn_small_fused_dispatch(q, q_cs, k, k_cs, v, v_cs, dz, dz_cs, lse, dvec, gamma, dq, dq_cs, dk, dk_cs,
                                        dv, dv_cs, B, N, C, ws, (hipStream_t)stream);
  BwdArgs a;
  a.q = q; a.k = k; a.v = v; a.dz = dz; a.lse = lse; a.dvec = dvec; a.gamma = gamma;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.q_cs = q_cs; a.k_cs = k_cs; a.v_cs = v_cs; a.dz_cs = dz_cs; a.dq_cs = dq_cs; a.dk_cs = dk_cs; a.dv_cs = dv_cs;
  a.B = B; a.C = C; a.N = N; a.nt = sf_cdiv(N, 128);
  a.dqp = nullptr;
  a.stagger = 0;
  if (C > 64) {  // d = 128: both kernels of the two-kernel form, their sweeps cut into parts so that B * N/128 < 2 x 256
    hipStream_t st = (hipStream_t)stream;  // workgroups become B * N/128 * z; the parts are summed in part order
    a.zs = sf_sweep_parts((long)B * a.nt, sf_cdiv(N, 32));
    if (a.zs <= 1) {
      a.zs = 1;
      return launch<128>(a, 3, st);
    }
    const long part = (long)B * a.zs * N * 128;
    a.dqp = ws; a.dkp = ws + part; a.dvp = ws + 2 * part;
    int rc = launch<128>(a, 3, st);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dqp, dq, dq_cs, B, N, C, 128, a.zs, st);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dkp, dk, dk_cs, B, N, C, 128, a.zs, st);
    if (rc == SF_OK) rc = sf_attn_dq_reduce(a.dvp, dv, dv_cs, B, N, C, 128, a.zs, st);
    return rc;
  }
  if (C <= 32) {
    const bool vec4 = (C % 4 == 0) && (q_cs % 4 == 0) && (k_cs % 4 == 0) && (v_cs % 4 == 0) && (dz_cs % 4 == 0) &&
                      (dk_cs % 4 == 0) && (dv_cs % 4 == 0) && sf_aligned16(q) && sf_aligned16(k) && sf_aligned16(v) &&
                      sf_aligned16(dz) && sf_aligned16(dk) && sf_aligned16(dv);
    if (vec4 && sf_attn_bx_level() >= 1) {
      // 256 keys (8 wavefronts, one workgroup per CU) per workgroup where that still fills the chip: half the dQ
      // planes to write and to sum (N = 25 088, B = 8: 9.20 -> 8.70 ms).  SF_ATTN_BX_NW=4|8 forces either.
      const bool wide = sf_attn_bx_wide(B, N);
      float* const bx_ws = ws + (sf_attn_bwd_fused_ws_floats(B, N, C) - sf_attn_bx_plane_elems(B, N));
      return wide ? launch_fused_bx<8>(a, ws, bx_ws, (hipStream_t)stream)
                  : launch_fused_bx<4>(a, ws, bx_ws, (hipStream_t)stream);
    }
    return launch_fused<32, FUSED_KEYS_32 / 32>(a, ws, (hipStream_t)stream);
  }
  {
    const bool vec4 = (C % 4 == 0) && (q_cs % 4 == 0) && (k_cs % 4 == 0) && (v_cs % 4 == 0) && (dz_cs % 4 == 0) &&
                      (dk_cs % 4 == 0) && (dv_cs % 4 == 0) && sf_aligned16(q) && sf_aligned16(k) && sf_aligned16(v) &&
                      sf_aligned16(dz) && sf_aligned16(dk) && sf_aligned16(dv);
    static const bool bx64 = [] { const char* e = getenv("SF_ATTN_BX64"); return !(e && e[0] == '0'); }();
    if (vec4 && bx64 && sf_attn_bx_level() >= 1)
      return launch_fused_bx2(a, ws, ws + (sf_attn_bwd_fused_ws_floats(B, N, C) - 2 * sf_attn_bx_plane_elems(B, N)),
                              (hipStream_t)stream);
  }
  return launch_fused<64, 4>(a, ws, (hipStream_t)stream);
}
