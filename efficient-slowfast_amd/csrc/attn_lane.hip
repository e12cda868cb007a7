// attn_lane.hip — flash SpatialAttention for head dims d = 4 and d = 8 with ONE QUERY PER LANE on
// v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer-product blocks per instruction, K = 1).
//
// At d <= 8 the 16x16x4 tiles of attn_small.hip spend the second product (O^T = V^T P^T, M = channels) on 16-row
// tiles of which d + 1 carry data: 9 of 16 at d = 8 (R50 s1_fuse, N = 25 088), 5 of 16 at d = 4 (GhostNet s1_fuse,
// N = 100 352).  The 4x4x1 form has the same FLOP rate (512 FLOP per ~8-cycle instruction, measured 112 TFLOP/s:
// tools/microbench/mfma4x4_probe.hip) and no padding at these sizes: block b = lane / 4 holds queries 4b .. 4b+3,
//   S^T[key 4g+i][query]  : A (lane 4b+i) = K[key 4g+i][c]  (depends on lane & 3 only: one broadcast LDS read),
//                           B (lane)      = Q[query = lane][c]  (registers, pre-scaled by log2 e), one instruction
//                           per channel c;  D leaves key i of the group in VGPR i and the QUERY ON THE LANE,
//   O^T[ch 4cg+i][query] += A (lane 4b+i) = V[key][4cg+i], B (lane) = p[key] — exactly the VGPR S^T left behind.
// Every key of a query passes through the same lane, so the online softmax (running max, denominator, rescale) is
// purely per-lane: no shuffles, no LDS exchange.  (wdf_attention_helper.py:41-54 + the CMDA tail
// custom_video_model_builder.py:143-146, as attn_flash.hip.)
#include "attn_args.h"
#include <stdlib.h>
#include <type_traits>

namespace {

using AttnArgs = SfAttnArgs;
constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;

template <int D>
__global__ __launch_bounds__(256) void attn_lane_kernel(const AttnArgs p) {
  constexpr int CG = D / 4;            // channel groups of 4
  constexpr int KT = 128;              // keys per LDS tile
  constexpr int NG = 4;                // key groups (of 4) per softmax step: 16 keys
  constexpr int VP = KT + 4;           // V^T row pitch
  constexpr int KF4 = KT * D / 4;      // float4 per K (or V) tile
  constexpr int LIT = (KF4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * (KT * D + D * VP)];
  float* const Ks = smem;                  // [2][KT][D]
  float* const Vt = smem + 2 * KT * D;     // [2][D][VP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l3 = lane & 3;
  const int bz = blockIdx.x / p.nqt;       // workgroup -> (clip b, key part z, query tile of 256)
  const int b = bz / p.zs, z = bz - b * p.zs;
  const int N = p.N;
  const int qrow = (blockIdx.x - bz * p.nqt) * 256 + wave * 64 + lane;
  const bool qok = qrow < N;
  const long brow = (long)b * N;

  float qf[D];
  {
    const float* qp = p.q + (brow + (qok ? qrow : 0)) * p.q_cs;
#pragma unroll
    for (int c4 = 0; c4 < CG; ++c4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) qf[4 * c4 + e] = qok ? v[e] * LOG2E : 0.f;  // scores in the exp2 domain
    }
  }
  // NACC accumulators per channel group (4 dependent chains in all): the keys of a group go round-robin, so an
  // accumulator is revisited after 3 other instructions
  constexpr int NACC = 4 / CG;
  f32x4 oa[NACC][CG];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int c4 = 0; c4 < CG; ++c4) oa[a][c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // m_run = the reference maximum of the online softmax.  It is subtracted from the scores by the score MFMAs
  // themselves: their accumulators start from negm = -m_run (the C operand of each group's first instruction; dst !=
  // srcC costs nothing), which removes a v_sub per score from a loop whose vector instructions are wall time.  Until
  // the first group has been seen (`first`) the reference is 0 and that group's own maximum replaces it.
  float m_run = 0.f, l_run = 0.f;
  f32x4 negm = {0.f, 0.f, 0.f, 0.f};
  bool first = true;

  f32x4 rk[LIT], rv[LIT];
  auto load_tile = [&](int j0) {
#pragma unroll
    for (int i = 0; i < LIT; ++i) {
      const int f = tid + i * 256;
      const int srow = f / CG, sc4 = (f - srow * CG) * 4;
      rk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int j = j0 + srow;
      if (f < KF4 && j < N) {
        rk[i] = *reinterpret_cast<const f32x4*>(p.k + (brow + j) * p.k_cs + sc4);
        rv[i] = *reinterpret_cast<const f32x4*>(p.v + (brow + j) * p.v_cs + sc4);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LIT; ++i) {
      const int f = tid + i * 256;
      const int srow = f / CG, sc4 = (f - srow * CG) * 4;
      if (f < KF4) {
        *reinterpret_cast<f32x4*>(Ks + (buf * KT + srow) * D + sc4) = rk[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) Vt[(buf * D + sc4 + e) * VP + srow] = rv[i][e];
      }
    }
  };

  const int tiles_all = (N + KT - 1) / KT;
  const int tz = (tiles_all + p.zs - 1) / p.zs;   // key tiles per part
  const int t0 = z * tz;
  const int t1 = min(tiles_all, t0 + tz);
  if (t0 < t1) {
    load_tile(t0 * KT);
    store_tile(0);
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    const bool more = (t + 1) < t1;
    if (more) load_tile((t + 1) * KT);
    const float* kb = Ks + (buf * KT + l3) * D;           // + 4*group*D + 4*c4
    const float* vb = Vt + (buf * D + l3) * VP;           // + 4*c4*VP + 4*group
    const int jbase = t * KT;
    // the tile's sweep, instantiated with and without the key-range mask (only the last tile of a clip is ragged)
    auto sweep = [&](auto ragged_tag) {
    constexpr bool ragged = decltype(ragged_tag)::value;
#pragma unroll 1
    for (int g0 = 0; g0 < KT / 4; g0 += NG) {
      // ---- scores of 16 keys: NG independent accumulators, channel-major issue order
      f32x4 s[NG];
#pragma unroll
      for (int c4 = 0; c4 < CG; ++c4) {
        f32x4 ka[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) ka[g] = *reinterpret_cast<const f32x4*>(kb + (g0 + g) * 4 * D + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int g = 0; g < NG; ++g)
            s[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ka[g][e], qf[4 * c4 + e], (c4 == 0 && e == 0) ? negm : s[g], 0, 0,
                                                      0);  // s = score - m_run
      }
      if (ragged) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (jbase + (g0 + g) * 4 + i >= N) s[g][i] = NEG_BIG;
      }
      // ---- online softmax, all in this lane.  The 4x4x1 instructions are short (8 cycles) and leave no issue slots
      // for another wavefront's VALU work, so VALU instructions cost wall time here: the running max is only raised
      // when a score exceeds it by more than 2^8 (scores stay bounded: p <= 256, the result does not depend on which
      // max is subtracted), which makes the rescale of the 4*D accumulators rare; the denominator is summed as a
      // float4 (packed adds).
      float mloc = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
      for (int g = 1; g < NG; ++g) mloc = fmaxf(mloc, fmaxf(fmaxf(s[g][0], s[g][1]), fmaxf(s[g][2], s[g][3])));
      if (first || __any(mloc > 8.0f)) {   // (the scores are already relative to m_run)
        const float delta = (first || mloc > 8.0f) ? mloc : 0.f;   // the reference moves by this much in this lane
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int c4 = 0; c4 < CG; ++c4) oa[a][c4] *= alpha;
        }
        m_run += delta;
        negm = (f32x4){-m_run, -m_run, -m_run, -m_run};
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i) s[g][i] -= delta;
        first = false;
      }
      f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s[g][i] = __builtin_amdgcn_exp2f(s[g][i]);
        l4 += s[g];
      }
      l_run += (l4[0] + l4[1]) + (l4[2] + l4[3]);
      // ---- O^T += V^T P^T: one instruction per (key, channel group)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x4 va[CG];
#pragma unroll
        for (int c4 = 0; c4 < CG; ++c4) va[c4] = *reinterpret_cast<const f32x4*>(vb + 4 * c4 * VP + (g0 + g) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c4 = 0; c4 < CG; ++c4)
            oa[i % NACC][c4] = __builtin_amdgcn_mfma_f32_4x4x1f32(va[c4][i], s[g][i], oa[i % NACC][c4], 0, 0, 0);
      }
    }
    };
    if (jbase + KT > N) sweep(std::true_type{});
    else sweep(std::false_type{});
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: the lane holds its query's d outputs
  if (first) m_run = NEG_BIG;   // a key part without tiles: its (m, l) = (-inf, 0) must not raise the merged maximum
  if (!qok) return;
  f32x4 o[CG];
#pragma unroll
  for (int c4 = 0; c4 < CG; ++c4) {
    o[c4] = oa[0][c4];
#pragma unroll
    for (int a = 1; a < NACC; ++a) o[c4] += oa[a][c4];
  }
  if (p.zs > 1) {  // this key part's (O^T, m, l): merged and finished by attn_fwd_merge_kernel (attn_flash.hip)
    const long prow = (long)bz * N + qrow;
    *reinterpret_cast<float2*>(p.part_ml + prow * 2) = make_float2(m_run, l_run);
#pragma unroll
    for (int c4 = 0; c4 < CG; ++c4) *reinterpret_cast<f32x4*>(p.part_o + prow * D + 4 * c4) = o[c4];
    return;
  }
  const float inv_l = 1.0f / l_run;
  if (p.lse_save) p.lse_save[brow + qrow] = m_run + __log2f(l_run);
  const float gamma = p.gamma ? p.gamma[0] : 1.0f;
  const int HW = p.H * p.W;
  const int tq = qrow / HW;
  const int hw = qrow - tq * HW;
  const float* xp = p.x + (brow + qrow) * p.x_cs;
  const long orow0 = ((long)b * p.T * p.alpha + (long)tq * p.alpha) * HW + hw;
#pragma unroll
  for (int c4 = 0; c4 < CG; ++c4) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 4 * c4);
    f32x4 on = o[c4] * inv_l;
    if (p.o_save) *reinterpret_cast<f32x4*>(p.o_save + (brow + qrow) * D + 4 * c4) = on;
    f32x4 y = gamma * on + xv;
    if (p.scale) y = y * *reinterpret_cast<const f32x4*>(p.scale + 4 * c4) + *reinterpret_cast<const f32x4*>(p.bias + 4 * c4);
    if (p.act == SF_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
    }
    for (int r = 0; r < p.alpha; ++r)
      *reinterpret_cast<f32x4*>(p.out + (orow0 + (long)r * HW) * p.out_cs + p.out_coff + 4 * c4) = y;
  }
}

template <int D>
int launch(const AttnArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((attn_lane_kernel<D>), dim3(a.B * a.zs * a.nqt), dim3(256), 0, s, a);
  SF_CHECK_LAUNCH();
  return a.zs > 1 ? sf_attn_fwd_merge(a, D, s) : SF_OK;
}

}  // namespace

// Takes d = 4 problems whose views are all 16-byte addressable (SF_ATTN_LANE=8: d = 8 too; SF_ATTN_LANE=0: none);
// returns 1 when it does not take the problem (attn_small.hip then runs).  Measured on MI355X
// (tools/microbench/attn_small_bench.py, same process pair): d = 4, N = 100 352, B = 2: 8.87 -> 6.3 ms; d = 8,
// N = 25 088, B = 8: 2.62 -> 2.60 ms — at d = 8 the four-times-smaller MFMA instructions make the loop
// instruction-issue bound (PMC: MFMA pipe busy 44 %, wavefronts waiting to issue 74 % of their cycles, 2.2 VALU
// instructions per MFMA), which cancels the padding saved, so d = 8 stays on the 16x16x4 kernel.
int sf_attn_lane_try(SfAttnArgs a, bool vec4, float* ws, hipStream_t stream) {
  static const int mode = [] { const char* e = getenv("SF_ATTN_LANE"); return e ? atoi(e) : 4; }();
  if (mode == 0 || !vec4 || !(a.C == 4 || (a.C == 8 && mode == 8))) return 1;
  if ((a.scale && !sf_aligned16(a.scale)) || (a.bias && !sf_aligned16(a.bias)) ||
      (a.o_save && !sf_aligned16(a.o_save)))
    return 1;
  a.nqt = sf_cdiv(a.N, 256);
  a.zs = 1; a.part_o = nullptr; a.part_ml = nullptr;
  if (ws) {
    a.zs = sf_sweep_parts((long)a.B * a.nqt, sf_cdiv(a.N, 128));
    sf_attn_place_parts(a, a.C, ws);
  }
  return a.C == 4 ? launch<4>(a, stream) : launch<8>(a, stream);
}
