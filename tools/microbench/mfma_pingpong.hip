// Ping-pong schedule probe (round 5): a 512-thread workgroup puts two wavefronts on every SIMD.  Each wavefront
// alternates an M segment (60 v_mfma_f32_32x32x16_bf16 in five accumulation chains: the attention backward's
// products of one 32 x 32 score block) with a V segment (16 v_exp_f32 + 48 v_cvt_pk_bf16_f32 + NF v_fma_f32: its
// exp2 / operand splits).  MODE 0: free-running (what attn_bwd_bx_kernel does: both wavefronts of a SIMD run M, V,
// M, V ... and meet only at the tile's barriers); MODE 1: the second half of the workgroup lags by one segment and
// every segment ends in s_barrier, so a SIMD always holds one wavefront in M and one in V; MODE 2: MODE 1 with
// s_setprio 1 inside M segments; MODE 3: M and V of one wavefront interleaved by hand (1 MFMA : ~5 vector
// instructions), two such wavefronts per SIMD, no barrier.  Prints cycles of SIMD time per (M + V) pair; the matrix
// pipe alone needs 60 x 32 = 1920.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_pingpong.hip -o mfma_pingpong ; run: ./mfma_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void vwork(float (&f)[8], float b, int n_exp, int n_cvt, int n_fma) {
#pragma unroll
  for (int q = 0; q < 64; ++q) {
    if (q < n_exp) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
  }
#pragma unroll
  for (int q = 0; q < 64; ++q) {
    if (q < n_cvt) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[q & 7]) : "v"(b));
  }
#pragma unroll
  for (int q = 0; q < 400; ++q) {
    if (q < n_fma) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q & 7]) : "v"(b));
  }
}

template <int MODE, int NF>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, unsigned long long* cyc) {
  f32x16 acc[5];
  for (int i = 0; i < 5; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 av, bv;
  for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b; }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = a + i;
  const int grp = threadIdx.x >> 8;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 1 || MODE == 2) {
    if (grp == 1) __builtin_amdgcn_s_barrier();
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int u = 0; u < 12; ++u) {
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[c], 0, 0, 0);
          const int m = c * 12 + u;  // 60 gaps: 16 exp in the first 16, 48 cvt in the next 48 (one each), fmas spread
          if (m < 16) asm volatile("v_exp_f32 %0, %0" : "+v"(f[m & 7]));
          else if (m < 64) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[m & 7]) : "v"(b));
#pragma unroll
          for (int q = 0; q < (NF + 59 - m) / 60 + (m < 4 ? 1 : 0) * 0; ++q)
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q & 7]) : "v"(b));
        }
      continue;
    }
    if (MODE == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int u = 0; u < 12; ++u) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[c], 0, 0, 0);
    if (MODE == 2) __builtin_amdgcn_s_setprio(0);
    if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_barrier();
    vwork(f, b, 16, 48, NF);
    if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_barrier();
  }
  if (MODE == 1 || MODE == 2) {
    if (grp == 0) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 5; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 3 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE, int NF>
void run(float* out, unsigned long long* cyc, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NF>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  static const char* names[] = {"free-running", "ping-pong (barriers)", "ping-pong + setprio", "hand-interleaved, no barrier"};
  // two wavefronts per SIMD: SIMD cycles per (M + V) pair of ONE wavefront = wave cycles / iters / 2
  printf("%-30s fma=%3d: %.3f ms; wave 0: %.0f cycles per M+V, wave 4: %.0f  -> %.0f SIMD cycles per block (matrix pipe alone: 1920)\n",
         names[MODE], NF, best, (double)h[0] / iters, (double)h[4] / iters, (double)(h[0] > h[4] ? h[0] : h[4]) / iters / 2.0);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, (size_t)256 * 512 * 4);
  hipMalloc(&cyc, 64);
  run<0, 150>(out, cyc, iters); run<1, 150>(out, cyc, iters); run<2, 150>(out, cyc, iters); run<3, 150>(out, cyc, iters);
  run<0, 200>(out, cyc, iters); run<1, 200>(out, cyc, iters); run<2, 200>(out, cyc, iters); run<3, 200>(out, cyc, iters);
  run<0, 260>(out, cyc, iters); run<1, 260>(out, cyc, iters); run<2, 260>(out, cyc, iters); run<3, 260>(out, cyc, iters);
  return 0;
}
