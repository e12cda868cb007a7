// Cost of DEPENDENT fp32 MFMA chains on gfx950: each wavefront issues `iters` x 8 MFMAs round-robin over NACC
// accumulators (NACC = 1: every MFMA waits for the previous one's result).  One wavefront per SIMD (256 blocks) and two.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void chain(float* out, int iters) {
  f32x16 a32[NACC];
  f32x4 a16[NACC];
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) a32[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) a16[i][r] = 0.f;
  }
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (SHAPE == 32) a32[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a32[u % NACC], 0, 0, 0);
      else a16[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a16[u % NACC], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) s += a32[i][r];
    for (int r = 0; r < 4; ++r) s += a16[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The same with DIFFERENT A / B registers on every MFMA (16 of each, as a real tile loop has), NACC chains.
template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void chain_regs(float* out, int iters) {
  f32x16 a32[NACC];
  f32x4 a16[NACC];
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) a32[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) a16[i][r] = 0.f;
  }
  float av[16], bv[16];
  for (int i = 0; i < 16; ++i) { av[i] = threadIdx.x * 1e-3f + i; bv[i] = 1.0f + threadIdx.x * 1e-4f * i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (SHAPE == 32) a32[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u * 5) & 15], a32[u % NACC], 0, 0, 0);
      else a16[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[(u * 5) & 15], a16[u % NACC], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) s += a32[i][r];
    for (int r = 0; r < 4; ++r) s += a16[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int SHAPE>
void run_regs(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((chain_regs<NACC, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters / 2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double per_simd = 16.0 * (iters / 2) * (blocks / 256.0);
  printf("shape=%dx chains=%d waves/SIMD=%d, 16 distinct A and B registers: %.3f ms, %.2f ns per MFMA on a SIMD\n", SHAPE,
         NACC, blocks / 256, best, best * 1e6 / per_simd);
}

template <int NACC, int SHAPE>
void run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((chain<NACC, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double per_simd = 8.0 * iters * (blocks / 256.0);
  printf("shape=%dx chains=%d waves/SIMD=%d: %.3f ms, %.2f ns per MFMA on a SIMD\n", SHAPE, NACC, blocks / 256, best,
         best * 1e6 / per_simd);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  float* out;
  (void)hipMalloc(&out, (size_t)512 * 256 * 4);
  for (int blocks = 256; blocks <= 512; blocks *= 2) {
    run<1, 32>(out, blocks, iters);
    run<2, 32>(out, blocks, iters);
    run<4, 32>(out, blocks, iters);
    run<1, 16>(out, blocks, iters);
    run<2, 16>(out, blocks, iters);
    run<4, 16>(out, blocks, iters);
    run_regs<1, 32>(out, blocks, iters);
    run_regs<2, 32>(out, blocks, iters);
    run_regs<1, 16>(out, blocks, iters);
    run_regs<4, 16>(out, blocks, iters);
  }
  return 0;
}
