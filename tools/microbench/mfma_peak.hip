// Sustained fp32 MFMA rate of the box (no memory traffic): each wavefront issues `iters` x 8 independent
// v_mfma_f32_32x32x2_f32 (or 16x16x4) from registers.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// bf16 pipe (round 4): the same loop on v_mfma_f32_32x32x16_bf16 with lane-dependent, non-trivial operand bits (the
// power a matrix instruction draws depends on the data it toggles): what the part SUSTAINS for this instruction is the
// ceiling of every bf16-piece kernel of the library.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void kbf(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[2], b[2];
  for (int v = 0; v < 2; ++v)
    for (int e = 0; e < 8; ++e) {
      a[v][e] = (__bf16)(0.37f + 0.011f * (float)((threadIdx.x * 7 + e * 13 + v * 5) & 63));
      b[v][e] = (__bf16)(-0.81f + 0.017f * (float)((threadIdx.x * 3 + e * 11 + v * 9) & 63));
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 1], b[(u >> 1) & 1], acc[u & 3], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 8;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  if (argc > 3) {  // bf16: mfma_peak <blocks> <iters> bf16
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kbf, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = 32768.0 * 8.0 * iters * 4.0 * blocks;
      printf("mfma_32x32x16_bf16 blocks=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", blocks, iters, ms, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
  }
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (which == 0)
        hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters);
      else
        hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (which == 0 ? 4096.0 : 2048.0) * 8.0 * iters * 4.0 * blocks;  // per wave, 4 waves/block
      printf("%s blocks=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", which == 0 ? "mfma_32x32x2_f32" : "mfma_16x16x4_f32",
             blocks, iters, ms, flop / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
