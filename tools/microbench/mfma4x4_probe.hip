// Probe of v_mfma_f32_4x4x1_16B_f32's operand layout on gfx950 and of its issue rate.
// Hypothesis: 16 independent blocks b = lane / 4; D[vgpr v][lane 4b + j] += A[lane 4b + v] * B[lane 4b + j].
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma4x4_probe.hip -o gpurun_out/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[v * 64 + l] = c[v];
}

template <int NACC>
__global__ void rate(float* out, int iters) {
  f32x4 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run_rate(float* o, int waves_per_simd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(rate<NACC>, dim3(1024 * waves_per_simd), dim3(256), 0, 0, o, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate<NACC>, dim3(1024 * waves_per_simd), dim3(256), 0, 0, o, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = 1024.0 * waves_per_simd * 4 * iters * 8;
  printf("4x4x1_16B, %d accumulators, %d waves/SIMD: %.2f TFLOP/s, %.1f cycles per instruction per SIMD at 2.1 GHz\n",
         NACC, waves_per_simd, mfmas * 512 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.1e9 / (mfmas / 1024.0));
}

int main() {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + l; }
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int v = 0; v < 4; ++v)
    for (int l = 0; l < 64; ++l) {
      const float want = ha[(l / 4) * 4 + v] * hb[l];
      if (hd[v * 64 + l] != want) ++bad;
    }
  printf("hypothesis D[v][4b+j] = A[4b+v]*B[4b+j]: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  if (bad) {
    for (int v = 0; v < 4; ++v) { printf("v%d:", v); for (int l = 0; l < 8; ++l) printf(" %g", hd[v * 64 + l]); printf("\n"); }
  }
  float* o; hipMalloc(&o, 4096 * 256 * 4);
  run_rate<1>(o, 1); run_rate<2>(o, 1); run_rate<4>(o, 1); run_rate<8>(o, 1);
  run_rate<1>(o, 4); run_rate<2>(o, 4); run_rate<4>(o, 4);
  return 0;
}
