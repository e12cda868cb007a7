// Do fp32 MFMAs and ordinary vector instructions overlap on a gfx950 SIMD?  Each wavefront runs `iters` x 8
// v_mfma_f32_32x32x2_f32 (or 16x16x4) with FILL independent v_fma_f32 (or v_exp_f32) placed after every MFMA; one or
// two wavefronts per SIMD.  If the fillers are hidden the time per MFMA stays at 64 (32) cycles until their issue cost
// exceeds the MFMA's shadow; if the two share the ALUs every filler adds its full cost.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_coexec.hip -o mfma_coexec ; run: ./mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FILL, int KIND, int SHAPE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x16 acc32[2];
  f32x4 acc16[2];
  for (int i = 0; i < 2; ++i) {
    for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
  }
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (SHAPE == 3216) {  // the bf16 matrix pipe (round 4: do ITS MFMAs hide vector work?)
        bf16x8 av, bv;
        for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b; }
        acc32[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc32[u & 1], 0, 0, 0);
      } else if (SHAPE == 32) acc32[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[u & 1], 0, 0, 0);
      else acc16[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc16[u & 1], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < FILL; ++q) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q & 7]) : "v"(b));
        else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[q & 7]) : "v"(b));
        else asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) {
    for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int r = 0; r < 4; ++r) s += acc16[i][r];
  }
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FILL, int KIND, int SHAPE>
void run(float* out, int blocks, int iters, const char* tag) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<FILL, KIND, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // ns per MFMA per wavefront slot: waves per SIMD = blocks / 256 (one wave of a block per SIMD)
  const double per_simd = 8.0 * iters * (blocks / 256.0);
  printf("%-10s shape=%dx fill=%d %s blocks=%d: %.3f ms, %.2f ns per MFMA on a SIMD\n", tag, SHAPE, FILL,
         KIND == 2 ? "v_cvt_pk" : KIND ? "v_exp" : "v_fma", blocks, best, best * 1e6 / per_simd);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  float* out;
  hipMalloc(&out, (size_t)512 * 256 * 4);
  for (int blocks = 256; blocks <= 512; blocks *= 2) {
    run<0, 0, 32>(out, blocks, iters, "bare");
    run<2, 0, 32>(out, blocks, iters, "fillers");
    run<4, 0, 32>(out, blocks, iters, "fillers");
    run<8, 0, 32>(out, blocks, iters, "fillers");
    run<16, 0, 32>(out, blocks, iters, "fillers");
    run<2, 1, 32>(out, blocks, iters, "fillers");
    run<4, 1, 32>(out, blocks, iters, "fillers");
    run<8, 1, 32>(out, blocks, iters, "fillers");
    run<0, 0, 3216>(out, blocks, iters, "bare");
    run<2, 0, 3216>(out, blocks, iters, "fillers");
    run<4, 0, 3216>(out, blocks, iters, "fillers");
    run<6, 0, 3216>(out, blocks, iters, "fillers");
    run<8, 0, 3216>(out, blocks, iters, "fillers");
    run<12, 0, 3216>(out, blocks, iters, "fillers");
    run<2, 1, 3216>(out, blocks, iters, "fillers");
    run<4, 1, 3216>(out, blocks, iters, "fillers");
    run<4, 2, 3216>(out, blocks, iters, "fillers");
    run<8, 2, 3216>(out, blocks, iters, "fillers");
    run<0, 0, 16>(out, blocks, iters, "bare");
    run<2, 0, 16>(out, blocks, iters, "fillers");
    run<4, 0, 16>(out, blocks, iters, "fillers");
    run<8, 0, 16>(out, blocks, iters, "fillers");
    run<2, 1, 16>(out, blocks, iters, "fillers");
    run<4, 1, 16>(out, blocks, iters, "fillers");
  }
  return 0;
}
