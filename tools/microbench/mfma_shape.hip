// bf16 MFMA shape vs sustained clock (MI355X_MICROARCH.md, DVFS give-back item 7): the same 64 x 64 x (32 K per step) wave
// tile computed with v_mfma_f32_32x32x16_bf16 (2 x 2 blocks x 2 k-steps = 8 MFMAs of 32 cycles) and with
// v_mfma_f32_16x16x32_bf16 (4 x 4 tiles = 16 MFMAs of 16 cycles), operands in registers, random data / zeros,
// one or two wavefronts per SIMD.  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
// hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_shape.hip -o tools/microbench/build/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, float* out, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63;
  // operand fragments: 64 rows x 32 k of A and B = per lane 2 x 2 u32x4 (32x32x16: [row block][k step]) or 4 x 1 (16x16x32)
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(threadIdx.x * 8 + i) & 4095];
    b[i] = src[(threadIdx.x * 8 + 4 + i) & 4095];
  }
  unsigned long long t0 = 0, r0 = 0;
  if (lane == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  if constexpr (SHAPE == 32) {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i * 2 + ks]),
                                                                   __builtin_bit_cast(bf16x8, b[j * 2 + ks]), c[i * 2 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 c[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                 __builtin_bit_cast(bf16x8, b[j]), c[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
  if (lane == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

int main() {
  const int n = 4096;
  std::vector<unsigned> h(n * 4);
  u32x4* src; float* out; unsigned long long* st;
  (void)hipMalloc(&src, n * 16); (void)hipMalloc(&out, 2048 * 256 * 4); (void)hipMalloc(&st, 2048 * 4 * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int data = 0; data < 2; ++data) {
    srand(7);
    for (auto& v : h) {  // random bf16 pairs with exponents around 1.0, or zeros
      unsigned lo = 0x3f00u + (rand() & 0x1ff) + ((rand() & 1) << 15), hi = 0x3f00u + (rand() & 0x1ff) + ((rand() & 1) << 15);
      v = data == 0 ? (lo | (hi << 16)) : 0u;
    }
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps)
      for (int shape = 0; shape < 2; ++shape) {
        const int blocks = 256 * wps, iters = 40000;
        double best = 0, clk = 0;
        for (int rep = 0; rep < 3; ++rep) {
          // ~1 s of back-to-back launches so that the clock settles, then time the last one
          for (int w = 0; w < 4; ++w) {
            if (shape == 0) k<32><<<blocks, 256>>>(src, out, st, iters); else k<16><<<blocks, 256>>>(src, out, st, iters);
          }
          (void)hipEventRecord(e0);
          if (shape == 0) k<32><<<blocks, 256>>>(src, out, st, iters); else k<16><<<blocks, 256>>>(src, out, st, iters);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          const double fl = 2.0 * 64 * 64 * 32 * (double)iters * blocks * 4;
          std::vector<unsigned long long> hs(blocks * 4 * 2);
          (void)hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
          std::vector<double> cl;
          for (int w = 0; w < blocks * 4; ++w) cl.push_back((double)hs[2 * w] / (double)hs[2 * w + 1] * 100.0);
          std::sort(cl.begin(), cl.end());
          if (fl / ms * 1e-9 > best) { best = fl / ms * 1e-9; clk = cl[cl.size() / 2]; }
        }
        printf("%s data, %d wave(s)/SIMD, %s: %.0f TFLOP/s, in-kernel clock %.0f MHz\n", data ? "zero  " : "random", wps,
               shape ? "16x16x32" : "32x32x16", best, clk);
      }
  }
  return 0;
}
