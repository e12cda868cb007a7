// Ping-pong probe, second round: which ingredient of the real attention-backward segments breaks the overlap that the
// register-only probe (mfma_pingpong.hip) shows?  Always: 512-thread workgroup, second half one segment behind, every
// segment ends in s_barrier, s_setprio 1 inside M.  M = 60 bf16 MFMAs in chains of 6; V = 16 v_exp + 48 v_cvt_pk +
// 150 other vector instructions.  Template switches:
//   LDSR: M reads its A fragments from LDS (6 ds_read_b64_tr_b16 x 2 per chain, one chain ahead, counted waits)
//   LDSW: V ends with 14 LDS stores
//   VARY: the MFMAs of a chain take different A / B registers (as a three-way split product does)
//   MIX:  V's 150 "other" instructions are v_lshlrev / v_and / v_sub triples instead of v_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x2 lds_read_tr(const unsigned short* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}
__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define BAR()                                                           \
  do {                                                                  \
    PIN();                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     \
    PIN();                                                              \
  } while (0)

template <bool LDSR, bool LDSW, bool VARY, bool MIX>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[8 * 6 * 1024];  // 96 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8 * 6 * 1024; i += 512) tile[i] = (unsigned short)(0x3f80 + (i & 7));
  f32x16 acc[5];
  for (int i = 0; i < 5; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float a = tid * 1e-3f, b = 1.0f + tid * 1e-4f;
  u32x4 bv[3], av[3];
  for (int i = 0; i < 3; ++i) { bv[i] = (u32x4){0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; av[i] = bv[i]; }
  float f[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { f[i] = a + i; u[i] = tid + i; }
  const int grp = wave >> 2;
  const unsigned short* const my = tile + wave * 6 * 1024 + ((lane & 15) >> 2) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + 128 * (lane >> 5);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (grp == 1) BAR();
  auto frag = [&](int c, u32x4 (&x)[3]) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      const u32x2 lo = lds_read_tr(my + pc * 1024 + (c & 1) * 512);
      const u32x2 hi = lds_read_tr(my + pc * 1024 + (c & 1) * 512 + 256);
      x[pc] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    }
  };
  auto chain = [&](const u32x4 (&x)[3], f32x16 c) -> f32x16 {
    if (VARY) {
      c = mfma(x[0], bv[2], c); c = mfma(x[2], bv[0], c); c = mfma(x[1], bv[1], c);
      c = mfma(x[0], bv[1], c); c = mfma(x[1], bv[0], c); return mfma(x[0], bv[0], c);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) c = mfma(x[0], bv[0], c);
    return c;
  };
  for (int it = 0; it < iters; ++it) {
    // ---- M
    __builtin_amdgcn_s_setprio(1);
    u32x4 x0[3], x1[3];
    for (int i = 0; i < 3; ++i) { x0[i] = av[i]; x1[i] = av[i]; }
    if (LDSR) { frag(0, x0); PIN(); }
#pragma unroll
    for (int c = 0; c < 10; c += 2) {
      if (LDSR) { frag(c + 1, x1); PIN(); }
      acc[c >> 1] = chain(x0, acc[c >> 1]);
      PIN();
      if (LDSR && c + 2 < 10) { frag(c + 2, x0); PIN(); }
      acc[c >> 1] = chain(x1, acc[c >> 1]);
      PIN();
    }
    __builtin_amdgcn_s_setprio(0);
    BAR();
    // ---- V
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("v_exp_f32 %0, %0" : "+v"(f[q & 7]));
#pragma unroll
    for (int q = 0; q < 48; ++q) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[q & 7]) : "v"(b));
    if (MIX) {
#pragma unroll
      for (int q = 0; q < 50; ++q) {
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[q & 7]) : "v"(f[(q + 1) & 7]));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[(q + 3) & 7]) : "v"(f[(q + 2) & 7]));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[(q + 4) & 7]) : "v"(u[(q + 5) & 7]));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 150; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q & 7]) : "v"(b));
    }
    if (LDSW) {
      unsigned short* const w = tile + wave * 6 * 1024 + 3 * 1024 + lane * 4;
#pragma unroll
      for (int q = 0; q < 14; ++q) *reinterpret_cast<u32x2*>(w + q * 256) = (u32x2){u[q & 7], __builtin_bit_cast(unsigned, f[q & 7])};
    }
    BAR();
  }
  if (grp == 0) BAR();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 5; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i] + u[i];
  out[blockIdx.x * 512 + tid] = s;
  if (blockIdx.x == 3 && lane == 0) cyc[wave] = t1 - t0;
}

template <bool LDSR, bool LDSW, bool VARY, bool MIX>
void run(float* out, unsigned long long* cyc, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<LDSR, LDSW, VARY, MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
  float best = 1e30f;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<LDSR, LDSW, VARY, MIX>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("LDS reads in M %d  LDS stores in V %d  varied A/B %d  shl/and/sub mix %d: %.3f ms, %.0f cycles per interval (matrix pipe "
         "alone: 1920)\n", LDSR, LDSW, VARY, MIX, best, (double)h[0] / iters / 2.0);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)256 * 512 * 4);
  (void)hipMalloc(&cyc, 64);
  run<false, false, false, false>(out, cyc, iters);
  run<false, false, true, false>(out, cyc, iters);
  run<false, false, false, true>(out, cyc, iters);
  run<false, true, false, false>(out, cyc, iters);
  run<true, false, false, false>(out, cyc, iters);
  run<true, false, true, false>(out, cyc, iters);
  run<true, true, true, true>(out, cyc, iters);
  return 0;
}
