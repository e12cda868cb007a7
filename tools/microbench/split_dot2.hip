// Is v_dot2c_f32_bf16 an exact (and cheaper) way to take the residual a - bf16(a) of the three-way operand split (bx.h)?
//   1. bit-compare the three pieces of both forms over 2^26 floats of every exponent class
//   2. time 64 dependent splits per thread, both forms
// hipcc --offload-arch=gfx950 -O3 tools/microbench/split_dot2.hip -o gpurun_out/split_dot2 && gpurun_out/split_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void split_shift(float a, float b, unsigned (&p)[3]) {
  p[0] = cvt_pk(a, b);
  a -= __builtin_bit_cast(float, p[0] << 16);
  b -= __builtin_bit_cast(float, p[0] & 0xffff0000u);
  p[1] = cvt_pk(a, b);
  a -= __builtin_bit_cast(float, p[1] << 16);
  b -= __builtin_bit_cast(float, p[1] & 0xffff0000u);
  p[2] = cvt_pk(a, b);
}
__device__ __forceinline__ void split_dot(float a, float b, unsigned (&p)[3]) {
#ifdef INLINE_CONST  // hipcc 7.2 emits the inline constant "-1.0" for (-1, 0); the hardware reads that as (0, -1)
  const bf16x2 mlo = {(__bf16)-1.0f, (__bf16)0.0f};
#else
  unsigned mraw;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(mraw));
  const bf16x2 mlo = __builtin_bit_cast(bf16x2, mraw);
#endif
  const bf16x2 mhi = {(__bf16)0.0f, (__bf16)-1.0f};
  p[0] = cvt_pk(a, b);
  a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[0]), mlo, a, false);
  b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[0]), mhi, b, false);
  p[1] = cvt_pk(a, b);
  a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[1]), mlo, a, false);
  b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[1]), mhi, b, false);
  p[2] = cvt_pk(a, b);
}
__global__ void check(const float* x, long n, unsigned long long* bad, unsigned* first) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned s[3], d[3];
  split_shift(x[2 * i], x[2 * i + 1], s);
  split_dot(x[2 * i], x[2 * i + 1], d);
  if (s[0] != d[0] || s[1] != d[1] || s[2] != d[2]) {
    if (atomicAdd(bad, 1ull) == 0) {
      first[0] = __builtin_bit_cast(unsigned, x[2 * i]); first[1] = __builtin_bit_cast(unsigned, x[2 * i + 1]);
      for (int k = 0; k < 3; ++k) { first[2 + k] = s[k]; first[5 + k] = d[k]; }
    }
  }
}
template <int DOT>
__global__ void rate(const float* x, unsigned* o, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float a = x[2 * i], b = x[2 * i + 1];
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      unsigned p[3];
      if (DOT) split_dot(a, b, p); else split_shift(a, b, p);
      acc ^= p[0] + p[1] * 3 + p[2] * 5;
      a = a * 1.0009765625f + 0.25f;
      b = b * 0.9990234375f - 0.125f;
    }
  }
  o[i] = acc;
}
int main() {
  const long n = 1L << 26;
  std::vector<float> h(n);
  srand(1);
  for (long i = 0; i < n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    const int cls = i & 7;
    if (cls < 3) {                       // every exponent but inf / nan and the top one (bf16 rounding overflows there)
      unsigned e = (u >> 23) & 0xff; if (e >= 0xfe) e = 0x7f; u = (u & 0x807fffffu) | (e << 23);
    } else if (cls < 6) {                // probabilities / activations: exponents 2^-30 .. 2^6
      unsigned e = 97 + ((u >> 23) & 0xff) % 36; u = (u & 0x807fffffu) | (e << 23);
    } else if (cls == 6) {               // denormals and the smallest normals
      u &= 0x80ffffffu;
    } else {                             // few significant bits (ties of the bf16 rounding)
      u &= 0xffff8000u; unsigned e = (u >> 23) & 0xff; if (e >= 0xfe || e == 0) u = (u & 0x807fffffu) | (0x80u << 23);
    }
    memcpy(&h[i], &u, 4);
  }
  float* x; unsigned long long* bad; unsigned* first; unsigned* o;
  hipMalloc(&x, n * 4); hipMalloc(&bad, 8); hipMalloc(&first, 32); hipMalloc(&o, (n / 2) * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 8);
  check<<<(unsigned)((n / 2 + 255) / 256), 256>>>(x, n, bad, first);
  unsigned long long nb; unsigned f[8];
  hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
  printf("pairs checked %ld, pairs whose pieces differ: %llu\n", n / 2, nb);
  if (nb) printf("  first: a=%08x b=%08x  shift %08x %08x %08x  dot %08x %08x %08x\n", f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
  // per class
  for (int cls = 0; cls < 8; ++cls) {
    std::vector<float> g; g.reserve(n / 8 + 2);
    for (long i = cls; i < n; i += 8) g.push_back(h[i]);
    if (g.size() & 1) g.pop_back();
    hipMemcpy(x, g.data(), g.size() * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 8);
    check<<<(unsigned)((g.size() / 2 + 255) / 256), 256>>>(x, (long)g.size(), bad, first);
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
    printf("  class %d: %zu pairs, differing %llu\n", cls, g.size() / 2, nb);
  }
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, iters = 2000;
  for (int form = 0; form < 2; ++form)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (form) rate<1><<<blocks, 256>>>(x, o, iters); else rate<0><<<blocks, 256>>>(x, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double splits = (double)blocks * 256 * iters * 16;
      printf("%s: %.3f ms, %.2f G pair-splits/s\n", form ? "dot2c " : "shifts", ms, splits / ms * 1e-6);
    }
  return 0;
}
