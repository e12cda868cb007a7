// How fast does a CU get operand bytes on gfx950, by path?  One workgroup of 4 / 8 wavefronts per CU (256 workgroups),
// every wavefront issues `iters` x 8 wave-level loads of 1 KiB (64 lanes x 16 bytes) from a 64 MB L2/MALL-resident or a
// 512 MB (HBM) region, with all of them in flight (one counted wait at the end of each group of 8):
//   dma     buffer_load_dwordx4 ... lds   contiguous 1 KiB pieces (direct-to-LDS, what conv_bx / conv_pw_bx / conv_rows use)
//   dma_oob the same with every offset out of range (nothing leaves the CU)
//   reg     buffer_load_dwordx4 into registers, contiguous 1 KiB pieces
//   frag    buffer_load_dwordx4 into registers, 16 rows x 64 bytes (row pitch 1 KiB): conv_wave's fragment shape
// Prints GB/s over the chip and cycles per 1 KiB piece and CU (at the 2.1 GHz the part holds).
// Build: hipcc --offload-arch=gfx950 -O3 lds_delivery.hip -o build/lds_delivery
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ __launch_bounds__(512) void deliver(const float* src, unsigned bytes, float* out, int iters, unsigned span) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // every wavefront walks its own region; piece p of iteration it
  unsigned base = ((blockIdx.x * nw + wave) * 7919u * 1024u) % span;
  for (int it = 0; it < iters; ++it) {
    f32x4 r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      unsigned off;
      if (MODE == 3) off = (base + (unsigned)u * 16384u + (unsigned)(lane >> 2) * 1024u + (unsigned)(lane & 3) * 16u) % span;
      else off = (base + (unsigned)u * 1024u + (unsigned)lane * 16u) % span;
      if (MODE == 1) off = 0x80000000u;
      if (MODE <= 1) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (wave * 8 + u) * 256), 16, off, 0, 0, 0);
      } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[u]) : "v"(off), "s"(rs) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("" : "+v"(r[u]));
        acc += r[u];
      }
    }
    base = (base + 8u * 1024u * (MODE == 3 ? 16u : 1u)) % span;
  }
  if (MODE <= 1) {
    __syncthreads();
    acc[0] = smem[threadIdx.x];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}

template <int MODE>
void run(const char* name, const float* src, size_t total, unsigned span, int threads, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (threads / 64) * 8 * 1024;
  deliver<MODE><<<256, threads, lds>>>(src, (unsigned)total, out, 4, span);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  deliver<MODE><<<256, threads, lds>>>(src, (unsigned)total, out, iters, span);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double pieces_per_cu = (double)iters * 8 * (threads / 64);
  const double gb = pieces_per_cu * 256 * 1024 / 1e9;
  printf("%-8s %d waves/CU span %4u MB: %7.1f GB/s  %6.1f cycles per 1 KiB piece and CU  (%.3f ms)\n", name, threads / 64,
         span >> 20, gb / (ms * 1e-3), ms * 1e-3 * 2.1e9 / pieces_per_cu, ms);
}

int main() {
  const size_t total = 1ull << 30;
  float *src, *out;
  hipMalloc(&src, total); hipMalloc(&out, 64);
  hipMemset(src, 0, total);
  for (unsigned span : {64u << 20, 512u << 20})
    for (int threads : {256, 512}) {
      run<0>("dma", src, total, span, threads, 2000, out);
      run<1>("dma_oob", src, total, span, threads, 2000, out);
      run<2>("reg", src, total, span, threads, 2000, out);
      run<3>("frag", src, total, span, threads, 2000, out);
    }
  return 0;
}
