// input_prologue.hip — the per-clip input step as ONE pass.
//
// Reference (CPU, per clip, datasets/kinetics.py:230-248): uint8 THWC frames -> tensor_normalize -> permute ->
// spatial_sampling (bilinear short-side scale, crop, horizontal flip) -> pack_pathway_output (slow = frames at
// linspace indices).  That is four full-size float tensors per clip on the host and a 4x larger H2D copy.
// Here the decoded uint8 clip is read once per pathway and the result lands directly in the stem's input layout
// (NDHWC, channels padded to 4, zero H/W borders, row pitch Wp): each thread owns one destination pixel, pulls
// its (up to 4) source taps, normalises them and writes one aligned 16-byte store.  HBM-bound: 3 B read per tap,
// 16 B written per pixel.
#include "common.h"

namespace {
constexpr int TPB = 256;

struct NormArgs {
  float m[3], s[3];
};

__device__ __forceinline__ float norm1(unsigned char u, float m, float s) {
  return __fdiv_rn(__fdiv_rn((float)u, 255.f) - m, s);  // tensor/255 - mean, then /std (datasets/utils.py:306-314)
}

// torch upsample_bilinear (align_corners=False): src = scale*(dst+0.5)-0.5 clamped at 0; i0 = min(int(src), n-1);
// lambda = src - i0 clamped to [0,1]; i1 = i0 + (i0 < n-1).
__device__ __forceinline__ void src_index(float scale, int dst, int n, int* i0, int* i1, float* l0, float* l1) {
  float r = scale * ((float)dst + 0.5f) - 0.5f;
  if (r < 0.f) r = 0.f;
  int a = (int)r;
  if (a > n - 1) a = n - 1;
  float l = r - (float)a;
  l = fminf(fmaxf(l, 0.f), 1.f);
  *i0 = a;
  *i1 = a + (a < n - 1 ? 1 : 0);
  *l1 = l;
  *l0 = 1.f - l;
}

__global__ void clip_prologue_kernel(const unsigned char* __restrict__ clip, int T, int H, int W, int new_h,
                                     int new_w, int y0, int x0, int crop, int flip, int reverse, NormArgs nm,
                                     const int* __restrict__ frame_idx, int n_frames, float* __restrict__ dst, int ph,
                                     int pw, int Wp, long total) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int Hp = crop + 2 * ph;
  const int xp = (int)(idx % Wp);
  long r = idx / Wp;
  const int yp = (int)(r % Hp);
  const int f = (int)(r / Hp);
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  const int yy = yp - ph, xx = xp - pw;
  if (yy >= 0 && yy < crop && xx >= 0 && xx < crop) {
    const int t = frame_idx ? frame_idx[f] : f;
    const int ys = y0 + yy;                            // row in the scaled image
    const int xs = x0 + (flip ? crop - 1 - xx : xx);   // column in the scaled image (flip is applied after the crop)
    const unsigned char* fr = clip + (long)t * H * W * 3;
    float v[3];
    if (new_h == H && new_w == W) {
      const unsigned char* p = fr + ((long)ys * W + xs) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = norm1(p[c], nm.m[c], nm.s[c]);
    } else {
      int ya, yb, xa, xb;
      float ly0, ly1, lx0, lx1;
      src_index((float)H / (float)new_h, ys, H, &ya, &yb, &ly0, &ly1);
      src_index((float)W / (float)new_w, xs, W, &xa, &xb, &lx0, &lx1);
      const unsigned char* p00 = fr + ((long)ya * W + xa) * 3;
      const unsigned char* p01 = fr + ((long)ya * W + xb) * 3;
      const unsigned char* p10 = fr + ((long)yb * W + xa) * 3;
      const unsigned char* p11 = fr + ((long)yb * W + xb) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float a = norm1(p00[c], nm.m[c], nm.s[c]), b = norm1(p01[c], nm.m[c], nm.s[c]);
        const float d = norm1(p10[c], nm.m[c], nm.s[c]), e = norm1(p11[c], nm.m[c], nm.s[c]);
        v[c] = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * d + lx1 * e);
      }
    }
    if (reverse) {  // DATA.REVERSE_INPUT_CHANNEL: frames[[2, 1, 0]]
      o[0] = v[2]; o[1] = v[1]; o[2] = v[0];
    } else {
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
  }
  *reinterpret_cast<f32x4*>(dst + idx * 4) = o;
}
}  // namespace

extern "C" int sf_clip_prologue(const unsigned char* clip, int T, int H, int W, int new_h, int new_w, int y0, int x0,
                                int crop, int flip, int reverse, const float* mean3, const float* std3,
                                const int* frame_idx, int n_frames, float* dst, int ph, int pw, int Wp, void* stream) {
  if (!clip || !dst || !mean3 || !std3 || T <= 0 || H <= 0 || W <= 0 || new_h <= 0 || new_w <= 0 || crop <= 0 ||
      n_frames <= 0 || ph < 0 || pw < 0)
    return SF_EINVAL;
  if (y0 < 0 || x0 < 0 || y0 + crop > new_h || x0 + crop > new_w || Wp < crop + 2 * pw) return SF_EINVAL;
  if (!frame_idx && n_frames != T) return SF_EINVAL;
  if (!sf_aligned16(dst)) return SF_EINVAL;
  NormArgs nm;
  for (int c = 0; c < 3; ++c) {
    nm.m[c] = mean3[c];   // HOST pointers: three floats each
    nm.s[c] = std3[c];
  }
  const long total = (long)n_frames * (crop + 2 * ph) * Wp;
  hipLaunchKernelGGL(clip_prologue_kernel, dim3(sf_cdiv(total, TPB)), dim3(TPB), 0, (hipStream_t)stream, clip, T, H, W,
                     new_h, new_w, y0, x0, crop, flip, reverse, nm, frame_idx, n_frames, dst, ph, pw, Wp, total);
  SF_CHECK_LAUNCH();
  return SF_OK;
}
