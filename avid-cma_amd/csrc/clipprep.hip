// Video clip front end on the GPU — SURVEY.md §8(f) rank 4 ("clip normalisation").
//
// Reference: utils/videotransforms/volume_transforms.py:14-66 (ClipToTensor: m frames of H x W x 3 uint8 ->
// float tensor [3][m][H][W], divided by 255) followed by utils/videotransforms/tensor_transforms.py:13-37 /
// utils/functional.py:8-17 (Normalize: tensor.sub_(mean).div_(std) per channel), as composed by
// datasets/preprocessing.py:45-48.  Same fp32 operation order (u / 255, - mean, / std: IEEE divisions), so the
// result is bit-identical to the reference's CPU transform.
//
// HBM-bound: 3 B read + 12 B written per pixel.  One thread = 4 consecutive pixels of a row (12 input bytes as
// three 32-bit loads when W % 4 == 0) -> one float4 store into each of the three channel planes.
#include "common.h"

namespace avid {

__global__ __launch_bounds__(256) void clip_normalize_kernel(const uint8_t* __restrict__ frames, float* __restrict__ out,
                                                             long long npix4, long long plane, int T_HW4, int W4,
                                                             float m0, float m1, float m2, float s0, float s1,
                                                             float s2, int vec) {
  // pixel quads are numbered over [B][T][H][W/4]; plane = T*H*W floats of one channel of one clip
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < npix4; q += stride) {
    const long long b = q / T_HW4, r = q - b * T_HW4;          // r: quad inside the clip
    unsigned char px[12];
    const uint8_t* src = frames + q * 12;
    if (vec) {
      const unsigned w0 = reinterpret_cast<const unsigned*>(src)[0], w1 = reinterpret_cast<const unsigned*>(src)[1],
                     w2 = reinterpret_cast<const unsigned*>(src)[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        px[k] = (w0 >> (8 * k)) & 0xff;
        px[4 + k] = (w1 >> (8 * k)) & 0xff;
        px[8 + k] = (w2 >> (8 * k)) & 0xff;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) px[k] = src[k];
    }
    floatx4 c0, c1, c2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c0[k] = ((float)px[3 * k + 0] / 255.f - m0) / s0;
      c1[k] = ((float)px[3 * k + 1] / 255.f - m1) / s1;
      c2[k] = ((float)px[3 * k + 2] / 255.f - m2) / s2;
    }
    float* o = out + b * 3 * plane + r * 4;
    *reinterpret_cast<floatx4*>(o) = c0;
    *reinterpret_cast<floatx4*>(o + plane) = c1;
    *reinterpret_cast<floatx4*>(o + 2 * plane) = c2;
  }
  (void)W4;
}

// generic widths: one thread per pixel
__global__ __launch_bounds__(256) void clip_normalize_px_kernel(const uint8_t* __restrict__ frames,
                                                                float* __restrict__ out, long long npix,
                                                                long long plane, float m0, float m1, float m2,
                                                                float s0, float s1, float s2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
    const long long b = i / plane, r = i - b * plane;
    const uint8_t* src = frames + i * 3;
    float* o = out + b * 3 * plane + r;
    o[0] = ((float)src[0] / 255.f - m0) / s0;
    o[plane] = ((float)src[1] / 255.f - m1) / s1;
    o[2 * plane] = ((float)src[2] / 255.f - m2) / s2;
  }
}

}  // namespace avid

using namespace avid;

extern "C" int avid_clip_normalize(int B, int T, int H, int W, const uint8_t* frames, const float* mean3,
                                   const float* std3, float* out, avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0, AVID_E_SHAPE, "clip_normalize: bad shape");
  AVID_REQUIRE(frames && mean3 && std3 && out, AVID_E_BADARG, "clip_normalize: null pointer");
  AVID_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, AVID_E_BADARG, "clip_normalize: zero std");
  hipStream_t s = (hipStream_t)stream;
  const long long plane = (long long)T * H * W, npix = (long long)B * plane;
  ScopedTimer t(s, "clip_normalize_kernel", 0.0, 15.0 * npix);
  if (W % 4 == 0) {
    const long long n4 = npix / 4;
    long long g = ceil_div(n4, 256);
    if (g > 4096) g = 4096;
    const int vec = (reinterpret_cast<uintptr_t>(frames) & 3) == 0 ? 1 : 0;
    hipLaunchKernelGGL(clip_normalize_kernel, dim3((unsigned)g), dim3(256), 0, s, frames, out, n4, plane,
                       (int)(plane / 4), W / 4, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], vec);
  } else {
    long long g = ceil_div(npix, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(clip_normalize_px_kernel, dim3((unsigned)g), dim3(256), 0, s, frames, out, npix, plane, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  }
  return check_launch("clip_normalize");
}
