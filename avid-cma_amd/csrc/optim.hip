// Flat fused Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient, bias
// correction as in torch's single-tensor path) over ONE contiguous fp32 buffer holding every
// parameter — replaces the 141 per-tensor optimizer launches of utils/main_utils.py:250-261.
#include <math.h>

#include "common.h"

namespace avid {

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n4,
                                                        long long n, float b1, float b2, float eps, float wd,
                                                        float step_size, float inv_sqrt_bc2, float grad_scale,
                                                        float lr, const long long* __restrict__ step_dev,
                                                        const float* __restrict__ lr_dev) {
  if (lr_dev) {   // graph-replay safe learning rate (scheduler writes the device word)
    const float nl = *lr_dev;
    step_size = lr != 0.f ? step_size * (nl / lr) : 0.f;   // (exact when step_dev is given: recomputed below)
    lr = nl;
  }
  if (step_dev) {  // graph-replay safe: bias corrections from the device-resident step counter
    const double t = (double)*step_dev;
    step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    floatx4 pv = reinterpret_cast<floatx4*>(p)[i];
    floatx4 gv = reinterpret_cast<const floatx4*>(g)[i];
    floatx4 mv = reinterpret_cast<floatx4*>(m)[i];
    floatx4 vv = reinterpret_cast<floatx4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * grad_scale + wd * pv[j];
      mv[j] = b1 * mv[j] + (1.f - b1) * gg;
      vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pv[j] -= step_size * (mv[j] / denom);
    }
    reinterpret_cast<floatx4*>(p)[i] = pv;
    reinterpret_cast<floatx4*>(m)[i] = mv;
    reinterpret_cast<floatx4*>(v)[i] = vv;
  }
  // tail (n % 4) handled by block 0
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
    const long long i = n4 * 4 + threadIdx.x;
    const float gg = g[i] * grad_scale + wd * p[i];
    m[i] = b1 * m[i] + (1.f - b1) * gg;
    v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
    p[i] -= step_size * (m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps));
  }
}

}  // namespace avid

using namespace avid;

extern "C" int avid_adam_flat(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t step, const int64_t* step_dev,
                              const float* lr_dev, float grad_scale, avid_stream_t stream) {
  AVID_REQUIRE(n > 0 && p && g && m && v && (step >= 1 || step_dev), AVID_E_BADARG, "adam_flat: bad argument");
  if (step < 1) step = 1;
  AVID_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, AVID_E_BADARG,
               "adam_flat: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const long long n4 = n / 4;
  long long grid = ceil_div(n4 > 0 ? n4 : 1, 256);
  if (grid > 4096) grid = 4096;
  ScopedTimer t((hipStream_t)stream, "adam_flat_kernel", 0.0, 28.0 * n);
  hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4,
                     (long long)n, beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2, grad_scale, lr,
                     (const long long*)step_dev, lr_dev);
  return check_launch("adam_flat");
}
