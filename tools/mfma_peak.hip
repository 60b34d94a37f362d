// Dev tool: sustained fp32-MFMA rate of this chip (v_mfma_f32_32x32x2_f32) with random register data,
// for 1 / 2 / 4 waves per SIMD — the practical ceiling the conv kernels are priced against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters, float seed) {
  floatx16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = seed + threadIdx.x * 0.37f, b = seed * 1.3f - threadIdx.x * 0.11f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    a = a * 0.999f + 0.001f; b = b * 1.0001f;
  }
  float s = 0;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4}) {
    const int iters = 20000, nacc = 4;
    dim3 grid(256), block(256 * wps);
    hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * 4 * wps * (double)iters * nacc * 2 * 32 * 32 * 2;
    printf("waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
  }
  return 0;
}
