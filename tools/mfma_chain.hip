// Dev tool: does a chain of DEPENDENT v_mfma_f32_32x32x2_f32 (same accumulator back to back) run at the rate of
// independent ones?  One wave per SIMD, NACC accumulators used round-robin in runs of RUN.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC, int RUN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  floatx16 a[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) a[i][r] = 0.f;
  float x = 1.f + threadIdx.x, y = 2.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 64 / (NACC * RUN); ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < RUN; ++r) a[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += a[i][r];
  if (s == 1234.5f) out[threadIdx.x] = s;
}
template <int NACC, int RUN>
static void t(float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, RUN>), dim3(256), dim3(256), 0, 0, out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, RUN>), dim3(256), dim3(256), 0, 0, out, 4000);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fl = 256.0 * 4 * 4000 * 64 * 4096.0;
  printf("accumulators %2d  run %2d : %7.1f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", NACC, RUN, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (4000.0 * 64));
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  t<1, 1>(out); t<2, 1>(out); t<4, 1>(out); t<2, 8>(out); t<16, 1>(out); t<16, 4>(out); t<8, 8>(out);
  return 0;
}
