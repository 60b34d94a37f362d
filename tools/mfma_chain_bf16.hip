// Dev tool: v_mfma_f32_32x32x16_bf16 in runs of RUN dependent instructions (same accumulator back to back) over NACC
// accumulators, one wave per SIMD — what wino2_kernel's "six products of a transform point, then the next point" costs against
// an interleaved order.  (tools/mfma_chain.hip is the fp32 instruction's version.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int RUN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  floatx16 a[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) a[i][r] = 0.f;
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(1.f + threadIdx.x + e); y[e] = (__bf16)2.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 96 / (NACC * RUN); ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < RUN; ++r) a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a[i], 0, 0, 0);
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
  printf("accumulators %2d  run %2d : %6.2f ns per MFMA\n", NACC, RUN, ms * 1e6 / (4000.0 * 96));
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  t<1, 1>(out); t<2, 1>(out); t<4, 1>(out); t<16, 6>(out); t<16, 3>(out); t<16, 2>(out); t<16, 1>(out); t<2, 6>(out); t<2, 3>(out);
  return 0;
}
