// Dev tool: issue time of the three matrix instructions the kernels choose between, one wave per SIMD, four independent
// accumulation chains: v_mfma_f32_32x32x16_bf16, the legacy K = 8 v_mfma_f32_32x32x8_bf16_1k, v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  floatx16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
  bf16x8 x8; shortx4 x4 = {1, 2, 3, 4}; float xf = threadIdx.x;
  for (int i = 0; i < 8; ++i) x8[i] = (__bf16)(float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, x8, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, x8, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, x8, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x8, x8, a3, 0, 0, 0);
      } else if (KIND == 1) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(x4, x4, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(x4, x4, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(x4, x4, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(x4, x4, a3, 0, 0, 0);
      } else {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, xf, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, xf, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, xf, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, xf, a3, 0, 0, 0);
      }
    }
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 1234.5f) out[threadIdx.x] = s;
}
template <int KIND> static float t(float* out, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, iters); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; (void)hipMalloc(&out, 1 << 16);
  const int it = 20000;
  const char* n[] = {"32x32x16 bf16", "32x32x8 bf16_1k", "32x32x2 f32"};
  float ms[3] = {t<0>(out, it), t<1>(out, it), t<2>(out, it)};
  for (int i = 0; i < 3; ++i) printf("%-18s %8.3f ms  %6.2f ns per instruction (one wave per SIMD, 4 independent chains)\n", n[i], ms[i], ms[i] * 1e6 / (it * 16.0));
  return 0;
}
