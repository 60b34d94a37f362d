// Dev tool: sustained fp32-MFMA rate of this chip (v_mfma_f32_32x32x2_f32) with random register data,
// for 1 / 2 / 4 waves per SIMD — the practical ceiling the conv kernels are priced against — plus the
// shader clock actually held under that load (s_memtime ticks / s_memrealtime ticks at 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int NACC, bool LDS>
__global__ void k(float* out, long long* clk, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float sm[192 * 36];
  floatx16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = threadIdx.x; i < 192 * 36; i += blockDim.x) sm[i] = seed + i * 1e-4f;
  __syncthreads();
  float a = seed + threadIdx.x * 0.37f, b = seed * 1.3f - threadIdx.x * 0.11f;
  const int lane = threadIdx.x & 63;
  const float* Ab = sm + (lane & 31) * 36 + (lane >> 5) * 4;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (LDS) {
      // the conv kernel's phase A: 3 b128 fragment reads feed 8 MFMAs (per-wave tile 32x64), 4 groups per tile
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        floatx4 af = *reinterpret_cast<const floatx4*>(Ab + g * 8);
        floatx4 b0 = *reinterpret_cast<const floatx4*>(Ab + 64 * 36 + g * 8);
        floatx4 b1 = *reinterpret_cast<const floatx4*>(Ab + 128 * 36 + g * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b0[s], acc[0], 0, 0, 0);
          acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], b1[s], acc[1 % NACC], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      a = a * 0.999f + 0.001f; b = b * 1.0001f;
    }
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  float s = 0;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC, bool LDS>
static void run(const char* name, int wps, int wgs_per_cu, int iters, int mfma_per_iter, float* d, long long* dc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * wgs_per_cu), block(256 * wps / wgs_per_cu);
  hipLaunchKernelGGL((k<NACC, LDS>), grid, block, 0, 0, d, dc, 100, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDS>), grid, block, 0, 0, d, dc, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc[2]; hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  double fl = 256.0 * 4 * wps * (double)iters * mfma_per_iter * 2 * 32 * 32 * 2;
  printf("%-10s waves/SIMD=%d wg/CU=%d  %8.3f ms  %6.1f TFLOP/s   shader clock %.0f MHz (memtime %lld / realtime %lld)\n",
         name, wps, wgs_per_cu, ms, fl / ms / 1e9, hc[0] / (hc[1] / 100.0), hc[0], hc[1]);
}
int main() {
  float* d; hipMalloc(&d, 4);
  long long* dc; hipMalloc(&dc, 16);
  for (int wps : {1, 2, 4}) run<4, false>("reg", wps, 1, 20000, 4, d, dc);
  run<4, false>("reg-long", 4, 1, 400000, 4, d, dc);
  run<2, false>("reg-2acc", 4, 1, 40000, 2, d, dc);
  run<2, false>("reg-2acc", 2, 1, 40000, 2, d, dc);
  run<2, false>("reg-2acc", 1, 1, 40000, 2, d, dc);
  run<2, true>("lds", 4, 2, 5000, 32, d, dc);
  run<2, true>("lds", 2, 1, 5000, 32, d, dc);
  run<2, true>("lds", 1, 1, 5000, 32, d, dc);
  return 0;
}
