// Dev tool: the compute loop of igemm_pk_kernel in isolation (4 waves, 64x64 register tile per wave,
// 4 b128 fragment reads per 16 MFMAs, one barrier per k-tile) — how close to the MFMA peak does the
// loop structure itself get at 1 and 2 waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 36;
template <int BM, int BN, int WN, bool BARRIER, bool PREFETCH>
__global__ __launch_bounds__(256, 2) void k(float* out, long long* clk, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int STAGE = (BM + BN) * LDK;
  for (int i = threadIdx.x; i < 2 * STAGE; i += blockDim.x) sm[i] = seed + i * 1e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int a_frag = (wm * 64 + l31) * LDK + h * 4, b_frag = (BM + wn * 64 + l31) * LDK + h * 4;
  floatx16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  long long c0 = clock64(), w0 = wall_clock64();
  int u = 0;
  for (int it = 0; it < iters; ++it, u ^= 1) {
    const float* Ab = sm + u * STAGE + a_frag;
    const float* Bb = sm + u * STAGE + b_frag;
    floatx4 af[2][2], bf[2][2];
    for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK);
    for (int j = 0; j < 2; ++j) bf[0][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (PREFETCH && g + 1 < 4) {
        for (int i = 0; i < 2; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + (g + 1) * 8);
        for (int j = 0; j < 2; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + (g + 1) * 8);
      }
      if (PREFETCH) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[PREFETCH ? (g & 1) : 0][i][s], bf[PREFETCH ? (g & 1) : 0][j][s], acc[i][j], 0, 0, 0);
      if (PREFETCH) __builtin_amdgcn_sched_barrier(0);
      if (!PREFETCH && g + 1 < 4) {
        for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + (g + 1) * 8);
        for (int j = 0; j < 2; ++j) bf[0][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + (g + 1) * 8);
      }
    }
    if (BARRIER) __syncthreads();
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}
template <int BM, int BN, int WN, bool BARRIER, bool PREFETCH>
static void run(const char* name, int wgs_per_cu, float* d, long long* dc) {
  const int iters = 2000;
  auto kern = k<BM, BN, WN, BARRIER, PREFETCH>;
  const size_t lds = sizeof(float) * 2 * (BM + BN) * LDK;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), lds, 0, d, dc, 10, 1.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), lds, 0, d, dc, iters, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long hc[2]; (void)hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  double fl = 256.0 * wgs_per_cu * 4 * (double)iters * 64 * 2 * 32 * 32 * 2;
  printf("%-28s wg/CU=%d lds=%zuK %8.3f ms %6.1f TFLOP/s  cycles/k-tile %.0f (ideal 4096)  clock %.0f MHz\n", name, wgs_per_cu,
         lds / 1024, ms, fl / ms / 1e9, (double)hc[0] / iters, hc[0] / (hc[1] / 100.0));
}
int main() {
  float* d; (void)hipMalloc(&d, 4);
  long long* dc; (void)hipMalloc(&dc, 16);
  run<256, 64, 1, true, true>("256x64 barrier prefetch", 1, d, dc);
  run<256, 64, 1, false, true>("256x64 nobarrier prefetch", 1, d, dc);
  run<256, 64, 1, true, false>("256x64 barrier noprefetch", 1, d, dc);
  run<256, 64, 1, false, false>("256x64 nobarrier noprefetch", 1, d, dc);
  run<128, 128, 2, true, true>("128x128 barrier prefetch", 1, d, dc);
  run<128, 128, 2, true, true>("128x128 barrier prefetch", 2, d, dc);
  run<128, 128, 2, false, true>("128x128 nobarrier prefetch", 2, d, dc);
  return 0;
}
