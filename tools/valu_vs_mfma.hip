// Dev tool: does a wave that streams back-to-back v_mfma_f32_32x32x2_f32 leave issue slots to ANOTHER wave of its SIMD?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run the MFMA stream, waves 4-7 a stream of one other
// instruction kind sized to take about as long when alone.  Host events only: together ~ max(a, b) means the two
// overlap, ~ a + b means the second stream only ran after the first.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int other_iters, int prio, int swap, int gap) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = swap ? ((tid >> 6) ^ 4) : (tid >> 6);     // swap: the other stream's waves are the older ones
  lds[tid] = tid; lds[tid + 512] = 1.f;
  if (prio && wave >= 4) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  if (wave < 4) {
    floatx16 a0, a1;
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
    float x = 1.f + lane, y = 2.f;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        if (gap == 1) asm volatile("s_nop 0");
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        if (gap == 1 || gap == 2) asm volatile("s_nop 0");
      }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    if (s == 1234.5f) out[tid] = s;
    return;
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane + i;
  int sacc = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll 1
  for (int it = 0; it < other_iters; ++it) {
    if (KIND == 0) {          // fp32 VALU, 8 independent chains
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(1.0001f));
    } else if (KIND == 2) {   // scalar ALU
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sacc));
    } else if (KIND == 3) {   // LDS reads
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += lds[(lane * 4 + i * 64 + it) & 1023];
    }
  }
  float s = sacc;
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 1234.5f) out[tid] = s;
}

template <int KIND>
static float time_it(float* out, int mi, int oi, int prio, int swap, int gap) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, mi, oi, prio, swap, gap);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, mi, oi, prio, swap, gap);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int KIND>
static void run(const char* name, float* out, int oi) {
  const int mi = 2000;
  const float a = time_it<KIND>(out, mi, 0, 0, 0, 0), b = time_it<KIND>(out, 0, oi, 0, 0, 0);
  printf("%-14s MFMA alone %6.3f ms, %s alone %6.3f ms | together %6.3f | other waves older %6.3f | s_setprio 3 on the other %6.3f |"
         " s_nop after every MFMA %6.3f / every 2nd %6.3f\n", name, a, name, b, time_it<KIND>(out, mi, oi, 0, 0, 0),
         time_it<KIND>(out, mi, oi, 0, 1, 0), time_it<KIND>(out, mi, oi, 1, 0, 0), time_it<KIND>(out, mi, oi, 0, 0, 1),
         time_it<KIND>(out, mi, oi, 0, 0, 2));
}

int main() {
  float* out;
  (void)hipMalloc(&out, 1 << 20);
  run<0>("v_fma_f32", out, 60000);
  run<2>("s_add_i32", out, 60000);
  run<3>("ds_read+v_add", out, 15000);
  return 0;
}
