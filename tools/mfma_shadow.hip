// Dev tool: what issues in the shadow of a v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles)?  One wave per SIMD streams
// MFMAs with K other instructions after each one; if they are free the time stays at the K = 0 value until the
// shadow is full.  Second table: the same instructions issued by ANOTHER wave of the SIMD instead (2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND, int K, bool OTHER>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ float lds[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  lds[tid] = tid; lds[tid + 512] = 1.f; lds[tid + 1024] = 2.f; lds[tid + 1536] = 2.f;
  __syncthreads();
  if (!OTHER && wave >= 4) return;
  floatx16 a0, a1;
  for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
  float x = 1.f + lane, y = 2.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = lane + i;
  int sa = wave, sb = 3;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (1 << 20)), 0, 1 << 25, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(out + 65536), 0, 1 << 18, 0x00020000);
  const bool mf = wave < 4, oth = OTHER ? wave >= 4 : true;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (mf) {
        if (q & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      }
      if (oth) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(1.0001f));
          if (KIND == 1) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(sa) : "s"(sb));
          if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"((lane * 4 + i * 256) & 8191));
          if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(3));
          if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(3.f));
          if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<float __attribute__((ext_vector_type(2)))*>(&v[(i & 7) * 2])) : "v"(*reinterpret_cast<float __attribute__((ext_vector_type(2)))*>(&v[((i + 1) & 7) * 2])));
          if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1.5f));
          if (KIND == 8) asm volatile("ds_write_b64 %0, %1" : : "v"((lane * 8 + i * 512) & 8191), "v"(*reinterpret_cast<float __attribute__((ext_vector_type(2)))*>(&v[(i & 7) * 2])));
          if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(x));
          if (KIND == 10) asm volatile("ds_write_b128 %0, %1" : : "v"((lane * 16 + i * 1024) & 8191), "v"(*reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[(i & 3) * 4])));
          if (KIND == 11) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(*reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[(i & 3) * 4])) : "v"(lane * 16 + i * 1024 + wave * 16384), "s"(rs));
          if (KIND == 12) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v[i]));
          if (KIND == 14) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(*reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[(i & 3) * 4])), "v"((lane & 31) * 512 + (lane >> 5) * 16 + (i & 7) * 32 + wave * 16384 + blockIdx.x * 65536), "s"(rsw));
          if (KIND == 15) asm volatile("buffer_store_dword %0, %1, %2, 0 offen" : : "v"(v[i]), "v"(lane * 4 + (i & 15) * 512 + wave * 16384 + blockIdx.x * 65536), "s"(rsw));
          if (KIND == 16) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(*reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[(i & 3) * 4])), "v"(lane * 16 + (i & 15) * 1024 + wave * 16384 + blockIdx.x * 65536), "s"(rsw));
          if (KIND == 13) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(*reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[(i & 3) * 4])) : "v"((lane & 31) * 256 + (lane >> 5) * 32 + i * 16), "s"(rs));
          if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<float __attribute__((ext_vector_type(4)))*>(&v[(i & 3) * 4])) : "v"((lane * 16 + i * 1024) & 8191));
        }
      }
    }
    if (KIND >= 2) asm volatile("s_waitcnt lgkmcnt(0)");
    if (KIND == 11 || KIND >= 13) asm volatile("s_waitcnt vmcnt(0)");
  }
  float s = sa;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + v[r];
  if (s == 1234.5f) out[tid] = s;
}

template <int KIND, int K, bool OTHER>
static float t(float* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, K, OTHER>), dim3(256), dim3(512), 0, 0, out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, K, OTHER>), dim3(256), dim3(512), 0, 0, out, 4000);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (4000.f * 8);   // ns per MFMA slot
}
template <int KIND, bool OTHER>
static void row(const char* name, float* out) {
  printf("%-14s %s  K=0 %5.1f  1 %5.1f  2 %5.1f  4 %5.1f  8 %5.1f  12 %5.1f  16 %5.1f   ns per MFMA (+ K instructions)\n", name,
         OTHER ? "other wave" : "same wave ", t<KIND, 0, OTHER>(out), t<KIND, 1, OTHER>(out), t<KIND, 2, OTHER>(out), t<KIND, 4, OTHER>(out),
         t<KIND, 8, OTHER>(out), t<KIND, 12, OTHER>(out), t<KIND, 16, OTHER>(out));
}
int main(int argc, char** argv) {
  float* out; (void)hipMalloc(&out, (1 << 22) + (1 << 25) + (1 << 20));
  if (argc > 1) {
    row<14, false>("store_x4 32 lines", out);
    row<15, false>("store_b32 2 lines", out);
    row<16, false>("store_x4 1KB", out);
    return 0;
  }
  row<0, false>("v_fma_f32", out);  row<0, true>("v_fma_f32", out);
  row<1, false>("s_mul_i32", out);  row<1, true>("s_mul_i32", out);
  row<2, false>("ds_read_b32", out); row<2, true>("ds_read_b32", out);
  row<3, false>("ds_read_b128", out); row<3, true>("ds_read_b128", out);
  row<4, false>("v_add_u32", out); row<4, true>("v_add_u32", out);
  row<5, false>("v_cndmask_b32", out); row<5, true>("v_cndmask_b32", out);
  row<6, false>("v_pk_add_f32", out); row<6, true>("v_pk_add_f32", out);
  row<7, false>("v_add_f32", out); row<7, true>("v_add_f32", out);
  row<8, false>("ds_write_b64", out); row<8, true>("ds_write_b64", out);
  row<9, false>("v_mov_b32", out); row<9, true>("v_mov_b32", out);
  row<10, false>("ds_write_b128", out); row<10, true>("ds_write_b128", out);
  row<11, false>("buf_load_x4 1KB", out); row<11, true>("buf_load_x4 1KB", out);
  row<13, false>("buf_load_x4 rows", out); row<13, true>("buf_load_x4 rows", out);
  row<12, false>("accvgpr_read", out); row<12, true>("accvgpr_read", out);
  row<14, false>("store_x4 32 lines", out);
  row<15, false>("store_b32 2 lines", out);
  row<16, false>("store_x4 1KB", out);
  return 0;
}
