// Dev tool: which part of igemm_pk_kernel<4,1,1,2> (128 x 64 tile, 4 waves, wave tile 32 x 64, BK = 32, two workgroups
// per CU) costs what?  The k-loop is rebuilt piece by piece on synthetic data:
//   F  fragment reads from LDS (3 b128 per 8 MFMAs, next group prefetched)      B  one barrier per k-tile
//   W  6 ds_write_b128 per k-tile (the staged tile)                             L  6 buffer_load_b128 per k-tile (HBM stream)
//   E  every 18 k-tiles an epilogue: 32 buffer_store_b32 per wave + accumulator reset
// Each variant reports TFLOP/s, cycles per k-tile of a workgroup pair (ideal 4096) and the shader clock it ran at.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 36, BM = 128, BN = 64, STAGE = (BM + BN) * LDK;

template <bool F, bool B, bool W, bool L, bool E>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, const float* __restrict__ wts, float* __restrict__ dst,
                                            long long* clk, int iters, long long src_floats) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 2 * STAGE; i += 256) sm[i] = 1.f + i * 1e-4f;
  __syncthreads();
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const int a_frag = (wave * 32 + l31) * LDK + h * 4, b_frag = (BM + l31) * LDK + h * 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, 64 * 576 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0x7fffffff, 0x00020000);
  floatx16 acc[2];
  for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  floatx4 va[4], vb[2];
  for (int i = 0; i < 4; ++i) va[i] = floatx4{1.f, 2.f, 3.f, 4.f};
  for (int i = 0; i < 2; ++i) vb[i] = floatx4{1.f, 2.f, 3.f, 4.f};
  // a workgroup streams over its own 128-pixel tiles of a [pixels][64] fp32 activation (256 B per pixel), 18 k-tiles each
  unsigned tile = blockIdx.x;
  const unsigned ntiles = (unsigned)(src_floats / (128 * 64));
  long long c0 = clock64(), w0 = wall_clock64();
  int u = 0, kt = 0;
  for (int it = 0; it < iters; ++it, u ^= 1) {
    const float* Ab = sm + u * STAGE + a_frag;
    const float* Bb = sm + u * STAGE + b_frag;
    float* nxt = sm + (u ^ 1) * STAGE;
    floatx4 af[2], bf[2][2];
    if (F) {
      af[0] = *reinterpret_cast<const floatx4*>(Ab);
      bf[0][0] = *reinterpret_cast<const floatx4*>(Bb);
      bf[0][1] = *reinterpret_cast<const floatx4*>(Bb + 32 * LDK);
    } else {
      af[0] = af[1] = floatx4{1.f, 2.f, 3.f, (float)lane};
      bf[0][0] = bf[0][1] = bf[1][0] = bf[1][1] = floatx4{0.5f, 2.f, 1.f, (float)wave};
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (F && g + 1 < 4) {
        af[(g + 1) & 1] = *reinterpret_cast<const floatx4*>(Ab + (g + 1) * 8);
        bf[(g + 1) & 1][0] = *reinterpret_cast<const floatx4*>(Bb + (g + 1) * 8);
        bf[(g + 1) & 1][1] = *reinterpret_cast<const floatx4*>(Bb + 32 * LDK + (g + 1) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (W && g == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<floatx4*>(&nxt[(lrow + 32 * i) * LDK + lcol]) = va[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<floatx4*>(&nxt[(BM + lrow + 32 * i) * LDK + lcol]) = vb[i];
      }
      if (L && g == 1) {
        const unsigned tap = (unsigned)(kt >> 1), cb = (unsigned)(kt & 1);
        const unsigned soff = (tile % ntiles) * (128u * 256u) + cb * 128u + (tap % 3) * 256u + (tap / 3) * (28u * 256u);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          va[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (unsigned)((lrow + 32 * i) * 256 + lcol * 4), (int)soff, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
          vb[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)((lrow + 32 * i) * 2304 + lcol * 4), (int)(kt * 128), 0));
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(F ? af[g & 1][s] : af[0][s], F ? bf[g & 1][j][s] : bf[0][j][s], acc[j], 0, 0, 0);
      if (W && g == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
      }
      if (L && g == 1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (B) __syncthreads();
    if (++kt == 18) {
      kt = 0;
      if (E) {
        const unsigned base = (tile % ntiles) * (128u * 256u);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned voff = (unsigned)(((wave * 32 + 4 * h) * 64 + j * 32 + l31) * 4);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[j][r]), rsD, voff, (int)(base + ((r & 3) + 8 * (r >> 2)) * 256), 0);
            acc[j][r] = 0.f;
          }
        }
      }
      tile += gridDim.x;
    }
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
  float s = 0;
  for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) dst[0] = s;
}

template <bool F, bool B, bool W, bool L, bool E>
static void run(const char* name, const float* src, const float* wts, float* dst, long long* dc, long long src_floats) {
  const int iters = 18 * 40, grid = 512;
  auto kern = k<F, B, W, L, E>;
  const size_t lds = sizeof(float) * 2 * STAGE;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, wts, dst, dc, 36, src_floats);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, wts, dst, dc, iters, src_floats);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long hc[2]; (void)hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  const double fl = (double)grid * 4 * iters * 32 * 2 * 32 * 32 * 2;
  printf("%-34s %8.3f ms %6.1f TFLOP/s  cycles per k-tile pair %.0f (ideal 4096)  clock %.0f MHz\n", name, ms, fl / ms / 1e9,
         (double)hc[0] / iters, hc[0] / (hc[1] / 100.0));
}

#include <stdlib.h>
int main(int argc, char** argv) {
  const long long src_floats = 401408ll * 64;          // conv2x activation: 103 MB
  float *src, *wts, *dst; long long* dc;
  (void)hipMalloc(&src, src_floats * 4 + (1 << 20)); (void)hipMalloc(&wts, 64 * 576 * 4 + 4096); (void)hipMalloc(&dst, src_floats * 4 + (1 << 20));
  (void)hipMalloc(&dc, 16);
  // operands: zeros (argv[1] == "zero") or N(0,1)-like noise — the matrix pipe's power, and with it the clock the
  // chip holds, depends on the data it multiplies
  const bool zero = argc > 1 && argv[1][0] == 'z';
  {
    float* hbuf = (float*)malloc(src_floats * 4);
    unsigned st = 12345u;
    for (long long i = 0; i < src_floats; ++i) {
      st = st * 1664525u + 1013904223u;
      const float u = (float)(st >> 8) * (1.f / 16777216.f);
      hbuf[i] = zero ? 0.f : (u - 0.5f) * 3.4f;
    }
    (void)hipMemcpy(src, hbuf, src_floats * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(wts, hbuf + 1000, 64 * 576 * 4, hipMemcpyHostToDevice);
    free(hbuf);
  }
  run<false, false, false, false, false>("MFMA only", src, wts, dst, dc, src_floats);
  run<true, false, false, false, false>("F", src, wts, dst, dc, src_floats);
  run<true, true, false, false, false>("F B", src, wts, dst, dc, src_floats);
  run<true, true, false, false, true>("F B E", src, wts, dst, dc, src_floats);
  run<true, true, true, false, false>("F B W", src, wts, dst, dc, src_floats);
  run<true, true, false, true, false>("F B L", src, wts, dst, dc, src_floats);
  run<true, true, true, true, false>("F B W L", src, wts, dst, dc, src_floats);
  run<true, true, true, true, true>("F B W L E  (the kernel's loop)", src, wts, dst, dc, src_floats);
  run<false, false, false, true, false>("L only (loads, no LDS)", src, wts, dst, dc, src_floats);
  run<false, false, true, false, false>("W only (LDS writes)", src, wts, dst, dc, src_floats);
  run<false, false, false, false, true>("E only (stores)", src, wts, dst, dc, src_floats);
  return 0;
}
