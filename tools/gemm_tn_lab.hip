// Dev tool: the wgrad-shaped GEMM (both operands reduction-major: C[M][N] = sum_k A[k][M] * B[k][N]) — what
// does the 4-wave double-buffered structure reach when the fragments are 4-byte LDS reads (ds_read2_b32)
// instead of the forward kernel's b128 reads?  Persistent workgroups, K split across them like wgrad.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// tile: (64*NB) x (64*KC) per workgroup of 4 waves (2x2), wave tile (32*NB) x (32*KC); chunk = 32 k rows
template <int NB, int KC, int OCC, bool PREFETCH>
__global__ __launch_bounds__(256, OCC) void gemm_tn(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ C, int M, int N, int K, int chunks_per_wg) {
  constexpr int BM = 64 * NB, BN = 64 * KC;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int STAGE = 32 * (LDA + LDB);
  constexpr int PA = 32 * BM / 4 / 256, PB = 32 * BN / 4 / 256;   // float4 per thread per chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int ntm = M / BM, ntn = N / BN, tiles = ntm * ntn;
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int k0 = split * chunks_per_wg * 32;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
  floatx4 va[PA], vb[PB];
  unsigned a_off[PA], b_off[PB];
  int a_r[PA], a_c[PA], b_r[PB], b_c[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int e = tid + 256 * i; a_r[i] = e / (BM / 4); a_c[i] = (e % (BM / 4)) * 4;
    a_off[i] = (unsigned)(((long long)(k0 + a_r[i]) * M + m0 + a_c[i]) * 4);
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int e = tid + 256 * i; b_r[i] = e / (BN / 4); b_c[i] = (e % (BN / 4)) * 4;
    b_off[i] = (unsigned)(((long long)(k0 + b_r[i]) * N + n0 + b_c[i]) * 4);
  }
  auto issue = [&](int ch) {
#pragma unroll
    for (int i = 0; i < PA; ++i) va[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[i], ch * 32 * M * 4, 0));
#pragma unroll
    for (int i = 0; i < PB; ++i) vb[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[i], ch * 32 * N * 4, 0));
  };
  auto store = [&](float* st) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<floatx4*>(&st[a_r[i] * LDA + a_c[i]]) = va[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<floatx4*>(&st[32 * LDA + b_r[i] * LDB + b_c[i]]) = vb[i];
  };
  floatx16 acc[NB][KC];
  for (int t = 0; t < NB; ++t) for (int j = 0; j < KC; ++j) for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
  issue(0); store(smem);
  if (chunks_per_wg > 1) issue(1);
  __syncthreads();
  for (int ch = 0; ch < chunks_per_wg; ++ch) {
    const float* cur = smem + (ch & 1) * STAGE;
    float* nxt = smem + ((ch & 1) ^ 1) * STAGE;
    const float* Ab = cur + wm * 32 * NB + l31;
    const float* Bb = cur + 32 * LDA + wn * 32 * KC + l31;
    float a[2][2][NB], b[2][2][KC];
    auto frag = [&](int kp, int buf) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kk = 2 * kp + q;
#pragma unroll
        for (int t = 0; t < NB; ++t) a[buf][q][t] = Ab[(2 * kk + h) * LDA + t * 32];
#pragma unroll
        for (int j = 0; j < KC; ++j) b[buf][q][j] = Bb[(2 * kk + h) * LDB + j * 32];
      }
    };
    if (PREFETCH) frag(0, 0);
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      if (PREFETCH) { if (kp + 1 < 8) frag(kp + 1, (kp + 1) & 1); __builtin_amdgcn_sched_barrier(0); }
      else frag(kp, kp & 1);
      if (kp == 0 && ch + 1 < chunks_per_wg) store(nxt);
      if (kp == 2 && ch + 2 < chunks_per_wg) issue(ch + 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
          for (int j = 0; j < KC; ++j)
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp & 1][q][t], b[kp & 1][q][j], acc[t][j], 0, 0, 0);
      if (PREFETCH) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  float* o = C + (long long)split * M * N;
  for (int t = 0; t < NB; ++t) for (int j = 0; j < KC; ++j) for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 * NB + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + wn * 32 * KC + j * 32 + l31;
    o[(long long)m * N + n] = acc[t][j][r];
  }
}
static float *dA, *dB, *dC;
template <int NB, int KC, int OCC, bool PF>
static void run(const char* name, int M, int N, int K) {
  auto kern = gemm_tn<NB, KC, OCC, PF>;
  const size_t lds = sizeof(float) * 2 * 32 * (64 * NB + 4 + 64 * KC + 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int tiles = (M / (64 * NB)) * (N / (64 * KC));
  int nsplit = 512 / tiles; if (nsplit < 1) nsplit = 1;
  int cps = K / 32 / nsplit; const int Ku = cps * nsplit * 32;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(tiles * nsplit), dim3(256), lds, 0, dA, dB, dC, M, N, Ku, cps);
  (void)hipDeviceSynchronize();
  const int reps = 10;
  (void)hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(tiles * nsplit), dim3(256), lds, 0, dA, dB, dC, M, N, Ku, cps);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  hipError_t err = hipGetLastError();
  printf("%-30s M=%4d N=%4d K=%7d tiles=%3d split=%3d lds=%3zuK %8.1f us %6.1f TFLOP/s %s\n", name, M, N, Ku, tiles, nsplit, lds / 1024,
         ms * 1e3, 2.0 * M * N * Ku / ms / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
}
int main() {
  (void)hipMalloc(&dA, 401408ull * 512 * 4); (void)hipMalloc(&dB, 401408ull * 640 * 4); (void)hipMalloc(&dC, 512ull * 4608 * 64 * 4);
  (void)hipMemset(dA, 0, 401408ull * 512 * 4); (void)hipMemset(dB, 0, 401408ull * 640 * 4);
  // conv2x spatial wgrad: dw 64 x 576 over 401408 pixels; conv3x: 128 x 1152 over 50176; conv5x: 512 x 4608 over 1024
  run<1, 3, 2, true>("64x192 pf", 64, 576, 401408);
  run<1, 3, 2, false>("64x192 nopf", 64, 576, 401408);
  run<2, 2, 2, true>("128x128 pf", 128, 1152, 50176);
  run<2, 2, 2, false>("128x128 nopf", 128, 1152, 50176);
  run<2, 2, 2, true>("128x128 pf", 256, 2304, 6272);
  run<2, 2, 2, true>("128x128 pf", 512, 4608, 1024);
  run<2, 4, 1, true>("128x256 pf occ1", 128, 1280, 50176);
  return 0;
}
