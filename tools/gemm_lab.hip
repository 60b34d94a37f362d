// Dev tool: which workgroup shape keeps the fp32 MFMA pipe fed when the operands really come from HBM/L2?
// C[M][N] = A[M][K] . B[N][K]^T with the conv kernels' staging (buffer loads -> registers -> padded LDS rows
// -> b128 fragment reads), persistent workgroups, k-tiles double-buffered in LDS with one barrier each.
// Template: NW waves arranged WM x WN, wave tile (TM*32) x (TN*32), BK floats per k-tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TN, int BK, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M, int N, int K) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDK = BK + 4;
  constexpr int CPR = BK / 4;                 // float4 chunks per row
  constexpr int RPP = NT / CPR;               // rows staged per pass
  constexpr int PA = BM / RPP, PB = (BN + RPP - 1) / RPP;
  static_assert(BM % RPP == 0, "");
  constexpr int STAGE = (BM + BN) * LDK;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int lrow = tid / CPR, lcol = (tid % CPR) * 4;
  const int ntn = N / BN, ntiles = (M / BM) * ntn, G = gridDim.x;
  const int slot = blockIdx.x;
  if (slot >= ntiles) return;
  const int nk = K / BK;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
  int ld_tile = slot, ld_k = 0;
  unsigned a_off[PA], b_off[PB];
  floatx4 va[PA], vb[PB];
  auto setup = [&](int tile) {
    const int mt = tile / ntn, nt = tile - mt * ntn;
#pragma unroll
    for (int i = 0; i < PA; ++i) a_off[i] = (unsigned)(((mt * BM + lrow + RPP * i) * (long long)K + lcol) * 4);
#pragma unroll
    for (int i = 0; i < PB; ++i) b_off[i] = (unsigned)(((nt * BN + lrow + RPP * i) * K + lcol) * 4);
  };
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i)
      va[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[i], ld_k * 4, 0));
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (lrow + RPP * i < BN) vb[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[i], ld_k * 4, 0));
  };
  auto advance = [&]() {
    ld_k += BK;
    if (ld_k == K) { ld_k = 0; ld_tile += G; if (ld_tile < ntiles) setup(ld_tile); }
  };
  auto store = [&](float* st) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<floatx4*>(&st[(lrow + RPP * i) * LDK + lcol]) = va[i];
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (lrow + RPP * i < BN) *reinterpret_cast<floatx4*>(&st[(BM + lrow + RPP * i) * LDK + lcol]) = vb[i];
  };
  setup(slot);
  issue(); advance(); store(smem);
  if (ld_tile < ntiles) { issue(); advance(); }
  __syncthreads();
  const int a_frag = (wm * TM * 32 + l31) * LDK + h * 4, b_frag = (BM + wn * TN * 32 + l31) * LDK + h * 4;
  const int n_my = (ntiles - slot + G - 1) / G;
  int left = n_my * nk, u = 0;
  floatx16 acc[TM][TN];
  auto ktile = [&](auto ST, auto LD) {
    const float* Ab = smem + u * STAGE + a_frag;
    const float* Bb = smem + u * STAGE + b_frag;
    float* nxt = smem + (u ^ 1) * STAGE;
    floatx4 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + (g + 1) * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + (g + 1) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (decltype(ST)::value && g == 0) store(nxt);
      if (decltype(LD)::value && g == (BK / 8 > 1 ? 1 : 0)) issue();
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j], 0, 0, 0);
      if (decltype(ST)::value && g == 0) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
      }
      if (decltype(LD)::value && g == (BK / 8 > 1 ? 1 : 0)) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int tile = slot; tile < ntiles; tile += G) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int nst = left - 2 < nk ? left - 2 : nk;
    if (nst < 0) nst = 0;
    for (int ks = 0; ks < nst; ++ks, u ^= 1) { ktile(std::true_type{}, std::true_type{}); advance(); __syncthreads(); }
    left -= nst;
    if (nst < nk) {
      if (left == 2) { ktile(std::true_type{}, std::false_type{}); __syncthreads(); u ^= 1; --left; ++nst; }
      if (nst < nk) { ktile(std::false_type{}, std::false_type{}); __syncthreads(); u ^= 1; --left; }
    }
    const int mt = tile / ntn, nt = tile - mt * ntn;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * BM + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int col = nt * BN + (wn * TN + j) * 32 + l31;
          C[(long long)row * N + col] = acc[i][j][r];
        }
  }
}

template <int WM, int WN, int TM, int TN, int BK, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void gemm3(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M, int N, int K) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDK = BK + 4;
  constexpr int CPR = BK / 4;                 // float4 chunks per row
  constexpr int RPP = NT / CPR;               // rows staged per pass
  constexpr int PA = BM / RPP, PB = (BN + RPP - 1) / RPP;
  static_assert(BM % RPP == 0, "");
  constexpr int STAGE = (BM + BN) * LDK;
  constexpr int NS = 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, h = lane >> 5, l31 = lane & 31;
  const int lrow = tid / CPR, lcol = (tid % CPR) * 4;
  const int ntn = N / BN, ntiles = (M / BM) * ntn, G = gridDim.x;
  const int slot = blockIdx.x;
  if (slot >= ntiles) return;
  const int nk = K / BK;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
  int ld_tile = slot, ld_k = 0;
  unsigned a_off[PA], b_off[PB];
  floatx4 va[PA], vb[PB];
  auto setup = [&](int tile) {
    const int mt = tile / ntn, nt = tile - mt * ntn;
#pragma unroll
    for (int i = 0; i < PA; ++i) a_off[i] = (unsigned)(((mt * BM + lrow + RPP * i) * (long long)K + lcol) * 4);
#pragma unroll
    for (int i = 0; i < PB; ++i) b_off[i] = (unsigned)(((nt * BN + lrow + RPP * i) * K + lcol) * 4);
  };
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i)
      va[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off[i], ld_k * 4, 0));
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (lrow + RPP * i < BN) vb[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[i], ld_k * 4, 0));
  };
  auto advance = [&]() {
    ld_k += BK;
    if (ld_k == K) { ld_k = 0; ld_tile += G; if (ld_tile < ntiles) setup(ld_tile); }
  };
  auto store = [&](float* st) {
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<floatx4*>(&st[(lrow + RPP * i) * LDK + lcol]) = va[i];
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (lrow + RPP * i < BN) *reinterpret_cast<floatx4*>(&st[(BM + lrow + RPP * i) * LDK + lcol]) = vb[i];
  };
  setup(slot);
  issue(); advance(); store(smem);
  if (ld_tile < ntiles) { issue(); advance(); store(smem + STAGE); }
  if (ld_tile < ntiles) { issue(); advance(); }
  __syncthreads();
  const int a_frag = (wm * TM * 32 + l31) * LDK + h * 4, b_frag = (BM + wn * TN * 32 + l31) * LDK + h * 4;
  const int n_my = (ntiles - slot + G - 1) / G;
  int left = n_my * nk, u = 0;
  floatx16 acc[TM][TN];
  floatx4 af[2][TM], bf[2][TN];
  {
    const float* Ab = smem + a_frag;
    const float* Bb = smem + b_frag;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK);
  }
  auto ktile = [&](auto ST, auto LD) {
    const float* Ab = smem + u * STAGE + a_frag;
    const float* Bb = smem + u * STAGE + b_frag;
    const int u1 = u + 1 == NS ? 0 : u + 1, u2 = u1 + 1 == NS ? 0 : u1 + 1;
    float* nxt = smem + u2 * STAGE;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + (g + 1) * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + (g + 1) * 8);
      } else {   // next k-tile's first fragments: its stage was completed one barrier ago
        const float* An = smem + u1 * STAGE + a_frag;
        const float* Bn = smem + u1 * STAGE + b_frag;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(g + 1) & 1][i] = *reinterpret_cast<const floatx4*>(An + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(g + 1) & 1][j] = *reinterpret_cast<const floatx4*>(Bn + j * 32 * LDK);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (decltype(ST)::value && g == 0) store(nxt);
      if (decltype(LD)::value && g == (BK / 8 > 1 ? 1 : 0)) issue();
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j], 0, 0, 0);
      if (decltype(ST)::value && g == 0) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
      }
      if (decltype(LD)::value && g == (BK / 8 > 1 ? 1 : 0)) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int tile = slot; tile < ntiles; tile += G) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int nst = left - 3 < nk ? left - 3 : nk;
    if (nst < 0) nst = 0;
    for (int ks = 0; ks < nst; ++ks, u = (u + 1 == NS ? 0 : u + 1)) { ktile(std::true_type{}, std::true_type{}); advance(); __syncthreads(); }
    left -= nst;
    if (nst < nk) {
      if (left == 2) { ktile(std::true_type{}, std::false_type{}); __syncthreads(); u = (u + 1 == NS ? 0 : u + 1); --left; ++nst; }
      if (nst < nk) { ktile(std::false_type{}, std::false_type{}); __syncthreads(); u = (u + 1 == NS ? 0 : u + 1); --left; }
    }
    const int mt = tile / ntn, nt = tile - mt * ntn;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * BM + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int col = nt * BN + (wn * TN + j) * 32 + l31;
          C[(long long)row * N + col] = acc[i][j][r];
        }
  }
}

static float *dA, *dB, *dC;
template <int WM, int WN, int TM, int TN, int BK, int OCC>
static void run(const char* name, int M, int N, int K, int wgs_per_cu) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  auto kern = gemm<WM, WN, TM, TN, BK, OCC>;
  const size_t lds = sizeof(float) * 2 * (BM + BN) * (BK + 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * wgs_per_cu;
  const int ntiles = (M / BM) * (N / BN);
  const int Mu = (ntiles / grid) * grid / (N / BN) * BM;   // whole rounds only
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, 0, dA, dB, dC, Mu, N, K);
  (void)hipDeviceSynchronize();
  const int reps = 10;
  (void)hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, 0, dA, dB, dC, Mu, N, K);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  hipError_t err = hipGetLastError();
  printf("%-34s M=%7d N=%3d K=%4d  wg/CU=%d lds=%3zuK  %8.1f us  %6.1f TFLOP/s  %s\n", name, Mu, N, K, wgs_per_cu, lds / 1024,
         ms * 1e3, 2.0 * Mu * N * K / ms / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
}
template <int WM, int WN, int TM, int TN, int BK, int OCC>
static void run3(const char* name, int M, int N, int K, int wgs_per_cu) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  auto kern = gemm3<WM, WN, TM, TN, BK, OCC>;
  const size_t lds = sizeof(float) * 3 * (BM + BN) * (BK + 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * wgs_per_cu;
  const int ntiles = (M / BM) * (N / BN);
  const int Mu = (ntiles / grid) * grid / (N / BN) * BM;   // whole rounds only
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, 0, dA, dB, dC, Mu, N, K);
  (void)hipDeviceSynchronize();
  const int reps = 10;
  (void)hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, 0, dA, dB, dC, Mu, N, K);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  hipError_t err = hipGetLastError();
  printf("%-34s M=%7d N=%3d K=%4d  wg/CU=%d lds=%3zuK  %8.1f us  %6.1f TFLOP/s  %s\n", name, Mu, N, K, wgs_per_cu, lds / 1024,
         ms * 1e3, 2.0 * Mu * N * K / ms / 1e9, err == hipSuccess ? "" : hipGetErrorString(err));
}
int main() {
  const size_t Mmax = 1 << 20;
  (void)hipMalloc(&dA, Mmax * 1152 * 4); (void)hipMalloc(&dB, 512 * 1152 * 4); (void)hipMalloc(&dC, Mmax * 128 * 4);
  (void)hipMemset(dA, 0, Mmax * 1152 * 4); (void)hipMemset(dB, 0, 512 * 1152 * 4);
  for (int K : {576, 1152}) {
    const int M = 400000;
    run<4, 1, 2, 2, 32, 1>("4w 256x64  w64x64 bk32", M, 64, K, 1);
    run<4, 2, 2, 1, 32, 1>("8w 256x64  w64x32 bk32", M, 64, K, 1);
    run<4, 1, 1, 2, 32, 2>("4w 128x64  w32x64 bk32", M, 64, K, 2);
    run<4, 1, 1, 2, 16, 4>("4w 128x64  w32x64 bk16 x4", M, 64, K, 4);
    run3<4, 1, 1, 2, 16, 3>("3st 128x64 bk16 x3", M, 64, K, 3);
    run3<4, 1, 1, 2, 32, 1>("3st 128x64 bk32 x1", M, 64, K, 1);
    run3<2, 2, 2, 2, 16, 2>("3st 128x128 bk16 x2", M, 128, K, 2);
    run<4, 1, 1, 2, 16, 3>("4w 128x64  w32x64 bk16 x3", M, 64, K, 3);
    run<2, 2, 2, 2, 16, 2>("4w 128x128 w64x64 bk16 x2", M, 128, K, 2);
    run<4, 1, 2, 2, 16, 2>("4w 256x64  w64x64 bk16", M, 64, K, 2);
    run<4, 2, 2, 1, 16, 1>("8w 256x64  w64x32 bk16 x2", M, 64, K, 2);
    run<2, 2, 2, 2, 32, 2>("4w 128x128 w64x64 bk32", M, 128, K, 2);
    run<4, 2, 2, 2, 32, 1>("8w 256x128 w64x64 bk32", M, 128, K, 1);
    run<2, 2, 2, 2, 16, 2>("4w 128x128 w64x64 bk16 x3", M, 128, K, 3);
  }
  return 0;
}
