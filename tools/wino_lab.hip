// Dev lab (next-round groundwork, not part of the product): Winograd F(2x2, 3x3) forward of the conv2x spatial layer
// (NHWC fp32, 512 frames x 28 x 28 x 64 -> 64, pad 1, stride 1) as ONE fused kernel on the fp32 MFMA — input
// transform -> 16 [32 tiles x 64 cin] x [64 cin x 64 cout] products -> output transform, nothing but x, U and y
// touches memory.  2.25x fewer multiply-adds than the direct form (6.58 vs 14.8 G); the question is what the
// transforms (VALU work, which DESIGN §8c shows is paid 1:1 in matrix-pipe time) leave of that.
//   workgroup = 4 waves, 32 tiles (2x2 outputs each) x 64 cout; wave w owns the four transform points xi = 4w..4w+3
//   LDS: V[16][32 tiles][32 cin + 4]  (one 32-channel chunk at a time, 73.7 KB: two workgroups per CU)
//   A operand: V[xi][tile = lane & 31][16 * (lane >> 5) + s],  B operand: U[xi][cout][16 * (lane >> 5) + s] (global)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int H = 28, W = 28, C = 64, K = 64, TH = 14, TW = 14, TPF = TH * TW;   // tiles per frame
constexpr int TB = 32, CK = 32, VLD = CK + 4;
constexpr int V_FLOATS = 16 * TB * VLD;

__global__ __launch_bounds__(256, 2) void wino_fwd(const float* __restrict__ x, const float* __restrict__ U,
                                                   float* __restrict__ y, int frames, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long long)frames * H * W * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, 16 * K * C * 4, 0x00020000);
  const long long ntiles = (long long)frames * TPF;
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    floatx16 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
    for (int ck = 0; ck < C / CK; ++ck) {
      __syncthreads();                         // the previous chunk's / block's LDS reads are done
      // ---- input transform: thread = (4 channels = tid & 7, tile = tid >> 3): one tile x 4 channels per thread,
      // 16 b128 loads, the transform on float4, 16 b128 LDS writes
      {
        const int c4 = (tid & 7) * 4, tl = tid >> 3;
        const long long t = (long long)blk * TB + tl;
        const int f = (int)(t / TPF), tt = (int)(t - (long long)f * TPF);
        const int ti = tt / TW, tj = tt - ti * TW;
        floatx4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int yy = 2 * ti - 1 + a, xx = 2 * tj - 1 + b;
            const bool ok = t < ntiles && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned off = (unsigned)((((long long)f * H + yy) * W + xx) * C + ck * CK + c4) * 4u;
            d[a][b] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : 0x80000000u, 0, 0));
          }
        floatx4 w_[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          w_[0][b] = d[0][b] - d[2][b];
          w_[1][b] = d[1][b] + d[2][b];
          w_[2][b] = d[2][b] - d[1][b];
          w_[3][b] = d[1][b] - d[3][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float* dst = sm + ((a * 4) * TB + tl) * VLD + c4;
          *reinterpret_cast<floatx4*>(dst + 0 * TB * VLD) = w_[a][0] - w_[a][2];
          *reinterpret_cast<floatx4*>(dst + 1 * TB * VLD) = w_[a][1] + w_[a][2];
          *reinterpret_cast<floatx4*>(dst + 2 * TB * VLD) = w_[a][2] - w_[a][1];
          *reinterpret_cast<floatx4*>(dst + 3 * TB * VLD) = w_[a][1] - w_[a][3];
        }
      }
      __syncthreads();
      // ---- 16 products: wave owns xi = 4*wave + c
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int xi = wave * 4 + c;
        const float* Ap = sm + (xi * TB + l31) * VLD + 16 * h;
        floatx4 av[4], bv[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const floatx4*>(Ap + 4 * q);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            bv[j][q] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                rsU, (unsigned)(((xi * K + j * 32 + l31) * C + ck * CK + 16 * h + 4 * q) * 4), 0, 0));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][e], bv[j][q][e], acc[c][j], 0, 0, 0);
      }
    }
    // ---- output transform.  Columns in registers: T[r][0] = M0 + M1 + M2, T[r][1] = M1 - M2 - M3 (r = wave)
    floatx16 T0[2], T1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      T0[j] = acc[0][j] + acc[1][j] + acc[2][j];
      T1[j] = acc[1][j] - acc[2][j] - acc[3][j];
    }
    __syncthreads();                           // V is dead: the LDS becomes the exchange buffer T[r][q][j][reg][lane]
    float* ex = sm;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ex[(((wave * 2 + 0) * 2 + j) * 16 + r) * 64 + lane] = T0[j][r];
        ex[(((wave * 2 + 1) * 2 + j) * 16 + r) * 64 + lane] = T1[j][r];
      }
    __syncthreads();
    // rows across waves: wave -> output (p = wave >> 1, q = wave & 1): Y[0][q] = T[0]+T[1]+T[2], Y[1][q] = T[1]-T[2]-T[3]
    const int p = wave >> 1, q = wave & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        auto Tq = [&](int rr) { return ex[(((rr * 2 + q) * 2 + j) * 16 + r) * 64 + lane]; };
        const float v = p == 0 ? Tq(0) + Tq(1) + Tq(2) : Tq(1) - Tq(2) - Tq(3);
        const int tl = (r & 3) + 8 * (r >> 2) + 4 * h;             // MFMA C layout: row = tile
        const long long t = (long long)blk * TB + tl;
        if (t < ntiles) {
          const int f = (int)(t / TPF), tt = (int)(t - (long long)f * TPF);
          const int ti = tt / TW, tj = tt - ti * TW;
          y[(((long long)f * H + 2 * ti + p) * W + 2 * tj + q) * K + j * 32 + l31] = v;
        }
      }
  }
}

// naive direct reference for a few pixels
static float ref_pixel(const std::vector<float>& x, const std::vector<float>& w, int f, int yy, int xx, int k) {
  double s = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const int iy = yy - 1 + a, ix = xx - 1 + b;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      for (int c = 0; c < C; ++c) s += (double)x[(((size_t)f * H + iy) * W + ix) * C + c] * w[((k * 3 + a) * 3 + b) * C + c];
    }
  return (float)s;
}

int main() {
  const int frames = 512;
  const size_t nx = (size_t)frames * H * W * C, ny = (size_t)frames * H * W * K;
  std::vector<float> hx(nx), hw((size_t)K * 9 * C), hU((size_t)16 * K * C);
  unsigned st = 1234567u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) * (1.f / 16777216.f) - 0.5f) * 2.f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.1f;
  // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  layout U[xi][cout][cin]
  const float G[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < C; ++c) {
      float g[3][3], t[4][3];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = hw[((k * 3 + a) * 3 + b) * C + c];
      for (int i = 0; i < 4; ++i) for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
      for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
        hU[((size_t)(i * 4 + j) * K + k) * C + c] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
    }
  float *dx, *dU, *dy;
  (void)hipMalloc(&dx, nx * 4); (void)hipMalloc(&dU, hU.size() * 4); (void)hipMalloc(&dy, ny * 4);
  (void)hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dU, hU.data(), hU.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(dy, 0, ny * 4);
  const int nblocks = (int)(((long long)frames * TPF + TB - 1) / TB);
  const size_t lds = sizeof(float) * V_FLOATS;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wino_fwd, dim3(512), dim3(256), lds, 0, dx, dU, dy, frames, nblocks);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wino_fwd, dim3(512), dim3(256), lds, 0, dx, dU, dy, frames, nblocks);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double direct_fl = 2.0 * frames * H * W * (double)K * 9 * C;
  printf("wino_fwd: %.1f us per launch; direct-equivalent %.1f TFLOP/s (the direct kernel: ~278 us, 106 TF); MFMA flops issued %.1f TF\n",
         ms * 1e3, direct_fl / ms / 1e9, direct_fl / 2.25 / ms / 1e9);
  std::vector<float> hy(ny);
  (void)hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int s = 0; s < 400; ++s) {
    st = st * 1664525u + 1013904223u; const int f = (st >> 8) % frames;
    st = st * 1664525u + 1013904223u; const int yy = s < 60 ? (s % 2 ? 0 : H - 1) : (st >> 8) % H;
    st = st * 1664525u + 1013904223u; const int xx = s < 60 ? (s % 3 ? W - 1 : 0) : (st >> 8) % W;
    st = st * 1664525u + 1013904223u; const int k = (st >> 8) % K;
    const float r = ref_pixel(hx, hw, f, yy, xx, k), g = hy[(((size_t)f * H + yy) * W + xx) * K + k];
    worst = fmax(worst, fabs((double)r - g)); scale = fmax(scale, fabs((double)r));
  }
  printf("check vs fp64 direct convolution on 400 sampled outputs (incl. borders): max |err| %.3e of scale %.3f -> %.2e relative\n",
         worst, scale, worst / scale);
  return 0;
}
