// Lab kernel: fp32-accurate convolution GEMM on the bf16 matrix instruction.
//
// Every fp32 operand is split into three bf16 terms x = hi + mid + lo (round-to-nearest at each step: the three terms
// carry 24+ mantissa bits, i.e. x exactly up to its last bit) and the product is assembled from SIX bf16 MFMAs with fp32
// accumulation — hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; the dropped mid.lo / lo.mid / lo.lo terms are <= 2^-24 of
// the product.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so six of them per fp32 product
// are 16/6 = 2.67x the exact-fp32 instruction's rate.  This is NOT a bf16 mode: the bar is the fp64 comparison of the
// fp32 kernels at their tolerances.
//
//   Y[m][n] = sum_tap sum_c X[m + toff[tap]][c] * W[n][tap][c]      (implicit GEMM over `ntap` row-shifted views of X)
//
// Layer A: conv2x temporal (3,1,1): M = 64 x 8 x 28 x 28 = 401408, C = 64, taps at -784 / 0 / +784 rows, N = 64 (K = 192).
// Layer B: conv5x spatial as a plain GEMM: M = 1024, K = 4608 (one "tap" of 4608 channels), N = 512, split over K.
//
// Kernel: 4 waves, tile 128 x 64 (wave: 32 rows x 64 columns), k-tile 32.  The activation tile is loaded as fp32, split
// in registers (5.5 VALU ops per element) and staged in LDS as three bf16 planes (double-buffered: 2 x 30 KB, two
// workgroups per CU); the weights are split ONCE by a prepass into the matrix instruction's fragment order and read
// straight from L2 (one 16-byte load per lane per fragment, prefetched one k-tile ahead).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// x0, x1 -> packed (hi, mid, lo) pairs
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - bf_lo(h), r1 = x1 - bf_hi(h);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - bf_lo(m), s1 = r1 - bf_hi(m);
  l = cvt_pk_bf16(s0, s1);
}

// ---- weights: W[N][K] fp32 -> fragment order Bf[K/16][N/32][3 planes][64 lanes][8 bf16]
//      lane l of fragment (kk, j): column n = 32 j + (l & 31), k = 16 kk + 8 (l >> 5) + [0, 8)
__global__ void split_weights(const float* __restrict__ W, uintx4* __restrict__ Bf, int N, int K) {
  const int frag = blockIdx.x, lane = threadIdx.x;
  const int nj = N / 32, kk = frag / nj, j = frag % nj;
  const int n = 32 * j + (lane & 31), k0 = 16 * kk + 8 * (lane >> 5);
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2(W[(size_t)n * K + k0 + 2 * i], W[(size_t)n * K + k0 + 2 * i + 1], h[i], m[i], l[i]);
  uintx4* dst = Bf + ((size_t)frag * 3) * 64 + lane;
  dst[0] = uintx4{h[0], h[1], h[2], h[3]};
  dst[64] = uintx4{m[0], m[1], m[2], m[3]};
  dst[128] = uintx4{l[0], l[1], l[2], l[3]};
}

__global__ void split_activation(const float* __restrict__ X, unsigned short* __restrict__ P, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  unsigned h, m, l;
  split2(X[i], X[i + 1], h, m, l);
  *reinterpret_cast<unsigned*>(P + i) = h;
  *reinterpret_cast<unsigned*>(P + n + i) = m;
  *reinterpret_cast<unsigned*>(P + 2 * n + i) = l;
}

struct Args {
  const float* X;        // [Mx][ldx]
  const uintx4* Bf;      // split weights, fragment order
  const unsigned short* Xp;   // variant 5: X as three bf16 planes [3][Mx][ldx]
  float* Y;              // [ksplit][M][N]
  long long xbytes;      // size of X in bytes (range check of the buffer loads: rows outside read zero)
  int M, N, K, ldx, Ctap, ntap, ksplit;
  int toff[4];           // row offset of each tap
};

constexpr int BM = 128, BN = 64, BK = 32, ROWB = 80;        // LDS row: 32 bf16 = 64 bytes + 16 pad
constexpr int PLANE = BM * ROWB, STAGE = 3 * PLANE;

template <int TERMS, bool CORR>
__global__ __launch_bounds__(256, 2) void conv_bf16x3(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = a.N / BN, ntm = a.M / BM;
  const int ntiles = ntm * ntn * a.ksplit;
  const int kt_per = a.K / BK / a.ksplit;                    // k-tiles per work item
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)(a.xbytes > 0x7fffffff ? 0x7fffffff : a.xbytes), 0x00020000);
  // loader: thread t stages row t / 2, 16 consecutive channels (t & 1) * 16 ..
  const int lrow = tid >> 1, lcol = (tid & 1) * 16;
  const int st_off = lrow * ROWB + lcol * 2;
  // fragments: row 32 wave + (lane & 31), k group lane >> 5
  const int fr_off = (32 * wave + (lane & 31)) * ROWB + (lane >> 5) * 16;
  floatx4 xa[4];
  uintx4 bcur[2][2][3], bnxt[2][2][3];

  for (int item = blockIdx.x; item < ntiles; item += gridDim.x) {
    const int ks = item % a.ksplit, t2 = item / a.ksplit, nt = t2 % ntn, mt = t2 / ntn;
    const int kt0 = ks * kt_per;
    auto load_a = [&](int kt) {
      const int k = (kt0 + kt) * BK, tap = k / a.Ctap, c = k - tap * a.Ctap;
      const long long row = (long long)mt * BM + lrow + a.toff[tap];
      const long long off = (row * a.ldx + c + lcol) * 4;
      const unsigned o = (row < 0 || off + 64 > a.xbytes) ? 0xfffffff0u : (unsigned)off;   // outside: the range check returns zeros
#pragma unroll
      for (int i = 0; i < 4; ++i) xa[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, o, 16 * i, 0));
    };
    auto load_b = [&](int kt, uintx4 (&b)[2][2][3]) {
      const int kk0 = (kt0 + kt) * 2;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uintx4* src = a.Bf + ((size_t)((kk0 + s) * (a.N / 32) + nt * 2 + j) * 3) * 64 + lane;
#pragma unroll
          for (int p = 0; p < 3; ++p) b[s][j][p] = src[64 * p];
        }
    };
    auto store_a = [&](char* st) {
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        split2(xa[i][0], xa[i][1], h[2 * i], m[2 * i], l[2 * i]);
        split2(xa[i][2], xa[i][3], h[2 * i + 1], m[2 * i + 1], l[2 * i + 1]);
      }
      char* d = st + st_off;
      *reinterpret_cast<uintx4*>(d) = uintx4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<uintx4*>(d + 16) = uintx4{h[4], h[5], h[6], h[7]};
      *reinterpret_cast<uintx4*>(d + PLANE) = uintx4{m[0], m[1], m[2], m[3]};
      *reinterpret_cast<uintx4*>(d + PLANE + 16) = uintx4{m[4], m[5], m[6], m[7]};
      *reinterpret_cast<uintx4*>(d + 2 * PLANE) = uintx4{l[0], l[1], l[2], l[3]};
      *reinterpret_cast<uintx4*>(d + 2 * PLANE + 16) = uintx4{l[4], l[5], l[6], l[7]};
    };
    floatx16 acc[2], cor[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; cor[j][r] = 0.f; }
    __syncthreads();                                          // (the previous item's last reads of stage 0)
    load_a(0);
    load_b(0, bcur);
    store_a(smem);
    __syncthreads();
    int u = 0;
    for (int kt = 0; kt < kt_per; ++kt) {
      const bool more = kt + 1 < kt_per;
      if (more) { load_a(kt + 1); load_b(kt + 1, bnxt); }
      const char* st = smem + u * STAGE + fr_off;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + s * 32);
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(st + PLANE + s * 32);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + 2 * PLANE + s * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bcur[s][j][0]);
          const bf16x8 bm = __builtin_bit_cast(bf16x8, bcur[s][j][1]);
          const bf16x8 bl = __builtin_bit_cast(bf16x8, bcur[s][j][2]);
          floatx16& c = CORR ? cor[j] : acc[j];
          if (TERMS >= 6) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
          }
          if (TERMS >= 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
          }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
        }
      }
      if (more) {
        store_a(smem + (u ^ 1) * STAGE);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bcur[s][j][p] = bnxt[s][j][p];
      }
      __syncthreads();
      u ^= 1;
    }
    // epilogue: C layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* Y = a.Y + (size_t)ks * a.M * a.N;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * BM + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = nt * BN + 32 * j + (lane & 31);
        Y[(size_t)row * a.N + col] = CORR ? acc[j][r] + cor[j][r] : acc[j][r];
      }
  }
}


// ---- variant 2: the weights of the item's column tile for ALL of K sit in LDS for the life of the (persistent) workgroup
// (K = 192, 64 columns: 72 KB in fragment order), the activation tile is prefetched TWO k-tiles ahead in registers; one
// workgroup of 4 waves per CU (135 KB of LDS).  Tells whether streaming the weight fragments from L2 (variant 1: 295 KB
// per 128-row tile and workgroup) is what holds variant 1 at 68 us with a single product.
template <int TERMS, bool CORR>
__global__ __launch_bounds__(256, 1) void conv_bf16x3_bs(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntm = a.M / BM;
  const int nkk = a.K / 16;
  char* ldsB = smem + 2 * STAGE;
  // weights -> LDS (column tile 0: N == 64 here)
  {
    const uintx4* src = a.Bf;
    uintx4* dst = reinterpret_cast<uintx4*>(ldsB);
    for (int i = tid; i < nkk * 2 * 3 * 64; i += 256) dst[i] = src[i];
  }
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)(a.xbytes > 0x7fffffff ? 0x7fffffff : a.xbytes), 0x00020000);
  const int lrow = tid >> 1, lcol = (tid & 1) * 16;
  const int st_off = lrow * ROWB + lcol * 2;
  const int fr_off = (32 * wave + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const int nkt = a.K / BK;
  floatx4 xa[2][4];
  // a flat stream of (tile, k-tile) pairs: loads run two steps ahead of the products, across tile boundaries
  const int n_my = (ntm - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int steps = n_my * nkt;
  auto load_a = [&](int step, floatx4 (&x)[4]) {
    const int it = step / nkt, kt = step - it * nkt, mt = blockIdx.x + it * gridDim.x;
    const int k = kt * BK, tap = k / a.Ctap, c = k - tap * a.Ctap;
    const long long row = (long long)mt * BM + lrow + a.toff[tap];
    const long long off = (row * a.ldx + c + lcol) * 4;
    const unsigned o = (row < 0 || off + 64 > a.xbytes) ? 0xfffffff0u : (unsigned)off;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, o, 16 * i, 0));
  };
  auto store_a = [&](char* st, const floatx4 (&x)[4]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split2(x[i][0], x[i][1], h[2 * i], m[2 * i], l[2 * i]);
      split2(x[i][2], x[i][3], h[2 * i + 1], m[2 * i + 1], l[2 * i + 1]);
    }
    char* d = st + st_off;
    *reinterpret_cast<uintx4*>(d) = uintx4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uintx4*>(d + 16) = uintx4{h[4], h[5], h[6], h[7]};
    *reinterpret_cast<uintx4*>(d + PLANE) = uintx4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<uintx4*>(d + PLANE + 16) = uintx4{m[4], m[5], m[6], m[7]};
    *reinterpret_cast<uintx4*>(d + 2 * PLANE) = uintx4{l[0], l[1], l[2], l[3]};
    *reinterpret_cast<uintx4*>(d + 2 * PLANE + 16) = uintx4{l[4], l[5], l[6], l[7]};
  };
  if (steps == 0) return;
  load_a(0, xa[0]);
  if (steps > 1) load_a(1, xa[1]);
  store_a(smem, xa[0]);
  __syncthreads();
  floatx16 acc[2], cor[2];
  int u = 0;
  for (int step = 0; step < steps; ++step) {
    const int it = step / nkt, kt = step - it * nkt;
    if (kt == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; cor[j][r] = 0.f; }
    }
    // registers: xa[(step + 1) & 1] holds step + 1 (in flight), xa[step & 1] is free -> load step + 2 into it
    if (step + 2 < steps) {
      if (step & 1) load_a(step + 2, xa[1]); else load_a(step + 2, xa[0]);
    }
    const char* st = smem + u * STAGE + fr_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + s * 32);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(st + PLANE + s * 32);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + 2 * PLANE + s * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* bp = ldsB + (size_t)(((kt * 2 + s) * 2 + j) * 3) * 1024 + lane * 16;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp);
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bp + 1024);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + 2048);
        floatx16& c = CORR ? cor[j] : acc[j];
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        }
        if (TERMS >= 3) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
    if (step + 1 < steps) {
      if (step & 1) store_a(smem + (u ^ 1) * STAGE, xa[0]); else store_a(smem + (u ^ 1) * STAGE, xa[1]);
    }
    if (kt == nkt - 1) {
      const int mt = blockIdx.x + it * gridDim.x;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * BM + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = 32 * j + (lane & 31);
          a.Y[(size_t)row * a.N + col] = CORR ? acc[j][r] + cor[j][r] : acc[j][r];
        }
    }
    __syncthreads();
    u ^= 1;
  }
}


// ---- variant 3: variant 1 (two workgroups per CU, weight fragments from L2) with the activation loads running THREE
// k-tiles ahead of the products in a register ring, across tile boundaries: a k-tile of this kernel holds 24 matrix
// instructions per wave (768 cycles) — one k-tile of lead does not cover an HBM access any more.
template <int TERMS, bool CORR>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_pf(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntm = a.M / BM;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)(a.xbytes > 0x7fffffff ? 0x7fffffff : a.xbytes), 0x00020000);
  const int lrow = tid >> 1, lcol = (tid & 1) * 16;
  const int st_off = lrow * ROWB + lcol * 2;
  const int fr_off = (32 * wave + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const int nkt = a.K / BK;
  floatx4 xa[3][4];
  uintx4 bcur[2][2][3], bnxt[2][2][3];
  const int n_my = (ntm - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int steps = n_my * nkt;
  if (steps <= 0) return;
  auto load_a = [&](int step, floatx4 (&x)[4]) {
    const int it = step / nkt, kt = step - it * nkt, mt = blockIdx.x + it * gridDim.x;
    const int k = kt * BK, tap = k / a.Ctap, c = k - tap * a.Ctap;
    const long long row = (long long)mt * BM + lrow + a.toff[tap];
    const long long off = (row * a.ldx + c + lcol) * 4;
    const unsigned o = (row < 0 || off + 64 > a.xbytes) ? 0xfffffff0u : (unsigned)off;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, o, 16 * i, 0));
  };
  auto load_b = [&](int step, uintx4 (&b)[2][2][3]) {
    const int kt = step % nkt;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uintx4* src = a.Bf + ((size_t)((kt * 2 + s) * (a.N / 32) + j) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[s][j][p] = src[64 * p];
      }
  };
  auto store_a = [&](char* st, const floatx4 (&x)[4]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split2(x[i][0], x[i][1], h[2 * i], m[2 * i], l[2 * i]);
      split2(x[i][2], x[i][3], h[2 * i + 1], m[2 * i + 1], l[2 * i + 1]);
    }
    char* d = st + st_off;
    *reinterpret_cast<uintx4*>(d) = uintx4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uintx4*>(d + 16) = uintx4{h[4], h[5], h[6], h[7]};
    *reinterpret_cast<uintx4*>(d + PLANE) = uintx4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<uintx4*>(d + PLANE + 16) = uintx4{m[4], m[5], m[6], m[7]};
    *reinterpret_cast<uintx4*>(d + 2 * PLANE) = uintx4{l[0], l[1], l[2], l[3]};
    *reinterpret_cast<uintx4*>(d + 2 * PLANE + 16) = uintx4{l[4], l[5], l[6], l[7]};
  };
  floatx16 acc[2], cor[2];
  int u = 0;
  // ring: step s lives in xa[s % 3]; at the top of step s the loads of s + 1, s + 2 are in flight and s + 3 is issued
  load_a(0, xa[0]);
  if (steps > 1) load_a(1, xa[1]);
  if (steps > 2) load_a(2, xa[2]);
  load_b(0, bcur);
  store_a(smem, xa[0]);
  __syncthreads();
  auto body = [&](int step, auto R) {
    constexpr int r = decltype(R)::value;                      // step % 3
    const int it = step / nkt, kt = step - it * nkt;
    if (kt == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; cor[j][q] = 0.f; }
    }
    if (step + 3 < steps) load_a(step + 3, xa[r]);             // (xa[r] was stored to LDS at the end of step - 1)
    if (step + 1 < steps) load_b(step + 1, bnxt);
    const char* st = smem + u * STAGE + fr_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + s * 32);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(st + PLANE + s * 32);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + 2 * PLANE + s * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bcur[s][j][0]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bcur[s][j][1]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bcur[s][j][2]);
        floatx16& c = CORR ? cor[j] : acc[j];
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        }
        if (TERMS >= 3) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
    if (step + 1 < steps) {
      store_a(smem + (u ^ 1) * STAGE, xa[(r + 1) % 3]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) bcur[s][j][p] = bnxt[s][j][p];
    }
    if (kt == nkt - 1) {
      const int mt = blockIdx.x + it * gridDim.x;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = mt * BM + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          const int col = 32 * j + (lane & 31);
          a.Y[(size_t)row * a.N + col] = CORR ? acc[j][q] + cor[j][q] : acc[j][q];
        }
    }
    __syncthreads();
    u ^= 1;
  };
  int step = 0;
  for (; step + 2 < steps; step += 3) {
    body(step, std::integral_constant<int, 0>{});
    body(step + 1, std::integral_constant<int, 1>{});
    body(step + 2, std::integral_constant<int, 2>{});
  }
  if (step < steps) body(step, std::integral_constant<int, 0>{});
  if (step + 1 < steps) body(step + 1, std::integral_constant<int, 1>{});
}

// ---- variant 4 = variant 3 + (a) tiles in XCD-contiguous order: workgroup b runs on XCD b % 8 (hardware order); each XCD
// takes one contiguous eighth of the tiles and its 64 workgroups walk it 64 tiles at a time, so the rows a tile reads at
// +-784 (six tiles away) are in that XCD's L2 — in hardware order every tap came from HBM (302 MB fetched per launch for
// 103 MB of activations); (b) unpadded 64-byte LDS rows with the 16-byte chunk index XOR-ed with (row >> 2) & 3: fragment
// reads and loader writes are both conflict-free (the 80-byte rows of variants 1-3 conflict on the writes: 46 % of the
// LDS cycles).  Variant 3 itself: variant 1 with the activation loads running THREE
// k-tiles ahead of the products in a register ring, across tile boundaries: a k-tile of this kernel holds 24 matrix
// instructions per wave (768 cycles) — one k-tile of lead does not cover an HBM access any more.
template <int TERMS, bool CORR>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_x(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntm = a.M / BM;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)(a.xbytes > 0x7fffffff ? 0x7fffffff : a.xbytes), 0x00020000);
  const int lrow = tid >> 1, lcol = (tid & 1) * 16;
  constexpr int XROW = 64, XPLANE = BM * XROW, XSTAGE = 3 * XPLANE;
  const int sw_st = (lrow >> 2) & 3;
  const int st_off0 = lrow * XROW + ((((tid & 1) * 2) ^ sw_st) << 4), st_off1 = lrow * XROW + ((((tid & 1) * 2 + 1) ^ sw_st) << 4);
  const int frow = 32 * wave + (lane & 31), sw_fr = (frow >> 2) & 3;
  const int fr_off0 = frow * XROW + ((((lane >> 5)) ^ sw_fr) << 4), fr_off1 = frow * XROW + (((2 + (lane >> 5)) ^ sw_fr) << 4);
  const int nkt = a.K / BK;
  floatx4 xa[3][4];
  uintx4 bcur[2][2][3], bnxt[2][2][3];
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, per = gridDim.x >> 3;      // workgroups per XCD
  const int q = ntm / 8, rem = ntm % 8;
  const int t_begin = xcd * q + (xcd < rem ? xcd : rem), t_cnt = q + (xcd < rem ? 1 : 0);
  const int n_my = lb < t_cnt ? (t_cnt - lb + per - 1) / per : 0;
  const int steps = n_my * nkt;
  if (steps <= 0) return;
  auto load_a = [&](int step, floatx4 (&x)[4]) {
    const int it = step / nkt, kt = step - it * nkt, mt = t_begin + lb + it * per;
    const int k = kt * BK, tap = k / a.Ctap, c = k - tap * a.Ctap;
    const long long row = (long long)mt * BM + lrow + a.toff[tap];
    const long long off = (row * a.ldx + c + lcol) * 4;
    const unsigned o = (row < 0 || off + 64 > a.xbytes) ? 0xfffffff0u : (unsigned)off;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, o, 16 * i, 0));
  };
  auto load_b = [&](int step, uintx4 (&b)[2][2][3]) {
    const int kt = step % nkt;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uintx4* src = a.Bf + ((size_t)((kt * 2 + s) * (a.N / 32) + j) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[s][j][p] = src[64 * p];
      }
  };
  auto store_a = [&](char* st, const floatx4 (&x)[4]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split2(x[i][0], x[i][1], h[2 * i], m[2 * i], l[2 * i]);
      split2(x[i][2], x[i][3], h[2 * i + 1], m[2 * i + 1], l[2 * i + 1]);
    }
    char* d0 = st + st_off0;
    char* d1 = st + st_off1;
    *reinterpret_cast<uintx4*>(d0) = uintx4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uintx4*>(d1) = uintx4{h[4], h[5], h[6], h[7]};
    *reinterpret_cast<uintx4*>(d0 + XPLANE) = uintx4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<uintx4*>(d1 + XPLANE) = uintx4{m[4], m[5], m[6], m[7]};
    *reinterpret_cast<uintx4*>(d0 + 2 * XPLANE) = uintx4{l[0], l[1], l[2], l[3]};
    *reinterpret_cast<uintx4*>(d1 + 2 * XPLANE) = uintx4{l[4], l[5], l[6], l[7]};
  };
  floatx16 acc[2], cor[2];
  int u = 0;
  // ring: step s lives in xa[s % 3]; at the top of step s the loads of s + 1, s + 2 are in flight and s + 3 is issued
  load_a(0, xa[0]);
  if (steps > 1) load_a(1, xa[1]);
  if (steps > 2) load_a(2, xa[2]);
  load_b(0, bcur);
  store_a(smem, xa[0]);
  __syncthreads();
  auto body = [&](int step, auto R) {
    constexpr int r = decltype(R)::value;                      // step % 3
    const int it = step / nkt, kt = step - it * nkt;
    if (kt == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; cor[j][q] = 0.f; }
    }
    if (step + 3 < steps) load_a(step + 3, xa[r]);             // (xa[r] was stored to LDS at the end of step - 1)
    if (step + 1 < steps) load_b(step + 1, bnxt);
    const char* st = smem + u * XSTAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int fo = s ? fr_off1 : fr_off0;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + fo);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(st + XPLANE + fo);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + 2 * XPLANE + fo);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bcur[s][j][0]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bcur[s][j][1]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bcur[s][j][2]);
        floatx16& c = CORR ? cor[j] : acc[j];
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        }
        if (TERMS >= 3) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
    if (step + 1 < steps) {
      store_a(smem + (u ^ 1) * XSTAGE, xa[(r + 1) % 3]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) bcur[s][j][p] = bnxt[s][j][p];
    }
    if (kt == nkt - 1) {
      const int mt = t_begin + lb + it * per;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = mt * BM + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          const int col = 32 * j + (lane & 31);
          a.Y[(size_t)row * a.N + col] = CORR ? acc[j][q] + cor[j][q] : acc[j][q];
        }
    }
    __syncthreads();
    u ^= 1;
  };
  int step = 0;
  for (; step + 2 < steps; step += 3) {
    body(step, std::integral_constant<int, 0>{});
    body(step + 1, std::integral_constant<int, 1>{});
    body(step + 2, std::integral_constant<int, 2>{});
  }
  if (step < steps) body(step, std::integral_constant<int, 0>{});
  if (step + 1 < steps) body(step + 1, std::integral_constant<int, 1>{});
}


// ---- variant 6 = variant 4 with the k-tile body as ONE basic block (loads / stores past the end are clamped instead of
// branched around) and an explicit issue order: one matrix instruction, then four vector instructions of the NEXT k-tile's
// split, and so on (sched_group_barrier) — the split and the LDS traffic inside the matrix pipe's 32-cycle issues instead
// of behind them.  Variant 4 = variant 3 + (a) tiles in XCD-contiguous order: workgroup b runs on XCD b % 8 (hardware order); each XCD
// takes one contiguous eighth of the tiles and its 64 workgroups walk it 64 tiles at a time, so the rows a tile reads at
// +-784 (six tiles away) are in that XCD's L2 — in hardware order every tap came from HBM (302 MB fetched per launch for
// 103 MB of activations); (b) unpadded 64-byte LDS rows with the 16-byte chunk index XOR-ed with (row >> 2) & 3: fragment
// reads and loader writes are both conflict-free (the 80-byte rows of variants 1-3 conflict on the writes: 46 % of the
// LDS cycles).  Variant 3 itself: variant 1 with the activation loads running THREE
// k-tiles ahead of the products in a register ring, across tile boundaries: a k-tile of this kernel holds 24 matrix
// instructions per wave (768 cycles) — one k-tile of lead does not cover an HBM access any more.
template <int TERMS, bool CORR>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_il(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntm = a.M / BM;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)(a.xbytes > 0x7fffffff ? 0x7fffffff : a.xbytes), 0x00020000);
  const int lrow = tid >> 1, lcol = (tid & 1) * 16;
  constexpr int XROW = 64, XPLANE = BM * XROW, XSTAGE = 3 * XPLANE;
  const int sw_st = (lrow >> 2) & 3;
  const int st_off0 = lrow * XROW + ((((tid & 1) * 2) ^ sw_st) << 4), st_off1 = lrow * XROW + ((((tid & 1) * 2 + 1) ^ sw_st) << 4);
  const int frow = 32 * wave + (lane & 31), sw_fr = (frow >> 2) & 3;
  const int fr_off0 = frow * XROW + ((((lane >> 5)) ^ sw_fr) << 4), fr_off1 = frow * XROW + (((2 + (lane >> 5)) ^ sw_fr) << 4);
  const int nkt = a.K / BK;
  floatx4 xa[3][4];
  uintx4 bcur[2][2][3], bnxt[2][2][3];
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, per = gridDim.x >> 3;      // workgroups per XCD
  const int q = ntm / 8, rem = ntm % 8;
  const int t_begin = xcd * q + (xcd < rem ? xcd : rem), t_cnt = q + (xcd < rem ? 1 : 0);
  const int n_my = lb < t_cnt ? (t_cnt - lb + per - 1) / per : 0;
  const int steps = n_my * nkt;
  if (steps <= 0) return;
  auto load_a = [&](int step, floatx4 (&x)[4]) {
    const int it = step / nkt, kt = step - it * nkt, mt = t_begin + lb + it * per;
    const int k = kt * BK, tap = k / a.Ctap, c = k - tap * a.Ctap;
    const long long row = (long long)mt * BM + lrow + a.toff[tap];
    const long long off = (row * a.ldx + c + lcol) * 4;
    const unsigned o = (row < 0 || off + 64 > a.xbytes) ? 0xfffffff0u : (unsigned)off;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, o, 16 * i, 0));
  };
  auto load_b = [&](int step, uintx4 (&b)[2][2][3]) {
    const int kt = step % nkt;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uintx4* src = a.Bf + ((size_t)((kt * 2 + s) * (a.N / 32) + j) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[s][j][p] = src[64 * p];
      }
  };
  auto store_a = [&](char* st, const floatx4 (&x)[4]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      split2(x[i][0], x[i][1], h[2 * i], m[2 * i], l[2 * i]);
      split2(x[i][2], x[i][3], h[2 * i + 1], m[2 * i + 1], l[2 * i + 1]);
    }
    char* d0 = st + st_off0;
    char* d1 = st + st_off1;
    *reinterpret_cast<uintx4*>(d0) = uintx4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uintx4*>(d1) = uintx4{h[4], h[5], h[6], h[7]};
    *reinterpret_cast<uintx4*>(d0 + XPLANE) = uintx4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<uintx4*>(d1 + XPLANE) = uintx4{m[4], m[5], m[6], m[7]};
    *reinterpret_cast<uintx4*>(d0 + 2 * XPLANE) = uintx4{l[0], l[1], l[2], l[3]};
    *reinterpret_cast<uintx4*>(d1 + 2 * XPLANE) = uintx4{l[4], l[5], l[6], l[7]};
  };
  floatx16 acc[2], cor[2];
  int u = 0;
  // ring: step s lives in xa[s % 3]; at the top of step s the loads of s + 1, s + 2 are in flight and s + 3 is issued
  load_a(0, xa[0]);
  if (steps > 1) load_a(1, xa[1]);
  if (steps > 2) load_a(2, xa[2]);
  load_b(0, bcur);
  store_a(smem, xa[0]);
  __syncthreads();
  auto body = [&](int step, auto R) {
    constexpr int r = decltype(R)::value;                      // step % 3
    const int it = step / nkt, kt = step - it * nkt;
    if (kt == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; cor[j][q] = 0.f; }
    }
    load_a(step + 3 < steps ? step + 3 : steps - 1, xa[r]);      // (clamped: a harmless re-load at the very end)
    load_b(step + 1, bnxt);
    const char* st = smem + u * XSTAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int fo = s ? fr_off1 : fr_off0;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + fo);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(st + XPLANE + fo);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + 2 * XPLANE + fo);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bcur[s][j][0]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bcur[s][j][1]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bcur[s][j][2]);
        floatx16& c = CORR ? cor[j] : acc[j];
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        }
        if (TERMS >= 3) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
    store_a(smem + (u ^ 1) * XSTAGE, xa[(r + 1) % 3]);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bcur[s][j][p] = bnxt[s][j][p];
    // issue order of the block above: the fragment reads first, then 1 MFMA : 4 VALU, the LDS writes as they become ready
    __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int g = 0; g < 4 * TERMS; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, TERMS >= 6 ? 4 : (TERMS >= 3 ? 8 : 24), 0);
      if (g % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, TERMS >= 6 ? 1 : (TERMS >= 3 ? 2 : 6), 0);
    }
    if (kt == nkt - 1) {
      const int mt = t_begin + lb + it * per;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = mt * BM + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          const int col = 32 * j + (lane & 31);
          a.Y[(size_t)row * a.N + col] = CORR ? acc[j][q] + cor[j][q] : acc[j][q];
        }
    }
    __syncthreads();
    u ^= 1;
  };
  int step = 0;
  for (; step + 2 < steps; step += 3) {
    body(step, std::integral_constant<int, 0>{});
    body(step + 1, std::integral_constant<int, 1>{});
    body(step + 2, std::integral_constant<int, 2>{});
  }
  if (step < steps) body(step, std::integral_constant<int, 0>{});
  if (step + 1 < steps) body(step + 1, std::integral_constant<int, 1>{});
}



// ---- variant 5 = variant 4 with the activation ALREADY split by its producer: X arrives as three bf16 planes
// [3][Mx][ldx] (6 bytes per element instead of 4: what a BatchNorm-apply epilogue would write), the loader moves 16-byte
// chunks global -> registers -> LDS and does no arithmetic.  Variant 4 = variant 3 + (a) tiles in XCD-contiguous order: workgroup b runs on XCD b % 8 (hardware order); each XCD
// takes one contiguous eighth of the tiles and its 64 workgroups walk it 64 tiles at a time, so the rows a tile reads at
// +-784 (six tiles away) are in that XCD's L2 — in hardware order every tap came from HBM (302 MB fetched per launch for
// 103 MB of activations); (b) unpadded 64-byte LDS rows with the 16-byte chunk index XOR-ed with (row >> 2) & 3: fragment
// reads and loader writes are both conflict-free (the 80-byte rows of variants 1-3 conflict on the writes: 46 % of the
// LDS cycles).  Variant 3 itself: variant 1 with the activation loads running THREE
// k-tiles ahead of the products in a register ring, across tile boundaries: a k-tile of this kernel holds 24 matrix
// instructions per wave (768 cycles) — one k-tile of lead does not cover an HBM access any more.
template <int TERMS, bool CORR>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_ps(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntm = a.M / BM;
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)a.Xp, 0, (int)(a.xbytes / 2 * 3 > 0x7fffffff ? 0x7fffffff : a.xbytes / 2 * 3), 0x00020000);
  const int lrow = tid >> 1, lcol = (tid & 1) * 16;
  constexpr int XROW = 64, XPLANE = BM * XROW, XSTAGE = 3 * XPLANE;
  const int sw_st = (lrow >> 2) & 3;
  const int st_off0 = lrow * XROW + ((((tid & 1) * 2) ^ sw_st) << 4), st_off1 = lrow * XROW + ((((tid & 1) * 2 + 1) ^ sw_st) << 4);
  const int frow = 32 * wave + (lane & 31), sw_fr = (frow >> 2) & 3;
  const int fr_off0 = frow * XROW + ((((lane >> 5)) ^ sw_fr) << 4), fr_off1 = frow * XROW + (((2 + (lane >> 5)) ^ sw_fr) << 4);
  const int nkt = a.K / BK;
  uintx4 xa[3][6];
  uintx4 bcur[2][2][3], bnxt[2][2][3];
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, per = gridDim.x >> 3;      // workgroups per XCD
  const int q = ntm / 8, rem = ntm % 8;
  const int t_begin = xcd * q + (xcd < rem ? xcd : rem), t_cnt = q + (xcd < rem ? 1 : 0);
  const int n_my = lb < t_cnt ? (t_cnt - lb + per - 1) / per : 0;
  const int steps = n_my * nkt;
  if (steps <= 0) return;
  auto load_a = [&](int step, uintx4 (&x)[6]) {
    const int it = step / nkt, kt = step - it * nkt, mt = t_begin + lb + it * per;
    const int k = kt * BK, tap = k / a.Ctap, c = k - tap * a.Ctap;
    const long long row = (long long)mt * BM + lrow + a.toff[tap];
    const long long off = (row * a.ldx + c + lcol) * 2;                 // bf16 planes
    const bool out = row < 0 || off + 32 > a.xbytes / 2;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned o = out ? 0xfffffff0u : (unsigned)(off + (long long)p * (a.xbytes / 2));
      x[2 * p] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rsP, o, 0, 0));
      x[2 * p + 1] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rsP, o, 16, 0));
    }
  };
  auto load_b = [&](int step, uintx4 (&b)[2][2][3]) {
    const int kt = step % nkt;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uintx4* src = a.Bf + ((size_t)((kt * 2 + s) * (a.N / 32) + j) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) b[s][j][p] = src[64 * p];
      }
  };
  auto store_a = [&](char* st, const uintx4 (&x)[6]) {
    char* d0 = st + st_off0;
    char* d1 = st + st_off1;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      *reinterpret_cast<uintx4*>(d0 + p * XPLANE) = x[2 * p];
      *reinterpret_cast<uintx4*>(d1 + p * XPLANE) = x[2 * p + 1];
    }
  };
  floatx16 acc[2], cor[2];
  int u = 0;
  // ring: step s lives in xa[s % 3]; at the top of step s the loads of s + 1, s + 2 are in flight and s + 3 is issued
  load_a(0, xa[0]);
  if (steps > 1) load_a(1, xa[1]);
  if (steps > 2) load_a(2, xa[2]);
  load_b(0, bcur);
  store_a(smem, xa[0]);
  __syncthreads();
  auto body = [&](int step, auto R) {
    constexpr int r = decltype(R)::value;                      // step % 3
    const int it = step / nkt, kt = step - it * nkt;
    if (kt == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[j][q] = 0.f; cor[j][q] = 0.f; }
    }
    if (step + 3 < steps) load_a(step + 3, xa[r]);             // (xa[r] was stored to LDS at the end of step - 1)
    if (step + 1 < steps) load_b(step + 1, bnxt);
    const char* st = smem + u * XSTAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int fo = s ? fr_off1 : fr_off0;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + fo);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(st + XPLANE + fo);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + 2 * XPLANE + fo);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bcur[s][j][0]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, bcur[s][j][1]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bcur[s][j][2]);
        floatx16& c = CORR ? cor[j] : acc[j];
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        }
        if (TERMS >= 3) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
    if (step + 1 < steps) {
      store_a(smem + (u ^ 1) * XSTAGE, xa[(r + 1) % 3]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) bcur[s][j][p] = bnxt[s][j][p];
    }
    if (kt == nkt - 1) {
      const int mt = t_begin + lb + it * per;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = mt * BM + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          const int col = 32 * j + (lane & 31);
          a.Y[(size_t)row * a.N + col] = CORR ? acc[j][q] + cor[j][q] : acc[j][q];
        }
    }
    __syncthreads();
    u ^= 1;
  };
  int step = 0;
  for (; step + 2 < steps; step += 3) {
    body(step, std::integral_constant<int, 0>{});
    body(step + 1, std::integral_constant<int, 1>{});
    body(step + 2, std::integral_constant<int, 2>{});
  }
  if (step < steps) body(step, std::integral_constant<int, 0>{});
  if (step + 1 < steps) body(step + 1, std::integral_constant<int, 1>{});
}



__global__ void reduce_slabs(const float* __restrict__ P, float* __restrict__ Y, long long n, int ks) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 >= n) return;
  floatx4 s = *reinterpret_cast<const floatx4*>(P + i * 4);
  for (int k = 1; k < ks; ++k) s += *reinterpret_cast<const floatx4*>(P + (size_t)k * n + i * 4);
  *reinterpret_cast<floatx4*>(Y + i * 4) = s;
}

// the fp32 result the product kernels compute (a fused-multiply-add chain in k order) for sampled rows
__global__ void ref_fp32(Args a, const float* __restrict__ W, const int* __restrict__ rows, int nrows, float* __restrict__ out, double* __restrict__ out64) {
  const int r = blockIdx.x, n = threadIdx.x + blockIdx.y * blockDim.x;
  if (r >= nrows || n >= a.N) return;
  const long long m = rows[r];
  float s = 0.f;
  double d = 0.0;
  for (int k = 0; k < a.K; ++k) {
    const int tap = k / a.Ctap, c = k - tap * a.Ctap;
    const long long row = m + a.toff[tap];
    const long long off = row * a.ldx + c;
    const float x = (row < 0 || off * 4 + 4 > a.xbytes) ? 0.f : a.X[off];
    const float w = W[(size_t)n * a.K + k];
    s = fmaf(x, w, s);
    d += (double)x * (double)w;
  }
  out[(size_t)r * a.N + n] = s;
  out64[(size_t)r * a.N + n] = d;
}

static float frand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffffff) / 8388608.0f - 1.0f;
}

template <int TERMS, bool CORR, int VARIANT = 1>
static void run(const char* what, Args a, const float* dW, int grid, int reps, const int* drows, int nrows, const std::vector<float>& ref32,
                const std::vector<double>& ref64, double flops, float* dYfinal) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t lds = VARIANT == 1 ? 2 * STAGE : 2 * STAGE + (size_t)(a.K / 16) * 2 * 3 * 1024;
  if (VARIANT == 1) CK(hipFuncSetAttribute((const void*)conv_bf16x3<TERMS, CORR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  else if (VARIANT == 2) CK(hipFuncSetAttribute((const void*)conv_bf16x3_bs<TERMS, CORR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  else if (VARIANT == 3) CK(hipFuncSetAttribute((const void*)conv_bf16x3_pf<TERMS, CORR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * STAGE)));
  else if (VARIANT == 4) CK(hipFuncSetAttribute((const void*)conv_bf16x3_x<TERMS, CORR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * STAGE)));
  else if (VARIANT == 6) CK(hipFuncSetAttribute((const void*)conv_bf16x3_il<TERMS, CORR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * STAGE)));
  else CK(hipFuncSetAttribute((const void*)conv_bf16x3_ps<TERMS, CORR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * STAGE)));
  const long long n = (long long)a.M * a.N;
  auto once = [&]() {
    if (VARIANT == 1) hipLaunchKernelGGL((conv_bf16x3<TERMS, CORR>), dim3(grid), dim3(256), lds, 0, a);
    else if (VARIANT == 2) hipLaunchKernelGGL((conv_bf16x3_bs<TERMS, CORR>), dim3(grid / 2), dim3(256), lds, 0, a);
    else if (VARIANT == 3) hipLaunchKernelGGL((conv_bf16x3_pf<TERMS, CORR>), dim3(grid), dim3(256), 2 * STAGE, 0, a);
    else if (VARIANT == 4) hipLaunchKernelGGL((conv_bf16x3_x<TERMS, CORR>), dim3(grid), dim3(256), 2 * STAGE, 0, a);
    else if (VARIANT == 6) hipLaunchKernelGGL((conv_bf16x3_il<TERMS, CORR>), dim3(grid), dim3(256), 2 * STAGE, 0, a);
    else hipLaunchKernelGGL((conv_bf16x3_ps<TERMS, CORR>), dim3(grid), dim3(256), 2 * STAGE, 0, a);
    if (a.ksplit > 1) hipLaunchKernelGGL(reduce_slabs, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, a.Y, dYfinal, n, a.ksplit);
  };
  for (int i = 0; i < 3; ++i) once();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) once();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  // accuracy on the sampled rows
  std::vector<int> rows(nrows);
  CK(hipMemcpy(rows.data(), drows, nrows * sizeof(int), hipMemcpyDeviceToHost));
  const float* Ysrc = a.ksplit > 1 ? dYfinal : a.Y;
  double e_max = 0, e32_max = 0, e_rms = 0, e32_rms = 0, scale = 0;
  std::vector<float> yrow(a.N);
  for (int r = 0; r < nrows; ++r) {
    CK(hipMemcpy(yrow.data(), Ysrc + (size_t)rows[r] * a.N, a.N * sizeof(float), hipMemcpyDeviceToHost));
    for (int j = 0; j < a.N; ++j) {
      const double t = ref64[(size_t)r * a.N + j];
      const double e = fabs(yrow[j] - t), e32 = fabs(ref32[(size_t)r * a.N + j] - t);
      e_max = fmax(e_max, e); e32_max = fmax(e32_max, e32);
      e_rms += e * e; e32_rms += e32 * e32; scale += t * t;
    }
  }
  const double cnt = (double)nrows * a.N;
  scale = sqrt(scale / cnt);
  printf("%-34s %8.1f us  %7.1f TFLOP/s (fp32-equivalent)   err vs fp64 / rms(out): max %.3e rms %.3e   [fp32 fma chain: max %.3e rms %.3e]\n",
         what, us, flops / us * 1e-6, e_max / scale, sqrt(e_rms / cnt) / scale, e32_max / scale, sqrt(e32_rms / cnt) / scale);
}

static void layer(const char* name, int M, int N, int K, int ldx, int Ctap, int ntap, const int* toff, int ksplit, long long Mx) {
  printf("== %s: M %d  N %d  K %d  (taps %d x %d channels)  split-K %d\n", name, M, N, K, ntap, Ctap, ksplit);
  unsigned seed = 12345u;
  std::vector<float> hX((size_t)Mx * ldx), hW((size_t)N * K);
  for (auto& v : hX) v = frand(seed) * 2.0f;                          // activations: after BatchNorm + ReLU-ish spread
  for (size_t i = 0; i < hX.size(); i += 3) hX[i] = fmaxf(hX[i], 0.f);
  const float ws = sqrtf(6.0f / K);
  for (auto& v : hW) v = frand(seed) * ws;
  float *dX, *dW, *dY, *dYf, *dref;
  double* dref64;
  uintx4* dBf;
  CK(hipMalloc(&dX, hX.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&dY, (size_t)ksplit * M * N * 4));
  CK(hipMalloc(&dYf, (size_t)M * N * 4));
  CK(hipMalloc(&dBf, (size_t)(K / 16) * (N / 32) * 3 * 64 * 16));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split_weights, dim3((K / 16) * (N / 32)), dim3(64), 0, 0, dW, dBf, N, K);
  CK(hipDeviceSynchronize());
  unsigned short* dXp;
  CK(hipMalloc(&dXp, hX.size() * 6));
  hipLaunchKernelGGL(split_activation, dim3((unsigned)((hX.size() / 2 + 255) / 256)), dim3(256), 0, 0, dX, dXp, (long long)hX.size());
  CK(hipDeviceSynchronize());
  Args a;
  a.X = dX; a.Bf = dBf; a.Y = dY; a.Xp = dXp; a.xbytes = (long long)hX.size() * 4;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.Ctap = Ctap; a.ntap = ntap; a.ksplit = ksplit;
  for (int i = 0; i < 4; ++i) a.toff[i] = i < ntap ? toff[i] : 0;
  const int nrows = 96;
  std::vector<int> rows(nrows);
  for (int i = 0; i < nrows; ++i) rows[i] = (int)(((long long)i * 7919 * 131) % M);
  rows[0] = 0; rows[1] = M - 1;
  int* drows;
  CK(hipMalloc(&drows, nrows * 4));
  CK(hipMemcpy(drows, rows.data(), nrows * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dref, (size_t)nrows * N * 4));
  CK(hipMalloc(&dref64, (size_t)nrows * N * 8));
  hipLaunchKernelGGL(ref_fp32, dim3(nrows, (N + 63) / 64), dim3(64), 0, 0, a, dW, drows, nrows, dref, dref64);
  CK(hipDeviceSynchronize());
  std::vector<float> ref32((size_t)nrows * N);
  std::vector<double> ref64((size_t)nrows * N);
  CK(hipMemcpy(ref32.data(), dref, ref32.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ref64.data(), dref64, ref64.size() * 8, hipMemcpyDeviceToHost));
  const double flops = 2.0 * M * N * (double)K;
  int dev_cus = 256;
  const int grid = 2 * dev_cus;
  const int reps = 20;
  run<6, false>("bf16x3 (6 products)", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
  run<6, true>("bf16x3 (6, corrections apart)", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
  run<3, false>("bf16x2-ish (3 products)", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
  run<1, false>("bf16 (hi.hi only)", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
  if (ksplit == 1 && N == 64 && K <= 256) {
    run<6, true, 2>("v2 weights in LDS: 6, corr. apart", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<1, false, 2>("v2 weights in LDS: hi.hi only", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<6, true, 3>("v3 loads 3 k-tiles ahead: 6, apart", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<6, false, 3>("v3 loads 3 k-tiles ahead: 6", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<1, false, 3>("v3 loads 3 k-tiles ahead: hi.hi", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<6, false, 4>("v4 XCD order + swizzle: 6", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<3, false, 4>("v4 XCD order + swizzle: 3", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<1, false, 4>("v4 XCD order + swizzle: hi.hi", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<6, false, 6>("v6 interleaved issue order: 6", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<1, false, 6>("v6 interleaved issue order: hi.hi", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<6, false, 5>("v5 producer-split operand: 6", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<6, true, 5>("v5 producer-split: 6, corr. apart", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
    run<1, false, 5>("v5 producer-split operand: hi.hi", a, dW, grid, reps, drows, nrows, ref32, ref64, flops, dYf);
  }
  CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dY)); CK(hipFree(dYf)); CK(hipFree(dBf)); CK(hipFree(drows)); CK(hipFree(dref)); CK(hipFree(dref64));
}

int main() {
  {   // conv2x temporal: 64 clips x 8 frames x 28 x 28, 64 -> 64 channels, taps one frame (784 rows) apart
    const int toff[3] = {-784, 0, 784};
    layer("conv2x temporal (3,1,1)", 401408, 64, 192, 64, 64, 3, toff, 1, 401408);
  }
  {   // conv5x spatial as a GEMM over its im2col matrix: M = 64 x 1 x 4 x 4, K = 9 x 512, N = 512
    const int toff[1] = {0};
    layer("conv5x spatial (GEMM form)", 1024, 512, 4608, 4608, 4608, 1, toff, 8, 1024);
  }
  return 0;
}
