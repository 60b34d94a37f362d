// Implicit-GEMM convolution for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   forward : y[m][n]  = sum_{tap,c} x[pix(m,tap)][c] * w[n][tap][c]           (mode 0)
//   dgrad   : dx[m][c] = sum_{tap,n} dy[pixT(m,tap)][n] * wT[c][tap][n]        (mode 1, wT = repacked w)
//   wgrad   : dw[n][tap][c] = sum_m dy[m][n] * x[pix(m,tap)][c]                (split over m, tree-reduced)
//
// Activations are channels-last, so a GEMM-A row is 32 contiguous floats of one (possibly padded)
// input pixel: one 128-B line per 8 lanes, staged through LDS as [row][36] (32 + 4 pad) so the
// MFMA fragment reads are conflict-free ds_read_b128.  The MFMA k-index is free to permute as long
// as A and B agree, so lane-half h consumes k = 8g + 4h + s: one b128 read feeds four MFMAs.
//
// Reference ops replaced: nn.Conv3d/Conv2d/Linear fwd+bwd — models/video.py:20,
// models/network_blocks.py:18,20,35,37,40,42,49, models/audio.py:22, models/av_wrapper.py:25.
#include "common.h"

namespace avid {

struct ConvArgs {
  const float* __restrict__ src;     // [B,Ts,Hs,Ws,Cs] channels-last (VEC) or strided (scalar)
  const float* __restrict__ wk;      // [Cd][ntaps][Cs]
  const float* __restrict__ addend;  // [M][Cd] or null
  const float* __restrict__ bias;    // [Cd] or null
  float* __restrict__ dst;           // [M][Cd]
  int B, Ts, Hs, Ws, Cs;
  int Td, Hd, Wd, Cd;
  int kt, kh, kw;
  int st, sh, sw;
  int pt, ph, pw;
  int M;      // B*Td*Hd*Wd
  int mode;   // 0: s = d*stride - pad + tap ;  1: s = (d + pad - tap) / stride (exact)
  int relu;
  long long ssB, ssT, ssH, ssW, ssC;  // scalar-gather source strides (elements)
};

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row (floats): 144 B => b128 fragment reads conflict-free
constexpr int KTAB_MAX = 512;

__device__ __forceinline__ void decode_row(int m, int M, int Wd, int Hd, int Td, int& b, int& td, int& hd,
                                           int& wd, bool& ok) {
  ok = m < M;
  int mm = ok ? m : 0;
  wd = mm % Wd;
  int r = mm / Wd;
  hd = r % Hd;
  r = r / Hd;
  td = r % Td;
  b = r / Td;
}

template <int WM, int WN, int TM, int TN, bool VEC>
__global__ __launch_bounds__(256) void igemm_kernel(const ConvArgs p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int PA = BM / 32, PB = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;   // [2][BN][LDK]
  int2* ktab = reinterpret_cast<int2*>(smem + 2 * (BM + BN) * LDK);  // scalar mode only

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;

  // XCD-aware tile order: consecutive M-tiles (which share input halos) stay on one XCD's L2.
  const unsigned ntm = (p.M + BM - 1) / BM, ntn = p.Cd / BN;
  const unsigned tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  const int ntaps = p.kt * p.kh * p.kw;
  const int K = ntaps * p.Cs;
  const int nk = VEC ? ntaps * (p.Cs / BK) : (K + BK - 1) / BK;

  // ---- per-thread row bookkeeping for the A loader (rows are fixed for the whole K loop)
  int a_t0[PA], a_h0[PA], a_w0[PA];
  long long a_base[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int b, td, hd, wd;
    bool ok;
    decode_row(m0 + lrow + 32 * i, p.M, p.Wd, p.Hd, p.Td, b, td, hd, wd, ok);
    if (p.mode == 0) {
      a_t0[i] = td * p.st - p.pt;
      a_h0[i] = hd * p.sh - p.ph;
      a_w0[i] = wd * p.sw - p.pw;
    } else {
      a_t0[i] = td + p.pt;
      a_h0[i] = hd + p.ph;
      a_w0[i] = wd + p.pw;
    }
    if (!ok) a_t0[i] = -(1 << 28);
    if (VEC)
      a_base[i] = (long long)b * p.Ts * p.Hs * p.Ws;  // pixel index of (b,0,0,0)
    else
      a_base[i] = (long long)b * p.ssB + (long long)a_t0[i] * p.ssT + (long long)a_h0[i] * p.ssH +
                  (long long)a_w0[i] * p.ssW;
  }

  if (!VEC) {
    for (int k = tid; k < nk * BK; k += 256) {
      int2 e;
      if (k < K) {
        int tap = k / p.Cs, c = k - tap * p.Cs;
        int dw = tap % p.kw, r = tap / p.kw;
        int dh = r % p.kh, dt = r / p.kh;
        e.x = (int)(dt * p.ssT + dh * p.ssH + dw * p.ssW + c * p.ssC);
        e.y = dt | (dh << 8) | (dw << 16);
      } else {
        e.x = 0;
        e.y = -1;
      }
      ktab[k] = e;
    }
    __syncthreads();
  }

  floatx4 va[PA], vb[PB];

  auto load_tile = [&](int ks) {
    if (VEC) {
      const int cpt = p.Cs / BK;
      const int tap = ks / cpt, c0 = (ks - tap * cpt) * BK;
      const int dw = tap % p.kw, r = tap / p.kw;
      const int dh = r % p.kh, dt = r / p.kh;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        int ts, hs, ws;
        bool ok;
        if (p.mode == 0) {
          ts = a_t0[i] + dt;
          hs = a_h0[i] + dh;
          ws = a_w0[i] + dw;
          ok = true;
        } else {
          ts = a_t0[i] - dt;
          hs = a_h0[i] - dh;
          ws = a_w0[i] - dw;
          ok = (ts >= 0) & (hs >= 0) & (ws >= 0);
          if (p.st == 2) { ok &= !(ts & 1); ts >>= 1; }
          if (p.sh == 2) { ok &= !(hs & 1); hs >>= 1; }
          if (p.sw == 2) { ok &= !(ws & 1); ws >>= 1; }
        }
        ok &= ((unsigned)ts < (unsigned)p.Ts) & ((unsigned)hs < (unsigned)p.Hs) & ((unsigned)ws < (unsigned)p.Ws);
        long long pix = a_base[i] + ((long long)ts * p.Hs + hs) * p.Ws + ws;
        pix = ok ? pix : 0;  // clamp: the load below is unconditional (branch-free), never out of bounds
        const floatx4 v = *reinterpret_cast<const floatx4*>(p.src + pix * p.Cs + c0 + lcol);
        const floatx4 z = {0.f, 0.f, 0.f, 0.f};
        va[i] = ok ? v : z;
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int n = n0 + lrow + 32 * i;
        const floatx4* ptr =
            reinterpret_cast<const floatx4*>(p.wk + ((long long)n * ntaps + tap) * p.Cs + c0 + lcol);
        vb[i] = *ptr;  // Cd % BN == 0 is enforced on the host
      }
    } else {
      const int kb = ks * BK + lcol;
      int2 e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = ktab[kb + j];
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        floatx4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int dt = e[j].y & 0xff, dh = (e[j].y >> 8) & 0xff, dw = (e[j].y >> 16) & 0xff;
          const int ts = a_t0[i] + dt, hs = a_h0[i] + dh, ws = a_w0[i] + dw;
          const bool ok = (e[j].y >= 0) & ((unsigned)ts < (unsigned)p.Ts) & ((unsigned)hs < (unsigned)p.Hs) &
                          ((unsigned)ws < (unsigned)p.Ws);
          const float t = p.src[ok ? a_base[i] + e[j].x : 0];
          v[j] = ok ? t : 0.f;
        }
        va[i] = v;
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int n = n0 + lrow + 32 * i;
        floatx4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = p.wk[(kb + j < K) ? (long long)n * K + kb + j : 0];
          v[j] = (kb + j < K) ? t : 0.f;
        }
        vb[i] = v;
      }
    }
  };

  auto store_tile = [&](int buf) {
    float* Ab = As + buf * BM * LDK;
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<floatx4*>(&Ab[(lrow + 32 * i) * LDK + lcol]) = va[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<floatx4*>(&Bb[(lrow + 32 * i) * LDK + lcol]) = vb[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int h = lane >> 5, l31 = lane & 31;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) load_tile(ks + 1);  // global loads in flight under the MFMAs below

    const float* Ab = As + cur * BM * LDK + (wm * TM * 32 + l31) * LDK + h * 4;
    const float* Bb = Bs + cur * BN * LDK + (wn * TN * 32 + l31) * LDK + h * 4;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      floatx4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + g * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }

    if (ks + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.M) {
          const long long o = (long long)row * p.Cd + col;
          float v = acc[i][j][r] + bv;
          if (p.addend) v += p.addend[o];
          if (p.relu) v = fmaxf(v, 0.f);
          p.dst[o] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad: one workgroup owns a 64(n) x 64(c) tile of dw for ONE tap and one m-range (split).
// GEMM-K runs over m (pixels).  LDS tiles are [32 m][64 + 4]; fragments are ds_read_b32 (lanes of
// a half-wave read consecutive n / c of one m-row => conflict-free).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* __restrict__ src;  // x
  const float* __restrict__ dy;   // [M][Cd]
  float* __restrict__ out;        // [nsplit][Cd][K]   (K = ntaps*Cs)
  int B, Ts, Hs, Ws, Cs;
  int Td, Hd, Wd, Cd;
  int kt, kh, kw;
  int st, sh, sw;
  int pt, ph, pw;
  int M;
  int nsplit, chunks_per_split;  // chunks of 32 rows
  int kt_tiles;                  // number of 64-wide k tiles (VEC: ntaps * Cs/64; scalar: ceil(K/64))
  long long ssB, ssT, ssH, ssW, ssC;
};

constexpr int WG_LD = 64 + 4;

template <bool VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ds = smem;                    // [2][32][WG_LD]  dy tile  (m, n)
  float* Xs = smem + 2 * 32 * WG_LD;   // [2][32][WG_LD]  x tile   (m, k)
  int2* ktab = reinterpret_cast<int2*>(smem + 4 * 32 * WG_LD);  // scalar mode: 64 entries

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // wm: n half (rows of dw), wn: k half
  const int ntaps = p.kt * p.kh * p.kw;
  const int K = ntaps * p.Cs;

  // tile decode: blockIdx.x = ktile + kt_tiles * ntile ; blockIdx.y = split
  const int ktile = blockIdx.x % p.kt_tiles, ntile = blockIdx.x / p.kt_tiles;
  const int n0 = ntile * 64;
  int tap = 0, c0 = 0, dt = 0, dh = 0, dw = 0;
  if (VEC) {
    const int cpt = p.Cs / 64;
    tap = ktile / cpt;
    c0 = (ktile - tap * cpt) * 64;
    dw = tap % p.kw;
    int r = tap / p.kw;
    dh = r % p.kh;
    dt = r / p.kh;
  } else {
    if (tid < 64) {
      int k = ktile * 64 + tid;
      int2 e;
      if (k < K) {
        int tp = k / p.Cs, c = k - tp * p.Cs;
        int ew = tp % p.kw, r = tp / p.kw;
        int eh = r % p.kh, et = r / p.kh;
        e.x = (int)(et * p.ssT + eh * p.ssH + ew * p.ssW + c * p.ssC);
        e.y = et | (eh << 8) | (ew << 16);
      } else {
        e.x = 0;
        e.y = -1;
      }
      ktab[tid] = e;
    }
    __syncthreads();
  }

  const int chunk0 = blockIdx.y * p.chunks_per_split;
  const int total_chunks = (p.M + 31) / 32;
  int chunk1 = chunk0 + p.chunks_per_split;
  if (chunk1 > total_chunks) chunk1 = total_chunks;

  // loader mapping: 32 rows x 16 float4 per tile = 512 float4 -> 2 per thread
  const int lrow = tid >> 4;         // 0..15 (+16)
  const int lcol = (tid & 15) * 4;   // 0..60

  floatx4 vd[2], vx[2];

  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = ch * 32 + lrow + 16 * i;
      int b, td, hd, wd;
      bool ok;
      decode_row(m, p.M, p.Wd, p.Hd, p.Td, b, td, hd, wd, ok);
      floatx4 z = {0.f, 0.f, 0.f, 0.f};
      const floatx4 tdy = *reinterpret_cast<const floatx4*>(p.dy + (long long)(ok ? m : 0) * p.Cd + n0 + lcol);
      vd[i] = ok ? tdy : z;
      const int t0 = td * p.st - p.pt, h0 = hd * p.sh - p.ph, w0 = wd * p.sw - p.pw;
      if (VEC) {
        const int ts = t0 + dt, hs = h0 + dh, ws = w0 + dw;
        const bool okx = ok & ((unsigned)ts < (unsigned)p.Ts) & ((unsigned)hs < (unsigned)p.Hs) &
                         ((unsigned)ws < (unsigned)p.Ws);
        long long pix = (((long long)b * p.Ts + ts) * p.Hs + hs) * p.Ws + ws;
        pix = okx ? pix : 0;
        const floatx4 tx = *reinterpret_cast<const floatx4*>(p.src + pix * p.Cs + c0 + lcol);
        vx[i] = okx ? tx : z;
      } else {
        const long long base = (long long)b * p.ssB + (long long)t0 * p.ssT + (long long)h0 * p.ssH +
                               (long long)w0 * p.ssW;
        floatx4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int2 e = ktab[lcol + j];
          const int et = e.y & 0xff, eh = (e.y >> 8) & 0xff, ew = (e.y >> 16) & 0xff;
          const int ts = t0 + et, hs = h0 + eh, ws = w0 + ew;
          const bool okx = ok & (e.y >= 0) & ((unsigned)ts < (unsigned)p.Ts) &
                           ((unsigned)hs < (unsigned)p.Hs) & ((unsigned)ws < (unsigned)p.Ws);
          const float t = p.src[okx ? base + e.x : 0];
          v[j] = okx ? t : 0.f;
        }
        vx[i] = v;
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<floatx4*>(&Ds[buf * 32 * WG_LD + (lrow + 16 * i) * WG_LD + lcol]) = vd[i];
      *reinterpret_cast<floatx4*>(&Xs[buf * 32 * WG_LD + (lrow + 16 * i) * WG_LD + lcol]) = vx[i];
    }
  };

  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int h = lane >> 5, l31 = lane & 31;

  if (chunk0 < chunk1) {
    load_chunk(chunk0);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = chunk0; ch < chunk1; ++ch) {
    const int cur = (ch - chunk0) & 1;
    if (ch + 1 < chunk1) load_chunk(ch + 1);
    const float* Db = Ds + cur * 32 * WG_LD + wm * 32 + l31;
    const float* Xb = Xs + cur * 32 * WG_LD + wn * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float a = Db[(2 * kk + h) * WG_LD];  // A[i = n][k = m]
      const float b = Xb[(2 * kk + h) * WG_LD];  // B[k = m][j = c]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (ch + 1 < chunk1) store_chunk(cur ^ 1);
    __syncthreads();
  }

  // write the partial tile: out[split][n][k]
  float* o = p.out + (long long)blockIdx.y * p.Cd * K;
  const int kcol = (VEC ? tap * p.Cs + c0 : ktile * 64) + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (kcol < K) o[(long long)n * K + kcol] = acc[r];
  }
}

// sum partials over splits (fixed order => deterministic)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long long n, int nsplit) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += part[(long long)k * n + i];
  dw[i] = s;
}

// w[co][tap][ci] -> wT[ci][tap][co]
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int ntaps, int Ci) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)Co * ntaps * Ci;
  if (i >= n) return;
  int co = (int)(i % Co);
  long long r = i / Co;
  int tap = (int)(r % ntaps);
  int ci = (int)(r / ntaps);
  wt[i] = w[((long long)co * ntaps + tap) * Ci + ci];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int validate(const avid_conv_desc* d) {
  AVID_REQUIRE(d != nullptr, AVID_E_BADARG, "conv: null descriptor");
  AVID_REQUIRE(d->B > 0 && d->Ti > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0 && d->Cout > 0, AVID_E_SHAPE,
               "conv: non-positive dims");
  AVID_REQUIRE(d->kt > 0 && d->kh > 0 && d->kw > 0 && d->kt < 256 && d->kh < 256 && d->kw < 256, AVID_E_SHAPE,
               "conv: bad kernel extent");
  AVID_REQUIRE(d->st > 0 && d->sh > 0 && d->sw > 0, AVID_E_SHAPE, "conv: bad stride");
  const int To = (d->Ti + 2 * d->pt - d->kt) / d->st + 1;
  const int Ho = (d->Hi + 2 * d->ph - d->kh) / d->sh + 1;
  const int Wo = (d->Wi + 2 * d->pw - d->kw) / d->sw + 1;
  AVID_REQUIRE(To == d->To && Ho == d->Ho && Wo == d->Wo, AVID_E_SHAPE,
               "conv: output extent (%d,%d,%d) does not match (%d,%d,%d)", d->To, d->Ho, d->Wo, To, Ho, Wo);
  AVID_REQUIRE(d->Cout % 64 == 0, AVID_E_UNSUPPORTED, "conv: Cout=%d must be a multiple of 64", d->Cout);
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  if (!vec) {
    AVID_REQUIRE(d->kt * d->kh * d->kw * d->Cin <= KTAB_MAX - 32, AVID_E_UNSUPPORTED,
                 "conv: gather path supports K <= %d (got %d)", KTAB_MAX - 32, d->kt * d->kh * d->kw * d->Cin);
  }
  const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
  const long long Mi = (long long)d->B * d->Ti * d->Hi * d->Wi;
  AVID_REQUIRE(M < (1ll << 31) && Mi < (1ll << 31), AVID_E_UNSUPPORTED, "conv: more than 2^31 pixels");
  return AVID_OK;
}

static void fill_src_strides(const avid_conv_desc* d, long long& sB, long long& sT, long long& sH, long long& sW,
                             long long& sC) {
  if (d->x_channel_first) {
    sW = 1;
    sH = d->Wi;
    sT = (long long)d->Hi * d->Wi;
    sC = sT * d->Ti;
    sB = sC * d->Cin;
  } else {
    sC = 1;
    sW = d->Cin;
    sH = (long long)d->Wi * d->Cin;
    sT = sH * d->Hi;
    sB = sT * d->Ti;
  }
}

template <int WM, int WN, int TM, int TN, bool VEC>
static int launch_igemm(const ConvArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const size_t lds = sizeof(float) * 2 * (BM + BN) * LDK + (VEC ? 0 : sizeof(int2) * KTAB_MAX);
  static bool attr_set = false;
  auto kern = igemm_kernel<WM, WN, TM, TN, VEC>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const unsigned ntm = (a.M + BM - 1) / BM, ntn = a.Cd / BN;
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_kernel<%d,%d,%d,%d,%d>", WM, WN, TM, TN, VEC ? 1 : 0);
  const double K = (double)a.kt * a.kh * a.kw * a.Cs;
  // algorithmic work: 2*M*N*K flops; bytes = one read of src + weights, one write of dst (+ addend)
  const double srcpix = (double)a.B * a.Ts * a.Hs * a.Ws;
  ScopedTimer t(s, name, 2.0 * a.M * a.Cd * K,
                4.0 * (srcpix * a.Cs + (double)a.Cd * K + (double)a.M * a.Cd * (a.addend ? 2 : 1)));
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(256), lds, s, a);
  return check_launch("igemm");
}

// Tile choice: the biggest tile that still gives every CU >= 2 workgroups; small-M late layers fall
// back to 64x64 / 32x128 tiles (split-K for them is a later optimisation).
static int pick_tile(long long M, int Cd) {
  const long long want = 2 * 256;
  if (Cd % 128 == 0 && ((M + 127) / 128) * (Cd / 128) >= want) return 0;  // 128 x 128
  if (((M + 127) / 128) * (Cd / 64) >= want) return 1;                    // 128 x 64
  if (Cd % 128 == 0 && M <= 2048) return 2;                               // 32 x 128
  return 3;                                                               // 64 x 64
}
static const char* kTileName[4] = {"2,2,2,2", "4,1,1,2", "1,4,1,1", "2,2,1,1"};

template <bool VEC>
static int dispatch_igemm(const ConvArgs& a, hipStream_t s) {
  switch (pick_tile(a.M, a.Cd)) {
    case 0: return launch_igemm<2, 2, 2, 2, VEC>(a, s);
    case 1: return launch_igemm<4, 1, 1, 2, VEC>(a, s);
    case 2: return launch_igemm<1, 4, 1, 1, VEC>(a, s);
    default: return launch_igemm<2, 2, 1, 1, VEC>(a, s);
  }
}

}  // namespace avid

using namespace avid;

extern "C" int avid_conv_fwd(const avid_conv_desc* d, const float* x, const float* w, const float* addend,
                             const float* bias, int relu, float* y, avid_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(x && w && y, AVID_E_BADARG, "conv_fwd: null pointer");
  ConvArgs a;
  a.src = x; a.wk = w; a.addend = addend; a.bias = bias; a.dst = y;
  a.B = d->B; a.Ts = d->Ti; a.Hs = d->Hi; a.Ws = d->Wi; a.Cs = d->Cin;
  a.Td = d->To; a.Hd = d->Ho; a.Wd = d->Wo; a.Cd = d->Cout;
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw;
  a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.M = d->B * d->To * d->Ho * d->Wo;
  a.mode = 0;
  a.relu = relu;
  fill_src_strides(d, a.ssB, a.ssT, a.ssH, a.ssW, a.ssC);
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  return vec ? dispatch_igemm<true>(a, (hipStream_t)stream) : dispatch_igemm<false>(a, (hipStream_t)stream);
}

extern "C" size_t avid_conv_dgrad_workspace_bytes(const avid_conv_desc* d) {
  if (!d) return 0;
  return sizeof(float) * (size_t)d->Cout * d->kt * d->kh * d->kw * d->Cin;
}

extern "C" int avid_conv_dgrad(const avid_conv_desc* d, const float* dy, const float* w, const float* addend,
                               float* dx, void* ws, size_t ws_bytes, avid_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(dy && w && dx && ws, AVID_E_BADARG, "conv_dgrad: null pointer");
  AVID_REQUIRE(!d->x_channel_first && d->Cin % 64 == 0 && d->Cout % 32 == 0, AVID_E_UNSUPPORTED,
               "conv_dgrad: needs channels-last x, Cin %% 64 == 0 and Cout %% 32 == 0 (Cin=%d Cout=%d)", d->Cin,
               d->Cout);
  AVID_REQUIRE(d->st <= 2 && d->sh <= 2 && d->sw <= 2, AVID_E_UNSUPPORTED, "conv_dgrad: stride > 2");
  AVID_REQUIRE(ws_bytes >= avid_conv_dgrad_workspace_bytes(d), AVID_E_BADARG, "conv_dgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int ntaps = d->kt * d->kh * d->kw;
  float* wt = static_cast<float*>(ws);
  const long long nw = (long long)d->Cout * ntaps * d->Cin;
  {
    ScopedTimer t(s, "weight_transpose_kernel", 0.0, 8.0 * nw);
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, s, w, wt, d->Cout,
                       ntaps, d->Cin);
  }
  rc = check_launch("weight_transpose");
  if (rc) return rc;
  ConvArgs a;
  a.src = dy; a.wk = wt; a.addend = addend; a.bias = nullptr; a.dst = dx;
  a.B = d->B; a.Ts = d->To; a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Cout;
  a.Td = d->Ti; a.Hd = d->Hi; a.Wd = d->Wi; a.Cd = d->Cin;
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw;
  a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.M = d->B * d->Ti * d->Hi * d->Wi;
  a.mode = 1;
  a.relu = 0;
  a.ssB = a.ssT = a.ssH = a.ssW = a.ssC = 0;
  return dispatch_igemm<true>(a, s);
}

static void wgrad_plan(const avid_conv_desc* d, int& kt_tiles, int& nsplit, int& cps, bool& vec) {
  const int ntaps = d->kt * d->kh * d->kw;
  const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
  vec = (d->Cin % 64 == 0) && !d->x_channel_first;
  kt_tiles = vec ? ntaps * (d->Cin / 64) : (int)ceil_div((long long)ntaps * d->Cin, 64);
  const long long tiles = (long long)kt_tiles * (d->Cout / 64);
  const long long chunks = ceil_div(M, 32);
  long long want = ceil_div(4 * 256, tiles);           // ~4 workgroups per CU
  long long max_split = chunks / 8 > 0 ? chunks / 8 : 1;  // >= 8 chunks (256 rows) per split
  nsplit = (int)(want < 1 ? 1 : (want > max_split ? max_split : want));
  if (nsplit > 256) nsplit = 256;
  cps = (int)ceil_div(chunks, nsplit);
  nsplit = (int)ceil_div(chunks, cps);
}

extern "C" size_t avid_conv_wgrad_workspace_bytes(const avid_conv_desc* d) {
  if (!d) return 0;
  int kt_tiles, nsplit, cps;
  bool vec;
  wgrad_plan(d, kt_tiles, nsplit, cps, vec);
  return sizeof(float) * (size_t)nsplit * d->Cout * d->kt * d->kh * d->kw * d->Cin;
}

extern "C" int avid_conv_wgrad(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws,
                               size_t ws_bytes, avid_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(x && dy && dw, AVID_E_BADARG, "conv_wgrad: null pointer");
  int kt_tiles, nsplit, cps;
  bool vec;
  wgrad_plan(d, kt_tiles, nsplit, cps, vec);
  AVID_REQUIRE(nsplit == 1 || (ws && ws_bytes >= avid_conv_wgrad_workspace_bytes(d)), AVID_E_BADARG,
               "conv_wgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  WgradArgs a;
  a.src = x; a.dy = dy;
  a.out = nsplit == 1 ? dw : static_cast<float*>(ws);
  a.B = d->B; a.Ts = d->Ti; a.Hs = d->Hi; a.Ws = d->Wi; a.Cs = d->Cin;
  a.Td = d->To; a.Hd = d->Ho; a.Wd = d->Wo; a.Cd = d->Cout;
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw;
  a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.M = d->B * d->To * d->Ho * d->Wo;
  a.nsplit = nsplit; a.chunks_per_split = cps; a.kt_tiles = kt_tiles;
  fill_src_strides(d, a.ssB, a.ssT, a.ssH, a.ssW, a.ssC);
  const size_t lds = sizeof(float) * 4 * 32 * WG_LD + sizeof(int2) * 64;
  dim3 grid((unsigned)(kt_tiles * (d->Cout / 64)), (unsigned)nsplit);
  {
    const double K = (double)a.kt * a.kh * a.kw * a.Cs;
    const double srcpix = (double)a.B * a.Ts * a.Hs * a.Ws;
    ScopedTimer t(s, vec ? "wgrad_kernel<1>" : "wgrad_kernel<0>", 2.0 * a.M * a.Cd * K,
                  4.0 * (srcpix * a.Cs + (double)a.M * a.Cd + (double)a.Cd * K));
    if (vec)
      hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(256), lds, s, a);
    else
      hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), lds, s, a);
  }
  rc = check_launch("wgrad");
  if (rc) return rc;
  if (nsplit > 1) {
    const long long n = (long long)d->Cout * d->kt * d->kh * d->kw * d->Cin;
    ScopedTimer t(s, "wgrad_reduce_kernel", 0.0, 4.0 * n * (nsplit + 1));
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s,
                       static_cast<const float*>(ws), dw, n, nsplit);
    rc = check_launch("wgrad_reduce");
  }
  return rc;
}

extern "C" int avid_conv_kernel_name(const avid_conv_desc* d, int which, char* buf, int len) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(buf && len > 0 && which >= 0 && which <= 2, AVID_E_BADARG, "conv_kernel_name: bad argument");
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  if (which == 0) {
    const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
    snprintf(buf, len, "igemm_kernel<%s,%d>", kTileName[pick_tile(M, d->Cout)], vec ? 1 : 0);
  } else if (which == 1) {
    const long long M = (long long)d->B * d->Ti * d->Hi * d->Wi;
    snprintf(buf, len, "igemm_kernel<%s,1>", kTileName[pick_tile(M, d->Cin)]);
  } else {
    int kt_tiles, nsplit, cps;
    bool v;
    wgrad_plan(d, kt_tiles, nsplit, cps, v);
    snprintf(buf, len, "wgrad_kernel<%d>", v ? 1 : 0);
  }
  return AVID_OK;
}
