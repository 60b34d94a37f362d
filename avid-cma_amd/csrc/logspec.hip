// Log-spectrogram front end on the GPU: the reference computes it per clip on CPU
// workers with librosa (datasets/preprocessing.py:158-186); here a whole batch of mono clips goes
//     frames (reflect-centred, Hann-windowed)  ->  DFT as one fp32-MFMA GEMM against a cos | -sin basis
//     ->  power, bin-pair mean, dB, per-clip top_db floor, optional z-score          -> [B, 1, T, n_fft/2 + 1]
// The DFT-by-GEMM costs 2*T*n*(n+2) FLOP per clip (0.42 GFLOP at n = 1024, T = 200) — 100x an FFT's count
// but one launch of the convolution kernel, with no FFT library or twiddle plumbing; at batch 64 it is
// 27 GFLOP, a fraction of a millisecond next to the 16 ms training step.
#include <math.h>

#include "common.h"

namespace avid {

// basis[j][k]: j < F: cos(2 pi j k / n);  F <= j < 2F: -sin(2 pi (j - F) k / n);  else 0  (rows = GEMM N, [N][K])
__global__ void logspec_basis_kernel(float* __restrict__ basis, int n, int F, int Npad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)Npad * n) return;
  const int k = (int)(i % n), j = (int)(i / n);
  float v = 0.f;
  if (j < 2 * F) {
    const int bin = j < F ? j : j - F;
    const int m = (int)(((long long)bin * k) % n);                 // exact argument reduction
    const double a = 2.0 * M_PI * (double)m / (double)n;
    v = j < F ? (float)cos(a) : (float)(-sin(a));
  }
  basis[i] = v;
}

// A[(b*T + t)][k] = hann[k] * x[b][reflect(t*hop + k - n/2)]
__global__ void logspec_frames_kernel(const float* __restrict__ sig, float* __restrict__ A, int B, int L, int n, int hop,
                                      int T) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * T * n) return;
  const int k = (int)(i % n);
  const long long r = i / n;
  const int t = (int)(r % T), b = (int)(r / T);
  int p = t * hop + k - n / 2;
  if (p < 0) p = -p;
  if (p >= L) p = 2 * (L - 1) - p;
  const float w = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)k / (double)n));
  A[i] = w * sig[(long long)b * L + p];
}

__device__ __forceinline__ int float_order(float v) {   // monotone float -> int map (for atomicMax)
  const int i = __float_as_int(v);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float order_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void logspec_init_kernel(int* __restrict__ mx, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) mx[i] = float_order(-INFINITY);
}

// C[(b*T + t)][Npad] (re | im) -> db[b][t][fo], per-clip maximum
__global__ void logspec_db_kernel(const float* __restrict__ C, float* __restrict__ db, int* __restrict__ mx, int B, int T,
                                  int F, int Fo, int Npad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float v = -INFINITY;
  int b = 0;
  if (i < (long long)B * T * Fo) {
    const int fo = (int)(i % Fo);
    const long long r = i / Fo;
    b = (int)(r / T);
    const float* row = C + r * Npad;
    auto power = [&](int f) { return row[f] * row[f] + row[F + f] * row[F + f]; };
    const float S = fo == 0 ? power(0) : 0.5f * (power(2 * fo - 1) + power(2 * fo));
    v = 10.f * log10f(fmaxf(1e-10f, S));
    db[i] = v;
  }
  // one atomic per wave when the wave sits inside one clip (the common case), else per lane
  const int b0 = __shfl(b, 0, 64);
  const bool uniform = __all(b == b0 || v == -INFINITY);
  if (uniform) {
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m != -INFINITY) atomicMax(&mx[b0], float_order(m));
  } else if (v != -INFINITY) {
    atomicMax(&mx[b], float_order(v));
  }
}

__global__ void logspec_finish_kernel(float* __restrict__ out, const int* __restrict__ mx, const float* __restrict__ mean,
                                      const float* __restrict__ stdv, int B, int T, int Fo, float top_db) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * T * Fo) return;
  const int fo = (int)(i % Fo);
  const int b = (int)(i / ((long long)T * Fo));
  float v = fmaxf(out[i], order_float(mx[b]) - top_db);
  if (mean) v = (v - mean[fo]) / (stdv[fo] + 1e-5f);
  out[i] = v;
}

}  // namespace avid

using namespace avid;

static int logspec_npad(int n) { return (2 * (n / 2 + 1) + 63) / 64 * 64; }

extern "C" size_t avid_logspec_basis_floats(int n_stft) {
  if (n_stft < 64 || n_stft % 32) return 0;
  return (size_t)logspec_npad(n_stft) * n_stft;
}

extern "C" int avid_logspec_basis(int n_stft, float* basis, avid_stream_t stream) {
  AVID_REQUIRE(n_stft >= 64 && n_stft % 32 == 0 && basis, AVID_E_BADARG, "logspec_basis: n_stft must be a multiple of 32");
  const int Npad = logspec_npad(n_stft);
  const long long tot = (long long)Npad * n_stft;
  hipLaunchKernelGGL(logspec_basis_kernel, dim3((unsigned)ceil_div(tot, 256)), dim3(256), 0, (hipStream_t)stream, basis, n_stft,
                     n_stft / 2 + 1, Npad);
  return check_launch("logspec_basis");
}

extern "C" size_t avid_logspec_workspace_bytes(int B, int n_stft, int T) {
  if (B <= 0 || T <= 0 || n_stft < 64) return 0;
  const size_t rows = (size_t)B * T;
  return sizeof(float) * rows * ((size_t)n_stft + logspec_npad(n_stft)) + sizeof(int) * (((size_t)B + 63) / 64 * 64);
}

extern "C" int avid_logspec(int B, int L, const float* sig, int n_stft, int hop, int T, const float* basis,
                            const float* mean, const float* stdv, float top_db, float* out, void* ws, size_t ws_bytes,
                            avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && L > n_stft / 2 && hop > 0 && T > 0 && sig && basis && out && ws, AVID_E_BADARG,
               "logspec: bad argument");
  AVID_REQUIRE(n_stft >= 64 && n_stft % 64 == 0, AVID_E_UNSUPPORTED, "logspec: the STFT size must be a multiple of 64");
  AVID_REQUIRE(T <= 1 + L / hop, AVID_E_SHAPE, "logspec: %d frames requested, the signal has %d", T, 1 + L / hop);
  AVID_REQUIRE((mean == nullptr) == (stdv == nullptr), AVID_E_BADARG, "logspec: mean and std go together");
  AVID_REQUIRE(ws_bytes >= avid_logspec_workspace_bytes(B, n_stft, T), AVID_E_BADARG, "logspec: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int F = n_stft / 2 + 1, Fo = n_stft / 4 + 1, Npad = logspec_npad(n_stft);
  const long long rows = (long long)B * T;
  float* A = static_cast<float*>(ws);
  float* Cm = A + rows * n_stft;
  int* mx = reinterpret_cast<int*>(Cm + rows * Npad);
  {
    ScopedTimer t(s, "logspec_frames_kernel", 0.0, 4.0 * rows * n_stft);
    hipLaunchKernelGGL(logspec_frames_kernel, dim3((unsigned)ceil_div(rows * n_stft, 256)), dim3(256), 0, s, sig, A, B, L,
                       n_stft, hop, T);
  }
  int rc = check_launch("logspec_frames");
  if (rc) return rc;
  rc = sim_gemm_nt(A, basis, Cm, nullptr, 0, rows, Npad, n_stft, s);
  if (rc) return rc;
  hipLaunchKernelGGL(logspec_init_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, s, mx, B);
  const long long no = rows * Fo;
  {
    ScopedTimer t(s, "logspec_db_kernel", 0.0, 4.0 * rows * (2.0 * F + Fo));
    hipLaunchKernelGGL(logspec_db_kernel, dim3((unsigned)ceil_div(no, 256)), dim3(256), 0, s, Cm, out, mx, B, T, F, Fo, Npad);
  }
  hipLaunchKernelGGL(logspec_finish_kernel, dim3((unsigned)ceil_div(no, 256)), dim3(256), 0, s, out, mx, mean, stdv, B, T, Fo,
                     top_db);
  return check_launch("logspec");
}
