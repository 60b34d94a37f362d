// Shared host/device helpers for libavid_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/avid_hip.h"

namespace avid {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return AVID_E_HIP;
  }
  return AVID_OK;
}

#define AVID_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) {                    \
      avid::set_error(__VA_ARGS__);   \
      return (code);                  \
    }                                 \
  } while (0)

// CUs the persistent kernels size their grids for: the device's, or avid_set_cu_budget()'s / AVID_CU_RESERVE's fewer
// (a multiple of 8: whole CUs per XCD) — common.hip
int device_cus();

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Optional per-kernel timing with HIP events recorded on the launch stream (avid_timing_enable /
// avid_timing_report): bench.py uses it for the roofline numbers.  Zero cost when disabled.
struct ScopedTimer {
  hipStream_t s;
  int slot;
  ScopedTimer(hipStream_t stream, const char* name, double flops, double bytes);
  ~ScopedTimer();
};

// conv.hip: C[M][N] = A[M][K].Bq[N][K]^T (op 0) / min(.,Cin) (1) / max(.,Cin) (2) on the MFMA igemm kernel
int sim_gemm_nt(const float* A, const float* Bq, float* C, const float* Cin, int op, long long M, int N, int K,
                hipStream_t s);

// stem.hip: direct-from-LDS-patch stem convolutions (Cin 3 / 1, 7x7 stride 2)
bool stem_fwd_supported(const avid_conv_desc* d);
bool stem_wgrad_supported(const avid_conv_desc* d);
size_t stem_fwd_ws_bytes(const avid_conv_desc* d);
size_t stem_wgrad_ws_bytes(const avid_conv_desc* d);
int stem_fwd(const avid_conv_desc* d, const float* x, const float* w, float* y, float* stats, void* ws, hipStream_t s);
int stem_fwd_grid(const avid_conv_desc* d);
bool stem_fwd_is_split(const avid_conv_desc* d);     // runs as stem_fwd3_kernel (fp32-accurate split-bf16 products)
bool stem_fwd_is_presplit(const avid_conv_desc* d);  // ... as stem_fwd3p_kernel (its patch split once, at commit time)
int stem_wgrad(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, hipStream_t s);

// wino.hip: Winograd F(2x2, 3x3) forward / input gradient of the (1,3,3) stride-1 layers (mode 0 / 1)
bool wino_supported(const avid_conv_desc* d, int mode);
int wino_variant(const avid_conv_desc* d, int mode);      // 1: wino_kernel, 2: wino2_kernel (the layout of a pre-transformed U follows it)
size_t wino_ws_bytes(const avid_conv_desc* d, int mode);
int wino_grid(const avid_conv_desc* d, int mode);
int wino_conv(const avid_conv_desc* d, int mode, const float* src, const float* w, const float* u_pre, float* dst,
              const float* addend, float* stats, const avid_bn_bwd_fuse* bn, void* ws, hipStream_t s);
// Winograd weight gradient of the same layers: slabs [nsplit][Cout][9][Cin] in ws (nsplit == 1: dw itself)
bool wino_wgrad_supported(const avid_conv_desc* d);
size_t wino_wgrad_ws_bytes(const avid_conv_desc* d);
int wino_wgrad(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, int* nsplit_out,
               hipStream_t s);

// 64-lane wave reductions (DPP/ds_swizzle chosen by the compiler from __shfl_xor).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware bijective remap of a linear workgroup id: XCD k (= hardware id % 8) receives one
// contiguous chunk of the logical tile space so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  if (nwg < nx) return bid;
  unsigned q = nwg / nx, r = nwg % nx;
  unsigned xcd = bid % nx, pos = bid / nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + pos;
}

// wino2_kernel computes its products as split-bf16 (three bf16 terms per operand, six v_mfma_f32_32x32x16_bf16 per product
// block: DESIGN.md 8e) and reads U already split; -DAVID_W2_FP32 (every source file) builds the fp32 instruction instead.
#ifdef AVID_W2_FP32
constexpr bool W2_SPLIT = false;
#else
constexpr bool W2_SPLIT = true;
#endif

// (x0, x1) -> three packed bf16 pairs h, m, l (element 0 in the low half) with x = h + m + l to fp32 accuracy: h = bf16(x),
// m = bf16(x - h), l = bf16(x - h - m), round to nearest even.
//
// Round 5: the residuals come from v_dot2c_f32_bf16 — r0 = x0 + (h0, h1) . (-1, 0), r1 = x1 + (h0, h1) . (0, -1): exact (a
// bf16 times -1 or 0 and one fp32 subtraction whose result is representable), two instructions per pair where expanding h
// to fp32 and subtracting took three (and / shift / packed subtract): SEVEN vector instructions per pair instead of nine.
// (In every split-bf16 kernel the splits are as many instructions as everything else: profiles/r04_b counts 5.6-13 vector
// instructions per matrix instruction, and a SIMD runs its waves largely as one serial stream — tools/tconv_trace.py,
// DESIGN.md 8g.)  Bit-identical to the classic form for every finite input, denormals included
// (tools/split_dot_check.hip: 2^25 random pairs); a non-finite element (or |x| > 3.39e38: h rounds to inf) now also turns
// its PAIR-MATE's lower terms into NaN (0 x inf) — the neighbouring k index of the same fragment row, i.e. of the same
// dot products, which the non-finite element has already made NaN (tests/test_gpu_precision.py).
// Two traps, both found by that lab: the pair (-1, 0) = 0x0000bf80 has to come from a REGISTER — as a constant the compiler
// and the assembler encode it as the inline constant -1.0, which the hardware reads as 0xbf800000 = (0, -1); and the
// builtin, not inline assembly — a dot instruction reading a register the previous instruction wrote needs a wait state
// that only the compiler's hazard recogniser inserts.  -DAVID_SPLIT_CLASSIC (every source file) builds the round-4 forms.
typedef float floatx2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_bf16_dot(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  unsigned k10 = 0x0000bf80u;
  asm volatile("" : "+s"(k10));
  const bf16x2_t m10 = __builtin_bit_cast(bf16x2_t, k10), m01 = {(__bf16)0.0f, (__bf16)-1.0f};
  const floatx2_t x = {x0, x1};
  const bf16x2_t hb = __builtin_convertvector(x, bf16x2_t);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, m10, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, m01, x1, false);
  const floatx2_t r = {r0, r1};
  const bf16x2_t mb = __builtin_convertvector(r, bf16x2_t);
  const float t0 = __builtin_amdgcn_fdot2_f32_bf16(mb, m10, r0, false), t1 = __builtin_amdgcn_fdot2_f32_bf16(mb, m01, r1, false);
  const floatx2_t t = {t0, t1};
  const bf16x2_t lb = __builtin_convertvector(t, bf16x2_t);
  h = __builtin_bit_cast(unsigned, hb);
  m = __builtin_bit_cast(unsigned, mb);
  l = __builtin_bit_cast(unsigned, lb);
}
// Where it pays (same box, alternating library builds, ms per step): igemm_pk_kernel<4,1,1,2,0> 0.594 -> 0.557, <4,1,1,2,1>
// 0.448 -> 0.42, wgrad_group_kernel 0.76 -> 0.74; where it does not: stem_fwd3_kernel 0.67 -> 0.73, wino2_kernel<1> 0.805 ->
// 0.83, tconv64_kernel<0> 0.29 -> 0.31 (v_dot2c issues a little slower than a plain vector instruction — 23.3 against 27.1 ns
// per pair for 7 against 9 instructions — and those kernels' own instruction interleave was tuned around the classic forms).
// So it is a third form, chosen per kernel; -DAVID_SPLIT_CLASSIC (every source file) maps it back to split2_bf16.

// The classic form.  Plain vector code on purpose — v_cvt_pk_bf16_f32, v_lshl /
// v_and, v_add: as inline assembly the same instructions came out with a wait state behind every packed subtraction
// and could not be moved by the scheduler (wino2_kernel: 160 -> 146 us on conv2x).
__device__ __forceinline__ void split2_bf16(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const floatx2_t x = {x0, x1};
  const bf16x2_t hb = __builtin_convertvector(x, bf16x2_t);
  const floatx2_t r = x - __builtin_convertvector(hb, floatx2_t);
  const bf16x2_t mb = __builtin_convertvector(r, bf16x2_t);
  const floatx2_t t = r - __builtin_convertvector(mb, floatx2_t);
  const bf16x2_t lb = __builtin_convertvector(t, bf16x2_t);
  h = __builtin_bit_cast(unsigned, hb);
  m = __builtin_bit_cast(unsigned, mb);
  l = __builtin_bit_cast(unsigned, lb);
}

// The same split with the conversions as inline assembly (the subtractions stay scalar): what the kernels with two
// waves per SIMD and their own instruction interleave run faster with (same box: stem_wgrad3_kernel 0.567 against 0.611
// ms, stem_fwd3_kernel 0.629 / 0.649, wgrad_tab_kernel 0.337 / 0.356; igemm_pk_kernel and wino2_kernel prefer the form above).
__device__ __forceinline__ void split2_bf16_asm(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
  const float t0 = r0 - __uint_as_float(m << 16), t1 = r1 - __uint_as_float(m & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(t0), "v"(t1));
}

#ifdef AVID_SPLIT_CLASSIC
#define split2_bf16_dot split2_bf16
#endif

// x = h + m + l to fp32 accuracy, each term a bf16 (round to nearest even): the scalar form of the packed splits in
// conv.hip / stem.hip / wino.hip (same instruction, same roundings)
__device__ __forceinline__ void split3_bf16(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  unsigned ph, pm, pl;
  split2_bf16(x, 0.f, ph, pm, pl);
  h = (unsigned short)ph; m = (unsigned short)pm; l = (unsigned short)pl;
}

// U[(xi * Cn + n) * Cr + k] = (G g G^T)[xi / 4][xi % 4],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], for the elements
// i = first, first + stride, ... of the Cn x Cr (n, k) pairs.  flip = 0: g[a][b] = w[n][a][b][k] (forward); flip = 1:
// g[a][b] = w[k][2-a][2-b][n] (input gradient).  Shared by wino_weight_kernel and weight_transpose_batched_kernel.
// U in the operand-fragment order of the kernel that consumes it (one contiguous KB per wave load instruction):
// frag = 2 (wino_kernel):  U[xi][n / 32][k / 32][q = (k % 16) / 4][lane = 32 ((k % 32) / 16) + n % 32][k % 4]
// frag = 1 (wino2_kernel): U[xi][n / 64][(n % 64) / 32][k / 16][q = (k % 8) / 4][lane = 32 ((k % 16) / 8) + n % 32][k % 4]
// frag = 3 (wino2_kernel, split-bf16): bf16 U[xi][n / 64][(n % 64) / 32][k / 16][term: hi, mid, lo][lane = 32 ((k % 16) / 8) + n % 32][k % 8]
//           (6 bytes per element: 16 * Cn * Cr * 6 bytes in all)
__device__ __forceinline__ void wino_weight_elements(const float* __restrict__ w, float* __restrict__ U, int Cn, int Cr, int Cin, int flip,
                                     long long first, long long stride, int frag) {
  for (long long i = first; i < (long long)Cn * Cr; i += stride) {
    const int k = (int)(i % Cr), n = (int)(i / Cr);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        g[a][b] = flip ? w[(((long long)k * 3 + (2 - a)) * 3 + (2 - b)) * Cin + n] : w[(((long long)n * 3 + a) * 3 + b) * Cin + k];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                  u3 = t[a][2];
      const long long st = (long long)Cn * Cr;
      long long o;
      if (frag == 3) {
        unsigned short* U16 = reinterpret_cast<unsigned short*>(U);
        const long long blk = ((long long)(n >> 6) * 2 + ((n >> 5) & 1)) * (Cr >> 4) + (k >> 4);
        const long long o16 = (long long)(a * 4) * st * 3 + blk * 1536 + ((((k & 15) >> 3) * 32 + (n & 31)) << 3) + (k & 7);
        const float uu[4] = {u0, u1, u2, u3};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          unsigned short th, tm, tl;
          split3_bf16(uu[b], th, tm, tl);
          U16[o16 + b * st * 3] = th;
          U16[o16 + b * st * 3 + 512] = tm;
          U16[o16 + b * st * 3 + 1024] = tl;
        }
        continue;
      }
      if (frag == 2) {
        const int kk = k & 31;
        o = (long long)(a * 4) * st +
            ((((long long)(n >> 5) * (Cr >> 5) + (k >> 5)) * 4 + ((kk & 15) >> 2)) * 64 + ((kk >> 4) * 32 + (n & 31))) * 4 + (kk & 3);
      } else {
        const int kk = k & 15;
        o = (long long)(a * 4) * st +
            ((((long long)(n >> 6) * 2 + ((n >> 5) & 1)) * (Cr >> 4) + (k >> 4)) * 2 + ((kk & 7) >> 2)) * 256 +
            (((kk >> 3) * 32 + (n & 31)) << 2) + (kk & 3);
      }
      U[o] = u0;
      U[o + st] = u1;
      U[o + 2 * st] = u2;
      U[o + 3 * st] = u3;
    }
  }
}


}  // namespace avid
