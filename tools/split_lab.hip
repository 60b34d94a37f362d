// Dev tool: what does the three-term bf16 split of 8 fp32 values cost a wave, alone and in the shadow of the six
// v_mfma_f32_32x32x16_bf16 it feeds?  Variants: RNE split with v_cvt_pk_bf16_f32 (pk_split8 of conv.hip) against an exact
// truncation split built from v_and / v_pk_add_f32 / v_perm_b32.  One wave per SIMD (256 threads per CU), host events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_cvt(const float (&v)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = v[2 * i], x1 = v[2 * i + 1];
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(x0), "v"(x1));
    const float r0 = x0 - __uint_as_float(h[i] << 16), r1 = x1 - __uint_as_float(h[i] & 0xffff0000u);
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m[i]) : "v"(r0), "v"(r1));
    const float t0 = r0 - __uint_as_float(m[i] << 16), t1 = r1 - __uint_as_float(m[i] & 0xffff0000u);
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l[i]) : "v"(t0), "v"(t1));
  }
  fh = __builtin_bit_cast(bf16x8, uintx4{h[0], h[1], h[2], h[3]});
  fm = __builtin_bit_cast(bf16x8, uintx4{m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(bf16x8, uintx4{l[0], l[1], l[2], l[3]});
}

// exact: hi = top 16 bits of x, r = x - hi (exact), mid = top 16 bits of r, lo = r - mid (exact, <= 8 significant bits)
__device__ __forceinline__ void split_trunc(const float (&v)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned u0 = __float_as_uint(v[2 * i]), u1 = __float_as_uint(v[2 * i + 1]);
    const float h0 = __uint_as_float(u0 & 0xffff0000u), h1 = __uint_as_float(u1 & 0xffff0000u);
    h[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = v[2 * i] - h0, r1 = v[2 * i + 1] - h1;
    const unsigned q0 = __float_as_uint(r0), q1 = __float_as_uint(r1);
    m[i] = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
    const float t0 = r0 - __uint_as_float(q0 & 0xffff0000u), t1 = r1 - __uint_as_float(q1 & 0xffff0000u);
    l[i] = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
  }
  fh = __builtin_bit_cast(bf16x8, uintx4{h[0], h[1], h[2], h[3]});
  fm = __builtin_bit_cast(bf16x8, uintx4{m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(bf16x8, uintx4{l[0], l[1], l[2], l[3]});
}

__device__ __forceinline__ floatx2 pk_sub(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// RNE split with packed subtractions: 36 instructions (w2_split8 of wino.hip)
__device__ __forceinline__ void split_pk(const float (&v)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const floatx2 x = {v[2 * i], v[2 * i + 1]};
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(x[0]), "v"(x[1]));
    const floatx2 r = pk_sub(x, floatx2{__uint_as_float(h[i] << 16), __uint_as_float(h[i] & 0xffff0000u)});
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m[i]) : "v"(r[0]), "v"(r[1]));
    const floatx2 t = pk_sub(r, floatx2{__uint_as_float(m[i] << 16), __uint_as_float(m[i] & 0xffff0000u)});
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l[i]) : "v"(t[0]), "v"(t[1]));
  }
  fh = __builtin_bit_cast(bf16x8, uintx4{h[0], h[1], h[2], h[3]});
  fm = __builtin_bit_cast(bf16x8, uintx4{m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(bf16x8, uintx4{l[0], l[1], l[2], l[3]});
}

// software-pipelined: the split of step u + 1 sits between the six MFMAs of step u.  PAIR: two accumulators alternate
// (no MFMA waits for the previous one's result); HINT: sched_group_barrier 1 MFMA : 6 VALU
template <bool PAIR, bool HINT>
__global__ __launch_bounds__(256) void kp(float* out, const float* in, int iters) {
  const int tid = threadIdx.x;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[(tid * 8 + i) & 4095];
  floatx16 acc0, acc1;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  bf16x8 bh, bm, bl, ah[2], am[2], al[2];
  split_pk(v, bh, bm, bl);
  split_pk(v, ah[0], am[0], al[0]);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = u & 1, n = c ^ 1;
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1e-3f));
      split_pk(v, ah[n], am[n], al[n]);
      floatx16& acc = (PAIR && c) ? acc1 : acc0;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[c], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[c], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[c], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[c], bh, acc, 0, 0, 0);
      if (HINT) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 1234.5f) out[tid] = s;
}

template <bool PAIR, bool HINT>
static float time_p(float* out, const float* in, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((kp<PAIR, HINT>), dim3(256), dim3(256), 0, 0, out, in, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((kp<PAIR, HINT>), dim3(256), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

// KIND: 0 cvt split only, 1 trunc split only, 2 six MFMAs only, 3 cvt split + MFMAs, 4 trunc split + MFMAs
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[(tid * 8 + i) & 4095];
  floatx16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 bh, bm, bl;
  split_cvt(v, bh, bm, bl);
  bf16x8 ah = bh, am = bm, al = bl;
  float sink = 0.f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0 || KIND == 3) split_cvt(v, ah, am, al);
      if (KIND == 1 || KIND == 4) split_trunc(v, ah, am, al);
      if (KIND >= 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      } else {
        const uintx4 q = __builtin_bit_cast(uintx4, ah) ^ __builtin_bit_cast(uintx4, am) ^ __builtin_bit_cast(uintx4, al);
        sink += __uint_as_float((q[0] ^ q[1] ^ q[2] ^ q[3]) & 0x3f800000u);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1e-3f));   // new data every step
    }
  }
  float s = sink;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 1234.5f) out[tid] = s;
}

__global__ void check(const float* in, float* err, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 8 >= n) return;
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = in[i * 8 + e];
  bf16x8 h, m, l, h2, m2, l2;
  split_cvt(v, h, m, l);
  split_trunc(v, h2, m2, l2);
  float e1 = 0.f, e2 = 0.f;
  for (int e = 0; e < 8; ++e) {
    const double a = (double)(float)h[e] + (double)(float)m[e] + (double)(float)l[e];
    const double b = (double)(float)h2[e] + (double)(float)m2[e] + (double)(float)l2[e];
    e1 = fmaxf(e1, (float)(fabs(a - (double)v[e]) / fabs((double)v[e])));
    e2 = fmaxf(e2, (float)(fabs(b - (double)v[e]) / fabs((double)v[e])));
  }
  err[2 * i] = e1; err[2 * i + 1] = e2;
}

template <int KIND>
static float time_it(float* out, const float* in, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, in, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out, *in, *err;
  const int n = 4096;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&in, n * 4); (void)hipMalloc(&err, n);
  float host[n];
  for (int i = 0; i < n; ++i) host[i] = (float)((sin(i * 12.9898) * 43758.5453) - floor(sin(i * 12.9898) * 43758.5453) - 0.5) * (i % 7 == 0 ? 1e-3f : 3.f);
  (void)hipMemcpy(in, host, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(2), dim3(256), 0, 0, in, err, n);
  float herr[n / 4];
  (void)hipMemcpy(herr, err, n, hipMemcpyDeviceToHost);
  float e1 = 0, e2 = 0;
  for (int i = 0; i < n / 8; ++i) { e1 = fmaxf(e1, herr[2 * i]); e2 = fmaxf(e2, herr[2 * i + 1]); }
  printf("reconstruction error (max relative): cvt (RNE) %.3g   trunc %.3g\n", e1, e2);
  const int iters = 20000;
  const char* names[] = {"split cvt alone", "split trunc alone", "6 MFMA alone", "split cvt + 6 MFMA", "split trunc + 6 MFMA"};
  float t[5] = {time_it<0>(out, in, iters), time_it<1>(out, in, iters), time_it<2>(out, in, iters), time_it<3>(out, in, iters), time_it<4>(out, in, iters)};
  for (int i = 0; i < 5; ++i) printf("%-22s %8.3f ms  %7.1f ns per step (8 values%s)\n", names[i], t[i], t[i] * 1e6 / (iters * 4.0), i >= 2 ? ", 6 MFMA = 192 matrix cycles" : "");
  printf("pipelined (split of the next step between this step's MFMAs), 36-instruction split:\n");
  printf("  one accumulator          %7.1f ns per step\n", time_p<false, false>(out, in, iters) * 1e6 / (iters * 4.0));
  printf("  one accumulator, hints   %7.1f ns per step\n", time_p<false, true>(out, in, iters) * 1e6 / (iters * 4.0));
  printf("  two accumulators         %7.1f ns per step\n", time_p<true, false>(out, in, iters) * 1e6 / (iters * 4.0));
  printf("  two accumulators, hints  %7.1f ns per step\n", time_p<true, true>(out, in, iters) * 1e6 / (iters * 4.0));
  return 0;
}
