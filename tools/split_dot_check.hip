// Lab: the three-term bf16 split with its residuals from v_dot2c_f32_bf16 (r = x - hi as hi . (-1, 0) + x: two instructions per
// pair instead of and / shift / packed subtract: 7 vector instructions per two values instead of 9) against the split of
// csrc/common.h, bit for bit over random bit patterns, special values and denormals; and the issue time of both forms.
//   hipcc --offload-arch=gfx950 -O3 tools/split_dot_check.hip -o tools/bin/split_dot_check && tools/bin/split_dot_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float floatx2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_ref(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const floatx2_t x = {x0, x1};
  const bf16x2_t hb = __builtin_convertvector(x, bf16x2_t);
  const floatx2_t r = x - __builtin_convertvector(hb, floatx2_t);
  const bf16x2_t mb = __builtin_convertvector(r, bf16x2_t);
  const floatx2_t t = r - __builtin_convertvector(mb, floatx2_t);
  const bf16x2_t lb = __builtin_convertvector(t, bf16x2_t);
  h = __builtin_bit_cast(unsigned, hb); m = __builtin_bit_cast(unsigned, mb); l = __builtin_bit_cast(unsigned, lb);
}
__device__ __forceinline__ void split_dot(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  // The pair (-1, 0) = 0x0000bf80 must come from a REGISTER: as a constant the compiler (and the assembler) encode it as the
  // inline constant -1.0, which the hardware reads as 0xbf800000 = (0, -1).  The builtin, not inline assembly: a dot
  // instruction that reads a register the instruction before it wrote needs a wait state only the compiler's hazard
  // recogniser inserts.
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
  h = __builtin_bit_cast(unsigned, hb); m = __builtin_bit_cast(unsigned, mb); l = __builtin_bit_cast(unsigned, lb);
}
__global__ void check(const float* x, int n, unsigned long long* bad, unsigned* first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned h0, m0, l0, h1, m1, l1;
  split_ref(x[2 * i], x[2 * i + 1], h0, m0, l0);
  split_dot(x[2 * i], x[2 * i + 1], h1, m1, l1);
  if (h0 != h1 || m0 != m1 || l0 != l1) {
    if (atomicAdd(bad, 1ull) == 0) { first[0] = __float_as_uint(x[2 * i]); first[1] = __float_as_uint(x[2 * i + 1]); first[2] = m0; first[3] = m1; first[4] = l0; first[5] = l1; }
  }
}
template <int WHICH>
__global__ void rate(const float* x, unsigned* out, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = x[threadIdx.x * 8 + i];
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h, m, l;
      if (WHICH == 0) split_ref(a[2 * i], a[2 * i + 1], h, m, l); else split_dot(a[2 * i], a[2 * i + 1], h, m, l);
      acc ^= h + m + l;
      a[2 * i] += __uint_as_float((l & 0xff) | 0x33000000u);      // keep the chain data-dependent
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const int n = 1 << 26;
  std::vector<float> hx(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned u = ((unsigned)rand() << 17) ^ ((unsigned)rand() << 2) ^ (unsigned)rand();
    if (i % 97 == 0) u &= 0x807fffffu;                        // denormals
    if (i % 101 == 0) u = (u & 0x80000000u) | 0x7f7fffffu;    // +-FLT_MAX
    if (i % 103 == 0) u = (u & 0x80000000u);                  // +-0
    float f; memcpy(&f, &u, 4);
    if (f != f) f = 1.0f;                                      // (NaN payloads are not compared)
    hx[i] = f;
  }
  float* dx; unsigned long long* bad; unsigned* first; unsigned* out;
  hipMalloc(&dx, n * 4); hipMalloc(&bad, 8); hipMalloc(&first, 32); hipMalloc(&out, 4 * 256 * 1024);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 8); hipMemset(first, 0, 32);
  check<<<n / 2 / 256, 256>>>(dx, n, bad, first);
  unsigned long long hb = 0; unsigned hf[8];
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost);
  printf("pairs %d: %llu differ", n / 2, hb);
  if (hb) printf("  first: x %08x %08x  mid %08x / %08x  lo %08x / %08x", hf[0], hf[1], hf[2], hf[3], hf[4], hf[5]);
  printf("\n");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) rate<0><<<1024, 256>>>(dx, out, 4096); else rate<1><<<1024, 256>>>(dx, out, 4096);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // 1024 blocks x 4 waves on 1024 SIMDs = 4 waves per SIMD; per wave iters x 4 pair-splits
      if (rep) printf("%s: %.3f ms -> %.2f ns per pair-split per SIMD\n", which ? "dot2c" : "and/shift/pk_add", ms, ms * 1e6 / (4.0 * 4096 * 4));
    }
  }
  return 0;
}
