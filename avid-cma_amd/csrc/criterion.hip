// Memory-bank NCE criterion kernels: wavefront-per-row gathers + DPP reductions, sized for banks of
// N x 128 fp32 rows (512 B = one fully coalesced wave64 float2 read) resident in HBM.
//
// Reference ops replaced: criterions/avid.py:47-129 (normalize, gather, bmm scores, EMA update),
// criterions/nce.py:21-58, utils/alias_method.py:56-71, criterions/avid_cma.py:196-209.
#include <math.h>

#include "common.h"

namespace avid {

// ---------------------------------------------------------------------------------------------
// F.normalize(x, p=2, dim=1): one wave per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm_out, int bs, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= bs) return;
  const float* p = x + (long long)row * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss += p[d] * p[d];
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  for (int d = lane; d < D; d += 64) y[(long long)row * D + d] = p[d] / nrm;
  if (lane == 0) norm_out[row] = nrm;
}

// dx = (dy - y * <y, dy>) / norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ norm,
                                                         const float* __restrict__ dy, float* __restrict__ dx, int bs,
                                                         int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= bs) return;
  const long long o = (long long)row * D;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot += y[o + d] * dy[o + d];
  dot = wave_sum(dot);
  const float inv = 1.f / norm[row];
  for (int d = lane; d < D; d += 64) dx[o + d] = (dy[o + d] - y[o + d] * dot) * inv;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al.): counter = (i_lo, i_hi, off_lo, off_hi), key = (seed_lo, seed_hi)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t& r0, uint32_t& r1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  r0 = c0;
  r1 = c1;
}

__global__ void counter_add_kernel(unsigned long long* ctr, unsigned long long inc) { *ctr += inc; }

__global__ void alias_draw_kernel(long long n, long long K, const float* __restrict__ prob,
                                  const long long* __restrict__ alias, int uniform, uint64_t seed, uint64_t offset,
                                  const unsigned long long* __restrict__ offset_dev,
                                  const long long* __restrict__ y, long long per_row, long long* __restrict__ out) {
  if (offset_dev) offset = *offset_dev;  // graph-replay safe: the draw counter lives in device memory
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t r0, r1;
    philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r0, r1);
    const long long kk = (long long)(((uint64_t)r0 * (uint64_t)K) >> 32);
    long long v = kk;
    if (!uniform) {
      const float u = (float)(r1 >> 8) * 5.9604644775390625e-8f;  // 2^-24
      v = (u < prob[kk]) ? kk : alias[kk];
    }
    if (y) v += (v >= y[i / per_row]) ? 1 : 0;
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// scores[b][j] = <bank[idx[b][j]], emb[b]> * inv_T.  grid = (row chunks, bs); 4 waves/block, each
// wave keeps its 16 independent 512-B row reads in flight.
// ---------------------------------------------------------------------------------------------
constexpr int SC_ROWS_PER_BLOCK = 64;

template <int DPL>  // floats per lane: D = 64 * DPL
__global__ __launch_bounds__(256) void bank_scores_fwd_kernel(const long long* __restrict__ idx,
                                                              const float* __restrict__ bank,
                                                              const float* __restrict__ emb, float inv_T,
                                                              float* __restrict__ scores,
                                                              float* __restrict__ rows_out, int R, long long N,
                                                              int* __restrict__ err) {
  constexpr int D = 64 * DPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  float e[DPL];
#pragma unroll
  for (int k = 0; k < DPL; ++k) e[k] = emb[(long long)b * D + k * 64 + lane];
  const int j0 = blockIdx.x * SC_ROWS_PER_BLOCK;
  const int j1 = min(j0 + SC_ROWS_PER_BLOCK, R);
  const long long* ib = idx + (long long)b * R;
  // a wave owns 16 consecutive rows: lanes 0..15 fetch their indices, then all 16 x DPL row loads are issued
  // before the first use (with 4 rows per trip and the index load inside the trip a wave made 8 dependent
  // round trips to random HBM rows: 1.8 TB/s on the 2M-row bank)
  constexpr int RPW = SC_ROWS_PER_BLOCK / 4;
  const int jw = j0 + wave * RPW;
  long long mine = 0;
  if (lane < RPW && jw + lane < j1) {
    mine = ib[jw + lane];
    if (mine < 0 || mine >= N) {                        // the reference raises an index error (avid.py:57-62):
      if (err) atomicOr(err, AVID_DEVERR_BANK_INDEX);   // flag it (raised by the host at its next poll) and
      mine = mine < 0 ? 0 : N - 1;                      // never read outside the bank
    }
  }
  float v[RPW][DPL];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const long long row = __shfl(mine, u, 64);
    const float* p = bank + row * D;
#pragma unroll
    for (int k = 0; k < DPL; ++k) v[u][k] = p[k * 64 + lane];   // (rows past j1 re-read row 0: discarded below)
  }
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int j = jw + u;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < DPL; ++k) acc += v[u][k] * e[k];
    if (rows_out && j < j1) {   // snapshot of the pre-update row for backward
      float* ro = rows_out + ((long long)b * R + j) * D;
#pragma unroll
      for (int k = 0; k < DPL; ++k) ro[k * 64 + lane] = v[u][k];
    }
    const float s = wave_sum(acc);
    if (lane == 0 && j < j1) scores[(long long)b * R + j] = s * inv_T;
  }
}

// demb[b] (+)= inv_T * sum_j ds[b][j] * bank[idx[b][j]] : one 1024-thread block per sample
template <int DPL>
__global__ __launch_bounds__(1024) void bank_scores_bwd_kernel(const float* __restrict__ rows,
                                                               const long long* __restrict__ idx,
                                                               const float* __restrict__ bank,
                                                               const float* __restrict__ ds, float inv_T,
                                                               int accumulate, float* __restrict__ demb, int R,
                                                               long long N) {
  constexpr int D = 64 * DPL;
  __shared__ float sh[16][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const long long* ib = idx + (long long)b * R;
  const float* db = ds + (long long)b * R;
  float acc[DPL];
#pragma unroll
  for (int k = 0; k < DPL; ++k) acc[k] = 0.f;
  // four trips' worth of rows (16 x DPL loads) are issued before the first use; the accumulation order is
  // the one-trip-at-a-time order (one trip = 4 rows cost a dependent memory round trip each: 16 per block)
  for (int j = wave * 4; j < R; j += 256) {
    float v[4][4][DPL], g[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = j + 64 * t + u;
        const bool ok = jj < R;
        const float* p;
        if (rows) {
          p = rows + ((long long)b * R + (ok ? jj : 0)) * D;
        } else {
          long long row = ib[ok ? jj : 0];
          row = row < 0 ? 0 : (row >= N ? N - 1 : row);
          p = bank + row * D;
        }
        g[t][u] = ok ? db[jj] : 0.f;
#pragma unroll
        for (int k = 0; k < DPL; ++k) v[t][u][k] = p[k * 64 + lane];
      }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j + 64 * t + u < R) {
#pragma unroll
          for (int k = 0; k < DPL; ++k) acc[k] += g[t][u] * v[t][u][k];
        }
  }
#pragma unroll
  for (int k = 0; k < DPL; ++k) sh[wave][k * 64 + lane] = acc[k];
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 1024) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += sh[w][d];
    s *= inv_T;
    float* o = demb + (long long)b * D + d;
    *o = accumulate ? *o + s : s;
  }
}

// ---------------------------------------------------------------------------------------------
// NCE (criterions/nce.py)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;  // valid on thread 0
}

__global__ __launch_bounds__(1024) void mean_exp_kernel(const float* __restrict__ s, int rows, int cols, int ld,
                                                        float* __restrict__ out) {
  __shared__ double sh[16];
  double acc = 0;
  const long long n = (long long)rows * cols;
  for (long long i = threadIdx.x; i < n; i += 1024) acc += (double)expf(s[(i / cols) * ld + (i % cols)]);
  const double t = block_sum_d(acc, sh);
  if (threadIdx.x == 0) out[0] = (float)(t / (double)n);
}

// One launch, `gridDim.x` blocks (a single 1024-thread block is compute-bound on ONE CU: 25-36 us for the
// 64 x 1025 exp / log pairs of a step).  Each block leaves its fp64 partial in `partial`, takes a ticket, and
// the block that draws the last ticket adds the partials in block order (fixed summation order) and writes the
// loss; it also re-arms the ticket counter, so the scratch needs zeroing only once.
__global__ __launch_bounds__(1024) void nce_fwd_kernel(const float* __restrict__ spos, const float* __restrict__ sneg,
                                                       const float* __restrict__ Zp, int bs, int P, int K, int ldp,
                                                       int ldn, float scale, int accumulate,
                                                       float* __restrict__ loss, double* partial, unsigned* ticket) {
  __shared__ double sh[16];
  const float KZ = (float)K * Zp[0];
  double acc = 0;
  const long long nn = (long long)bs * K, np = (long long)bs * P;
  const long long stride = (long long)gridDim.x * 1024, first = (long long)blockIdx.x * 1024 + threadIdx.x;
  // element i = first + stride * m -> (row, col) kept incrementally (a 64-bit divide + modulo per element is most
  // of this kernel's instructions otherwise), four independent loads per trip
  {
    const int sr = (int)(stride / K), sc = (int)(stride % K);
    int row = (int)(first / K), col = (int)(first % K);
    auto next = [&]() {
      long long o = (long long)row * ldn + col;
      row += sr; col += sc;
      if (col >= K) { col -= K; ++row; }
      return o;
    };
    long long i = first;
    for (; i + 3 * stride < nn; i += 4 * stride) {
      const long long o0 = next(), o1 = next(), o2 = next(), o3 = next();
      const float s0 = sneg[o0], s1 = sneg[o1], s2 = sneg[o2], s3 = sneg[o3];
      acc += (double)(-logf(KZ / (expf(s0) + KZ)));
      acc += (double)(-logf(KZ / (expf(s1) + KZ)));
      acc += (double)(-logf(KZ / (expf(s2) + KZ)));
      acc += (double)(-logf(KZ / (expf(s3) + KZ)));
    }
    for (; i < nn; i += stride) acc += (double)(-logf(KZ / (expf(sneg[next()]) + KZ)));
  }
  const double invP = 1.0 / (double)P;
  for (long long i = first; i < np; i += stride) {
    const float e = expf(spos[(i / P) * ldp + (i % P)]);
    acc += (double)(-logf(e / (e + KZ))) * invP;
  }
  double t = block_sum_d(acc, sh);
  if (threadIdx.x != 0) return;
  if (gridDim.x > 1) {
    __hip_atomic_store(&partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();                                     // the partial is visible device-wide before the ticket
    if (atomicAdd(ticket, 1u) != gridDim.x - 1) return;
    __threadfence();
    t = 0;
    for (unsigned g = 0; g < gridDim.x; ++g)
      t += __hip_atomic_load(&partial[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const float v = (float)(t / (double)bs) * scale;
  loss[0] = accumulate ? loss[0] + v : v;
}

__global__ void nce_bwd_kernel(const float* __restrict__ spos, const float* __restrict__ sneg,
                               const float* __restrict__ Zp, const float* __restrict__ dloss, int bs, int P, int K,
                               int ldp, int ldn, float scale, float* __restrict__ dpos, float* __restrict__ dneg) {
  const float KZ = (float)K * Zp[0];
  const float g = dloss[0] * scale / (float)bs;
  const long long nn = (long long)bs * K, np = (long long)bs * P;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nn + np; i += stride) {
    if (i < nn) {
      const float e = expf(sneg[(i / K) * ldn + (i % K)]);
      dneg[i] = g * (e / (e + KZ));
    } else {
      const long long k = i - nn;
      const float e = expf(spos[(k / P) * ldp + (k % P)]);
      dpos[k] = -(g / (float)P) * (KZ / (e + KZ));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bank[y[i]] = normalize(m * bank[y[i]] + (1-m) * emb[i]); last duplicate wins. One wave per sample.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bank_update_kernel(float* __restrict__ bank, const long long* __restrict__ y,
                                                          const float* __restrict__ emb, float mom, int B, int D,
                                                          long long N, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const long long row = y[i];
  if (row < 0 || row >= N) {   // index_copy_ would raise (avid.py:124): flag it, touch nothing
    if (err && lane == 0) atomicOr(err, AVID_DEVERR_UPDATE_INDEX);
    return;
  }
  int dup = 0;
  for (int k = i + 1 + lane; k < B; k += 64) dup |= (y[k] == row);
  if (__any(dup)) return;  // a later sample owns this row
  float* p = bank + row * D;
  float ss = 0.f;
  float v[8];
  int cnt = 0;
  for (int d = lane; d < D && cnt < 8; d += 64, ++cnt) {
    const float t = __fadd_rn(__fmul_rn(p[d], mom), __fmul_rn(emb[(long long)i * D + d], 1.f - mom));
    v[cnt] = t;
    ss += t * t;
  }
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  cnt = 0;
  for (int d = lane; d < D && cnt < 8; d += 64, ++cnt) p[d] = v[cnt] / nrm;
}

// criterions/avid_cma.py:196-209
__global__ void cma_negatives_kernel(const int32_t* __restrict__ pset, const long long* __restrict__ y,
                                     const long long* __restrict__ rnd, long long* __restrict__ pos_out,
                                     long long* __restrict__ neg_out, int bs, int K, int P, long long N,
                                     int* __restrict__ err) {
  const long long n = (long long)bs * K;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int b = (int)(i / K), k = (int)(i % K);
    long long yb = y[b];
    if (yb < 0 || yb >= N) {   // positive_set[y] would raise (avid_cma.py:199): flag it, stay inside the table
      if (err && k == 0) atomicOr(err, AVID_DEVERR_CMA_INDEX);
      yb = yb < 0 ? 0 : N - 1;
    }
    const int32_t* ps = pset + yb * (long long)P;
    const long long r = rnd[i];
    int cnt = 0;
    for (int j = 0; j < P; ++j) cnt += (r >= (long long)ps[j] - j) ? 1 : 0;
    neg_out[i] = r + cnt;
    if (k < P) pos_out[(long long)b * P + k] = ps[k];
  }
  // K < P is legal: finish the positive copy
  if (K < P)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)bs * P; i += stride) {
      const int b = (int)(i / P), k = (int)(i % P);
      long long yb = y[b];
      yb = yb < 0 ? 0 : (yb >= N ? N - 1 : yb);
      if (k >= K) pos_out[i] = pset[yb * (long long)P + k];
    }
}

}  // namespace avid

using namespace avid;

extern "C" int avid_l2norm_fwd(int bs, int D, const float* x, float* y, float* norm_out, avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && D > 0 && x && y && norm_out, AVID_E_BADARG, "l2norm_fwd: bad argument");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)ceil_div(bs, 4)), dim3(256), 0, (hipStream_t)stream, x, y,
                     norm_out, bs, D);
  return check_launch("l2norm_fwd");
}

extern "C" int avid_l2norm_bwd(int bs, int D, const float* y, const float* norm, const float* dy, float* dx,
                               avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && D > 0 && y && norm && dy && dx, AVID_E_BADARG, "l2norm_bwd: bad argument");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)ceil_div(bs, 4)), dim3(256), 0, (hipStream_t)stream, y, norm,
                     dy, dx, bs, D);
  return check_launch("l2norm_bwd");
}

extern "C" int avid_counter_add(uint64_t* counter, uint64_t inc, avid_stream_t stream) {
  AVID_REQUIRE(counter, AVID_E_BADARG, "counter_add: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)counter,
                     (unsigned long long)inc);
  return check_launch("counter_add");
}

extern "C" int avid_alias_draw(int64_t n, int64_t K, const float* prob, const int64_t* alias, int uniform,
                               uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const int64_t* y,
                               int64_t per_row, int64_t* out, avid_stream_t stream) {
  AVID_REQUIRE(n > 0 && K > 0 && K < (1ll << 32) && out, AVID_E_BADARG, "alias_draw: bad argument");
  AVID_REQUIRE(uniform || (prob && alias), AVID_E_BADARG, "alias_draw: tables missing");
  AVID_REQUIRE(!y || per_row > 0, AVID_E_BADARG, "alias_draw: per_row must be > 0 with y");
  long long g = ceil_div(n, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(alias_draw_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (long long)n,
                     (long long)K, prob, (const long long*)alias, uniform, seed, offset,
                     (const unsigned long long*)offset_dev, (const long long*)y,
                     (long long)(per_row > 0 ? per_row : 1), (long long*)out);
  return check_launch("alias_draw");
}

extern "C" int avid_bank_scores_fwd(int bs, int R, int D, int64_t N, const int64_t* idx, const float* bank,
                                    const float* emb, float inv_T, float* scores, float* rows_out, int32_t* err,
                                    avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && R > 0 && N > 0 && idx && bank && emb && scores, AVID_E_BADARG, "bank_scores_fwd: bad argument");
  AVID_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512, AVID_E_UNSUPPORTED, "bank_scores_fwd: D=%d unsupported", D);
  dim3 grid((unsigned)ceil_div(R, SC_ROWS_PER_BLOCK), (unsigned)bs);
  hipStream_t s = (hipStream_t)stream;
  const long long* ix = (const long long*)idx;
  ScopedTimer t(s, "bank_scores_fwd_kernel", 2.0 * bs * R * D, 4.0 * bs * R * ((double)D * (rows_out ? 2 : 1) + 3));
  switch (D / 64) {
    case 1: hipLaunchKernelGGL(bank_scores_fwd_kernel<1>, grid, dim3(256), 0, s, ix, bank, emb, inv_T, scores, rows_out, R, (long long)N, err); break;
    case 2: hipLaunchKernelGGL(bank_scores_fwd_kernel<2>, grid, dim3(256), 0, s, ix, bank, emb, inv_T, scores, rows_out, R, (long long)N, err); break;
    case 4: hipLaunchKernelGGL(bank_scores_fwd_kernel<4>, grid, dim3(256), 0, s, ix, bank, emb, inv_T, scores, rows_out, R, (long long)N, err); break;
    default: hipLaunchKernelGGL(bank_scores_fwd_kernel<8>, grid, dim3(256), 0, s, ix, bank, emb, inv_T, scores, rows_out, R, (long long)N, err); break;
  }
  return check_launch("bank_scores_fwd");
}

extern "C" int avid_bank_scores_bwd(int bs, int R, int D, int64_t N, const float* rows, const int64_t* idx,
                                    const float* bank, const float* dscores, float inv_T, int accumulate,
                                    float* demb, avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && R > 0 && dscores && demb && (rows || (N > 0 && idx && bank)), AVID_E_BADARG,
               "bank_scores_bwd: bad argument");
  AVID_REQUIRE(D == 64 || D == 128 || D == 256 || D == 512, AVID_E_UNSUPPORTED, "bank_scores_bwd: D=%d unsupported", D);
  hipStream_t s = (hipStream_t)stream;
  const long long* ix = (const long long*)idx;
  ScopedTimer t(s, "bank_scores_bwd_kernel", 2.0 * bs * R * D, 4.0 * bs * R * ((double)D + 1));
  switch (D / 64) {
    case 1: hipLaunchKernelGGL(bank_scores_bwd_kernel<1>, dim3(bs), dim3(1024), 0, s, rows, ix, bank, dscores, inv_T, accumulate, demb, R, (long long)N); break;
    case 2: hipLaunchKernelGGL(bank_scores_bwd_kernel<2>, dim3(bs), dim3(1024), 0, s, rows, ix, bank, dscores, inv_T, accumulate, demb, R, (long long)N); break;
    case 4: hipLaunchKernelGGL(bank_scores_bwd_kernel<4>, dim3(bs), dim3(1024), 0, s, rows, ix, bank, dscores, inv_T, accumulate, demb, R, (long long)N); break;
    default: hipLaunchKernelGGL(bank_scores_bwd_kernel<8>, dim3(bs), dim3(1024), 0, s, rows, ix, bank, dscores, inv_T, accumulate, demb, R, (long long)N); break;
  }
  return check_launch("bank_scores_bwd");
}

extern "C" int avid_mean_exp(int rows, int cols, int ld, const float* s, float* out, avid_stream_t stream) {
  AVID_REQUIRE(rows > 0 && cols > 0 && ld >= cols && s && out, AVID_E_BADARG, "mean_exp: bad argument");
  hipLaunchKernelGGL(mean_exp_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, s, rows, cols, ld, out);
  return check_launch("mean_exp");
}

extern "C" size_t avid_nce_workspace_bytes(void) { return 8 * 64 + 64; }

extern "C" int avid_nce_fwd(int bs, int P, int K, const float* spos, int ld_pos, const float* sneg, int ld_neg,
                            const float* Z, float scale, int accumulate, float* loss, void* ws, size_t ws_bytes,
                            avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && P > 0 && K > 0 && spos && sneg && Z && loss && ld_pos >= P && ld_neg >= K, AVID_E_BADARG,
               "nce_fwd: bad argument");
  // ws (zero-filled once by the caller, then owned by this op): [64] fp64 partials + the ticket counter
  const bool multi = ws && ws_bytes >= avid_nce_workspace_bytes() && (long long)bs * K >= 16 * 1024;
  double* partial = static_cast<double*>(ws);
  unsigned* ticket = multi ? reinterpret_cast<unsigned*>(partial + 64) : nullptr;
  hipLaunchKernelGGL(nce_fwd_kernel, dim3(multi ? 16 : 1), dim3(1024), 0, (hipStream_t)stream, spos, sneg, Z, bs, P, K,
                     ld_pos, ld_neg, scale, accumulate, loss, partial, ticket);
  return check_launch("nce_fwd");
}

extern "C" int avid_nce_bwd(int bs, int P, int K, const float* spos, int ld_pos, const float* sneg, int ld_neg,
                            const float* Z, const float* dloss, float scale, float* dpos, float* dneg,
                            avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && P > 0 && K > 0 && spos && sneg && Z && dloss && dpos && dneg && ld_pos >= P && ld_neg >= K,
               AVID_E_BADARG, "nce_bwd: bad argument");
  const long long n = (long long)bs * (K + P);
  long long g = ceil_div(n, 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(nce_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, spos, sneg, Z, dloss, bs, P,
                     K, ld_pos, ld_neg, scale, dpos, dneg);
  return check_launch("nce_bwd");
}

extern "C" int avid_bank_update(int B, int D, int64_t N, float* bank, const int64_t* y, const float* emb,
                                float momentum, int32_t* err, avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && D > 0 && D <= 512 && N > 0 && bank && y && emb, AVID_E_BADARG, "bank_update: bad argument");
  hipLaunchKernelGGL(bank_update_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, bank,
                     (const long long*)y, emb, momentum, B, D, (long long)N, err);
  return check_launch("bank_update");
}

extern "C" int avid_cma_negatives(int bs, int K, int P, int64_t N, const int32_t* positive_set, const int64_t* y,
                                  const int64_t* rand_idx, int64_t* pos_out, int64_t* neg_out, int32_t* err,
                                  avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && K > 0 && P > 0 && N > 0 && positive_set && y && rand_idx && pos_out && neg_out, AVID_E_BADARG,
               "cma_negatives: bad argument");
  const long long n = (long long)bs * (K > P ? K : P);
  long long g = ceil_div(n, 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(cma_negatives_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, positive_set,
                     (const long long*)y, (const long long*)rand_idx, (long long*)pos_out, (long long*)neg_out, bs, K,
                     P, (long long)N, err);
  return check_launch("cma_negatives");
}
