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
constexpr int XM_ROWS_DEFAULT = 64;      // xmodal_fused_kernel / cma_fused_kernel: rows per block, non-temporal row loads
constexpr int XM_NT_DEFAULT = 0;

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
__device__ __forceinline__ void bank_update_rows(float* __restrict__ bank, const long long* __restrict__ y,
                                                 const float* __restrict__ emb, float mom, int B, int D, long long N,
                                                 int* __restrict__ err) {
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

__global__ __launch_bounds__(256) void bank_update_kernel(float* __restrict__ bank, const long long* __restrict__ y,
                                                          const float* __restrict__ emb, float mom, int B, int D,
                                                          long long N, int* __restrict__ err) {
  bank_update_rows(bank, y, emb, mom, B, D, N, err);
}

// ---------------------------------------------------------------------------------------------
// Fused cross-modal criterion (criterions/avid.py:52-71 + criterions/nce.py:38-58 with the partition constant already
// frozen, nce.py:22-24): normalise both embeddings, gather the SAME rows of both banks, scores / T, the NCE terms —
// and, because Z is a constant after the first batch, the gradient with respect to the embeddings in the same pass,
// while a gathered row is still in registers:  d emb_hat = sum_j (dL / ds_j) row_j / T,  then the backward of
// F.normalize.  No score tensors, no snapshot of the gathered rows (the unfused path keeps one because the banks are
// updated before backward runs, criterions/avid.py:78), no second gather.
//   grid (splits of the K + 1 rows, sample); block = 4 waves x 16 rows (x 2 banks) in flight, as bank_scores_fwd_kernel.
//   Every block leaves its partial gradients / loss terms in ws; xmodal_finish_kernel (second launch, one block per
//   sample) folds them in split order and applies the normalisation's backward; the last of its blocks folds the
//   per-sample losses in sample order: fixed summation order, one re-armed ticket (the scratch is zeroed once by the
//   caller).  (A first version folded inside the first kernel behind two device-scope fences per block and per-sample
//   tickets: 129 us for the 1088 blocks of a step instead of ~25.)
// ---------------------------------------------------------------------------------------------
struct XModalArgs {
  const float* v_emb; const float* a_emb;          // raw embeddings [bs][128]
  const long long* y; const long long* idx;        // [bs], [bs][K]
  const float* bank_v; const float* bank_a;        // view1_mem (video), view2_mem (audio)
  const float* Z;
  float* v_hat; float* a_hat;                      // normalised embeddings [bs][128]
  float* losses;                                   // [4]: L_v2a, L_a2v, (L_v2a + L_a2v) / 2, coeff * that
  float* dv; float* da;                            // d(total) / d(raw embedding) [bs][128]
  float* part_g;                                   // [bs][S][2][128]
  double* part_l;                                  // [bs][S][2]
  double* samp_l;                                  // [bs][2]
  unsigned* tickets;                               // [bs + 1]
  float* norms;                                    // [2][bs]: |v_emb|, |a_emb| (clamped as F.normalize does)
  int* err;
  long long N;
  int bs, K, S;
  float inv_T, coeff;
  // CMA form (criterions/avid_cma.py:150-194 with cross-modal instance + within-modal positive terms): rows = self | P
  // positives (pos[b][.]) | K negatives; the first Kw negatives also serve the within-modal terms; cI / cP: the two
  // groups' normalised coefficients (folded into the gradients as they are accumulated: coeff = 1 then)
  const long long* pos;
  int P, Kw;
  float cI, cP;
};

// Lane layout (round 4): SIXTEEN lanes per bank row (8 consecutive floats each: two 16-byte loads per row and bank), four
// rows per wave at a time — a row's dot product is a 4-step reduction inside its lane group instead of a 6-step one over
// the wave, the exp / log / divide chain of four rows runs in parallel lanes, and a wave issues 64 16-byte loads instead of
// 64 4-byte ones for the same 16 rows (29.3 -> see DESIGN 3.6).
// CMA: four score sets per gathered row pair instead of two — inst-v2a / inst-a2v (self row positive, the K negatives) and
// pos-v2v / pos-a2a (the P positives, each 1 / P of the positive term; the first Kw negatives): the same rows of both
// banks, two more dot products per row, and both embeddings receive gradient from both banks.
// RB rows (of both banks) per block: 64 = 16 per wave in flight (round 4), 128 = 32 per wave — the gather is a chain of two
// dependent memory round trips (indices, rows) per block whatever its size, and what bounds it on the 2 M-row banks is how
// many row requests the memory system holds at once (round 5).  NT: the rows with non-temporal loads (read
// once per step; at 2 M rows nothing of them survives in L2 / MALL until the next step anyway).
template <bool CMA, int RB, bool NT>
__global__ __launch_bounds__(256) void xmodal_fused_kernel(const XModalArgs p) {
  constexpr int D = 128, RPW = RB / 4, NU = RPW / 4, NL = CMA ? 4 : 2;
  constexpr int SC_ROWS_PER_BLOCK = RB;
  __shared__ float sh_g[4][2][D];
  __shared__ double sh_l[4][NL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, grp = lane >> 4;
  const int b = blockIdx.y, split = blockIdx.x, R = (CMA ? p.P : 0) + p.K + 1;
  auto group_sum = [](float v) {                       // over the 16 lanes of a row group
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
  };
  // normalised embeddings: this lane's 8 components (every lane group computes the norms)
  float ev[8], ea[8], nv, na;
  {
    const floatx4 v0 = *reinterpret_cast<const floatx4*>(p.v_emb + (long long)b * D + sub * 8);
    const floatx4 v1 = *reinterpret_cast<const floatx4*>(p.v_emb + (long long)b * D + sub * 8 + 4);
    const floatx4 a0 = *reinterpret_cast<const floatx4*>(p.a_emb + (long long)b * D + sub * 8);
    const floatx4 a1 = *reinterpret_cast<const floatx4*>(p.a_emb + (long long)b * D + sub * 8 + 4);
    float sv = 0.f, sa = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ev[k] = v0[k]; ev[4 + k] = v1[k]; ea[k] = a0[k]; ea[4 + k] = a1[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { sv = fmaf(ev[k], ev[k], sv); sa = fmaf(ea[k], ea[k], sa); }
    nv = fmaxf(sqrtf(group_sum(sv)), 1e-12f);
    na = fmaxf(sqrtf(group_sum(sa)), 1e-12f);
#pragma unroll
    for (int k = 0; k < 8; ++k) { ev[k] /= nv; ea[k] /= na; }
    if (split == 0 && wave == 0 && grp == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        p.v_hat[(long long)b * D + sub * 8 + k] = ev[k];
        p.a_hat[(long long)b * D + sub * 8 + k] = ea[k];
      }
      if (sub == 0) { p.norms[b] = nv; p.norms[p.bs + b] = na; }
    }
  }
  const float KZ = (float)p.K * p.Z[0];
  const float KZw = CMA ? (float)p.Kw * p.Z[0] : 0.f;
  const int PP = CMA ? p.P : 0;
  // Cross-modal form: the blocks split the K NEGATIVES (rows 1 .. K) and the sample's own row 0 rides as an extra unit in
  // wave 0 of split 0 — K + 1 = 1025 rows cut into blocks of 64 were 17 blocks per sample, the 17th with one row: 1088 blocks
  // for the 1024 that are resident at once (108 registers: four waves per SIMD), i.e. a second round for 64 one-row blocks
  // behind the first one's whole latency chain (round 5).
  const int j0 = (CMA ? 0 : 1) + split * SC_ROWS_PER_BLOCK, j1 = min(j0 + SC_ROWS_PER_BLOCK, R);
  const int jw = j0 + wave * RPW;
  const bool extra = !CMA && split == 0 && wave == 0;      // (wave-uniform)
  // row j = 0 is the positive (the sample's own row y), [CMA: rows 1..P the positives pos[b][j - 1],] then the negatives
  long long mine = 0;
  if ((lane < RPW && jw + lane < j1) || (extra && lane == RPW)) {
    const int jj = lane == RPW ? 0 : jw + lane;
    mine = jj == 0 ? p.y[b] : (jj <= PP ? p.pos[(long long)b * PP + jj - 1] : p.idx[(long long)b * p.K + jj - 1 - PP]);
    if (mine < 0 || mine >= p.N) {
      if (p.err) atomicOr(p.err, AVID_DEVERR_BANK_INDEX);
      mine = mine < 0 ? 0 : p.N - 1;
    }
  }
  floatx4 xa[2] = {}, xv[2] = {};        // the extra unit's rows (row 0 of the sample)
  if (extra) {
    const long long row = __shfl(mine, RPW, 64);
    const float* pa = p.bank_a + row * D + sub * 8;
    const float* pv = p.bank_v + row * D + sub * 8;
    xa[0] = *reinterpret_cast<const floatx4*>(pa); xa[1] = *reinterpret_cast<const floatx4*>(pa + 4);
    xv[0] = *reinterpret_cast<const floatx4*>(pv); xv[1] = *reinterpret_cast<const floatx4*>(pv + 4);
  }
  floatx4 ra[NU][2], rv[NU][2];          // audio-bank rows (scored against the video embedding) and video-bank rows
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const long long row = __shfl(mine, u * 4 + grp, 64);
    const float* pa = p.bank_a + row * D + sub * 8;
    const float* pv = p.bank_v + row * D + sub * 8;
    if (NT) {
      ra[u][0] = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(pa)); ra[u][1] = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(pa + 4));
      rv[u][0] = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(pv)); rv[u][1] = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(pv + 4));
    } else {
      ra[u][0] = *reinterpret_cast<const floatx4*>(pa); ra[u][1] = *reinterpret_cast<const floatx4*>(pa + 4);
      rv[u][0] = *reinterpret_cast<const floatx4*>(pv); rv[u][1] = *reinterpret_cast<const floatx4*>(pv + 4);
    }
  }
  float gv[8], ga[8];                    // d L_v2a / d v_hat, d L_a2v / d a_hat (x T, unscaled): this group's rows
#pragma unroll
  for (int k = 0; k < 8; ++k) gv[k] = ga[k] = 0.f;
  double lv = 0, la = 0, lw = 0, lx = 0;
  const float invP = CMA ? 1.f / (float)p.P : 0.f;
  if (!CMA) {
    // Cross-modal form: the exp / log / divide chain of a row runs ONCE — lane `sub` = u of a row group takes unit u's two
    // scores, so the NU units' chains run side by side in NU lanes of every group instead of one after the other in all 16
    // (round 5: the chain was 3/4 of the kernel's instructions; the kernel is bound by them, not by the gather — consecutive
    // rows instead of random ones take the same time, tools/xmodal_bench.py XB_SEQ=1).
    float ms1 = 0.f, ms2 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d1 = fmaf(ra[u][0][k], ev[k], d1); d1 = fmaf(ra[u][1][k], ev[4 + k], d1);
        d2 = fmaf(rv[u][0][k], ea[k], d2); d2 = fmaf(rv[u][1][k], ea[4 + k], d2);
      }
      const float s1 = group_sum(d1) * p.inv_T, s2 = group_sum(d2) * p.inv_T;
      if (sub == u) { ms1 = s1; ms2 = s2; }
    }
    bool xlane = false;                          // this lane carries the extra unit (row 0): lane sub = NU of group 0
    if (extra) {
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d1 = fmaf(xa[0][k], ev[k], d1); d1 = fmaf(xa[1][k], ev[4 + k], d1);
        d2 = fmaf(xv[0][k], ea[k], d2); d2 = fmaf(xv[1][k], ea[4 + k], d2);
      }
      const float s1 = group_sum(d1) * p.inv_T, s2 = group_sum(d2) * p.inv_T;
      xlane = sub == NU && grp == 0;
      if (xlane) { ms1 = s1; ms2 = s2; }
    }
    float g1m = 0.f, g2m = 0.f;
    {
      const int j = xlane ? 0 : jw + sub * 4 + grp;          // the row of unit `sub` of this group
      if (xlane || (sub < NU && j < j1)) {
        const float e1 = expf(ms1), e2 = expf(ms2);
        if (j == 0) {            // -log(e / (e + KZ));  d / ds = -KZ / (e + KZ)
          lv += (double)(-logf(e1 / (e1 + KZ))); la += (double)(-logf(e2 / (e2 + KZ)));
          g1m = -(KZ / (e1 + KZ)); g2m = -(KZ / (e2 + KZ));
        } else {                 // -log(KZ / (e + KZ));  d / ds = e / (e + KZ)
          lv += (double)(-logf(KZ / (e1 + KZ))); la += (double)(-logf(KZ / (e2 + KZ)));
          g1m = e1 / (e1 + KZ); g2m = e2 / (e2 + KZ);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const float g1 = __shfl(g1m, (lane & 48) | u, 64), g2 = __shfl(g2m, (lane & 48) | u, 64);   // 0 for rows past the end
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gv[k] = fmaf(g1, ra[u][0][k], gv[k]); gv[4 + k] = fmaf(g1, ra[u][1][k], gv[4 + k]);
        ga[k] = fmaf(g2, rv[u][0][k], ga[k]); ga[4 + k] = fmaf(g2, rv[u][1][k], ga[4 + k]);
      }
    }
    if (extra) {
      const float gx1 = __shfl(g1m, NU, 64), gx2 = __shfl(g2m, NU, 64);
      const float g1 = grp == 0 ? gx1 : 0.f, g2 = grp == 0 ? gx2 : 0.f;      // (every group loaded the row: count it once)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gv[k] = fmaf(g1, xa[0][k], gv[k]); gv[4 + k] = fmaf(g1, xa[1][k], gv[4 + k]);
        ga[k] = fmaf(g2, xv[0][k], ga[k]); ga[4 + k] = fmaf(g2, xv[1][k], ga[4 + k]);
      }
    }
    // the units' loss terms sit in lanes sub = 0 .. NU of a group: into lane sub = 0
#pragma unroll
    for (int o = 1; o < 2 * NU; o <<= 1) { lv += __shfl_xor(lv, o, 64); la += __shfl_xor(la, o, 64); }
  }
#pragma unroll
  for (int u = 0; CMA && u < NU; ++u) {
    const int j = jw + u * 4 + grp;
    float d1 = 0.f, d2 = 0.f, d3 = 0.f, d4 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d1 = fmaf(ra[u][0][k], ev[k], d1); d1 = fmaf(ra[u][1][k], ev[4 + k], d1);
      d2 = fmaf(rv[u][0][k], ea[k], d2); d2 = fmaf(rv[u][1][k], ea[4 + k], d2);
      if (CMA) {
        d3 = fmaf(rv[u][0][k], ev[k], d3); d3 = fmaf(rv[u][1][k], ev[4 + k], d3);
        d4 = fmaf(ra[u][0][k], ea[k], d4); d4 = fmaf(ra[u][1][k], ea[4 + k], d4);
      }
    }
    const float s1 = group_sum(d1) * p.inv_T;     // v2a: video embedding . audio bank
    const float s2 = group_sum(d2) * p.inv_T;     // a2v
    if (CMA) {
      const float s3 = group_sum(d3) * p.inv_T;   // v2v: video embedding . video bank
      const float s4 = group_sum(d4) * p.inv_T;   // a2a
      if (j < j1) {
        const int n = j - 1 - PP;                 // index among the negatives (< 0: self / positive)
        float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
        if (j == 0 || n >= 0) {                   // instance terms: self positive, every negative
          const float e1 = expf(s1), e2 = expf(s2);
          if (j == 0) {
            if (sub == 0) { lv += (double)(-logf(e1 / (e1 + KZ))); la += (double)(-logf(e2 / (e2 + KZ))); }
            g1 = -(KZ / (e1 + KZ)); g2 = -(KZ / (e2 + KZ));
          } else {
            if (sub == 0) { lv += (double)(-logf(KZ / (e1 + KZ))); la += (double)(-logf(KZ / (e2 + KZ))); }
            g1 = e1 / (e1 + KZ); g2 = e2 / (e2 + KZ);
          }
        }
        if (j != 0 && n < p.Kw) {                 // within-modal terms: the positives (mean over P), the first Kw negatives
          const float e3 = expf(s3), e4 = expf(s4);
          if (n < 0) {
            if (sub == 0) { lw += (double)(-logf(e3 / (e3 + KZw)) * invP); lx += (double)(-logf(e4 / (e4 + KZw)) * invP); }
            g3 = -(KZw / (e3 + KZw)) * invP; g4 = -(KZw / (e4 + KZw)) * invP;
          } else {
            if (sub == 0) { lw += (double)(-logf(KZw / (e3 + KZw))); lx += (double)(-logf(KZw / (e4 + KZw))); }
            g3 = e3 / (e3 + KZw); g4 = e4 / (e4 + KZw);
          }
        }
        g1 *= p.cI; g2 *= p.cI; g3 *= p.cP; g4 *= p.cP;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          gv[k] = fmaf(g1, ra[u][0][k], gv[k]); gv[4 + k] = fmaf(g1, ra[u][1][k], gv[4 + k]);
          gv[k] = fmaf(g3, rv[u][0][k], gv[k]); gv[4 + k] = fmaf(g3, rv[u][1][k], gv[4 + k]);
          ga[k] = fmaf(g2, rv[u][0][k], ga[k]); ga[4 + k] = fmaf(g2, rv[u][1][k], ga[4 + k]);
          ga[k] = fmaf(g4, ra[u][0][k], ga[k]); ga[4 + k] = fmaf(g4, ra[u][1][k], ga[4 + k]);
        }
      }
    } else if (j < j1) {
      const float e1 = expf(s1), e2 = expf(s2);
      float g1, g2;
      if (j == 0) {            // -log(e / (e + KZ));  d / ds = -KZ / (e + KZ)
        if (sub == 0) { lv += (double)(-logf(e1 / (e1 + KZ))); la += (double)(-logf(e2 / (e2 + KZ))); }
        g1 = -(KZ / (e1 + KZ)); g2 = -(KZ / (e2 + KZ));
      } else {                 // -log(KZ / (e + KZ));  d / ds = e / (e + KZ)
        if (sub == 0) { lv += (double)(-logf(KZ / (e1 + KZ))); la += (double)(-logf(KZ / (e2 + KZ))); }
        g1 = e1 / (e1 + KZ); g2 = e2 / (e2 + KZ);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gv[k] = fmaf(g1, ra[u][0][k], gv[k]); gv[4 + k] = fmaf(g1, ra[u][1][k], gv[4 + k]);
        ga[k] = fmaf(g2, rv[u][0][k], ga[k]); ga[4 + k] = fmaf(g2, rv[u][1][k], ga[4 + k]);
      }
    }
  }
  // fold the four row groups of the wave (fixed order: (g0 + g1) + (g2 + g3)), then the four waves through LDS
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    gv[k] += __shfl_xor(gv[k], 16, 64); gv[k] += __shfl_xor(gv[k], 32, 64);
    ga[k] += __shfl_xor(ga[k], 16, 64); ga[k] += __shfl_xor(ga[k], 32, 64);
  }
  lv += __shfl_xor(lv, 16, 64); lv += __shfl_xor(lv, 32, 64);
  la += __shfl_xor(la, 16, 64); la += __shfl_xor(la, 32, 64);
  if (CMA) {
    lw += __shfl_xor(lw, 16, 64); lw += __shfl_xor(lw, 32, 64);
    lx += __shfl_xor(lx, 16, 64); lx += __shfl_xor(lx, 32, 64);
  }
  if (grp == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { sh_g[wave][0][sub * 8 + k] = gv[k]; sh_g[wave][1][sub * 8 + k] = ga[k]; }
  }
  if (lane == 0) {
    sh_l[wave][0] = lv; sh_l[wave][1] = la;
    if (CMA) { sh_l[wave][2] = lw; sh_l[wave][3] = lx; }
  }
  __syncthreads();
  float* pg = p.part_g + ((long long)b * p.S + split) * 2 * D;
  {
    const int t = threadIdx.x;                       // 256 threads = 2 x 128 gradient components
    pg[t] = ((&sh_g[0][0][0])[t] + (&sh_g[1][0][0])[t]) + ((&sh_g[2][0][0])[t] + (&sh_g[3][0][0])[t]);   // [which][d] contiguous
  }
  if (threadIdx.x < NL) {
    double* pl = p.part_l + ((long long)b * p.S + split) * NL;
    const int q = threadIdx.x;
    pl[q] = (sh_l[0][q] + sh_l[1][q]) + (sh_l[2][q] + sh_l[3][q]);
  }
}

// second launch of the fused criterion: one block per sample folds the S partial gradients in split order (the kernel
// boundary makes them visible: no fences around the 1088 producer blocks), applies the backward of F.normalize and
// leaves the sample's two loss terms; the last of the bs blocks (device ticket, re-armed) folds those in sample order.
template <bool CMA>
__global__ __launch_bounds__(256) void xmodal_finish_kernel(const XModalArgs p) {
  constexpr int D = 128, NL = CMA ? 4 : 2;
  __shared__ float sh_dot[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, which = t >> 7, d = t & 127;
  const int b = blockIdx.x;
  const float eh = (which == 0 ? p.v_hat : p.a_hat)[(long long)b * D + d];      // as the main kernel normalised it
  const float nrm = p.norms[which * p.bs + b];
  const float* pg = p.part_g + (long long)b * p.S * 2 * D + t;
  float g = 0.f;
  int sidx = 0;
  for (; sidx + 4 <= p.S; sidx += 4) {               // four independent loads per trip, summed in split order
    const float g0 = pg[(long long)sidx * 2 * D], g1 = pg[(long long)(sidx + 1) * 2 * D];
    const float g2 = pg[(long long)(sidx + 2) * 2 * D], g3 = pg[(long long)(sidx + 3) * 2 * D];
    g += g0; g += g1; g += g2; g += g3;
  }
  for (; sidx < p.S; ++sidx) g += pg[(long long)sidx * 2 * D];
  // d total / d score = coeff * 1/2 * 1/bs * dL/ds;  d emb_hat = that * row / T
  g *= p.coeff * 0.5f / (float)p.bs * p.inv_T;
  // dx = (dy - y <y, dy>) / norm
  const float dot = wave_sum(g * eh);
  if (lane == 0) sh_dot[wave] = dot;
  __syncthreads();
  const float full = sh_dot[which * 2] + sh_dot[which * 2 + 1];
  (which == 0 ? p.dv : p.da)[(long long)b * D + d] = (g - eh * full) / nrm;
  if (wave != 0) return;
  // the sample's loss terms: lanes 0..S-1 fetch the partials, fixed-order sum by lane 0
  double l[NL];
  {
    const double* pl = p.part_l + (long long)b * p.S * NL;
#pragma unroll
    for (int q = 0; q < NL; ++q) l[q] = 0;
    for (int base = 0; base < p.S; base += 64) {
      const int i = base + lane;
      double v[NL];
#pragma unroll
      for (int q = 0; q < NL; ++q) v[q] = i < p.S ? pl[NL * i + q] : 0.0;
      for (int k = 0; k < 64 && base + k < p.S; ++k) {
#pragma unroll
        for (int q = 0; q < NL; ++q) l[q] += __shfl(v[q], k, 64);
      }
    }
  }
  unsigned last = 0;
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < NL; ++q) __hip_atomic_store(p.samp_l + NL * b + q, l[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(&p.tickets[0], 1u) == (unsigned)p.bs - 1 ? 1u : 0u;
  }
  last = __shfl(last, 0, 64);
  if (!last) return;
  __threadfence();
  double tot[NL];
#pragma unroll
  for (int q = 0; q < NL; ++q) tot[q] = 0;
  for (int base = 0; base < p.bs; base += 64) {      // 64 samples per trip in flight, summed in sample order
    const int i = base + lane;
    double v[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) v[q] = i < p.bs ? __hip_atomic_load(p.samp_l + NL * i + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    for (int k = 0; k < 64 && base + k < p.bs; ++k) {
#pragma unroll
      for (int q = 0; q < NL; ++q) tot[q] += __shfl(v[q], k, 64);
    }
  }
  if (lane == 0) {
    float L[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) { L[q] = (float)(tot[q] / (double)p.bs); p.losses[q] = L[q]; }
    if (CMA) {     // criterions/avid_cma.py:338-358: every group the mean of its two directions, the total their weighted sum
      const float gi = L[0] / 2.f + L[1] / 2.f, gp = L[2] / 2.f + L[3] / 2.f;
      p.losses[4] = gi;
      p.losses[5] = gp;
      p.losses[6] = gi * p.cI + gp * p.cP;
    } else {
      const float xl = L[0] / 2.f + L[1] / 2.f;      // criterions/avid.py:221-222
      p.losses[2] = xl;
      p.losses[3] = xl * p.coeff;
    }
    __hip_atomic_store(&p.tickets[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-arm
  }
}

// both banks in one launch (criterions/avid.py:118-129): blockIdx.y picks the bank
__global__ __launch_bounds__(256) void bank_update2_kernel(float* __restrict__ bank0, float* __restrict__ bank1,
                                                           const long long* __restrict__ y, const float* __restrict__ emb0,
                                                           const float* __restrict__ emb1, float mom0, float mom1, int B,
                                                           int D, long long N, int* __restrict__ err) {
  if (blockIdx.y == 0)
    bank_update_rows(bank0, y, emb0, mom0, B, D, N, err);
  else
    bank_update_rows(bank1, y, emb1, mom1, B, D, N, err);
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

// rows per block / non-temporal row loads of the fused criterion kernels: AVID_XM_ROWS (64 | 128), AVID_XM_NT (0 | 1)
static int xm_rows() {
  static int v = 0;
  if (!v) {
    const char* e = getenv("AVID_XM_ROWS");
    v = e && atoi(e) == 64 ? 64 : (e && atoi(e) == 128 ? 128 : XM_ROWS_DEFAULT);
  }
  return v;
}
static bool xm_nt() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AVID_XM_NT");
    v = e ? (atoi(e) != 0) : XM_NT_DEFAULT;
  }
  return v != 0;
}
static size_t fused_ws_bytes(int bs, int rows, int nl) {
  const size_t S = (size_t)ceil_div(rows, 64);      // (sized for the smaller block: either setting fits)
  return (size_t)bs * S * 2 * 128 * 4 + (size_t)bs * S * nl * 8 + (size_t)bs * nl * 8 + ((size_t)bs + 1) * 4 + (size_t)bs * 2 * 4 + 64;
}
extern "C" size_t avid_xmodal_fused_workspace_bytes(int bs, int K) { return fused_ws_bytes(bs, K + 1, 2); }
extern "C" size_t avid_cma_fused_workspace_bytes(int bs, int P, int K) { return fused_ws_bytes(bs, 1 + P + K, 4); }

// scratch layout + the two launches shared by avid_xmodal_fused / avid_cma_fused
template <bool CMA>
static int fused_launch(XModalArgs& a, int rows, void* ws, hipStream_t s) {
  constexpr int NL = CMA ? 4 : 2;
  const int rb = xm_rows();
  a.S = (int)ceil_div(CMA ? rows : rows - 1, rb);       // cross-modal: the negatives in blocks, row 0 rides in the first
  char* w = static_cast<char*>(ws);                 // (zero-filled once by the caller: the tickets re-arm themselves)
  a.part_l = reinterpret_cast<double*>(w); w += (size_t)a.bs * a.S * NL * 8;
  a.samp_l = reinterpret_cast<double*>(w); w += (size_t)a.bs * NL * 8;
  a.part_g = reinterpret_cast<float*>(w); w += (size_t)a.bs * a.S * 2 * 128 * 4;
  a.tickets = reinterpret_cast<unsigned*>(w); w += ((size_t)a.bs + 1) * 4;
  a.norms = reinterpret_cast<float*>(w);
  // bytes: the gathered rows of both banks, read once
  ScopedTimer t(s, CMA ? "cma_fused_kernel" : "xmodal_fused_kernel", (CMA ? 2.0 : 1.0) * 2.0 * 2.0 * 2.0 * a.bs * rows * 128,
                2.0 * 4.0 * a.bs * rows * 128.0);
  const dim3 grid((unsigned)a.S, (unsigned)a.bs);
  if (rb == 128) {
    if (xm_nt()) hipLaunchKernelGGL((xmodal_fused_kernel<CMA, 128, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((xmodal_fused_kernel<CMA, 128, false>), grid, dim3(256), 0, s, a);
  } else {
    if (xm_nt()) hipLaunchKernelGGL((xmodal_fused_kernel<CMA, 64, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((xmodal_fused_kernel<CMA, 64, false>), grid, dim3(256), 0, s, a);
  }
  int rc = check_launch(CMA ? "cma_fused" : "xmodal_fused");
  if (rc) return rc;
  hipLaunchKernelGGL(xmodal_finish_kernel<CMA>, dim3((unsigned)a.bs), dim3(256), 0, s, a);
  return check_launch(CMA ? "cma_finish" : "xmodal_finish");
}

extern "C" int avid_xmodal_fused(int bs, int K, int D, int64_t N, const float* v_emb, const float* a_emb, const int64_t* y,
                                 const int64_t* idx, const float* bank_v, const float* bank_a, float inv_T, const float* Z,
                                 float coeff, float* v_hat, float* a_hat, float* losses, float* dv, float* da, void* ws,
                                 size_t ws_bytes, int32_t* err, avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && K > 0 && N > 0 && v_emb && a_emb && y && idx && bank_v && bank_a && Z && v_hat && a_hat && losses &&
                   dv && da && ws, AVID_E_BADARG, "xmodal_fused: bad argument");
  AVID_REQUIRE(D == 128, AVID_E_UNSUPPORTED, "xmodal_fused: D=%d unsupported (128 only; use the unfused ops)", D);
  AVID_REQUIRE(ws_bytes >= avid_xmodal_fused_workspace_bytes(bs, K), AVID_E_BADARG, "xmodal_fused: workspace too small");
  XModalArgs a;
  a.v_emb = v_emb; a.a_emb = a_emb; a.y = (const long long*)y; a.idx = (const long long*)idx;
  a.bank_v = bank_v; a.bank_a = bank_a; a.Z = Z; a.v_hat = v_hat; a.a_hat = a_hat; a.losses = losses; a.dv = dv; a.da = da;
  a.err = err; a.N = N; a.bs = bs; a.K = K; a.inv_T = inv_T; a.coeff = coeff;
  a.pos = nullptr; a.P = 0; a.Kw = 0; a.cI = a.cP = 0.f;
  return fused_launch<false>(a, K + 1, ws, (hipStream_t)stream);
}

extern "C" int avid_cma_fused(int bs, int P, int K, int Kw, int D, int64_t N, const float* v_emb, const float* a_emb,
                              const int64_t* y, const int64_t* pos, const int64_t* idx, const float* bank_v, const float* bank_a,
                              float inv_T, const float* Z, float coeff_inst, float coeff_pos, float* v_hat, float* a_hat,
                              float* losses, float* dv, float* da, void* ws, size_t ws_bytes, int32_t* err, avid_stream_t stream) {
  AVID_REQUIRE(bs > 0 && P > 0 && K > 0 && Kw > 0 && Kw <= K && N > 0 && v_emb && a_emb && y && pos && idx && bank_v && bank_a && Z &&
                   v_hat && a_hat && losses && dv && da && ws, AVID_E_BADARG, "cma_fused: bad argument");
  AVID_REQUIRE(D == 128, AVID_E_UNSUPPORTED, "cma_fused: D=%d unsupported (128 only; use the unfused ops)", D);
  AVID_REQUIRE(ws_bytes >= avid_cma_fused_workspace_bytes(bs, P, K), AVID_E_BADARG, "cma_fused: workspace too small");
  XModalArgs a;
  a.v_emb = v_emb; a.a_emb = a_emb; a.y = (const long long*)y; a.idx = (const long long*)idx;
  a.bank_v = bank_v; a.bank_a = bank_a; a.Z = Z; a.v_hat = v_hat; a.a_hat = a_hat; a.losses = losses; a.dv = dv; a.da = da;
  a.err = err; a.N = N; a.bs = bs; a.K = K; a.inv_T = inv_T; a.coeff = 1.f;
  a.pos = (const long long*)pos; a.P = P; a.Kw = Kw; a.cI = coeff_inst; a.cP = coeff_pos;
  return fused_launch<true>(a, 1 + P + K, ws, (hipStream_t)stream);
}

extern "C" int avid_bank_update2(int B, int D, int64_t N, float* bank0, float* bank1, const int64_t* y, const float* emb0,
                                 const float* emb1, float momentum0, float momentum1, int32_t* err, avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && N > 0 && bank0 && bank1 && y && emb0 && emb1, AVID_E_BADARG, "bank_update2: bad argument");
  AVID_REQUIRE(D > 0 && D <= 512, AVID_E_UNSUPPORTED, "bank_update2: D=%d unsupported", D);
  hipLaunchKernelGGL(bank_update2_kernel, dim3((unsigned)ceil_div(B, 4), 2), dim3(256), 0, (hipStream_t)stream, bank0, bank1,
                     (const long long*)y, emb0, emb1, momentum0, momentum1, B, D, (long long)N, err);
  return check_launch("bank_update2");
}

