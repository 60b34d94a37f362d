// HBM-bound epilogue kernels over channels-last [M, C] activations: train/eval BatchNorm (+ReLU)
// forward/backward, the stem max-pool, global (adaptive) max-pool, ReLU backward, column sums.
// All accesses are 16-B vectors with consecutive lanes on consecutive channels (fully coalesced).
//
// Reference ops replaced: nn.BatchNorm3d/2d + nn.ReLU (models/network_blocks.py:19-27,36-59,
// models/video.py:21-22, models/audio.py:23-24), nn.MaxPool3d (models/video.py:23),
// nn.AdaptiveMaxPool3d/2d (models/video.py:41, models/audio.py:31).
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace avid {

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics: grid of row-blocks, each thread owns one float4 channel group.
// partial[blk][0][C] = sum x, partial[blk][1][C] = sum x^2   (fp32 partials, fp64 final combine)
// ---------------------------------------------------------------------------------------------
struct BnPlan {
  int G;               // float4 groups per row = C/4
  int rows_per_pass;   // 256 / G
  int rows_per_block;
  int nblk;
};

static BnPlan bn_plan(int64_t M, int C) {
  BnPlan p;
  p.G = C / 4;
  p.rows_per_pass = 256 / p.G;
  int64_t rpb = ceil_div(M, 1024);
  if (rpb < 16) rpb = 16;   // small layers: more, shorter blocks (64-row blocks left conv5x with 16 blocks of 8 dependent iterations)
  rpb = ceil_div(rpb, p.rows_per_pass) * p.rows_per_pass;
  p.rows_per_block = (int)rpb;
  p.nblk = (int)ceil_div(M, rpb);
  return p;
}

__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                               long long M, int C, int G, int rows_per_pass,
                                                               int rows_per_block) {
  __shared__ floatx4 sh[2][256];
  const int tid = threadIdx.x;
  const int g = tid % G, r = tid / G;
  floatx4 s = {0, 0, 0, 0}, ss = {0, 0, 0, 0};
  const long long row0 = (long long)blockIdx.x * rows_per_block;
  if (r < rows_per_pass) {
    // 4 independent row streams per thread: 4 x 16-B loads in flight instead of one dependent chain
    floatx4 s1 = {0, 0, 0, 0}, s2 = s1, s3 = s1, q1 = s1, q2 = s1, q3 = s1;
    const floatx4 z = {0, 0, 0, 0};
    for (int k = r; k < rows_per_block; k += 4 * rows_per_pass) {
      const long long r0 = row0 + k, r1 = r0 + rows_per_pass, r2 = r1 + rows_per_pass, r3 = r2 + rows_per_pass;
      const bool o1 = k + rows_per_pass < rows_per_block, o2 = k + 2 * rows_per_pass < rows_per_block,
                 o3 = k + 3 * rows_per_pass < rows_per_block;
      const floatx4 v0 = r0 < M ? *reinterpret_cast<const floatx4*>(x + r0 * C + g * 4) : z;
      const floatx4 v1 = (o1 && r1 < M) ? *reinterpret_cast<const floatx4*>(x + r1 * C + g * 4) : z;
      const floatx4 v2 = (o2 && r2 < M) ? *reinterpret_cast<const floatx4*>(x + r2 * C + g * 4) : z;
      const floatx4 v3 = (o3 && r3 < M) ? *reinterpret_cast<const floatx4*>(x + r3 * C + g * 4) : z;
      s += v0; ss += v0 * v0;
      s1 += v1; q1 += v1 * v1;
      s2 += v2; q2 += v2 * v2;
      s3 += v3; q3 += v3 * v3;
    }
    s += s1 + (s2 + s3);
    ss += q1 + (q2 + q3);
  }
  sh[0][tid] = s;
  sh[1][tid] = ss;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rows_per_pass; ++k) {
      s += sh[0][k * G + g];
      ss += sh[1][k * G + g];
    }
    float* o = part + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<floatx4*>(o + g * 4) = s;
    *reinterpret_cast<floatx4*>(o + C + g * 4) = ss;
  }
}

// Combine the per-block partials in fp64.  Block = 16 channels (4 float4 groups) x 256 slices of the partial
// list: with up to 1024 partial rows a thread has at most 4 rows and all of its 8 float4 loads are in flight
// at once — the kernel is one memory round trip plus the fold (lane shuffles across the 16 slices that share a
// wave, LDS across the 16 waves).  History: 64 channels x 16 slices with one dependent chain of <= 64 scalar
// loads per thread took 11-13 us per BatchNorm layer (1 ms per training step); 16 x 64 slices with 4 scalar
// streams 6-10 us (rocprofv3; 82 launches per step).
constexpr int FIN_CH = 16, FIN_SLICES = 256;

__device__ __forceinline__ void finalize_sums(const float* __restrict__ part, int nblk, int C, int blk,
                                              double (*sh)[2][FIN_CH], double& s0, double& s1) {
  const int g = threadIdx.x & 3, slice = threadIdx.x >> 2;
  const int c0 = blk * FIN_CH + g * 4;
  double a[4][4], b[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[u][j] = b[u][j] = 0.0;
  if (c0 < C) {
    const long long st = 2ll * C;
    for (int k = slice; k < nblk; k += 4 * FIN_SLICES) {
      floatx4 x[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = k + u * FIN_SLICES;
        const floatx4 z = {0.f, 0.f, 0.f, 0.f};
        const float* q = part + (long long)(row < nblk ? row : k) * st + c0;
        x[u] = row < nblk ? *reinterpret_cast<const floatx4*>(q) : z;
        y[u] = row < nblk ? *reinterpret_cast<const floatx4*>(q + C) : z;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[u][j] += (double)x[u][j]; b[u][j] += (double)y[u][j]; }
    }
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double sa = (a[0][j] + a[1][j]) + (a[2][j] + a[3][j]), sb = (b[0][j] + b[1][j]) + (b[2][j] + b[3][j]);
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { sa += __shfl_xor(sa, m, 64); sb += __shfl_xor(sb, m, 64); }
    if ((threadIdx.x & 63) < 4) {
      sh[wave][0][g * 4 + j] = sa;
      sh[wave][1][g * 4 + j] = sb;
    }
  }
  __syncthreads();
  s0 = 0;
  s1 = 0;
  if (threadIdx.x < FIN_CH)
    for (int k = 0; k < 16; ++k) {
      s0 += sh[k][0][threadIdx.x];
      s1 += sh[k][1][threadIdx.x];
    }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nblk, long long M, int C,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float momentum, float eps,
                                                           float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, float* __restrict__ scale,
                                                           float* __restrict__ shift,
                                                           long long* __restrict__ num_batches_tracked) {
  __shared__ double sh[16][2][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  double s, ss;
  finalize_sums(part, nblk, C, blockIdx.x, sh, s, ss);
  if (threadIdx.x >= FIN_CH || c >= C) return;
  const double mean = s / (double)M;
  double var = ss / (double)M - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (running_mean) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd_out,
                                     float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  mean[c] = rm[c];
  invstd_out[c] = invstd;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// The grid stride is a multiple of G (= C/4, a divisor of 256 for every width of the two towers) whenever
// possible: a thread then stays on one channel group, its coefficients are loaded once, and the loop body is
// load / 4 fma / store — no 64-bit modulo and no coefficient re-reads per element.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       long long n4, int G, int relu) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  auto one = [&](long long i, const floatx4& sc, const floatx4& sh) {
    const floatx4 v = reinterpret_cast<const floatx4*>(x)[i];
    floatx4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = fmaf(v[j], sc[j], sh[j]);   // explicit fma: the backward kernels recompute exactly this for the ReLU mask
      if (relu) o[j] = fmaxf(o[j], 0.f);
    }
    reinterpret_cast<floatx4*>(y)[i] = o;
  };
  if (stride % G == 0) {
    const int g = (int)(i0 % G);
    const floatx4 sc = reinterpret_cast<const floatx4*>(scale)[g];
    const floatx4 sh = reinterpret_cast<const floatx4*>(shift)[g];
    long long i = i0;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four independent 16-B loads in flight per thread
      floatx4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const floatx4*>(x)[i + u * stride];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        floatx4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = fmaf(v[u][j], sc[j], sh[j]);
          if (relu) o[j] = fmaxf(o[j], 0.f);
        }
        reinterpret_cast<floatx4*>(y)[i + u * stride] = o;
      }
    }
    for (; i + stride < n4; i += 2 * stride) {   // two independent elements in flight
      const floatx4 v0 = reinterpret_cast<const floatx4*>(x)[i];
      const floatx4 v1 = reinterpret_cast<const floatx4*>(x)[i + stride];
      floatx4 o0, o1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o0[j] = fmaf(v0[j], sc[j], sh[j]);
        o1[j] = fmaf(v1[j], sc[j], sh[j]);
        if (relu) { o0[j] = fmaxf(o0[j], 0.f); o1[j] = fmaxf(o1[j], 0.f); }
      }
      reinterpret_cast<floatx4*>(y)[i] = o0;
      reinterpret_cast<floatx4*>(y)[i + stride] = o1;
    }
    if (i < n4) one(i, sc, sh);
    return;
  }
  for (long long i = i0; i < n4; i += stride) {
    const int g = (int)(i % G);
    one(i, reinterpret_cast<const floatx4*>(scale)[g], reinterpret_cast<const floatx4*>(shift)[g]);
  }
}

// eval path computes its coefficients in-kernel (no workspace): y = gamma*(x-rm)/sqrt(rv+eps)+beta
__global__ __launch_bounds__(256) void bn_apply_eval_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ rm, const float* __restrict__ rv,
                                                            float eps, long long n4, int G, int relu) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int g = (int)(i % G);
    const floatx4 v = reinterpret_cast<const floatx4*>(x)[i];
    const floatx4 ga = reinterpret_cast<const floatx4*>(gamma)[g];
    const floatx4 be = reinterpret_cast<const floatx4*>(beta)[g];
    const floatx4 m = reinterpret_cast<const floatx4*>(rm)[g];
    const floatx4 vv = reinterpret_cast<const floatx4*>(rv)[g];
    floatx4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float invstd = 1.f / sqrtf(vv[j] + eps);
      o[j] = (v[j] - m[j]) * invstd * ga[j] + be[j];
      if (relu) o[j] = fmaxf(o[j], 0.f);
    }
    reinterpret_cast<floatx4*>(y)[i] = o;
  }
}

// backward partials: part[blk][0][C] = sum dy_m, part[blk][1][C] = sum dy_m * xhat
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ part,
                                                             long long M, int C, int G, int rows_per_pass,
                                                             int rows_per_block, int relu) {
  __shared__ floatx4 sh[2][256];
  const int tid = threadIdx.x;
  const int g = tid % G, r = tid / G;
  floatx4 s = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
  const long long row0 = (long long)blockIdx.x * rows_per_block;
  if (r < rows_per_pass) {
    const floatx4 mu = reinterpret_cast<const floatx4*>(mean)[g];
    const floatx4 is = reinterpret_cast<const floatx4*>(invstd)[g];
    const floatx4 sc = reinterpret_cast<const floatx4*>(scale)[g];
    const floatx4 sh = reinterpret_cast<const floatx4*>(shift)[g];
    floatx4 sA[4], xA[4];   // four independent row streams per thread (latency: x and dy both come from HBM/MALL)
#pragma unroll
    for (int u = 0; u < 4; ++u) sA[u] = xA[u] = floatx4{0, 0, 0, 0};
    for (int k = r; k < rows_per_block; k += 4 * rows_per_pass) {
      floatx4 d[4], xv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long row = row0 + k + u * rows_per_pass;
        ok[u] = (k + u * rows_per_pass < rows_per_block) && row < M;
        const long long pos = (ok[u] ? row : 0) * C + g * 4;
        d[u] = *reinterpret_cast<const floatx4*>(dy + pos);
        xv[u] = *reinterpret_cast<const floatx4*>(x + pos);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (relu) {   // ReLU mask recomputed from x with the forward's exact fma (no read of the saved output)
#pragma unroll
          for (int j = 0; j < 4; ++j) d[u][j] = fmaf(xv[u][j], sc[j], sh[j]) > 0.f ? d[u][j] : 0.f;
        }
        if (!ok[u]) d[u] = floatx4{0, 0, 0, 0};
        sA[u] += d[u];
        xA[u] += d[u] * ((xv[u] - mu) * is);
      }
    }
    s = (sA[0] + sA[1]) + (sA[2] + sA[3]);
    sx = (xA[0] + xA[1]) + (xA[2] + xA[3]);
  }
  sh[0][tid] = s;
  sh[1][tid] = sx;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rows_per_pass; ++k) {
      s += sh[0][k * G + g];
      sx += sh[1][k * G + g];
    }
    float* o = part + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<floatx4*>(o + g * 4) = s;
    *reinterpret_cast<floatx4*>(o + C + g * 4) = sx;
  }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, long long M,
                                                               int C, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ k1,
                                                               float* __restrict__ k2, int frozen) {
  __shared__ double sh[16][2][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1));
  double s, sx;
  finalize_sums(part, nblk, C, blockIdx.x, sh, s, sx);
  if (threadIdx.x >= FIN_CH || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)sx;
  // frozen (eval-mode) statistics are constants of the graph: dx = gamma * invstd * dy_m, no mean terms
  k1[c] = frozen ? 0.f : (float)(s / (double)M);
  k2[c] = frozen ? 0.f : (float)(sx / (double)M);
}

// dx = gamma*invstd * (dy_m - mean(dy_m) - xhat * mean(dy_m*xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ dy, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ k1,
                                                           const float* __restrict__ k2, float* __restrict__ dx,
                                                           long long n4, int G, int relu) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  struct Coef { floatx4 sc, sh, mu, is, gis, a, b; };
  auto coef = [&](int g) {
    Coef c;
    c.sc = reinterpret_cast<const floatx4*>(scale)[g];
    c.sh = reinterpret_cast<const floatx4*>(shift)[g];
    c.mu = reinterpret_cast<const floatx4*>(mean)[g];
    c.is = reinterpret_cast<const floatx4*>(invstd)[g];
    c.gis = reinterpret_cast<const floatx4*>(gamma)[g] * c.is;
    c.a = reinterpret_cast<const floatx4*>(k1)[g];
    c.b = reinterpret_cast<const floatx4*>(k2)[g];
    return c;
  };
  auto eval = [&](floatx4 d, const floatx4& xv, const Coef& c) {
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = fmaf(xv[j], c.sc[j], c.sh[j]) > 0.f ? d[j] : 0.f;
    }
    const floatx4 xh = (xv - c.mu) * c.is;
    return c.gis * (d - c.a - xh * c.b);
  };
  if (stride % G == 0) {   // one channel group per thread: coefficients live in registers (see bn_apply_kernel)
    const Coef c = coef((int)(i0 % G));
    long long i = i0;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // eight independent 16-B loads in flight per thread
      floatx4 d[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = reinterpret_cast<const floatx4*>(dy)[i + u * stride];
        xv[u] = reinterpret_cast<const floatx4*>(x)[i + u * stride];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) reinterpret_cast<floatx4*>(dx)[i + u * stride] = eval(d[u], xv[u], c);
    }
    for (; i + stride < n4; i += 2 * stride) {
      const floatx4 d0 = reinterpret_cast<const floatx4*>(dy)[i], d1 = reinterpret_cast<const floatx4*>(dy)[i + stride];
      const floatx4 x0 = reinterpret_cast<const floatx4*>(x)[i], x1 = reinterpret_cast<const floatx4*>(x)[i + stride];
      reinterpret_cast<floatx4*>(dx)[i] = eval(d0, x0, c);
      reinterpret_cast<floatx4*>(dx)[i + stride] = eval(d1, x1, c);
    }
    if (i < n4) reinterpret_cast<floatx4*>(dx)[i] = eval(reinterpret_cast<const floatx4*>(dy)[i], reinterpret_cast<const floatx4*>(x)[i], c);
    return;
  }
  for (long long i = i0; i < n4; i += stride) {
    const Coef c = coef((int)(i % G));
    reinterpret_cast<floatx4*>(dx)[i] = eval(reinterpret_cast<const floatx4*>(dy)[i], reinterpret_cast<const floatx4*>(x)[i], c);
  }
}

// ---------------------------------------------------------------------------------------------
// Small layers (M*C <= 8 M elements, C % 16 == 0): the finalize folded into the apply launch.  Every block owns 16
// channels x a slice of the rows and folds the partial rows of ITS 16 channels itself (<= ~512 rows x 128 B from
// L2, the same fp64 fold in every block of a channel group, so all of them apply identical coefficients); the
// blocks of slice 0 also publish the saved statistics / running statistics / parameter gradients.  One launch
// (~6 us) instead of finalize (8-11 us, 4-32 blocks on an otherwise idle chip) + apply: conv4x / conv5x / audio
// layers, 29 of the step's 42 BatchNorms, forward and backward.
// ---------------------------------------------------------------------------------------------
constexpr int FA_CH = 16;

__device__ __forceinline__ void fold16(const float* __restrict__ part, int nblk, int C, int cg, double (*sh)[2][FA_CH],
                                       double& s0, double& s1) {
  const int tid = threadIdx.x, g = tid & 3, slice = tid >> 2;     // 4 float4 groups x 64 row slices
  const int c0 = cg * FA_CH + g * 4;
  const long long st = 2ll * C;
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = slice; k < nblk; k += 256) {                       // rows k, k+64, k+128, k+192 in flight
    floatx4 x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = k + u * 64;
      const floatx4 z = {0.f, 0.f, 0.f, 0.f};
      const float* q = part + (long long)(row < nblk ? row : k) * st + c0;
      x[u] = row < nblk ? *reinterpret_cast<const floatx4*>(q) : z;
      y[u] = row < nblk ? *reinterpret_cast<const floatx4*>(q + C) : z;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] += (double)x[u][j]; b[j] += (double)y[u][j]; }
  }
  const int wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double sa = a[j], sb = b[j];
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { sa += __shfl_xor(sa, m, 64); sb += __shfl_xor(sb, m, 64); }
    if ((tid & 63) < 4) {
      sh[wave][0][g * 4 + j] = sa;
      sh[wave][1][g * 4 + j] = sb;
    }
  }
  __syncthreads();
  s0 = s1 = 0.0;
  if (tid < FA_CH) {
    s0 = (sh[0][0][tid] + sh[1][0][tid]) + (sh[2][0][tid] + sh[3][0][tid]);
    s1 = (sh[0][1][tid] + sh[1][1][tid]) + (sh[2][1][tid] + sh[3][1][tid]);
  }
}

__global__ __launch_bounds__(256) void bn_fin_apply_kernel(const float* __restrict__ part, int nblk, long long M, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float momentum, float eps, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, float* __restrict__ scale,
                                                           float* __restrict__ shift, long long* __restrict__ num_batches_tracked,
                                                           const float* __restrict__ x, float* __restrict__ y, int relu) {
  __shared__ double sh[4][2][FA_CH];
  __shared__ float cf[2][FA_CH];
  const int tid = threadIdx.x;
  // a workgroup reads 16 channels = 64 bytes of every row: the workgroups that read the other half of the same 128-byte
  // lines (and the rest of the row) must sit on the SAME XCD, or every L2 fetches whole lines for half of their bytes —
  // in hardware order (id % 8 = XCD) and with C = 128 each XCD had exactly one channel group
  const int lid = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int ncg = C / FA_CH, cg = lid % ncg, slice = lid / ncg, nsl = gridDim.x / ncg;
  if (num_batches_tracked && lid == 0 && tid == 0) *num_batches_tracked += 1;
  // the first batch of x is requested BEFORE the partial sums are folded: its memory latency hides under the fold
  // (these launches are 7-19 us long, ~3 us of which is the fold every workgroup repeats for its 16 channels)
  const long long pstep = (long long)nsl * 64, pcol = cg * 4 + (tid & 3);
  const long long pr = (long long)slice * 64 + (tid >> 2);
  const bool pre = pr + 3 * pstep < M;
  floatx4 pv[4];
  if (pre) {
#pragma unroll
    for (int u = 0; u < 4; ++u) pv[u] = reinterpret_cast<const floatx4*>(x)[(pr + u * pstep) * (C >> 2) + pcol];
  }
  double s, ss;
  fold16(part, nblk, C, cg, sh, s, ss);
  if (tid < FA_CH) {
    const int c = cg * FA_CH + tid;
    const double mean = s / (double)M;
    double var = ss / (double)M - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    const float sf = beta[c] - (float)mean * sc;
    cf[0][tid] = sc;
    cf[1][tid] = sf;
    if (slice == 0) {
      save_mean[c] = (float)mean;
      save_invstd[c] = invstd;
      scale[c] = sc;
      shift[c] = sf;
      if (running_mean) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
  __syncthreads();
  const int g = tid & 3, G = C >> 2;
  const floatx4 sc = *reinterpret_cast<const floatx4*>(&cf[0][g * 4]);
  const floatx4 sf = *reinterpret_cast<const floatx4*>(&cf[1][g * 4]);
  const long long step = (long long)nsl * 64;
  const long long col = cg * 4 + g;
  auto one = [&](const floatx4& v) {
    floatx4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = fmaf(v[j], sc[j], sf[j]);   // the same fma the backward kernels recompute for the ReLU mask
      if (relu) o[j] = fmaxf(o[j], 0.f);
    }
    return o;
  };
  long long r = (long long)slice * 64 + (tid >> 2);
  if (pre) {
#pragma unroll
    for (int u = 0; u < 4; ++u) reinterpret_cast<floatx4*>(y)[(r + u * step) * G + col] = one(pv[u]);
    r += 4 * step;
  }
  for (; r + 3 * step < M; r += 4 * step) {
    floatx4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const floatx4*>(x)[(r + u * step) * G + col];
#pragma unroll
    for (int u = 0; u < 4; ++u) reinterpret_cast<floatx4*>(y)[(r + u * step) * G + col] = one(v[u]);
  }
  for (; r < M; r += step) reinterpret_cast<floatx4*>(y)[r * G + col] = one(reinterpret_cast<const floatx4*>(x)[r * G + col]);
}

__global__ __launch_bounds__(256) void bn_bwd_fin_apply_kernel(const float* __restrict__ part, int nblk, long long M, int C,
                                                               const float* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ dy,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dx, int relu,
                                                               int frozen) {
  __shared__ double sh[4][2][FA_CH];
  __shared__ float cf[2][FA_CH];
  const int tid = threadIdx.x;
  const int lid = (int)xcd_remap(blockIdx.x, gridDim.x);      // (see bn_fin_apply_kernel)
  const int ncg = C / FA_CH, cg = lid % ncg, slice = lid / ncg, nsl = gridDim.x / ncg;
  // (first batch of dy / x requested before the fold: see bn_fin_apply_kernel)
  const long long pstep = (long long)nsl * 64;
  const long long pr = (long long)slice * 64 + (tid >> 2);
  const bool pre = pr + pstep < M;
  floatx4 pd0, pd1, px0, px1;
  if (pre) {
    const long long i0 = pr * (C >> 2) + cg * 4 + (tid & 3), i1 = (pr + pstep) * (C >> 2) + cg * 4 + (tid & 3);
    pd0 = reinterpret_cast<const floatx4*>(dy)[i0]; pd1 = reinterpret_cast<const floatx4*>(dy)[i1];
    px0 = reinterpret_cast<const floatx4*>(x)[i0]; px1 = reinterpret_cast<const floatx4*>(x)[i1];
  }
  double s, sx;
  fold16(part, nblk, C, cg, sh, s, sx);
  if (tid < FA_CH) {
    const int c = cg * FA_CH + tid;
    if (slice == 0) {
      dbeta[c] = (float)s;
      dgamma[c] = (float)sx;
    }
    cf[0][tid] = frozen ? 0.f : (float)(s / (double)M);
    cf[1][tid] = frozen ? 0.f : (float)(sx / (double)M);
  }
  __syncthreads();
  const int g = tid & 3, G = C >> 2;
  const int cgi = cg * 4 + g;
  const floatx4 a = *reinterpret_cast<const floatx4*>(&cf[0][g * 4]);
  const floatx4 b = *reinterpret_cast<const floatx4*>(&cf[1][g * 4]);
  const floatx4 mu = reinterpret_cast<const floatx4*>(mean)[cgi], is = reinterpret_cast<const floatx4*>(invstd)[cgi];
  const floatx4 gis = reinterpret_cast<const floatx4*>(gamma)[cgi] * is;
  floatx4 sc = {0.f, 0.f, 0.f, 0.f}, sf = {0.f, 0.f, 0.f, 0.f};
  if (relu) { sc = reinterpret_cast<const floatx4*>(scale)[cgi]; sf = reinterpret_cast<const floatx4*>(shift)[cgi]; }
  auto eval = [&](floatx4 d, const floatx4& xv) {
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = fmaf(xv[j], sc[j], sf[j]) > 0.f ? d[j] : 0.f;
    }
    const floatx4 xh = (xv - mu) * is;
    return gis * (d - a - xh * b);
  };
  const long long step = (long long)nsl * 64;
  long long r = (long long)slice * 64 + (tid >> 2);
  if (pre) {
    reinterpret_cast<floatx4*>(dx)[r * G + cgi] = eval(pd0, px0);
    reinterpret_cast<floatx4*>(dx)[(r + step) * G + cgi] = eval(pd1, px1);
    r += 2 * step;
  }
  for (; r + step < M; r += 2 * step) {
    const long long i0 = r * G + cgi, i1 = (r + step) * G + cgi;
    const floatx4 d0 = reinterpret_cast<const floatx4*>(dy)[i0], d1 = reinterpret_cast<const floatx4*>(dy)[i1];
    const floatx4 x0 = reinterpret_cast<const floatx4*>(x)[i0], x1 = reinterpret_cast<const floatx4*>(x)[i1];
    reinterpret_cast<floatx4*>(dx)[i0] = eval(d0, x0);
    reinterpret_cast<floatx4*>(dx)[i1] = eval(d1, x1);
  }
  if (r < M) {
    const long long i0 = r * G + cgi;
    reinterpret_cast<floatx4*>(dx)[i0] = eval(reinterpret_cast<const floatx4*>(dy)[i0], reinterpret_cast<const floatx4*>(x)[i0]);
  }
}

// ---------------------------------------------------------------------------------------------
// MaxPool (1,3,3) stride (1,2,2) pad (0,1,1), channels-last.  argmax = window slot (dh*3+dw) of the
// FIRST maximum in scan order (ATen CPU: `val > max || isnan(val)`).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ am, int BT, int H, int W, int Ho,
                                                          int Wo, int G) {
  const long long n = (long long)BT * Ho * Wo * G;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long long bt = r / Ho;
    floatx4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int slot[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int h = ho * 2 - 1 + dh;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int w = wo * 2 - 1 + dw;
        if ((unsigned)w >= (unsigned)W) continue;
        const floatx4 v = *reinterpret_cast<const floatx4*>(x + (((bt * H + h) * W + w) * (long long)G + g) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (first || v[j] > best[j] || v[j] != v[j]) {
            best[j] = v[j];
            slot[j] = dh * 3 + dw;
          }
        }
        first = false;
      }
    }
    reinterpret_cast<floatx4*>(y)[i] = best;
    reinterpret_cast<uchar4*>(am)[i] = make_uchar4((unsigned char)slot[0], (unsigned char)slot[1],
                                                   (unsigned char)slot[2], (unsigned char)slot[3]);
  }
}

// gather-style backward: every input pixel sums dy over the (<= 4) windows whose argmax is this pixel.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ am,
                                                          float* __restrict__ dx, int BT, int H, int W, int Ho, int Wo,
                                                          int G) {
  const long long n = (long long)BT * H * W * G;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const long long bt = r / H;
    floatx4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int nh = h + 1 - dh;
      if (nh < 0 || (nh & 1)) continue;
      const int ho = nh >> 1;
      if (ho >= Ho) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int nw = w + 1 - dw;
        if (nw < 0 || (nw & 1)) continue;
        const int wo = nw >> 1;
        if (wo >= Wo) continue;
        const long long o = ((bt * Ho + ho) * Wo + wo) * (long long)G + g;
        const uchar4 a = reinterpret_cast<const uchar4*>(am)[o];
        const floatx4 d = reinterpret_cast<const floatx4*>(dy)[o];
        const int slot = dh * 3 + dw;
        acc.x += a.x == slot ? d.x : 0.f;
        acc.y += a.y == slot ? d.y : 0.f;
        acc.z += a.z == slot ? d.z : 0.f;
        acc.w += a.w == slot ? d.w : 0.f;
      }
    }
    reinterpret_cast<floatx4*>(dx)[i] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Stem tail fused: BatchNorm(train) + ReLU + MaxPool(1,3,3)/(1,2,2)/(0,1,1)  (models/video.py:21-23).
// The 411 MB normalised activation is never written: the pooled output is produced straight from the conv
// output, and the backward kernels rebuild the un-pooled gradient from the pooled one + the argmax slots.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ y,
                                                          uint8_t* __restrict__ am, int BT, int H, int W, int Ho, int Wo,
                                                          int G) {
  const long long n = (long long)BT * Ho * Wo * G;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int g = (int)(i % G);
    long long r = i / G;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long long bt = r / Ho;
    const floatx4 sc = reinterpret_cast<const floatx4*>(scale)[g];
    const floatx4 sh = reinterpret_cast<const floatx4*>(shift)[g];
    floatx4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int slot[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int h = ho * 2 - 1 + dh;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int w = wo * 2 - 1 + dw;
        if ((unsigned)w >= (unsigned)W) continue;
        const floatx4 xv = *reinterpret_cast<const floatx4*>(x + (((bt * H + h) * W + w) * (long long)G + g) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = fmaxf(fmaf(xv[j], sc[j], sh[j]), 0.f);   // the same fma the backward recomputes
          if (first || v > best[j] || v != v) {
            best[j] = v;
            slot[j] = dh * 3 + dw;
          }
        }
        first = false;
      }
    }
    reinterpret_cast<floatx4*>(y)[i] = best;
    reinterpret_cast<uchar4*>(am)[i] = make_uchar4((unsigned char)slot[0], (unsigned char)slot[1],
                                                   (unsigned char)slot[2], (unsigned char)slot[3]);
  }
}

// A thread owns a 2x2 cell of un-pooled positions (h = 2a, 2a+1; w = 2b, 2b+1) of one channel group: the four
// pooling windows (a|a+1, b|b+1) that can have selected them are loaded once (argmax slots + pooled gradient)
// and serve all four positions — 3 loads per position instead of 5.5, four positions in flight per thread.
struct PoolCell {
  floatx4 d[2][2];   // gradient reaching position (dh2, dw2) of the cell (before the ReLU mask)
};
__device__ __forceinline__ PoolCell pool_cell_grad(const float* __restrict__ dy, const uint8_t* __restrict__ am,
                                                   long long bt, int a, int b, int g, int Ho, int Wo, int G) {
  uchar4 A[2][2];
  floatx4 D[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = a + i < Ho && b + j < Wo;
      const long long o = ((bt * Ho + (ok ? a + i : 0)) * Wo + (ok ? b + j : 0)) * (long long)G + g;
      A[i][j] = reinterpret_cast<const uchar4*>(am)[o];
      D[i][j] = reinterpret_cast<const floatx4*>(dy)[o];
      if (!ok) A[i][j] = make_uchar4(255, 255, 255, 255);
    }
  PoolCell c;
  auto pick = [](uchar4 s, int slot, floatx4 d) {
    floatx4 r;
    r.x = s.x == slot ? d.x : 0.f; r.y = s.y == slot ? d.y : 0.f;
    r.z = s.z == slot ? d.z : 0.f; r.w = s.w == slot ? d.w : 0.f;
    return r;
  };
  // position (2a + dh2, 2b + dw2) sits in window (a + i, b + j) at slot ((dh2 + 1 - 2i), (dw2 + 1 - 2j)):
  // even coordinate -> only i (j) = 0, slot row (col) 1; odd -> i = 0 slot 2 and i = 1 slot 0
  c.d[0][0] = pick(A[0][0], 1 * 3 + 1, D[0][0]);
  c.d[0][1] = pick(A[0][0], 1 * 3 + 2, D[0][0]) + pick(A[0][1], 1 * 3 + 0, D[0][1]);
  c.d[1][0] = pick(A[0][0], 2 * 3 + 1, D[0][0]) + pick(A[1][0], 0 * 3 + 1, D[1][0]);
  c.d[1][1] = pick(A[0][0], 2 * 3 + 2, D[0][0]) + pick(A[0][1], 2 * 3 + 0, D[0][1]) + pick(A[1][0], 0 * 3 + 2, D[1][0]) +
              pick(A[1][1], 0 * 3 + 0, D[1][1]);
  return c;
}

// bn_bwd_partial_kernel with dy rebuilt from (pooled dy, argmax); ReLU mask from x.  Rows of the block = cells.
__global__ __launch_bounds__(256) void bn_pool_bwd_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ dyp, const uint8_t* __restrict__ am, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ part, long long cells, int C, int G, int cells_per_pass,
    int cells_per_block, int H, int W, int Ho, int Wo) {
  __shared__ floatx4 shm[2][256];
  const int tid = threadIdx.x;
  const int g = tid % G, r = tid / G;
  const int Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;
  floatx4 s = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
  const long long cell0 = (long long)blockIdx.x * cells_per_block;
  if (r < cells_per_pass) {
    const floatx4 mu = reinterpret_cast<const floatx4*>(mean)[g];
    const floatx4 is = reinterpret_cast<const floatx4*>(invstd)[g];
    const floatx4 sc = reinterpret_cast<const floatx4*>(scale)[g];
    const floatx4 sh = reinterpret_cast<const floatx4*>(shift)[g];
    for (int k = r; k < cells_per_block; k += cells_per_pass) {
      const long long cell = cell0 + k;
      if (cell >= cells) break;
      const int b = (int)(cell % Wc);
      const long long q = cell / Wc;
      const int a = (int)(q % Hc);
      const long long bt = q / Hc;
      const PoolCell pc = pool_cell_grad(dyp, am, bt, a, b, g, Ho, Wo, G);
      floatx4 xv[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool ok = 2 * a + i < H && 2 * b + j < W;
          xv[i][j] = *reinterpret_cast<const floatx4*>(
              x + (((bt * H + (ok ? 2 * a + i : 0)) * W + (ok ? 2 * b + j : 0)) * (long long)G + g) * 4);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (2 * a + i >= H || 2 * b + j >= W) continue;
          floatx4 d = pc.d[i][j];
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = fmaf(xv[i][j][e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
          s += d;
          sx += d * ((xv[i][j] - mu) * is);
        }
    }
  }
  shm[0][tid] = s;
  shm[1][tid] = sx;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < cells_per_pass; ++k) {
      s += shm[0][k * G + g];
      sx += shm[1][k * G + g];
    }
    float* o = part + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<floatx4*>(o + g * 4) = s;
    *reinterpret_cast<floatx4*>(o + C + g * 4) = sx;
  }
}

__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ dyp, const uint8_t* __restrict__ am, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ k1,
    const float* __restrict__ k2, float* __restrict__ dx, long long ncell4, int G, int H, int W, int Ho, int Wo) {
  const int Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell4; i += stride) {
    const int g = (int)(i % G);
    const long long cell = i / G;
    const int b = (int)(cell % Wc);
    const long long q = cell / Wc;
    const int a = (int)(q % Hc);
    const long long bt = q / Hc;
    const PoolCell pc = pool_cell_grad(dyp, am, bt, a, b, g, Ho, Wo, G);
    const floatx4 sc = reinterpret_cast<const floatx4*>(scale)[g];
    const floatx4 sh = reinterpret_cast<const floatx4*>(shift)[g];
    const floatx4 mu = reinterpret_cast<const floatx4*>(mean)[g];
    const floatx4 is = reinterpret_cast<const floatx4*>(invstd)[g];
    const floatx4 ga = reinterpret_cast<const floatx4*>(gamma)[g];
    const floatx4 ka = reinterpret_cast<const floatx4*>(k1)[g];
    const floatx4 kb = reinterpret_cast<const floatx4*>(k2)[g];
    floatx4 xv[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const bool ok = 2 * a + u < H && 2 * b + v < W;
        xv[u][v] = *reinterpret_cast<const floatx4*>(
            x + (((bt * H + (ok ? 2 * a + u : 0)) * W + (ok ? 2 * b + v : 0)) * (long long)G + g) * 4);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        if (2 * a + u >= H || 2 * b + v >= W) continue;
        floatx4 d = pc.d[u][v];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = fmaf(xv[u][v][e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
        const floatx4 xh = (xv[u][v] - mu) * is;
        *reinterpret_cast<floatx4*>(dx + (((bt * H + 2 * a + u) * W + 2 * b + v) * (long long)G + g) * 4) =
            ga * is * (d - ka - xh * kb);
      }
  }
}

// global max over S positions: x [B,S,C] -> y [B,C]; first maximum wins
__global__ void global_maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ am,
                                          int B, int S, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * C) return;
  const int c = (int)(i % C);
  const long long b = i / C;
  const float* p = x + b * S * C + c;
  float best = p[0];
  int arg = 0;
  for (int s = 1; s < S; ++s) {
    const float v = p[(long long)s * C];
    if (v > best || v != v) {
      best = v;
      arg = s;
    }
  }
  y[i] = best;
  am[i] = arg;
}

__global__ void global_maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ am,
                                          float* __restrict__ dx, int B, int S, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * S * C) return;
  const int c = (int)(i % C);
  const long long r = i / C;
  const int s = (int)(r % S);
  const long long b = r / S;
  dx[i] = am[b * C + c] == s ? dy[b * C + c] : 0.f;
}

__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// out[c] = sum_m x[m][c] — one wave column-slab per block: 64 lanes = 64 consecutive channels
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long M,
                                                     int C) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C)
    for (long long m = wave; m < M; m += 4) s += x[m * C + c];
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C) out[c] = sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane];
}

// the fused finalize + apply path of small layers: element bound and grid
static bool bn_fused_ok(int64_t M, int C) {
  const long long cap = 1ll << 23;      // conv3x-sized layers (6.4 M elements) still gain a launch; conv2x-sized ones lose
  return C % FA_CH == 0 && (long long)M * C <= cap;
}
static unsigned bn_fused_grid(int64_t M, int C) {
  const int ncg = C / FA_CH;
  const int cap = 256;            // more blocks repeat the fold more often: 512 / 1024 / 2048 measured slower
  // each block folds its 16 channels' partial rows itself (a fixed ~3 us): at least 4 x 64 rows of apply work per
  // block, at most `cap` blocks
  const long long chunks = (M + 63) / 64;
  long long nsl = (chunks + 3) / 4;
  if (nsl > cap / ncg) nsl = cap / ncg;
  if (nsl < 1) nsl = 1;
  return (unsigned)(ncg * nsl);
}

static unsigned ew_grid(long long n) {
  const int cap = 256 * 32;            // 8192 blocks of 256 threads at most: conv2x-sized apply 35.2 -> 32.8 us, backward apply 53.7 -> 49.9 us
  long long g = ceil_div(n, 256);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace avid

using namespace avid;

extern "C" size_t avid_bn_workspace_bytes(int64_t M, int C) {
  if (M <= 0 || C <= 0 || C % 4 || 256 % (C / 4)) return 0;
  BnPlan p = bn_plan(M, C);
  return sizeof(float) * ((size_t)p.nblk * 2 * C + 4 * (size_t)C);
}

static int bn_check(int64_t M, int C, const char* who) {
  AVID_REQUIRE(M > 0 && C > 0, AVID_E_SHAPE, "%s: empty tensor", who);
  AVID_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, AVID_E_UNSUPPORTED,
               "%s: C=%d must be a power of two in [4, 1024]", who, C);
  return AVID_OK;
}

extern "C" int avid_bn_fwd_train(int64_t M, int C, const float* x, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, int relu,
                                 float* y, float* save_mean, float* save_invstd, float* save_scale,
                                 float* save_shift, int64_t* num_batches_tracked, const float* partials, int nparts,
                                 void* ws, size_t ws_bytes, avid_stream_t stream) {
  int rc = bn_check(M, C, "bn_fwd_train");
  if (rc) return rc;
  AVID_REQUIRE(x && gamma && beta && save_mean && save_invstd && save_scale && save_shift && ws, AVID_E_BADARG,
               "bn_fwd_train: null pointer");
  AVID_REQUIRE(ws_bytes >= avid_bn_workspace_bytes(M, C), AVID_E_BADARG, "bn_fwd_train: workspace too small");
  // y == NULL: statistics only — mean / invstd / scale / shift (and the running statistics) are made, the normalised tensor is
  // not: its consumer applies fma(x, scale, shift) (+ReLU) while it stages x (avid_conv_fwd_in / avid_conv_wgrad_in)

  hipStream_t s = (hipStream_t)stream;
  BnPlan p = bn_plan(M, C);
  float* part = static_cast<float*>(ws);
  float* scale = save_scale;
  float* shift = save_shift;
  int nblk = p.nblk;
  if (partials && nparts > 0) {   // statistics already reduced to partial rows by the producing convolution
    part = const_cast<float*>(partials);
    nblk = nparts;
  } else {
    ScopedTimer t(s, "bn_stats_partial_kernel", 0.0, 4.0 * M * C);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(p.nblk), dim3(256), 0, s, x, part, (long long)M, C, p.G,
                       p.rows_per_pass, p.rows_per_block);
  }
  if (y && bn_fused_ok(M, C)) {
    ScopedTimer t(s, "bn_fin_apply_kernel", 0.0, 8.0 * M * C);
    hipLaunchKernelGGL(bn_fin_apply_kernel, dim3(bn_fused_grid(M, C)), dim3(256), 0, s, part, nblk, (long long)M, C, gamma,
                       beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift,
                       reinterpret_cast<long long*>(num_batches_tracked), x, y, relu);
    return check_launch("bn_fwd_train");
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div(C, FIN_CH)), dim3(1024), 0, s, part, nblk, (long long)M,
                     C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift,
                     reinterpret_cast<long long*>(num_batches_tracked));
  const long long n4 = (long long)M * p.G;
  if (y) {
    ScopedTimer t(s, "bn_apply_kernel", 0.0, 8.0 * M * C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, x, y, scale, shift, n4, p.G, relu);
  }
  return check_launch("bn_fwd_train");
}

extern "C" int avid_bn_fwd_eval(int64_t M, int C, const float* x, const float* gamma, const float* beta,
                                const float* running_mean, const float* running_var, float eps, int relu, float* y,
                                float* save4, avid_stream_t stream) {
  int rc = bn_check(M, C, "bn_fwd_eval");
  if (rc) return rc;
  AVID_REQUIRE(x && gamma && beta && running_mean && running_var && y, AVID_E_BADARG, "bn_fwd_eval: null pointer");
  const long long n4 = (long long)M * (C / 4);
  if (save4) {   // a backward will follow (fine-tuning with frozen BatchNorm): y = fma(x, scale, shift), the
                 // expression avid_bn_bwd recomputes the ReLU mask with; mean / invstd / scale / shift are kept
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, C,
                       gamma, beta, running_mean, running_var, eps, save4, save4 + C, save4 + 2 * C, save4 + 3 * C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, y, save4 + 2 * C,
                       save4 + 3 * C, n4, C / 4, relu);
    return check_launch("bn_fwd_eval");
  }
  hipLaunchKernelGGL(bn_apply_eval_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta,
                     running_mean, running_var, eps, n4, C / 4, relu);
  return check_launch("bn_fwd_eval");
}

extern "C" int avid_bn_bwd(int64_t M, int C, const float* x, const float* dy, const float* gamma,
                           const float* save_mean, const float* save_invstd, const float* save_scale,
                           const float* save_shift, int relu, float* dx, float* dgamma, float* dbeta,
                           const float* partials, int nparts, int frozen, void* ws, size_t ws_bytes,
                           avid_stream_t stream) {
  int rc = bn_check(M, C, "bn_bwd");
  if (rc) return rc;
  AVID_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && ws, AVID_E_BADARG,
               "bn_bwd: null pointer");
  AVID_REQUIRE(!relu || (save_scale && save_shift), AVID_E_BADARG, "bn_bwd: relu needs the saved scale/shift");
  AVID_REQUIRE(ws_bytes >= avid_bn_workspace_bytes(M, C), AVID_E_BADARG, "bn_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  BnPlan p = bn_plan(M, C);
  float* wsf = static_cast<float*>(ws);
  float* k1 = wsf + (size_t)p.nblk * 2 * C;
  float* k2 = k1 + C;
  const float* part = wsf;
  int nblk = p.nblk;
  if (partials && nparts > 0) {   // the dgrad that produced dy already reduced it to partial rows
    part = partials;
    nblk = nparts;
  } else {
    ScopedTimer t(s, "bn_bwd_partial_kernel", 0.0, 4.0 * M * C * 2);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(p.nblk), dim3(256), 0, s, x, save_scale, save_shift, dy, save_mean,
                       save_invstd, wsf,
                       (long long)M, C, p.G, p.rows_per_pass, p.rows_per_block, relu);
  }
  if (bn_fused_ok(M, C)) {
    ScopedTimer t(s, "bn_bwd_fin_apply_kernel", 0.0, 4.0 * M * C * 3);
    hipLaunchKernelGGL(bn_bwd_fin_apply_kernel, dim3(bn_fused_grid(M, C)), dim3(256), 0, s, part, nblk, (long long)M, C, x,
                       save_scale, save_shift, dy, gamma, save_mean, save_invstd, dgamma, dbeta, dx, relu, frozen);
    return check_launch("bn_bwd");
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(C, FIN_CH)), dim3(1024), 0, s, part, nblk,
                     (long long)M, C, dgamma, dbeta, k1, k2, frozen);
  const long long n4 = (long long)M * p.G;
  {
    ScopedTimer t(s, "bn_bwd_apply_kernel", 0.0, 4.0 * M * C * 3);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, x, save_scale, save_shift, dy, gamma,
                       save_mean,
                       save_invstd, k1, k2, dx, n4, p.G, relu);
  }
  return check_launch("bn_bwd");
}

extern "C" int avid_bn_relu_maxpool_fwd(int B, int T, int H, int W, int C, const float* x, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float momentum,
                                        float eps, float* y, uint8_t* argmax, float* save_mean, float* save_invstd,
                                        float* save_scale, float* save_shift, int64_t* num_batches_tracked,
                                        const float* partials, int nparts, void* ws, size_t ws_bytes,
                                        avid_stream_t stream) {
  const int64_t M = (int64_t)B * T * H * W;
  int rc = bn_check(M, C, "bn_relu_maxpool_fwd");
  if (rc) return rc;
  AVID_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0, AVID_E_SHAPE, "bn_relu_maxpool_fwd: bad shape");
  AVID_REQUIRE(x && gamma && beta && y && argmax && save_mean && save_invstd && save_scale && save_shift && ws,
               AVID_E_BADARG, "bn_relu_maxpool_fwd: null pointer");
  AVID_REQUIRE(ws_bytes >= avid_bn_workspace_bytes(M, C), AVID_E_BADARG, "bn_relu_maxpool_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  BnPlan p = bn_plan(M, C);
  const float* part = static_cast<float*>(ws);
  int nblk = p.nblk;
  if (partials && nparts > 0) {   // statistics already reduced to partial rows by the producing convolution
    part = partials;
    nblk = nparts;
  } else {
    ScopedTimer t(s, "bn_stats_partial_kernel", 0.0, 4.0 * M * C);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(p.nblk), dim3(256), 0, s, x, static_cast<float*>(ws), (long long)M,
                       C, p.G, p.rows_per_pass, p.rows_per_block);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div(C, FIN_CH)), dim3(1024), 0, s, part, nblk,
                     (long long)M, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                     save_scale, save_shift, reinterpret_cast<long long*>(num_batches_tracked));
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long n = (long long)B * T * Ho * Wo * p.G;
  {
    ScopedTimer t(s, "bn_pool_fwd_kernel", 0.0, 4.0 * B * T * C * ((double)H * W + 1.25 * Ho * Wo));
    hipLaunchKernelGGL(bn_pool_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, s, x, save_scale, save_shift, y, argmax, B * T,
                       H, W, Ho, Wo, p.G);
  }
  return check_launch("bn_relu_maxpool_fwd");
}

extern "C" int avid_bn_relu_maxpool_bwd(int B, int T, int H, int W, int C, const float* x, const float* dy,
                                        const uint8_t* argmax, const float* gamma, const float* save_mean,
                                        const float* save_invstd, const float* save_scale, const float* save_shift,
                                        float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                        avid_stream_t stream) {
  const int64_t M = (int64_t)B * T * H * W;
  int rc = bn_check(M, C, "bn_relu_maxpool_bwd");
  if (rc) return rc;
  AVID_REQUIRE(x && dy && argmax && gamma && save_mean && save_invstd && save_scale && save_shift && dx && dgamma &&
                   dbeta && ws,
               AVID_E_BADARG, "bn_relu_maxpool_bwd: null pointer");
  AVID_REQUIRE(ws_bytes >= avid_bn_workspace_bytes(M, C), AVID_E_BADARG, "bn_relu_maxpool_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  BnPlan p = bn_plan(M, C);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  float* part = static_cast<float*>(ws);
  float* k1 = part + (size_t)p.nblk * 2 * C;
  float* k2 = k1 + C;
  // blocks over 2x2 cells of positions; at most p.nblk partial rows (the workspace is sized for that)
  const long long cells = (long long)B * T * ((H + 1) / 2) * ((W + 1) / 2);
  long long cpb = ceil_div(cells, p.nblk);
  cpb = ceil_div(cpb, p.rows_per_pass) * p.rows_per_pass;
  const int nblk = (int)ceil_div(cells, cpb);
  {
    ScopedTimer t(s, "bn_pool_bwd_partial_kernel", 0.0, 4.0 * M * C * 1.3);
    hipLaunchKernelGGL(bn_pool_bwd_partial_kernel, dim3(nblk), dim3(256), 0, s, x, save_scale, save_shift, dy, argmax,
                       save_mean, save_invstd, part, cells, C, p.G, p.rows_per_pass, (int)cpb, H, W, Ho, Wo);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(C, FIN_CH)), dim3(1024), 0, s, part, nblk,
                     (long long)M, C, dgamma, dbeta, k1, k2, 0);
  const long long nc4 = cells * p.G;
  {
    ScopedTimer t(s, "bn_pool_bwd_apply_kernel", 0.0, 4.0 * M * C * 2.3);
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(ew_grid(nc4)), dim3(256), 0, s, x, save_scale, save_shift, dy, argmax,
                       gamma, save_mean, save_invstd, k1, k2, dx, nc4, p.G, H, W, Ho, Wo);
  }
  return check_launch("bn_relu_maxpool_bwd");
}

extern "C" int avid_maxpool_hw3s2_fwd(int B, int T, int H, int W, int C, const float* x, float* y, uint8_t* argmax,
                                      avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, AVID_E_SHAPE, "maxpool_fwd: bad shape");
  AVID_REQUIRE(x && y && argmax, AVID_E_BADARG, "maxpool_fwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long n = (long long)B * T * Ho * Wo * (C / 4);
  ScopedTimer t((hipStream_t)stream, "maxpool_fwd_kernel", 0.0,
                4.0 * B * T * C * ((double)H * W + 1.25 * Ho * Wo));
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, B * T, H,
                     W, Ho, Wo, C / 4);
  return check_launch("maxpool_fwd");
}

extern "C" int avid_maxpool_hw3s2_bwd(int B, int T, int H, int W, int C, const float* dy, const uint8_t* argmax,
                                      float* dx, avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, AVID_E_SHAPE, "maxpool_bwd: bad shape");
  AVID_REQUIRE(dy && dx && argmax, AVID_E_BADARG, "maxpool_bwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long n = (long long)B * T * H * W * (C / 4);
  ScopedTimer t((hipStream_t)stream, "maxpool_bwd_kernel", 0.0,
                4.0 * B * T * C * ((double)H * W + 1.25 * Ho * Wo));
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, B * T, H,
                     W, Ho, Wo, C / 4);
  return check_launch("maxpool_bwd");
}

extern "C" int avid_global_maxpool_fwd(int B, int S, int C, const float* x, float* y, int32_t* argmax,
                                       avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && S > 0 && C > 0, AVID_E_SHAPE, "global_maxpool_fwd: bad shape");
  AVID_REQUIRE(x && y && argmax, AVID_E_BADARG, "global_maxpool_fwd: null pointer");
  const long long n = (long long)B * C;
  hipLaunchKernelGGL(global_maxpool_fwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     y, argmax, B, S, C);
  return check_launch("global_maxpool_fwd");
}

extern "C" int avid_global_maxpool_bwd(int B, int S, int C, const float* dy, const int32_t* argmax, float* dx,
                                       avid_stream_t stream) {
  AVID_REQUIRE(B > 0 && S > 0 && C > 0, AVID_E_SHAPE, "global_maxpool_bwd: bad shape");
  AVID_REQUIRE(dy && dx && argmax, AVID_E_BADARG, "global_maxpool_bwd: null pointer");
  const long long n = (long long)B * S * C;
  hipLaunchKernelGGL(global_maxpool_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     dy, argmax, dx, B, S, C);
  return check_launch("global_maxpool_bwd");
}

extern "C" int avid_relu_bwd(int64_t n, const float* y, const float* dy, float* dx, avid_stream_t stream) {
  AVID_REQUIRE(n > 0 && y && dy && dx, AVID_E_BADARG, "relu_bwd: bad argument");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, (long long)n);
  return check_launch("relu_bwd");
}

extern "C" int avid_colsum(int64_t M, int C, const float* x, float* out, avid_stream_t stream) {
  AVID_REQUIRE(M > 0 && C > 0 && x && out, AVID_E_BADARG, "colsum: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(C, 64)), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long long)M, C);
  return check_launch("colsum");
}
