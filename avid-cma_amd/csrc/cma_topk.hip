// CMA correspondence search (criterions/avid_cma.py:42-73) — SURVEY.md §8(f) rank 1.
//
// For a batch of nq query rows q: sim[n][q] = combine(<V[n],V[q]>, <A[n],A[q]>) over all N bank rows
// (two fp32-MFMA GEMMs through the conv kernel, the second with a fused min / max epilogue), then the
// top (pos_k + 1) per query by similarity, the best one dropped (assumed self, avid_cma.py:69) and the
// rest sorted by index (avid_cma.py:70).  The [N][nq] score slab is kept small (nq = 64..256) so it
// lives in the 256 MB Infinity Cache between the GEMM that writes it and the scan that reads it.
//
// Selection (banks of >= 4096 rows) = threshold filter, two streaming passes over the slab with lanes = 64
// consecutive queries (a score row is one coalesced 256-B read):
//   1. every (row-split, wave, query) lane takes the maximum of its rows; the K-th largest of a query's
//      P = 4 * splits lane maxima is a lower bound T of its K-th best score (K distinct rows reach it);
//   2. rows with score >= T are appended to a per-query candidate list (a few dozen of N rows);
//   3. one block per query ranks its candidates, keeps the best K, drops the best, sorts by index.
// A query whose candidate list overflows (heavy ties / clustered scores) raises a flag and the exact
// per-lane insertion-list scan + merge below (the path for small banks, where it is cheap) redoes the
// batch: both of its kernels are always launched and return at once when the flag is clear, so the host
// never synchronises.  The insertion lists alone took 6.5 ms per 1024 queries x 240k rows (every wave
// iteration has SOME lane inserting while the lists warm up: 64 queries share a wave); the filter takes
// ~0.7 ms.  Ties are ordered (value desc, index asc) — the reference's torch.topk leaves tie order unspecified.
#include <math.h>

#include "common.h"

namespace avid {

constexpr int TK_MAX = 64;       // pos_k + 1 <= 64
constexpr int TK_SPLITS = 64;    // row splits per query group

__device__ __forceinline__ bool better(float v, int i, float ev, int ei) { return v > ev || (v == ev && i < ei); }

constexpr int TK_CAP = 1024;     // candidate slots per query of the threshold filter

// grid = (nq / 64, splits); block = 256 = 4 waves; wave w takes rows r0 + w, r0 + w + 4, ... of its split.
// pmax[(split * 4 + wave)][nq]
__global__ __launch_bounds__(256) void topk_max_kernel(const float* __restrict__ sim, long long N, int nq,
                                                       float* __restrict__ pmax) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane;
  const long long per = (N + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per;
  const long long r1 = r0 + per < N ? r0 + per : N;
  float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
  long long r = r0 + wave;
  for (; r + 12 < r1; r += 16) {
    m0 = fmaxf(m0, sim[r * nq + q]);
    m1 = fmaxf(m1, sim[(r + 4) * nq + q]);
    m2 = fmaxf(m2, sim[(r + 8) * nq + q]);
    m3 = fmaxf(m3, sim[(r + 12) * nq + q]);
  }
  for (; r < r1; r += 4) m0 = fmaxf(m0, sim[r * nq + q]);
  pmax[((long long)blockIdx.y * 4 + wave) * nq + q] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

// one block (256 threads) per query: T = K-th largest of its P <= 256 lane maxima; resets the query's
// candidate counter (and, block 0, the overflow flag)
__global__ __launch_bounds__(256) void topk_thresh_kernel(const float* __restrict__ pmax, int P, int nq, int K,
                                                          float* __restrict__ thr, int* __restrict__ count,
                                                          int* __restrict__ flag) {
  __shared__ float v[256];
  const int q = blockIdx.x, i = threadIdx.x;
  v[i] = i < P ? pmax[(long long)i * nq + q] : -INFINITY;
  __syncthreads();
  if (i < P) {
    const float vi = v[i];
    int rank = 0;
    for (int j = 0; j < P; ++j) rank += (v[j] > vi || (v[j] == vi && j < i)) ? 1 : 0;
    if (rank == K - 1) thr[q] = vi;
  }
  if (i == 0) count[q] = 0;
  if (q == 0 && i == 0) *flag = 0;
}

// same traversal as topk_max_kernel: rows with score >= T[q] go to the query's candidate list
__global__ __launch_bounds__(256) void topk_collect_kernel(const float* __restrict__ sim, long long N, int nq,
                                                           const float* __restrict__ thr, int* __restrict__ count,
                                                           float* __restrict__ cval, int* __restrict__ cidx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane;
  const long long per = (N + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per;
  const long long r1 = r0 + per < N ? r0 + per : N;
  const float t = thr[q];
  auto take = [&](float v, long long r) {
    if (v >= t) {
      const int slot = atomicAdd(&count[q], 1);
      if (slot < TK_CAP) {
        cval[(long long)q * TK_CAP + slot] = v;
        cidx[(long long)q * TK_CAP + slot] = (int)r;
      }
    }
  };
  long long r = r0 + wave;
  for (; r + 12 < r1; r += 16) {
    const float v0 = sim[r * nq + q], v1 = sim[(r + 4) * nq + q], v2 = sim[(r + 8) * nq + q],
                v3 = sim[(r + 12) * nq + q];
    if (__any((v0 >= t) | (v1 >= t) | (v2 >= t) | (v3 >= t))) {
      take(v0, r); take(v1, r + 4); take(v2, r + 8); take(v3, r + 12);
    }
  }
  for (; r < r1; r += 4) take(sim[r * nq + q], r);
}

// one block per query: rank the candidates by (value desc, index asc); ranks 1..K-1 sorted by index -> out.
// A list that overflowed raises the flag instead (the exact scan below then redoes the batch).
__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ cval, const int* __restrict__ cidx,
                                                          const int* __restrict__ count, int K, long long q0,
                                                          long long N, int* __restrict__ flag,
                                                          int32_t* __restrict__ out) {
  __shared__ float sv[TK_CAP];
  __shared__ int si[TK_CAP];
  __shared__ int chosen[TK_MAX];
  const int q = blockIdx.x;
  if (q0 + q >= N) return;
  const int n = count[q];
  if (n > TK_CAP || n < K) {      // (n < K cannot happen: K lane maxima reach the threshold)
    if (threadIdx.x == 0) atomicOr(flag, 1);
    return;
  }
  for (int c = threadIdx.x; c < n; c += 256) {
    sv[c] = cval[(long long)q * TK_CAP + c];
    si[c] = cidx[(long long)q * TK_CAP + c];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n; c += 256) {
    const float v = sv[c];
    const int i = si[c];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += better(sv[j], si[j], v, i) ? 1 : 0;
    if (rank < K) chosen[rank] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // drop rank 0 (assumed self), insertion-sort the rest ascending by index
    int32_t* o = out + (long long)q * (K - 1);
    for (int k = 1; k < K; ++k) {
      const int v = chosen[k];
      int j = k - 1;
      while (j > 0 && o[j - 1] > v) {
        o[j] = o[j - 1];
        --j;
      }
      o[j] = v;
    }
  }
}

// grid = (nq / 64, TK_SPLITS); block = 256 = 4 waves; wave w scans rows r0 + w, r0 + w + 4, ...
// (`flag`: when given and clear, the threshold filter above already produced this batch — nothing to do)
__global__ __launch_bounds__(256) void topk_scan_kernel(const float* __restrict__ sim, long long N, int nq, int K,
                                                        float* __restrict__ pval, int* __restrict__ pidx,
                                                        const int* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (flag && *flag == 0) return;
  float* lv = smem;                                         // [4][K][64]
  int* li = reinterpret_cast<int*>(smem + 4 * K * 64);      // [4][K][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane;
  float* mv = lv + wave * K * 64 + lane;
  int* mi = li + wave * K * 64 + lane;
  for (int k = 0; k < K; ++k) {
    mv[k * 64] = -INFINITY;
    mi[k * 64] = 0x7fffffff;
  }
  const long long per = (N + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per;
  const long long r1 = r0 + per < N ? r0 + per : N;
  float thr_v = -INFINITY;   // current K-th best of this lane's list
  int thr_i = 0x7fffffff;
  for (long long r = r0 + wave; r < r1; r += 4) {
    const float v = sim[r * nq + q];
    const int idx = (int)r;
    if (better(v, idx, thr_v, thr_i)) {
      // insertion into the descending list (position K-1 is evicted)
      int k = K - 1;
      while (k > 0 && better(v, idx, mv[(k - 1) * 64], mi[(k - 1) * 64])) {
        mv[k * 64] = mv[(k - 1) * 64];
        mi[k * 64] = mi[(k - 1) * 64];
        --k;
      }
      mv[k * 64] = v;
      mi[k * 64] = idx;
      thr_v = mv[(K - 1) * 64];
      thr_i = mi[(K - 1) * 64];
    }
  }
  const long long slot = ((long long)blockIdx.y * 4 + wave) * nq + q;   // [P][nq][K]
  for (int k = 0; k < K; ++k) {
    pval[slot * K + k] = mv[k * 64];
    pidx[slot * K + k] = mi[k * 64];
  }
}

// one block per query: K rounds of block-wide arg-best over the P*K candidates, then drop the best and
// sort the remaining indices ascending.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                                         int P, int nq, int K, long long q0, long long N,
                                                         int32_t* __restrict__ out, const int* __restrict__ flag,
                                                         int32_t* __restrict__ fallbacks) {
  if (flag && *flag == 0) return;
  if (flag && fallbacks && blockIdx.x == 0 && threadIdx.x == 0) *fallbacks += 1;   // this batch overflowed its filter
  __shared__ float sv[256];
  __shared__ int si[256], sp[256];
  __shared__ int chosen[TK_MAX];
  __shared__ unsigned char taken[TK_SPLITS * 4 * TK_MAX];
  const int q = blockIdx.x;
  if (q0 + q >= N) return;
  const int ncand = P * K;
  for (int c = threadIdx.x; c < ncand; c += 256) taken[c] = 0;
  __syncthreads();
  for (int round = 0; round < K; ++round) {
    float bv = -INFINITY;
    int bi = 0x7fffffff, bp = -1;
    for (int c = threadIdx.x; c < ncand; c += 256) {
      if (taken[c]) continue;
      const int pslot = c / K, k = c - pslot * K;
      const long long o = ((long long)pslot * nq + q) * K + k;
      const float v = pval[o];
      const int i = pidx[o];
      if (bp < 0 || better(v, i, bv, bi)) {
        bv = v; bi = i; bp = c;
      }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi; sp[threadIdx.x] = bp;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        const int o = threadIdx.x + s;
        if (sp[o] >= 0 && (sp[threadIdx.x] < 0 || better(sv[o], si[o], sv[threadIdx.x], si[threadIdx.x]))) {
          sv[threadIdx.x] = sv[o]; si[threadIdx.x] = si[o]; sp[threadIdx.x] = sp[o];
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      chosen[round] = si[0];
      if (sp[0] >= 0) taken[sp[0]] = 1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // drop rank 0 (assumed self), insertion-sort the rest ascending by index
    int32_t* o = out + (long long)q * (K - 1);
    for (int k = 1; k < K; ++k) {
      const int v = chosen[k];
      int j = k - 1;
      while (j > 0 && o[j - 1] > v) {
        o[j] = o[j - 1];
        --j;
      }
      o[j] = v;
    }
  }
}

}  // namespace avid

using namespace avid;

static int scan_splits(int64_t N) {
  int s = (int)(N / 2048);
  if (s < 1) s = 1;
  if (s > TK_SPLITS) s = TK_SPLITS;
  return s;
}

// threshold filter: row splits (0 = bank too small for it: P = 4 * splits lane maxima must cover K)
static int filter_splits(int64_t N) {
  if (N < 4096) return 0;
  int s = (int)(N / 256);
  return s > TK_SPLITS ? TK_SPLITS : s;
}
// floats of the filter's scratch: lane maxima [P][nq], thresholds, counters + flag, candidate lists
static size_t filter_floats(int64_t N, int nq) {
  const int S = filter_splits(N);
  return S ? (size_t)S * 4 * nq + 2 * (size_t)nq + 64 + 2 * (size_t)nq * TK_CAP : 0;
}

extern "C" size_t avid_cma_topk_workspace_bytes(int64_t N, int nq, int pos_k) {
  if (N <= 0 || nq <= 0 || nq % 64 || pos_k <= 0 || pos_k + 1 > TK_MAX) return 0;
  const size_t P = (size_t)scan_splits(N) * 4;
  return sizeof(float) * ((size_t)N * nq + filter_floats(N, nq)) +
         (sizeof(float) + sizeof(int)) * P * nq * (pos_k + 1) + 256;
}

extern "C" int avid_cma_topk(int64_t N, int D, const float* view1, const float* view2, int64_t q0, int nq, int pos_k,
                             int kind, int32_t* out, int32_t* fallbacks, void* ws, size_t ws_bytes,
                             avid_stream_t stream) {
  AVID_REQUIRE(N > 0 && view1 && view2 && out && ws, AVID_E_BADARG, "cma_topk: bad argument");
  AVID_REQUIRE(D % 32 == 0 && nq > 0 && nq % 64 == 0, AVID_E_UNSUPPORTED, "cma_topk: D %% 32 and nq %% 64 required");
  AVID_REQUIRE(pos_k > 0 && pos_k + 1 <= TK_MAX && pos_k < N, AVID_E_UNSUPPORTED, "cma_topk: pos_k must be in [1, %d]",
               TK_MAX - 1);
  AVID_REQUIRE(kind >= 0 && kind <= 3, AVID_E_BADARG, "cma_topk: kind must be 0..3");
  AVID_REQUIRE(q0 >= 0 && q0 + nq <= N, AVID_E_BADARG, "cma_topk: query range [%lld, %lld) outside the bank",
               (long long)q0, (long long)(q0 + nq));
  AVID_REQUIRE(ws_bytes >= avid_cma_topk_workspace_bytes(N, nq, pos_k), AVID_E_BADARG, "cma_topk: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int K = pos_k + 1;
  const int S = scan_splits(N), P = S * 4;
  float* sim = static_cast<float*>(ws);
  float* pval = sim + (size_t)N * nq;
  int* pidx = reinterpret_cast<int*>(pval + (size_t)P * nq * K);
  int rc;
  // similarity slab [N][nq]
  if (kind == 2 || kind == 3) {
    const float* bank = kind == 2 ? view1 : view2;
    rc = sim_gemm_nt(bank, bank + q0 * D, sim, nullptr, 0, N, nq, D, s);
  } else {
    rc = sim_gemm_nt(view1, view1 + q0 * D, sim, nullptr, 0, N, nq, D, s);
    if (!rc) rc = sim_gemm_nt(view2, view2 + q0 * D, sim, sim, kind == 0 ? 1 : 2, N, nq, D, s);
  }
  if (rc) return rc;
  const size_t lds = (sizeof(float) + sizeof(int)) * 4 * K * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_scan_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)((sizeof(float) + sizeof(int)) * 4 * TK_MAX * 64));
    attr_set = true;
  }
  const int* flag = nullptr;
  const int FS = filter_splits(N);
  if (FS) {
    float* pmax = reinterpret_cast<float*>(pidx + (size_t)P * nq * K);
    float* thr = pmax + (size_t)FS * 4 * nq;
    int* count = reinterpret_cast<int*>(thr + nq);
    int* fl = count + nq;
    float* cval = reinterpret_cast<float*>(fl + 64);
    int* cidx = reinterpret_cast<int*>(cval + (size_t)nq * TK_CAP);
    hipLaunchKernelGGL(topk_max_kernel, dim3(nq / 64, FS), dim3(256), 0, s, sim, (long long)N, nq, pmax);
    hipLaunchKernelGGL(topk_thresh_kernel, dim3(nq), dim3(256), 0, s, pmax, FS * 4, nq, K, thr, count, fl);
    hipLaunchKernelGGL(topk_collect_kernel, dim3(nq / 64, FS), dim3(256), 0, s, sim, (long long)N, nq, thr, count, cval,
                       cidx);
    hipLaunchKernelGGL(topk_select_kernel, dim3(nq), dim3(256), 0, s, cval, cidx, count, K, (long long)q0,
                       (long long)N, fl, out);
    rc = check_launch("topk_filter");
    if (rc) return rc;
    flag = fl;
  }
  hipLaunchKernelGGL(topk_scan_kernel, dim3(nq / 64, S), dim3(256), lds, s, sim, (long long)N, nq, K, pval, pidx, flag);
  rc = check_launch("topk_scan");
  if (rc) return rc;
  hipLaunchKernelGGL(topk_merge_kernel, dim3(nq), dim3(256), 0, s, pval, pidx, P, nq, K, (long long)q0, (long long)N,
                     out, flag, fallbacks);
  return check_launch("topk_merge");
}
