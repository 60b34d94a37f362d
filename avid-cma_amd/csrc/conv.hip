// Implicit-GEMM convolution for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   forward : y[m][n]  = sum_{tap,c} x[pix(m,tap)][c] * w[n][tap][c]           (MODE 0)
//   dgrad   : dx[m][c] = sum_{tap,n} dy[pixT(m,tap)][n] * wT[c][tap][n]        (MODE 1, wT = repacked w)
//   wgrad   : dw[n][tap][c] = sum_m dy[m][n] * x[pix(m,tap)][c]                (split over m, tree-reduced)
//
// Activations are channels-last, so a GEMM-A row is 32 contiguous floats of one (possibly padded)
// input pixel: one 128-B line per 8 lanes, staged through LDS as [row][36] (32 + 4 pad) so the
// MFMA fragment reads are conflict-free ds_read_b128.  The MFMA k-index is free to permute as long
// as A and B agree, so lane-half h consumes k = 8g + 4h + s: one b128 read feeds four MFMAs.
//
// Staging uses buffer loads: a padded / out-of-image row gets a voffset past num_records and the
// hardware returns zeros — no select on the loaded data, so the loads of tile k+1 stay in flight
// under the 32..64 MFMAs of tile k (register prefetch, double-buffered LDS, one barrier per tile).
// The per-row tap validity is a 27-bit mask computed once per workgroup; the loader is straight-line.
//
// Small-M layers (conv4x/conv5x, audio, heads) keep the big tile and split GEMM-K across workgroups
// (grid.y), summing the fp32 partial slabs in a fixed order in a reduce kernel that also applies the
// epilogue (deterministic; no atomics).
//
// Reference ops replaced: nn.Conv3d/Conv2d/Linear fwd+bwd — models/video.py:20,
// models/network_blocks.py:18,20,35,37,40,42,49, models/audio.py:22, models/av_wrapper.py:25.
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"

namespace avid {

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* __restrict__ src;     // [B,Ts,Hs,Ws,Cs] channels-last (vector path) or strided (gather path)
  const float* __restrict__ wk;      // [Cd][ntaps][Cs], rows w_row floats apart (a trimmed tap range of a wider row)
  int w_row, w_nrec;                 // floats per weight row; bytes addressable from wk
  const float* __restrict__ addend;  // [M][Cd] or null
  const float* __restrict__ bias;    // [Cd] or null
  float* __restrict__ dst;           // [M][Cd]
  float* __restrict__ part;          // split-K partial slabs [nsplit][M][Cd] (nsplit > 1)
  int B, Ts, Hs, Ws, Cs;
  int Td, Hd, Wd, Cd;
  int kt, kh, kw;
  int st, sh, sw;
  int pt, ph, pw;
  int M;      // B*Td*Hd*Wd
  int mode;   // 0: s = d*stride - pad + tap ;  1: s = (d + pad - tap) / stride (exact)
  int relu;
  int epi_op;  // how `addend` combines: 0 add, 1 min, 2 max (CMA agreement scores)
  int nsplit, ksteps_per_split;
  int mt2_begin, mt2_count;           // window of M pair-tiles this launch covers (tail launches split K)
  int part_row_begin;                 // first destination row of the partial slabs
  long long ssB, ssT, ssH, ssW, ssC;  // gather-path source strides (elements)
  // Strided dgrad only: destination pixels are processed per stride-parity class (only taps of matching
  // parity contribute), each class a dense sub-problem with its own tap subset.
  int ncls, cls_ptiles_total;
  int cls_begin[9];                     // first pair-tile of class c (prefix sums)
  int cls_p0[8][3], cls_n[8][3];        // first position / position count per (t,h,w)
  int cls_d0[8][3], cls_nd[8][3];       // first tap / tap count per (t,h,w) (tap step = stride)
  unsigned mgW, mgH, mgT;               // multiply-shift division by Wd, Hd, Td (igemm_pk_kernel)
  int shW, shH, shT;
  // igemm_pk_kernel work list: tiles [0, pk_full) go round-robin to the workgroups as whole tiles; the
  // pk_tail_units = tail_tiles * pk_f units after them are (tile, K-range) pieces, one per workgroup, that
  // write partial slabs to `part` (rows [part_row_begin, M))
  int pk_full, pk_tail_units, pk_f, pk_kps;
  unsigned cls_mg[8][3];   // igemm_pk_kernel<STRIDED>: multiply-shift division by a class's (T,H,W) extents
  int cls_shf[8][3];
  // Strided dgrad: `addend` may be COMPACT — the gradient of a sub-sampled view x[:, ::add_s[0], ::add_s[1], ::add_s[2]]
  // (the input gradient of the block's 1x1x1 strided residual convolution), shape [B][add_n[0]][add_n[1]][add_n[2]][Cd]:
  // it contributes only at positions divisible by the strides.  add_s = {1,1,1}: the usual dx-shaped addend.
  int add_s[3], add_n[3];
  int cls_f[8];            // igemm_pk_kernel<STRIDED>: K pieces per tile of class c (1: tiles written directly)
  int cls_ubegin[9];       // ... and the first work unit (tile, piece) of class c (prefix sums; [ncls] = total)
  float* stats;    // BatchNorm partial sums [rows][2][Cd] of the output (igemm_pk_kernel forward), or null
  // dgrad whose output is the gradient of a BatchNorm(+ReLU) output: the BN's backward partial sums
  // (sum dy_m, sum dy_m * xhat; dy_m = dy masked by the recomputed ReLU) go to `stats` from the epilogue
  const float* bnb_x;        // the BN's input (this dgrad's dx has its shape), or null
  const float *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd;
  int bnb_relu;
  int pk_rot;      // tail unit u runs on workgroup (u + pk_rot) mod G: the ones that got one full tile less
  int pk_paired;   // grid = 2 workgroups per CU: number them so that v and v + G/2 share a CU
  // Weight-stationary order of the tail units (0: tile-major — unit u = (tile u / f, piece u % f): the pieces and column blocks
  // of an M-tile are neighbours; > 0 = the tail's M-tile count: unit u = (M-tile u % mtt, piece (u / mtt) % f, column block
  // u / mtt / f): the M-tiles of one (column block, K piece) are neighbours — consecutive units sit on one XCD, so each weight
  // byte is fetched into ONE L2 and the (small) activations are what the XCDs replicate: plan_pk_order)
  int pk_ws;
  // igemm_pk_kernel<..., BS>: the weights pre-split into three bf16 terms, in the kernel's operand-fragment order
  // (avid_wt_desc mode 5 / 6): base of the first live tap's chunks, bytes addressable from it, bytes between two k-tiles
  const void* wsp;
  int wsp_nrec, wsp_kstep;
  int stats_rows;  // tconv64_kernel: rows of `stats` the caller will fold (avid_conv_fwd_stats_rows / avid_conv_dgrad_bn_rows)
  // avid_conv_fwd_in: `src` is the INPUT of a BatchNorm (+ReLU) whose output this convolution consumes; the loader applies
  // fma(x, in_scale[c], in_shift[c]) (+ max(., 0)) to every element it stages — bn_apply_kernel's expression, bit for bit — so
  // the normalised tensor is never written (tconv64_kernel only; null = src is read as it is)
  const float* in_scale;
  const float* in_shift;
  int in_relu;
};

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row (floats): 144 B => b128 fragment reads conflict-free
constexpr int KTAB_MAX = 512;
constexpr unsigned OOB = 0x80000000u;  // voffset beyond any num_records (< 2 GiB): buffer load returns 0

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

__device__ __forceinline__ floatx4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

// ------------------------------------------------------------------------------------------------
// Vector path (Cs % 32 == 0, channels-last source): all layers except the two stems.
//
// Ping-pong schedule.  A workgroup is 8 waves = two independent groups of 4 waves, each owning its own
// BM x BN output tile (consecutive M-tiles) and its own single-buffered LDS stage.  The groups run
// the same loop shifted by one phase:
//     phase A : ds_read fragments + 32..64 MFMAs of tile k          (matrix pipe)
//     phase B : ds_write tile k+1 (prefetched in registers), issue buffer loads of tile k+2
// so on every SIMD exactly one wave is in its MFMA phase while its partner stages — the matrix pipe
// stays busy without relying on two co-resident workgroups happening to drift out of phase
// (measured before: both in phase => pipe 52 % busy, SQ_WAIT_INST_ANY 60 %).
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int MODE, bool STRIDED = false>
__global__ __launch_bounds__(512) void igemm_kernel(const ConvArgs p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int PA = BM / 32, PB = BN / 32;
  static_assert(!STRIDED || MODE == 1, "parity classes are a dgrad construct");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);   // wave-uniform by construction
  float* As = smem + grp * (BM + BN) * LDK;   // [BM][LDK]
  float* Bs = As + BM * LDK;                  // [BN][LDK]
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;

  // XCD-aware tile order: consecutive M-tiles (which share input halos) stay on one XCD's L2.
  const unsigned ntn = p.Cd / BN;
  const unsigned ntm2 = STRIDED ? (unsigned)p.cls_ptiles_total : (unsigned)p.mt2_count, ntiles = ntm2 * ntn;
  const unsigned lin = xcd_remap(blockIdx.x, ntiles * p.nsplit);
  const unsigned split = lin / ntiles, tile = lin - split * ntiles + (STRIDED ? 0u : (unsigned)p.mt2_begin * ntn);
  const int n0 = (tile % ntn) * BN;
  int m0 = ((tile / ntn) * 2 + grp) * BM;     // first row of this group's tile (class-local row when STRIDED)

  const int ntaps = p.kt * p.kh * p.kw;
  const int cpt = p.Cs / BK;
  // class geometry (trivial single class unless STRIDED)
  int cT = p.Td, cH = p.Hd, cW = p.Wd, cM = p.M;
  int p0t = 0, p0h = 0, p0w = 0, d0t = 0, d0h = 0, d0w = 0, ndt = p.kt, ndh = p.kh, ndw = p.kw;
  if (STRIDED) {
    const int ptile = tile / ntn;
    int c = 0;
    while (c + 1 < p.ncls && ptile >= p.cls_begin[c + 1]) ++c;
    m0 = ((ptile - p.cls_begin[c]) * 2 + grp) * BM;
    cT = p.cls_n[c][0]; cH = p.cls_n[c][1]; cW = p.cls_n[c][2];
    cM = p.B * cT * cH * cW;
    p0t = p.cls_p0[c][0]; p0h = p.cls_p0[c][1]; p0w = p.cls_p0[c][2];
    d0t = p.cls_d0[c][0]; d0h = p.cls_d0[c][1]; d0w = p.cls_d0[c][2];
    ndt = p.cls_nd[c][0]; ndh = p.cls_nd[c][1]; ndw = p.cls_nd[c][2];
  }
  const int nk_total = STRIDED ? ndt * ndh * ndw * cpt : ntaps * cpt;
  const int ks0 = split * p.ksteps_per_split;
  const int ks1 = min(ks0 + p.ksteps_per_split, nk_total);
  int* drow = reinterpret_cast<int*>(smem + 2 * (BM + BN) * LDK) + grp * BM;   // STRIDED: tile row -> dst row

  // Buffer descriptors.  A: based at the first batch item this group touches, so 32-bit byte
  // offsets are enough for any tensor size.  B: the (small) weight tensor.
  const int pix_per_b = p.Ts * p.Hs * p.Ws;
  int b_lo = m0 / (cT * cH * cW);
  if (b_lo >= p.B) b_lo = p.B - 1;            // group past the end of M: every row is masked anyway
  const long long a_base = (long long)b_lo * pix_per_b * p.Cs;
  long long a_bytes = ((long long)p.B * pix_per_b * p.Cs - a_base) * 4;
  if (a_bytes > 0x7fffffffll) a_bytes = 0x7fffffffll;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + a_base), 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wk, 0, p.w_nrec, 0x00020000);

  // ---- per-thread row bookkeeping (rows are fixed for the whole K loop)
  int a_t0[PA], a_h0[PA], a_w0[PA];
  unsigned a_off[PA], a_mask[PA];
  const int cs4 = p.Cs * 4;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int b, td, hd, wd;
    bool ok;
    decode_row(m0 + lrow + 32 * i, cM, cW, cH, cT, b, td, hd, wd, ok);
    if (STRIDED) {   // class-local coordinates -> destination pixel
      td = p0t + td * p.st; hd = p0h + hd * p.sh; wd = p0w + wd * p.sw;
      if ((tid & 7) == 0) drow[lrow + 32 * i] = ok ? ((b * p.Td + td) * p.Hd + hd) * p.Wd + wd : -1;
    }
    if (MODE == 0) {
      a_t0[i] = td * p.st - p.pt;
      a_h0[i] = hd * p.sh - p.ph;
      a_w0[i] = wd * p.sw - p.pw;
    } else {
      a_t0[i] = td + p.pt;
      a_h0[i] = hd + p.ph;
      a_w0[i] = wd + p.pw;
    }
    // per-dimension tap validity (bit d set <=> tap offset d lands inside the source along that axis)
    unsigned mt = 0, mh = 0, mw = 0;
    for (int dt = 0; dt < p.kt; ++dt) {
      int ts = MODE == 0 ? a_t0[i] + dt : a_t0[i] - dt;
      bool v = ts >= 0;
      if (MODE == 1) { v &= (ts & (p.st - 1)) == 0; ts >>= (p.st - 1); }
      mt |= ((v & (ts < p.Ts)) ? 1u : 0u) << dt;
    }
    for (int dh = 0; dh < p.kh; ++dh) {
      int hs = MODE == 0 ? a_h0[i] + dh : a_h0[i] - dh;
      bool v = hs >= 0;
      if (MODE == 1) { v &= (hs & (p.sh - 1)) == 0; hs >>= (p.sh - 1); }
      mh |= ((v & (hs < p.Hs)) ? 1u : 0u) << dh;
    }
    for (int dw = 0; dw < p.kw; ++dw) {
      int ws = MODE == 0 ? a_w0[i] + dw : a_w0[i] - dw;
      bool v = ws >= 0;
      if (MODE == 1) { v &= (ws & (p.sw - 1)) == 0; ws >>= (p.sw - 1); }
      mw |= ((v & (ws < p.Ws)) ? 1u : 0u) << dw;
    }
    a_mask[i] = ok ? (mt | (mh << 8) | (mw << 16)) : 0u;
    if (MODE == 0)  // pixel offset of tap (0,0,0); taps add a uniform offset
      a_off[i] = (unsigned)((((b - b_lo) * p.Ts + a_t0[i]) * p.Hs + a_h0[i]) * p.Ws + a_w0[i]) * cs4 + lcol * 4;
    else
      a_off[i] = (unsigned)((b - b_lo) * pix_per_b) * cs4 + lcol * 4;
  }
  unsigned b_off[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) b_off[i] = (unsigned)((n0 + lrow + 32 * i) * p.w_row + lcol) * 4;

  floatx4 va[PA], vb[PB];

  auto load_tile = [&](int ks) {
    const int tapi = ks / cpt, c0 = (ks - tapi * cpt) * BK;
    int dw = tapi % ndw, r = tapi / ndw;
    int dh = r % ndh, dt = r / ndh;
    if (STRIDED) { dt = d0t + dt * p.st; dh = d0h + dh * p.sh; dw = d0w + dw * p.sw; }
    const int tap = (dt * p.kh + dh) * p.kw + dw;
    const unsigned tap_off = (MODE == 0 ? (unsigned)(((dt * p.Hs + dh) * p.Ws + dw) * cs4) : 0u) + c0 * 4;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      unsigned off;
      if (MODE == 0) {
        off = a_off[i] + tap_off;
      } else {
        const int ts = (a_t0[i] - dt) >> (p.st - 1), hs = (a_h0[i] - dh) >> (p.sh - 1),
                  ws = (a_w0[i] - dw) >> (p.sw - 1);
        off = a_off[i] + (unsigned)((ts * p.Hs + hs) * p.Ws + ws) * cs4 + tap_off;
      }
      const unsigned okb = (a_mask[i] >> dt) & (a_mask[i] >> (8 + dh)) & (a_mask[i] >> (16 + dw)) & 1u;
      off = okb ? off : OOB;
      va[i] = buf_load4(rsA, off);
    }
    const unsigned wtap = (unsigned)(tap * p.Cs + c0) * 4;
#pragma unroll
    for (int i = 0; i < PB; ++i) vb[i] = buf_load4(rsB, b_off[i] + wtap);
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i) *reinterpret_cast<floatx4*>(&As[(lrow + 32 * i) * LDK + lcol]) = va[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<floatx4*>(&Bs[(lrow + 32 * i) * LDK + lcol]) = vb[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int h = lane >> 5, l31 = lane & 31;
  const float* Ab = As + (wm * TM * 32 + l31) * LDK + h * 4;
  const float* Bb = Bs + (wn * TN * 32 + l31) * LDK + h * 4;

  // prologue: tile ks0 -> LDS, tile ks0+1 -> registers  (a parity class may have no taps at all)
  if (ks0 < ks1) {
    load_tile(ks0);
    store_tile();
    if (ks0 + 1 < ks1) load_tile(ks0 + 1);
  }
  __syncthreads();
  if (grp == 1) __syncthreads();   // shift group 1 by one phase

  for (int ks = ks0; ks < ks1; ++ks) {
    // ---- phase A: matrix pipe.  Fragments of k-group g+1 are fetched BEFORE the MFMAs of group g are
    // issued (order pinned with sched_barrier: the compiler otherwise sinks the ds_reads to just before
    // their use and exposes the LDS latency four times per phase).
    floatx4 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[(g + 1) & 1][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + (g + 1) * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[(g + 1) & 1][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + (g + 1) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // ---- phase B: stage (the partner group is in its phase A)
    if (ks + 1 < ks1) {
      store_tile();
      if (ks + 2 < ks1) load_tile(ks + 2);
    }
    __syncthreads();
  }
  if (grp == 0) __syncthreads();   // balance the barrier count of the two groups

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool direct = p.nsplit == 1;
  float* outp = direct ? p.dst : p.part + ((long long)split * (p.M - p.part_row_begin) - p.part_row_begin) * p.Cd;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      const float bv = (direct && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int trow = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int row = STRIDED ? drow[trow] : m0 + trow;
        if (STRIDED ? row >= 0 : row < p.M) {
          const long long o = (long long)row * p.Cd + col;
          float v = acc[i][j][r] + bv;
          if (direct) {
            if (p.addend) {
              const float ad = p.addend[o];
              v = p.epi_op == 0 ? v + ad : (p.epi_op == 1 ? fminf(v, ad) : fmaxf(v, ad));
            }
            if (p.relu) v = fmaxf(v, 0.f);
          }
          outp[o] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent, K-pipelined variant for the big-M layers (conv2x / conv3x forward + unit-stride dgrad).
//
// What the hardware rewards (tools/mfma_peak.hip, tools/gemm_lab.hip): an MFMA blocks its own wave's
// instruction stream for its 64 cycles, so everything that is not an MFMA (address math, LDS and VMEM
// issue) is paid once per wave and only overlaps with the MFMAs of OTHER waves on the SIMD.  A plain GEMM
// staged exactly like this kernel reaches 122-134 TFLOP/s; the convolution gets there by making its
// loader as cheap as the GEMM's:
//   * per-row voffsets are recomputed only when the filter tap changes (every Cs/32 k-tiles); the channel
//     block and the weight tap ride in the scalar soffset, so a k-tile issues PA+PB loads and no VALU;
//   * row decode uses multiply-shift division (magic numbers from the host), once per tile;
//   * the epilogue is buffer stores whose row step rides in soffset and whose bound is num_records.
// The workgroup is persistent: it walks tiles slot, slot+G, ... as one flattened (tile, k-tile) stream
// whose loader runs two k-tiles ahead ACROSS tile boundaries (LDS double-buffered, one barrier per k-tile),
// so a tile's row decode and first loads hide under the previous tile's last MFMAs and its stores drain
// under the next tile's.  The host hands it whole rounds of tiles; the remaining rows go to igemm_kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned magic_div(unsigned n, unsigned magic, int shift) {
  return (unsigned)(((unsigned long long)n * magic) >> shift);
}

// EPI: the epilogue of directly written tiles, fixed at compile time (one variant per instantiation):
//   forward  0 plain   1 + addend   2 + bias   3 + bias, ReLU   4 min(., addend)   5 max(., addend)
//   dgrad    0 plain   1 + addend   8 BatchNorm-backward sums   9 + addend and those sums
//   15       any other combination, told apart at run time (the pre-specialisation code)
// With several LOADING variants in one kernel the compiler hoisted their common loads above the variant branch;
// on the paths that do not consume them they stayed "pending" into the k-loop header, and the waitcnt pass then
// put `s_waitcnt vmcnt(1)` in front of the first MFMA of EVERY k-tile (the tile loads were supposed to be
// waited for one by one, behind the first six MFMAs): 2-7 % of every dense kernel.
constexpr int EPI_ANY = 15;
#ifdef AVID_PK_TRACE
__device__ long long g_pk_trace[1024 * 64];
extern "C" int avid_debug_pk_trace(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pk_trace), sizeof(long long) * 1024 * 64);
}
#define PK_STAMP(idx) do { if (threadIdx.x == 0 && (idx) < 32) { g_pk_trace[blockIdx.x * 64 + (idx)] = wall_clock64(); g_pk_trace[blockIdx.x * 64 + 32 + (idx)] = clock64(); } } while (0)
#else
#define PK_STAMP(idx) do {} while (0)
#endif
// Split-bf16 products (round 4): every fp32 operand as three bf16 terms, six v_mfma_f32_32x32x16_bf16 per
// product tile — fp32 accuracy (error against fp64 at or below the fp32 instruction's) at 6/16 of its issue time.  Build
// with -DAVID_PK_FP32 for the exact-fp32 instruction (tools/build_variant.sh: A/B on one box).
#ifdef AVID_PK_FP32
constexpr bool PK_SPLIT = false;
#else
constexpr bool PK_SPLIT = true;
#endif
typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pk_uintx4 __attribute__((ext_vector_type(4)));
// DOT: the residuals from v_dot2c_f32_bf16 (common.h split2_bf16_dot: 7 instead of 9 instructions per pair; igemm_pk_kernel's
// loops gain 6 %, tconv64_kernel's hand-ordered one loses 6 %)
template <bool DOT = true>
__device__ __forceinline__ void pk_split8(const floatx4& v0, const floatx4& v1, pk_bf16x8& fh, pk_bf16x8& fm, pk_bf16x8& fl) {
  const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (DOT) split2_bf16_dot(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
    else split2_bf16(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
  }
  fh = __builtin_bit_cast(pk_bf16x8, pk_uintx4{h[0], h[1], h[2], h[3]});
  fm = __builtin_bit_cast(pk_bf16x8, pk_uintx4{m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(pk_bf16x8, pk_uintx4{l[0], l[1], l[2], l[3]});
}

// BS: the weight operand arrives pre-split (ConvArgs::wsp): one (k-tile, 64 rows) chunk is PK_BCH contiguous bytes
// [row block j of 32][k-step st of 16][term: hi, mid, lo][lane][8 bf16] — a straight 16-byte-per-thread copy into LDS, read
// back as whole fragments (lane * 16: conflict-free), no split instructions for this operand.
constexpr int PK_BCH = 2 * 2 * 3 * 1024;
template <int WM, int WN, int TM, int TN, int MODE, bool STRIDED = false, int EPI = EPI_ANY, bool BS = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void igemm_pk_kernel(const ConvArgs p) {
  static_assert(!STRIDED || MODE == 1, "parity classes are a dgrad construct");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RPP = NT / 8;                   // rows staged per pass: 8 lanes x 16 B cover a 32-float row
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  static_assert(!BS || (NT == 256 && (BN == 64 || BN == 128) && PK_SPLIT), "pre-split weights: the 4-wave tiles of the split-bf16 build");
  constexpr int BCH = (BN / 64) * PK_BCH;       // bytes of a pre-split (k-tile, column block) chunk: one PK_BCH per 64 columns
  constexpr int PB3 = BCH / (NT * 16);          // 16-byte items of a pre-split chunk per thread
  // BS with the 128-column tile: the input stage unpadded (32 floats per row) with its 16-byte pieces XOR-swizzled by bits 1-3 of
  // the row — 16 KB + 24 KB of weight fragments = 40 KB per stage, two stages = 80 KB: TWO workgroups in the CU's 160 KB (the padded
  // rows made it 43 KB and one workgroup: round 5, first form).  A read phase (16 consecutive rows, one piece each) and a store
  // pass (8 rows x 8 pieces per wave) both touch every bank once.
  constexpr bool ASWZ = BS && BN == 128;
  constexpr int ALD = ASWZ ? BK : LDK;
  constexpr int STAGE = BS ? BM * ALD + BCH / 4 : (BM + BN) * LDK;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // The class tables of the strided input gradient are indexed with run-time class numbers.  Read through `p` (the by-value
  // argument = a private copy the optimiser has to dissolve) such an index keeps the copy alive whenever the optimiser's
  // forwarding to the argument block gives up — the whole 1 KB argument then lives in scratch (1056 B of private segment, seen
  // with tools/kernel_resources.py after an unrelated edit).  So they are read where they are: from the kernel-argument segment
  // itself (constant address space, scalar loads at computed offsets); `p` is only ever accessed at constant offsets.
  typedef const ConvArgs __attribute__((address_space(4)))* KArgs;
  const KArgs kp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const int h = lane >> 5, l31 = lane & 31;
  PK_STAMP(0);

  const int ntn = p.Cd / BN;
  const int G = gridDim.x;
  // Workgroup numbering.  Hardware deals consecutive ids round-robin to the 8 XCDs and, inside an XCD, one
  // per CU before doubling up.  Logical slots keep 32 consecutive tiles on one XCD (shared L2 halos) and,
  // for a 2-per-CU grid, put the CU-mates at v and v + G/2 — the planner balances work per CU with that.
  int slot_;
  if (p.pk_paired) {
    const int per = G >> 4;                               // CUs per XCD
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int second = pos >= per ? 1 : 0;
    slot_ = second * (G >> 1) + xcd * per + (pos - second * per);
  } else {
    slot_ = (int)xcd_remap(blockIdx.x, G);
  }
  const int slot = __builtin_amdgcn_readfirstlane(slot_);
  const int ntaps = p.kt * p.kh * p.kw;
  const int cpt = p.Cs / BK;
  const int nk = ntaps * cpt;
  const int cs4 = p.Cs * 4;
  const int pix_per_b = p.Ts * p.Hs * p.Ws;
  const int pix_d = p.Td * p.Hd * p.Wd;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      BS ? const_cast<void*>(p.wsp) : (void*)p.wk, 0, BS ? p.wsp_nrec : p.w_nrec, 0x00020000);

  // ---- this workgroup's segments: whole tiles slot, slot+G, ... below pk_full, then at most one
  // (tile, K-range) piece of the split tail
  const int n_full = slot < p.pk_full ? (p.pk_full - slot + G - 1) / G : 0;
  int tslot = slot - p.pk_rot;
  tslot += tslot < 0 ? G : 0;
  const int nseg = n_full + (tslot < p.pk_tail_units ? 1 : 0);
  // running column sum / sum of squares of this wave's outputs (p.stats); two lanes of accumulation per column
  // (accumulator registers r even / odd) so that the sums pair up as packed fp32 ops on adjacent registers
  float cs[TN][2], cq[TN][2];
#pragma unroll
  for (int jj = 0; jj < TN; ++jj) cs[jj][0] = cs[jj][1] = cq[jj][0] = cq[jj][1] = 0.f;
  // BatchNorm statistics fused into the epilogue: every workgroup owns one partial row [2][Cd] (its tiles
  // all share one column block — the planner keeps full, rot and the grid multiples of Cd/BN); columns it
  // does not cover, and the rows of workgroups without direct tiles, are written as zeros.
  auto write_stats = [&]() {
    if (!p.stats) return;
    float* red = smem;                                   // [2][WM][BN] (the stages are dead by now)
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      const float a0 = cs[jj][0] + cs[jj][1], b0 = cq[jj][0] + cq[jj][1];
      const float a = a0 + __shfl_xor(a0, 32, 64), b = b0 + __shfl_xor(b0, 32, 64);
      if (h == 0) {
        red[wm * BN + (wn * TN + jj) * 32 + l31] = a;
        red[WM * BN + wm * BN + (wn * TN + jj) * 32 + l31] = b;
      }
    }
    __syncthreads();
    // this workgroup's column block (weight-stationary tails — no full tiles then: plan_pk_order — deal column blocks by unit)
    const int nb = p.pk_ws ? (tslot < p.pk_tail_units ? (tslot / p.pk_ws / p.pk_f) % ntn : 0) : slot % ntn;
    float* row = p.stats + (long long)slot * 2 * p.Cd;
    for (int c = tid; c < p.Cd; c += NT) {
      float a = 0.f, b = 0.f;
      if (c / BN == nb) {
#pragma unroll
        for (int w = 0; w < WM; ++w) { a += red[w * BN + c % BN]; b += red[WM * BN + w * BN + c % BN]; }
      }
      row[c] = a;
      row[p.Cd + c] = b;
    }
  };
  if (nseg == 0) {
    write_stats();
    return;
  }
  // Strided dgrad (STRIDED): the tile list is the concatenation of the stride-parity classes' tile lists
  // (cls_begin, in tiles); a class is a dense unit-stride problem over its own pixel grid and tap subset,
  // so its K extent differs (and may be empty: those tiles only write zeros / the addend).
  auto cls_of = [&](int tile) {
    int c = 0;
    while (c + 1 < p.ncls && tile >= kp->cls_begin[c + 1]) ++c;
    return c;
  };
  auto seg_info = [&](int j, int& tile, int& k0, int& k1, int& split) {
    if (STRIDED) {   // units = (tile, piece of its own K range), all dealt round-robin; pieces per tile vary by class
      const int u = slot + j * G;
      int c = 0;
      while (c + 1 < p.ncls && u >= kp->cls_ubegin[c + 1]) ++c;
      const int fc = kp->cls_f[c];
      const int local = u - kp->cls_ubegin[c];
      const int lt = local / fc;
      const int piece = local - lt * fc;
      tile = kp->cls_begin[c] + lt;
      const int nkc = kp->cls_nd[c][0] * kp->cls_nd[c][1] * kp->cls_nd[c][2] * cpt;
      const int kps = (nkc + fc - 1) / fc;
      k0 = piece * kps < nkc ? piece * kps : nkc;
      k1 = k0 + kps < nkc ? k0 + kps : nkc;
      split = fc > 1 ? piece : -1;
    } else if (j < n_full) {
      tile = slot + j * G; k0 = 0; k1 = nk; split = -1;
    } else {
      int piece;
      if (p.pk_ws) {                            // weight-stationary order (ConvArgs::pk_ws = M-tiles of the tail)
        const int r = tslot / p.pk_ws, mt = tslot - r * p.pk_ws;
        const int nt = r / p.pk_f;
        piece = r - nt * p.pk_f;
        tile = p.pk_full + mt * ntn + nt;
      } else {
        tile = p.pk_full + tslot / p.pk_f;
        piece = tslot % p.pk_f;
      }
      split = p.pk_f > 1 ? piece : -1;          // unsplit tail tiles take the direct epilogue
      k0 = piece * p.pk_kps;
      k1 = k0 + p.pk_kps < nk ? k0 + p.pk_kps : nk;
    }
  };

  // ---- loader state (two k-tiles ahead of the MFMAs, possibly already in the next segment)
  int ld_seg = 0, ld_ks = 0, ld_kend = 0;
  int ld_dt = 0, ld_dh = 0, ld_dw = 0, ld_c0 = 0, ld_tap = 0;   // position in the K loop (tap, channel block)
  int ld_nh = p.kh, ld_nw = p.kw, ld_cls = 0;                  // tap extents of the loader's class (STRIDED)
  int ld_plo = 0, ld_phi = 0, ld_nrec = 0;                     // A descriptor of the loader tile: base pointer, bytes
  int ld_soff_a = 0, ld_soff_b = 0;                            // scalar offsets of the current k-tile (A: channel block; B: tap + block)
  unsigned a_base[PA], a_mask[PA], a_cur[PA];                  // tap-(0,0,0) offset, tap validity, current voffset
  unsigned b_off[PB];
  floatx4 va[PA], vb[PB];
  pk_uintx4 vb3[PB3];                                          // BS: this thread's items of the weight chunk
  int ld_bnt = 0;                                              // BS: byte offset of the loader tile's column block in a k-tile's chunks

  auto range_mask = [](int lo, int hi, int k) -> unsigned {    // bits lo..hi clipped to [0, k)
    lo = lo < 0 ? 0 : lo;
    hi = hi > k - 1 ? k - 1 : hi;
    return hi >= lo ? (2u << hi) - (1u << lo) : 0u;
  };
  auto retap = [&]() {   // voffset of every staged row for the loader's current tap (OOB: tap outside the source)
    const int pixoff = ((ld_dt * p.Hs + ld_dh) * p.Ws + ld_dw) * cs4;
    const unsigned tap_off = (unsigned)(MODE == 0 ? pixoff : -pixoff);
    const unsigned sh_t = ld_dt, sh_h = 8 + ld_dh, sh_w = 16 + ld_dw;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const unsigned okb = (a_mask[i] >> sh_t) & (a_mask[i] >> sh_h) & (a_mask[i] >> sh_w) & 1u;
      a_cur[i] = okb ? a_base[i] + tap_off : OOB;
    }
  };
  auto setup_tile = [&](int tile) {
    int c = 0, cT = p.Td, cH = p.Hd, cW = p.Wd, cM = p.M, cpix = pix_d, lt = tile;
    unsigned mgW = p.mgW, mgH = p.mgH, mgT = p.mgT;
    int shW = p.shW, shH = p.shH, shT = p.shT;
    int offt = p.pt, offh = p.ph, offw = p.pw, ndt = p.kt, ndh = p.kh, ndw = p.kw;
    if (STRIDED) {
      c = ld_cls;
      lt = tile - kp->cls_begin[c];
      cT = kp->cls_n[c][0]; cH = kp->cls_n[c][1]; cW = kp->cls_n[c][2];
      cpix = cT * cH * cW;
      cM = p.B * cpix;
      mgT = kp->cls_mg[c][0]; mgH = kp->cls_mg[c][1]; mgW = kp->cls_mg[c][2];
      shT = kp->cls_shf[c][0]; shH = kp->cls_shf[c][1]; shW = kp->cls_shf[c][2];
      // source coordinate of class tap j for class-local position q: q + off - j  (off is exact by construction)
      offt = (kp->cls_p0[c][0] + p.pt - kp->cls_d0[c][0]) / p.st;
      offh = (kp->cls_p0[c][1] + p.ph - kp->cls_d0[c][1]) / p.sh;
      offw = (kp->cls_p0[c][2] + p.pw - kp->cls_d0[c][2]) / p.sw;
      ndt = kp->cls_nd[c][0]; ndh = kp->cls_nd[c][1]; ndw = kp->cls_nd[c][2];
    }
    const int mt = lt / ntn, nt = lt - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    int b_lo = m0 / cpix;
    if (b_lo >= p.B) b_lo = p.B - 1;
    b_lo = __builtin_amdgcn_readfirstlane(b_lo);
    {   // descriptor based at the first batch item the tile touches: 32-bit offsets for any tensor size
      const long long base = (long long)b_lo * pix_per_b * p.Cs;
      long long a_bytes = ((long long)p.B * pix_per_b * p.Cs - base) * 4;
      if (a_bytes > 0x7fffffffll) a_bytes = 0x7fffffffll;
      const unsigned long long ptr = reinterpret_cast<unsigned long long>(p.src + base);
      ld_plo = __builtin_amdgcn_readfirstlane((int)(unsigned)ptr);
      ld_phi = __builtin_amdgcn_readfirstlane((int)(unsigned)(ptr >> 32));
      ld_nrec = __builtin_amdgcn_readfirstlane((int)a_bytes);
    }
    // The 8 lanes that stage one row (lcol = 0..28) used to decode each of their PA rows themselves: ~65 VALU per
    // row, PA times per tile, all on the wave's critical path between two tiles.  Lane j of such a group now decodes
    // ONE row (i = j mod PA) and the group exchanges the results with PA lane permutes.
    static_assert(PA <= 8, "one decoded row per lane of an 8-lane staging group");
    unsigned dec_base, dec_mask;
    {
      const int i = (tid & 7) % PA;
      const unsigned m = m0 + lrow + RPP * i;
      const bool ok = m < (unsigned)cM;
      const unsigned mm = ok ? m : 0u;
      const unsigned q1 = magic_div(mm, mgW, shW);
      const int wd = mm - q1 * cW;
      const unsigned q2 = magic_div(q1, mgH, shH);
      const int hd = q1 - q2 * cH;
      const int b = magic_div(q2, mgT, shT);
      const int td = q2 - b * cT;
      int t0, h0, w0;
      unsigned mt_, mh, mw;
      if (MODE == 0) {   // source coordinate of tap d: t0 + d
        t0 = td * p.st - p.pt; h0 = hd * p.sh - p.ph; w0 = wd * p.sw - p.pw;
        mt_ = range_mask(-t0, p.Ts - 1 - t0, p.kt);
        mh = range_mask(-h0, p.Hs - 1 - h0, p.kh);
        mw = range_mask(-w0, p.Ws - 1 - w0, p.kw);
      } else {           // dgrad: t0 - d (unit stride, or class-local coordinates of a strided one)
        t0 = td + offt; h0 = hd + offh; w0 = wd + offw;
        mt_ = range_mask(t0 - p.Ts + 1, t0, ndt);
        mh = range_mask(h0 - p.Hs + 1, h0, ndh);
        mw = range_mask(w0 - p.Ws + 1, w0, ndw);
      }
      dec_mask = ok ? (mt_ | (mh << 8) | (mw << 16)) : 0u;
      dec_base = (unsigned)((((b - b_lo) * p.Ts + t0) * p.Hs + h0) * p.Ws + w0) * cs4;
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int src = (lane & ~7) | i;             // the lane of this row's group that decoded row i
      a_mask[i] = (unsigned)__shfl((int)dec_mask, src, 64);
      a_base[i] = (unsigned)__shfl((int)dec_base, src, 64) + lcol * 4;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) b_off[i] = (unsigned)((n0 + lrow + RPP * i) * p.w_row + lcol) * 4;
    if (BS) ld_bnt = __builtin_amdgcn_readfirstlane(nt * BCH);
  };
  auto issue_loads = [&]() {   // k-tile (loader tile; tap, channel block) -> registers: PA + PB loads
    // The descriptor words and scalar offsets were prepared when they last changed (setup_tile / advance); here
    // they are only pinned to SGPRs — the compiler's divergence analysis gives up on this loop-carried state and
    // would otherwise wrap every buffer load in a readfirstlane waterfall loop.
    const unsigned long long ptr = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(ld_phi) << 32) |
                                   (unsigned)__builtin_amdgcn_readfirstlane(ld_plo);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(ptr), 0, __builtin_amdgcn_readfirstlane(ld_nrec), 0x00020000);
    const int soff_a = __builtin_amdgcn_readfirstlane(ld_soff_a);
    const int soff_b = __builtin_amdgcn_readfirstlane(ld_soff_b);
#pragma unroll
    for (int i = 0; i < PA; ++i)
      va[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_cur[i], soff_a, 0));
    if (BS) {
      const int soff_b3 = __builtin_amdgcn_readfirstlane(ld_soff_b + ld_bnt);
#pragma unroll
      for (int i = 0; i < PB3; ++i)
        vb3[i] = __builtin_bit_cast(pk_uintx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (tid + NT * i) * 16, soff_b3, 0));
      return;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      vb[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[i], soff_b, 0));
  };
  const auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  // weight tap index of the loader's current (class-local) tap
  auto set_tap = [&]() {
    if (STRIDED) {
      const int c = ld_cls;
      ld_tap = sgpr(((kp->cls_d0[c][0] + ld_dt * p.st) * p.kh + kp->cls_d0[c][1] + ld_dh * p.sh) * p.kw + kp->cls_d0[c][2] +
                    ld_dw * p.sw);
    } else {
      ld_tap = sgpr((ld_dt * p.kh + ld_dh) * p.kw + ld_dw);
    }
    ld_soff_a = sgpr(ld_c0 * 4);
    ld_soff_b = BS ? sgpr((ld_tap * cpt + ld_c0 / BK) * p.wsp_kstep) : sgpr((ld_tap * p.Cs + ld_c0) * 4);
  };
  auto setup_seg = [&](int j) {   // loader enters segment j (STRIDED: the next segment that has any k-tiles)
    int tile, k0, k1, split;
    seg_info(j, tile, k0, k1, split);
    if (STRIDED) {
      while (k1 == k0 && j + 1 < nseg) seg_info(++j, tile, k0, k1, split);
      ld_seg = sgpr(k1 == k0 ? nseg : j);
      if (k1 == k0) return;
      ld_cls = sgpr(cls_of(tile));
      ld_nh = sgpr(kp->cls_nd[ld_cls][1]);
      ld_nw = sgpr(kp->cls_nd[ld_cls][2]);
    }
    ld_ks = sgpr(k0);
    ld_kend = sgpr(k1);
    const int tapi = k0 / cpt;
    ld_c0 = sgpr((k0 - tapi * cpt) * BK);
    ld_dw = sgpr(tapi % ld_nw);
    const int r = tapi / ld_nw;
    ld_dh = sgpr(r % ld_nh);
    ld_dt = sgpr(r / ld_nh);
    set_tap();
    setup_tile(tile);
  };
  // (one retap() site for the new-tile and the new-tap case: the row offsets are updated in place instead of being
  // merged from three branches, which cost 8 register copies per k-tile)
  auto advance = [&]() {
    ld_ks = sgpr(ld_ks + 1);
    ld_c0 = sgpr(ld_c0 + BK);
    bool moved = false;
    if (ld_ks == ld_kend) {          // next segment of this workgroup
      ld_seg = sgpr(ld_seg + 1);
      if (ld_seg < nseg) {
        setup_seg(ld_seg);
        moved = ld_seg < nseg;
      }
    } else if (ld_c0 != p.Cs) {      // next channel block of the same tap
      ld_soff_a = sgpr(ld_soff_a + BK * 4);
      ld_soff_b = sgpr(ld_soff_b + (BS ? p.wsp_kstep : BK * 4));
    } else {                         // next tap
      ld_c0 = 0;
      int dw = ld_dw + 1, dh = ld_dh, dt = ld_dt;
      if (dw == ld_nw) { dw = 0; ++dh; }
      if (dh == ld_nh) { dh = 0; ++dt; }
      ld_dw = sgpr(dw);
      ld_dh = sgpr(dh);
      ld_dt = sgpr(dt);
      set_tap();
      moved = true;
    }
    if (moved) retap();
  };
  auto store_stage = [&](float* st) {
#pragma unroll
    for (int i = 0; i < PA; ++i)   // (RPP is a multiple of 16: the swizzle key of a thread's rows is that of lrow)
      *reinterpret_cast<floatx4*>(&st[(lrow + RPP * i) * ALD + (ASWZ ? (((tid & 7) ^ ((lrow >> 1) & 7)) << 2) : lcol)]) = va[i];
    if (BS) {
#pragma unroll
      for (int i = 0; i < PB3; ++i)
        *reinterpret_cast<pk_uintx4*>(reinterpret_cast<char*>(st + BM * ALD) + (tid + NT * i) * 16) = vb3[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) *reinterpret_cast<floatx4*>(&st[(BM + lrow + RPP * i) * LDK + lcol]) = vb[i];
  };

  // ---- prologue: k-tile 0 -> LDS stage 0, k-tile 1 -> registers
  setup_seg(0);
  if (!STRIDED || ld_seg < nseg) {  // (a strided workgroup may own nothing but empty classes)
    retap();
    issue_loads();
    advance();
    store_stage(smem);
    if (ld_seg < nseg) {            // registers hold the k-tile after the one in LDS
      issue_loads();
      advance();
    }
  }
  __syncthreads();
  PK_STAMP(1);

  const int a_frag = (wm * TM * 32 + l31) * LDK + h * 4;
  const int b_frag = (BM + wn * TN * 32 + l31) * LDK + h * 4;
  int left = n_full * nk;   // k-tiles still to be multiplied, including the one in LDS
  if (STRIDED) {
    left = 0;
    for (int j = 0; j < nseg; ++j) {
      int tile, k0, k1, split;
      seg_info(j, tile, k0, k1, split);
      left += k1 - k0;
    }
  } else if (nseg > n_full) {
    int tile, k0, k1, split;
    seg_info(n_full, tile, k0, k1, split);
    left += k1 - k0;
  }
  int u = 0;
  floatx16 acc[TM][TN];

  // One k-tile: 4 k-groups of (fragment prefetch | staging work in the shadow of the group's MFMAs).
  // ST: the registers hold the next k-tile (write it to the other LDS stage); LD: one more exists (load it).
  auto ktile = [&](auto ST, auto LD) {
    const float* cur = smem + u * STAGE;
    float* nxt = smem + (u ^ 1) * STAGE;
    const float* Ab = cur + a_frag;
    const float* Bb = cur + b_frag;
    // (64 x 64 wave tiles only: with 32 x 64 the split of three fragments feeds 12 matrix instructions instead of four
    //  fragments 24 — measured: <4,1,1,2,0> 0.98 -> 1.06 ms per step, <2,2,2,2,0> 0.59 -> 0.52)
    if (BS) {
      // the weight fragments come whole from LDS; only this wave's input rows are split in registers
      const float* As = ASWZ ? cur + (wm * TM * 32 + l31) * ALD : Ab - h * 4 + h * 8;
      const int akey = (l31 >> 1) & 7;                 // ASWZ: piece c of this lane's rows sits at c ^ akey
      const char* Bc = reinterpret_cast<const char*>(cur + BM * ALD) + lane * 16;
      // (-DAVID_PK_PREFETCH: all fragments of k-step st + 1 requested before the products of k-step st are issued — the compiler
      // places every fragment read right in front of the matrix instructions that use it, "two reads, wait, six products".
      // Measured on the 128 x 64 tile (134 -> 181 registers): nothing — `<4,1,1,2,0>` 0.509-0.514 against 0.514-0.520 ms per step,
      // `<4,1,1,2,1>` 0.754-0.761 / 0.755-0.762, conv4x temporal 32.2 / 32.1 us: the LDS round trip of the fragments is not what a
      // k-tile waits for either)
      floatx4 ar[2][TM][2];
      pk_bf16x8 bq[2][TN][3];
      auto fetch = [&](int st, int slot) {
        const int o0 = ASWZ ? (((st * 4 + h * 2) ^ akey) << 2) : st * 16;
        const int o1 = ASWZ ? (((st * 4 + h * 2 + 1) ^ akey) << 2) : st * 16 + 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ar[slot][i][0] = *reinterpret_cast<const floatx4*>(As + i * 32 * ALD + o0);
          ar[slot][i][1] = *reinterpret_cast<const floatx4*>(As + i * 32 * ALD + o1);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const char* Bj = Bc + (((wn * TN + j) * 2 + st) * 3) * 1024;
          bq[slot][j][0] = *reinterpret_cast<const pk_bf16x8*>(Bj);
          bq[slot][j][1] = *reinterpret_cast<const pk_bf16x8*>(Bj + 1024);
          bq[slot][j][2] = *reinterpret_cast<const pk_bf16x8*>(Bj + 2048);
        }
      };
#ifdef AVID_PK_PREFETCH
      constexpr bool PF = TM * TN <= 2 && !STRIDED;     // (the 128 x 128 tiles have no registers for a second set: spills)
#else
      constexpr bool PF = false;
#endif
      fetch(0, 0);
#pragma unroll
      for (int st = 0; st < BK / 16; ++st) {
        if (PF) {
          if (st + 1 < BK / 16) fetch(st + 1, (st + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
        } else if (st > 0) {
          fetch(st, st & 1);
        }
        if (decltype(ST)::value && st == 0) store_stage(nxt);
        if (decltype(LD)::value && st == 1) issue_loads();
        pk_bf16x8 ah[TM], am[TM], al[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) pk_split8(ar[st & 1][i][0], ar[st & 1][i][1], ah[i], am[i], al[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const pk_bf16x8 bh = bq[st & 1][j][0], bm = bq[st & 1][j][1], bl = bq[st & 1][j][2];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
          }
        }
      }
      return;
    }
    if (PK_SPLIT && TM * TN >= 4) {
      // two k-steps of 16: lane (row l31, half h) reads its 8 consecutive k (two 16-byte reads; the 36-float row pitch puts
      // the 16 lanes of a read phase on disjoint banks), splits them in registers, six matrix instructions per tile
      const float* As = Ab - h * 4 + h * 8;            // (a_frag / b_frag carry the fp32 layout's h * 4)
      const float* Bs = Bb - h * 4 + h * 8;
#pragma unroll
      for (int st = 0; st < BK / 16; ++st) {
        if (decltype(ST)::value && st == 0) store_stage(nxt);
        if (decltype(LD)::value && st == 1) issue_loads();
        pk_bf16x8 ah[TM], am[TM], al[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const floatx4 v0 = *reinterpret_cast<const floatx4*>(As + i * 32 * LDK + st * 16);
          const floatx4 v1 = *reinterpret_cast<const floatx4*>(As + i * 32 * LDK + st * 16 + 4);
          pk_split8(v0, v1, ah[i], am[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const floatx4 v0 = *reinterpret_cast<const floatx4*>(Bs + j * 32 * LDK + st * 16);
          const floatx4 v1 = *reinterpret_cast<const floatx4*>(Bs + j * 32 * LDK + st * 16 + 4);
          pk_bf16x8 bh, bm, bl;
          pk_split8(v0, v1, bh, bm, bl);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
          }
        }
      }
      return;
    }
    floatx4 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (g + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[(g + 1) & 1][i] = *reinterpret_cast<const floatx4*>(Ab + i * 32 * LDK + (g + 1) * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[(g + 1) & 1][j] = *reinterpret_cast<const floatx4*>(Bb + j * 32 * LDK + (g + 1) * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (decltype(ST)::value && g == 0) store_stage(nxt);
      if (decltype(LD)::value && g == 1) issue_loads();
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][s], bf[g & 1][j][s], acc[i][j], 0, 0, 0);
      // issue order inside the group: one staging instruction per MFMA, never a burst
      if (decltype(ST)::value && g == 0) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        }
      }
      if (decltype(LD)::value && g == 1) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int row_bytes = p.Cd * 4;
  for (int j = 0; j < nseg; ++j) {
    int tile, k0, k1, split;
    seg_info(j, tile, k0, k1, split);
    const int nkj = k1 - k0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // steady state: a k-tile is pending in registers and another one exists beyond it — one straight-line
    // body; only the last two k-tiles of the workgroup's whole stream take the drain variants.
    int nst = left - 2 < nkj ? left - 2 : nkj;
    if (nst < 0) nst = 0;
    PK_STAMP(2 + 3 * j);
    for (int ks = 0; ks < nst; ++ks, u ^= 1) {
      ktile(std::true_type{}, std::true_type{});
      advance();
      __syncthreads();
    }
    left -= nst;
    if (nst < nkj) {
      if (left == 2) {
        ktile(std::true_type{}, std::false_type{});
        __syncthreads();
        u ^= 1;
        --left;
        ++nst;
      }
      if (nst < nkj) {
        ktile(std::false_type{}, std::false_type{});
        __syncthreads();
        u ^= 1;
        --left;
      }
    }

    PK_STAMP(3 + 3 * j);
    // ---- epilogue of this segment: buffer stores bounded by num_records (rows past M are dropped by the
    // hardware), the row step in soffset; they drain under the next tile's MFMAs.  A K-split piece writes
    // its raw partial sums to the slab of its split; splitk_reduce_kernel applies the epilogue.
    if (STRIDED) {
      // class-local rows -> destination pixels are not an affine map: one decode per tile row into LDS, then
      // buffer stores through the looked-up row offsets (rows past the class fall off num_records)
      int* drow = reinterpret_cast<int*>(smem + 2 * STAGE);
      int* arow = drow + BM;               // row offsets into the addend (== drow unless the addend is compact)
      const bool sparse_add = p.add_s[0] * p.add_s[1] * p.add_s[2] > 1;
      const int c = cls_of(tile);
      const int lt = tile - kp->cls_begin[c];
      const int mt = lt / ntn, nt = lt - mt * ntn;
      const int n0 = nt * BN;
      const int cT = kp->cls_n[c][0], cH = kp->cls_n[c][1], cW = kp->cls_n[c][2];
      __syncthreads();
      if (tid < BM) {
        const unsigned m = mt * BM + tid;
        const bool ok = m < (unsigned)(p.B * cT * cH * cW);
        const unsigned mm = ok ? m : 0u;
        const unsigned q1 = magic_div(mm, kp->cls_mg[c][2], kp->cls_shf[c][2]);
        const int wd = mm - q1 * cW;
        const unsigned q2 = magic_div(q1, kp->cls_mg[c][1], kp->cls_shf[c][1]);
        const int hd = q1 - q2 * cH;
        const int b = magic_div(q2, kp->cls_mg[c][0], kp->cls_shf[c][0]);
        const int td = q2 - b * cT;
        const int dst = ((b * p.Td + kp->cls_p0[c][0] + td * p.st) * p.Hd + kp->cls_p0[c][1] + hd * p.sh) * p.Wd +
                        kp->cls_p0[c][2] + wd * p.sw;
        drow[tid] = ok ? dst * row_bytes : (int)OOB;
        if (sparse_add) {   // strides are 1 or 2: position -> compact position where every strided coordinate is even
          const int t = kp->cls_p0[c][0] + td * p.st, hh = kp->cls_p0[c][1] + hd * p.sh, w = kp->cls_p0[c][2] + wd * p.sw;
          const bool on = ok && (t & (p.add_s[0] - 1)) == 0 && (hh & (p.add_s[1] - 1)) == 0 && (w & (p.add_s[2] - 1)) == 0;
          const int ar = ((b * p.add_n[0] + (t >> (p.add_s[0] - 1))) * p.add_n[1] + (hh >> (p.add_s[1] - 1))) * p.add_n[2] +
                         (w >> (p.add_s[2] - 1));
          arow[tid] = on ? ar * row_bytes : (int)OOB;
        } else {
          arow[tid] = ok ? dst * row_bytes : (int)OOB;
        }
      }
      __syncthreads();
      const bool direct = split < 0;   // K-split pieces write raw sums to their slab, in destination row order
      const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(direct ? p.dst : p.part + (long long)split * p.M * p.Cd), 0, (int)((long long)p.M * row_bytes), 0x00020000);
      const long long add_rows = sparse_add ? (long long)p.B * p.add_n[0] * p.add_n[1] * p.add_n[2] : (long long)p.M;
      const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.addend ? p.addend : p.dst), 0, (int)(add_rows * row_bytes), 0x00020000);
      const __amdgpu_buffer_rsrc_t rsXb = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.bnb_x ? p.bnb_x : p.dst), 0, (int)((long long)p.M * row_bytes), 0x00020000);
      auto emit = [&](auto HAS_ADD, auto BNB) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int jj = 0; jj < TN; ++jj) {
            const int col = n0 + (wn * TN + jj) * 32 + l31;
            const int col4 = col * 4;
            unsigned voff[16];
            float ad[16], xb[16], bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int trow = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
              voff[r] = (unsigned)drow[trow] + col4;
              if (HAS_ADD)    // (a row the compact addend does not cover is out of range: the load returns 0)
                ad[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsE, (unsigned)arow[trow] + col4, 0, 0));
              if (BNB) xb[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsXb, voff[r], 0, 0));
            }
            if (BNB) { bsc = p.bnb_scale[col]; bsh = p.bnb_shift[col]; bmu = p.bnb_mean[col]; bis = p.bnb_invstd[col]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[i][jj][r];
              if (HAS_ADD) v += ad[r];
              if (BNB) {       // BatchNorm-backward partial sums of the gradient being written (see the dense epilogue)
                const float dm = (!p.bnb_relu || fmaf(xb[r], bsc, bsh) > 0.f) ? v : 0.f;
                cs[jj][r & 1] += dm;
                cq[jj][r & 1] = fmaf(dm, (xb[r] - bmu) * bis, cq[jj][r & 1]);
              }
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsD, voff[r], 0, 0);
            }
          }
        }
      };
      constexpr std::true_type Y{};
      constexpr std::false_type N{};
      if (!direct) emit(N, N);
      else if (EPI == 9) emit(Y, Y);
      else if (EPI == 8) emit(N, Y);
      else if (EPI == 1) emit(Y, N);
      else if (EPI == 0) emit(N, N);
      else if (p.bnb_x) { if (p.addend) emit(Y, Y); else emit(N, Y); }
      else if (p.addend) emit(Y, N);
      else emit(N, N);
      continue;
    }
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    int rows = p.M - m0;
    rows = rows > BM ? BM : rows;
    const bool direct = split < 0;
    const long long d_base = (long long)m0 * p.Cd;
    const long long s_base = ((long long)split * (p.M - p.part_row_begin) + (m0 - p.part_row_begin)) * p.Cd;
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(direct ? p.dst + d_base : p.part + s_base), 0, rows * row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.addend ? p.addend + d_base : p.dst + d_base), 0, rows * row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsXb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.bnb_x ? p.bnb_x + d_base : p.dst + d_base), 0, rows * row_bytes, 0x00020000);
    // The uniform cases are told apart once per tile, not per element: the element loops below are straight
    // lines of (load,) VALU, store.
    auto emit = [&](auto DIRECT, auto HAS_ADD, auto RELU, auto OP, auto BNB, auto BIAS) {   // OP: addend combines by 0 add, 1 min, 2 max; BIAS: 0 none, 1 yes, 2 if p.bias
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
          const int col = n0 + (wn * TN + jj) * 32 + l31;
          const unsigned voff = (unsigned)(((wm * TM + i) * 32 + 4 * h) * p.Cd + col) * 4;
          float ad[16];
          if (HAS_ADD) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              ad[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rsE, voff, ((r & 3) + 8 * (r >> 2)) * row_bytes, 0));
          }
          const float bv = (DIRECT && decltype(BIAS)::value != 0 && (decltype(BIAS)::value == 1 || p.bias)) ? p.bias[col] : 0.f;
          float xb[16], bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
          if (BNB) {   // the BatchNorm input at the positions of this tile + this column's saved coefficients
#pragma unroll
            for (int r = 0; r < 16; ++r)
              xb[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rsXb, voff, ((r & 3) + 8 * (r >> 2)) * row_bytes, 0));
            bsc = p.bnb_scale[col]; bsh = p.bnb_shift[col]; bmu = p.bnb_mean[col]; bis = p.bnb_invstd[col];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int soff = ((r & 3) + 8 * (r >> 2)) * row_bytes;
            float v = acc[i][jj][r];
            if (DIRECT) {
              if (decltype(BIAS)::value != 0) v += bv;   // (x + 0.f is not a no-op the compiler may drop: -0.f)
              if (HAS_ADD) v = decltype(OP)::value == 0 ? v + ad[r] : (decltype(OP)::value == 1 ? fminf(v, ad[r]) : fmaxf(v, ad[r]));
              if (RELU) v = fmaxf(v, 0.f);
              if (MODE == 0) { cs[jj][r & 1] += v; cq[jj][r & 1] = fmaf(v, v, cq[jj][r & 1]); }   // rows past M are exact zeros
              if (BNB) {       // bn_bwd_partial_kernel's sums, from the gradient while it is in registers
                const float dm = (!p.bnb_relu || fmaf(xb[r], bsc, bsh) > 0.f) ? v : 0.f;
                cs[jj][r & 1] += dm;
                cq[jj][r & 1] = fmaf(dm, (xb[r] - bmu) * bis, cq[jj][r & 1]);
              }
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsD, voff, soff, 0);
          }
        }
      }
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    constexpr std::integral_constant<int, 0> ADD{};
    constexpr std::integral_constant<int, 0> B0{};
    constexpr std::integral_constant<int, 1> B1{};
    constexpr std::integral_constant<int, 2> BQ{};
    if (!direct) emit(N, N, N, ADD, N, B0);
    else if (EPI == 0) emit(Y, N, N, ADD, N, B0);
    else if (EPI == 1) emit(Y, Y, N, ADD, N, B0);
    else if (MODE == 0 && EPI == 2) emit(Y, N, N, ADD, N, B1);
    else if (MODE == 0 && EPI == 3) emit(Y, N, Y, ADD, N, B1);
    else if (MODE == 0 && EPI == 4) emit(Y, Y, N, std::integral_constant<int, 1>{}, N, B0);      // CMA agreement scores
    else if (MODE == 0 && EPI == 5) emit(Y, Y, N, std::integral_constant<int, 2>{}, N, B0);
    else if (MODE == 1 && EPI == 8) emit(Y, N, N, ADD, Y, B0);
    else if (MODE == 1 && EPI == 9) emit(Y, Y, N, ADD, Y, B0);
    else if (MODE == 1 && p.bnb_x) {      // EPI_ANY: every combination, at run time
      if (p.addend) emit(Y, Y, N, ADD, Y, BQ); else emit(Y, N, N, ADD, Y, BQ);
    } else if (p.addend) {
      if (MODE == 0 && p.epi_op == 1) emit(Y, Y, N, std::integral_constant<int, 1>{}, N, BQ);
      else if (MODE == 0 && p.epi_op == 2) emit(Y, Y, N, std::integral_constant<int, 2>{}, N, BQ);
      else if (p.relu) emit(Y, Y, Y, ADD, N, BQ);
      else emit(Y, Y, N, ADD, N, BQ);
    } else {
      if (p.relu) emit(Y, N, Y, ADD, N, BQ); else emit(Y, N, N, ADD, N, BQ);
    }
    PK_STAMP(4 + 3 * j);
  }
  write_stats();
  PK_STAMP(31);
}

// ------------------------------------------------------------------------------------------------
// tconv64_kernel<MODE, EPI> — the (3,1,1) stride-1 pad-1 layers of conv2x (models/network_blocks.py:37,42 at 64 -> 64
// channels, 8 frames): forward (MODE 0) and input gradient (MODE 1), split-bf16 products.  (round 5)
//
// igemm_pk_kernel stages one A tile per (tap, 32-channel block): a temporal layer's input rows travel HBM/L2 -> registers
// -> LDS three times, once per tap, and a k-tile holds 24 matrix instructions per wave — too few to cover the round trip
// (58 us without the global loads, 83 with).  Here a tile is 32 POSITIONS x ALL 8 FRAMES (256 GEMM rows,
// wave w = output frame w): a 32-channel block of it is staged ONCE and serves the three taps as LDS rows shifted by a
// frame (tap d of output frame t reads frame t + d - 1 forward, t + 1 - d for the input gradient; a frame outside the
// clip is the zero padding: the wave skips that tap), and the 72 KB of pre-split weights (avid_wt_desc mode 5 / 6: six
// PK_BCH chunks) sit in LDS for the whole kernel.  Per staged block and wave: 72 matrix instructions, 4 global loads and
// 4 LDS stores per thread, ONE barrier — a third of the loads / stores / barriers per product, no weight traffic, and
// every input row is read by exactly one workgroup (HBM traffic = the algorithmic bytes).
// One persistent workgroup of 8 waves per CU (147 KB of LDS), tiles slot, slot + G, ...; the loader runs two
// blocks ahead through registers, across tile boundaries.
// Epilogues as igemm_pk_kernel's direct ones: BatchNorm partial sums of the output (forward, p.stats), addend, the
// BatchNorm-backward sums of the gradient being written (input gradient, EPI 8 / 9).
// ------------------------------------------------------------------------------------------------
constexpr int WG_LD_C = 64 + 4;               // row pitch (floats) of a 64-channel LDS row read by columns (== WG_LD below)
constexpr int TC_P = 32;                      // positions per tile
constexpr int TC_T = 8;                       // frames (= waves)
constexpr int TC_ROWS = TC_P * TC_T;
constexpr int TC_B_BYTES = 6 * PK_BCH;        // 3 taps x 2 channel blocks
constexpr int TC_STAGE = TC_ROWS * LDK;       // floats per A stage (one 32-channel block)
constexpr size_t TC_LDS = TC_B_BYTES + (2 * TC_STAGE + TC_P * LDK + 128) * sizeof(float);     // weights | two A stages | a frame of zeros | AFF: scale, shift

// AFF (forward): the staged rows are the input of a BatchNorm (+ReLU); the staging threads apply it (ConvArgs::in_scale)
template <int MODE, int EPI, bool AFF = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void tconv64_kernel(const ConvArgs p) {
  static_assert(!AFF || MODE == 0, "the input affine map belongs to the forward");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Bs = reinterpret_cast<char*>(smem);
  float* As = smem + TC_B_BYTES / 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int lcol = (tid & 7) * 4;
  const int G = gridDim.x;
  const int slot = __builtin_amdgcn_readfirstlane((int)xcd_remap(blockIdx.x, G));
  PK_STAMP(0);
  const int HW = p.Hs * p.Ws;
  const int NP = p.B * HW;                                  // positions
  const int ntiles = (NP + TC_P - 1) / TC_P;
  const int nmine = slot < ntiles ? (ntiles - slot + G - 1) / G : 0;
  const int nitems = 2 * nmine;                             // (tile, channel block) items of this workgroup
  const int frame_bytes = HW * 256;                         // one frame of 64-channel fp32 rows

  float cs[2][2], cq[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) cs[j][0] = cs[j][1] = cq[j][0] = cq[j][1] = 0.f;
  auto write_stats = [&]() {                                // one partial row [2][64] per workgroup + zero rows up to the promise
    if (!p.stats) return;
    float* red = As;                                        // [2][8 waves][64]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a0 = cs[j][0] + cs[j][1], b0 = cq[j][0] + cq[j][1];
      const float a = a0 + __shfl_xor(a0, 32, 64), b = b0 + __shfl_xor(b0, 32, 64);
      if (h == 0) {
        red[wave * 64 + j * 32 + l31] = a;
        red[512 + wave * 64 + j * 32 + l31] = b;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[which * 512 + w * 64 + c];
      p.stats[(long long)slot * 128 + which * 64 + c] = a;
      for (int r = slot + G; r < p.stats_rows; r += G) p.stats[(long long)r * 128 + which * 64 + c] = 0.f;
    }
  };
  if (nitems == 0) {
    write_stats();
    return;
  }

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.src, 0, (int)((long long)p.M * 256), 0x00020000);
  // ---- roles.  What the time stamps say (tools/tconv_trace.py, -DAVID_PK_TRACE): a SIMD executes its two waves' instructions
  // as ONE serial stream — the older wave (0-3) runs its 72 products + 216 split instructions in 1.75 us, the younger one
  // (4-7) gets the pipe when the older has finished (another 1.45-1.75 us); nothing of one wave hides under the matrix
  // instructions of the other, and a tile's time is the SUM of everything both waves issue (288 matrix instructions at
  // 17 ns = 4.9 us + ~860 split instructions at 2.7 ns + the rest = 7.5 us).  So the arrangement only decides who idles at
  // the barriers: the younger waves do everything that is not a product while the older multiply — they alone stage (global
  // -> registers -> LDS, all 256 rows) and they write their tile's output one item late, at the start of the next tile; the
  // older waves go straight into their products and write their output right after a tile's last product, while the
  // younger ones still multiply.  (60.5 -> 59.2 us; what would move it is fewer instructions: the split once per input
  // row instead of once per tap — needs the rows in LDS as bf16 planes, 6 B per element: 16-channel stages — DESIGN.md 8g.)
  const bool lead = wave < 4;
  const int trow = (tid & 255) >> 3;                        // trailing waves: position of this thread's staged rows (frames 0..7)
  // ---- loader (trailing waves): item li = (tile slot + (li >> 1) G, channel block li & 1); 8 rows per thread, one per frame
  unsigned ld_voff = OOB;
  auto ld_tile = [&](int li) {                              // this thread's row offset in the loader's tile
    const int tile = slot + (li >> 1) * G;
    const unsigned pg = (unsigned)(tile * TC_P + trow);
    const unsigned b = magic_div(pg, p.mgW, p.shW);         // position -> clip (division by HW)
    const unsigned hw = pg - b * HW;
    ld_voff = pg < (unsigned)NP ? (unsigned)((b * TC_T * HW + hw) * 256 + lcol * 4) : OOB;
  };
  floatx4 va[8];
  auto issue_loads = [&](int li) {
    const int c128 = (li & 1) * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      va[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                              rsA, ld_voff, __builtin_amdgcn_readfirstlane(c128 + i * frame_bytes), 0));
  };
  // AFF: scale / shift of this thread's four channels in either channel block; rows past the last position stay zero (their
  // loads fell off the buffer: the outputs computed from them must not reach the BatchNorm partial sums)
  // (the vectors sit in LDS behind the padding frame — scale[64] | shift[64], written by the leading waves in the prologue — and
  //  are read per item: held in registers for the whole kernel they were 16 of them, and 8 / 19 spilled ones)
  const float* aff = As + 2 * TC_STAGE + TC_P * LDK;
  auto store_stage = [&](float* st, int blk, bool first = false) {   // blk: the channel block of the item in the registers
    floatx4 asc = {0.f, 0.f, 0.f, 0.f}, ash = asc;
    if (AFF) {
      if (first) {                                          // the prologue's item: before the barrier that publishes the LDS copy
        asc = *reinterpret_cast<const floatx4*>(p.in_scale + blk * 32 + lcol);
        ash = *reinterpret_cast<const floatx4*>(p.in_shift + blk * 32 + lcol);
      } else {
        asc = *reinterpret_cast<const floatx4*>(aff + blk * 32 + lcol);
        ash = *reinterpret_cast<const floatx4*>(aff + 64 + blk * 32 + lcol);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      floatx4 v = va[i];
      if (AFF) {
        const bool ok = ld_voff != OOB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float o = fmaf(v[j], asc[j], ash[j]);             // bn_apply_kernel's expression, bit for bit
          if (p.in_relu) o = fmaxf(o, 0.f);
          v[j] = ok ? o : 0.f;
        }
      }
      *reinterpret_cast<floatx4*>(&st[(trow + 32 * i) * LDK + lcol]) = v;
    }
  };

  // ---- prologue: the weights -> LDS (72 KB, once; the older waves), item 0 -> stage 0, item 1 -> registers (the younger)
  if (lead) {
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wsp), 0, TC_B_BYTES, 0x00020000);
    pk_uintx4 wv[18];
#pragma unroll
    for (int i = 0; i < 18; ++i)
      wv[i] = __builtin_bit_cast(pk_uintx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (tid + 256 * i) * 16, 0, 0));
    for (int i = tid; i < TC_P * LDK; i += 256) As[2 * TC_STAGE + i] = 0.f;       // the padding frame
    if (AFF && tid < 128) As[2 * TC_STAGE + TC_P * LDK + tid] = tid < 64 ? p.in_scale[tid] : p.in_shift[tid - 64];
#pragma unroll
    for (int i = 0; i < 18; ++i) *reinterpret_cast<pk_uintx4*>(Bs + (tid + 256 * i) * 16) = wv[i];
  } else {
    ld_tile(0);
    issue_loads(0);
    store_stage(As, 0, true);
    if (nitems > 1) issue_loads(1);
  }
  __syncthreads();

  floatx16 acc[2];
  const int row_bytes = 256;                                // Cd = 64
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, (int)((long long)p.M * row_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.addend ? p.addend : p.dst), 0, (int)((long long)p.M * row_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsXb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.bnb_x ? p.bnb_x : p.dst), 0, (int)((long long)p.M * row_bytes), 0x00020000);
  constexpr bool HAS_ADD = (EPI & 1) != 0, BNB = (EPI & 8) != 0;
  float bsc[2] = {0.f, 0.f}, bsh[2] = {0.f, 0.f}, bmu[2] = {0.f, 0.f}, bis[2] = {0.f, 0.f};
  if (BNB) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j * 32 + l31;
      bsc[j] = p.bnb_scale[col]; bsh[j] = p.bnb_shift[col]; bmu[j] = p.bnb_mean[col]; bis[j] = p.bnb_invstd[col];
    }
  }

  // products of one staged block: six steps s = (tap d = s >> 1, k-step st = s & 1) of 12 matrix instructions each,
  // software-pipelined inside the wave: the fragments of step s + 2 are requested from LDS and the input rows of step s + 1
  // are split while step s is multiplied.  Source frame of tap d: wave + d - 1 (forward) / wave + 1 - d (input gradient);
  // a frame outside the clip reads the ZERO block (the padding in time), so every wave runs the same straight-line steps —
  // the barrier at the end of the block waits for the slowest SIMD anyway.
  const float* Alane = As + l31 * LDK + h * 8;              // this lane's fragment position inside a frame of a stage
  // products of one staged block: six steps s = (tap d = s >> 1, k-step st = s & 1) of 12 matrix instructions each,
  // software-pipelined inside the wave: the fragments of step s + 2 are requested from LDS and the input rows of step s + 1
  // are split while step s is multiplied.  Source frame of tap d: wave + d - 1 (forward) / wave + 1 - d (input gradient);
  // a frame outside the clip reads the ZERO block (the padding in time), so every wave runs the same straight-line steps —
  // the barrier at the end of the block waits for the slowest SIMD anyway.
  auto products = [&](int stage_off, auto CB) {
    constexpr int cb = decltype(CB)::value;
    int aoff[3];                                            // frame of tap d, as a scalar offset from As (floats)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int f = MODE == 0 ? wave + d - 1 : wave + 1 - d;
      aoff[d] = __builtin_amdgcn_readfirstlane((f < 0 || f >= TC_T) ? 2 * TC_STAGE : stage_off + f * TC_P * LDK);
    }
    const float* Af[3] = {Alane + aoff[0], Alane + aoff[1], Alane + aoff[2]};
    const char* Bc = Bs + cb * PK_BCH + lane * 16;
    floatx4 ar[3][2];
    pk_bf16x8 bf[3][6], sp[2][3];
    auto request = [&](int s_, int buf) {
      const int d = s_ >> 1, st = s_ & 1;
      ar[buf][0] = *reinterpret_cast<const floatx4*>(Af[d] + st * 16);
      ar[buf][1] = *reinterpret_cast<const floatx4*>(Af[d] + st * 16 + 4);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 3; ++t)
          bf[buf][j * 3 + t] = *reinterpret_cast<const pk_bf16x8*>(Bc + d * 2 * PK_BCH + ((j * 2 + st) * 3 + t) * 1024);
    };
    request(0, 0);
    request(1, 1);
    pk_split8<false>(ar[0][0], ar[0][1], sp[0][0], sp[0][1], sp[0][2]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s_ = 0; s_ < 6; ++s_) {
      const int b = s_ % 3, q2 = s_ & 1;
      if (s_ + 2 < 6) request(s_ + 2, (s_ + 2) % 3);      // lands under this step's and the next step's products
      if (s_ + 1 < 6) pk_split8<false>(ar[(s_ + 1) % 3][0], ar[(s_ + 1) % 3][1], sp[q2 ^ 1][0], sp[q2 ^ 1][1], sp[q2 ^ 1][2]);
      const pk_bf16x8 ah = sp[q2][0], am = sp[q2][1], al = sp[q2][2];
      // (the two accumulators alternate: no matrix instruction waits for the one issued just before it)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[b][2], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[b][5], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bf[b][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bf[b][3], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[b][1], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[b][4], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[b][1], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[b][4], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[b][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[b][3], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[b][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[b][3], acc[1], 0, 0, 0);
      // issue order of the step: the LDS requests first, then the next step's split spread under the matrix instructions
      if (s_ + 2 < 6) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (s_ + 1 < 6) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Epilogue of a tile: wave = frame, accumulator register r of lane (l31, h) = position (r & 3) + 8 (r >> 2) + 4 h, column
  // j * 32 + l31.  Destination row of (position pg, frame w): pg + (7 b + w) HW, b = pg / HW.  A tile that lies inside one
  // clip and inside the tensor (all but one in ~25) has ONE vector offset per lane, the 32 row / column steps ride in the
  // scalar offset: no address arithmetic per element (it was 9 vector instructions per row, all waves of the CU at once,
  // nothing for the matrix pipe to do meanwhile: 1.05 of a tile's 8.3 us).  SPLIT: the general form — a clip boundary or the
  // end of the tensor inside the tile.
  auto emit = [&](int tile, const floatx16 (&res)[2], auto SPLIT) {
    const unsigned p0 = (unsigned)(tile * TC_P);
    const unsigned b0 = magic_div(p0, p.mgW, p.shW);
    const unsigned pb = (b0 + 1) * HW;                      // first position of the next clip
    const unsigned base = (p0 + (7 * b0 + wave) * HW) * 256u + l31 * 4 + h * 4 * 256;
    const unsigned step_b = 7u * HW * 256u;
    unsigned voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      voff[r] = base;
      if (decltype(SPLIT)::value) {
        const unsigned pg = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        voff[r] = pg < (unsigned)NP ? base + (pg >= pb ? step_b : 0u) : OOB;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float ad[16], xb[16];
      if (HAS_ADD) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ad[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsE, voff[r], ((r & 3) + 8 * (r >> 2)) * 256 + j * 128, 0));
      }
      if (BNB) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xb[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsXb, voff[r], ((r & 3) + 8 * (r >> 2)) * 256 + j * 128, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = res[j][r];
        if (HAS_ADD) v += ad[r];
        if (MODE == 0) { cs[j][r & 1] += v; cq[j][r & 1] = fmaf(v, v, cq[j][r & 1]); }   // rows past the end are exact zeros
        if (BNB) {
          const float dm = (!p.bnb_relu || fmaf(xb[r], bsc[j], bsh[j]) > 0.f) ? v : 0.f;
          cs[j][r & 1] += dm;
          cq[j][r & 1] = fmaf(dm, (xb[r] - bmu[j]) * bis[j], cq[j][r & 1]);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsD, voff[r], ((r & 3) + 8 * (r >> 2)) * 256 + j * 128, 0);
      }
    }
  };
  auto epilogue = [&](int tile, const floatx16 (&res)[2]) {
    const unsigned p0 = (unsigned)(tile * TC_P);
    const unsigned pb = (magic_div(p0, p.mgW, p.shW) + 1) * HW;
    if (p0 + TC_P <= pb && p0 + TC_P <= (unsigned)NP) emit(tile, res, std::false_type{});
    else emit(tile, res, std::true_type{});
  };
  // one item (trailing waves): the registers hold item it + 1 -> the other stage (every wave left it at the last barrier),
  // item it + 2's loads go out; then, all waves, the products of item it from its stage; one barrier per item
  auto stage_next = [&](int it, int cb) {
    if (it + 1 < nitems) store_stage(As + (cb ^ 1) * TC_STAGE, cb ^ 1);
    if (it + 2 < nitems) {
      if (cb == 0) ld_tile(it + 2);
      issue_loads(it + 2);
    }
  };
  floatx16 out[2];                                          // trailing waves: the finished tile, written one item late
  PK_STAMP(1);
  for (int t = 0; t < nmine; ++t) {
    PK_STAMP(2 + 7 * t);
    if (!lead) {
      stage_next(2 * t, 0);
      if (t > 0) epilogue(slot + (t - 1) * G, out);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    PK_STAMP(3 + 7 * t);
#ifdef AVID_PK_TRACE   // tile 3: every wave's start and end of its products, in the rows of the (unused) workgroups 512...
    if (t == 3 && lane == 0) g_pk_trace[(512 + blockIdx.x) * 64 + wave] = wall_clock64();
#endif
    products(0, std::integral_constant<int, 0>{});
#ifdef AVID_PK_TRACE
    if (t == 3 && lane == 0) g_pk_trace[(512 + blockIdx.x) * 64 + 8 + wave] = wall_clock64();
#endif
    PK_STAMP(4 + 7 * t);
    __syncthreads();
    PK_STAMP(5 + 7 * t);
    if (!lead) stage_next(2 * t + 1, 1);
    products(TC_STAGE, std::integral_constant<int, 1>{});
    PK_STAMP(6 + 7 * t);
    if (lead) {
      epilogue(slot + t * G, acc);
    } else {
      out[0] = acc[0];
      out[1] = acc[1];
    }
    PK_STAMP(7 + 7 * t);
    __syncthreads();
    PK_STAMP(8 + 7 * t);
  }
  if (!lead) epilogue(slot + (nmine - 1) * G, out);
  PK_STAMP(31);
  write_stats();
}

// ------------------------------------------------------------------------------------------------
// twgrad64_kernel — the weight gradient of the same layers (conv2x's (3,1,1) stride-1 convolutions, 64 -> 64 channels, 8
// frames; backward of models/network_blocks.py:37,42): dw[n][dt][c] = sum over (clip, frame t, position) dy[t][n] x[t + dt - 1][c],
// split-bf16 products.  (round 5)
//
// wgrad_tab_kernel<1,3> splits every fragment where it uses it: 13 vector instructions per matrix instruction, and a SIMD
// runs them as one serial stream with its matrix instructions — the splits, not the products, are the kernel.
// The contraction index is the pixel, so a fragment is a lane's COLUMN of 8 pixel rows (8 ds_read_b32 + the split) and
// cannot be prepared once in LDS; what can be shared is the TAPS: a stage holds 16 positions x ALL 8 frames of dy and of x,
// wave (n half jn, c half jc, frame half kh) walks its four frames t with the split dy[t] fragment and a rolling window of
// the split x[t - 1], x[t], x[t + 1] fragments — one new dy and one new x fragment per 18 matrix instructions (4 vector
// instructions per matrix instruction instead of 13), the three taps' 32 x 32 blocks of dw in 48 accumulation registers for
// the whole kernel.  Persistent workgroups of 8 waves (136 KB of LDS: two stages), loads two stages ahead through
// registers, one barrier per stage; at the end the two frame halves are summed through LDS and every workgroup leaves one
// slab [64][3][64], which wgrad_reduce_kernel folds in fixed order.
// ------------------------------------------------------------------------------------------------
constexpr int TWG_P = 16;                          // positions per stage = one k-step
constexpr int TWG_PLANE = TC_T * TWG_P * WG_LD_C;  // floats of one operand's stage: [8 frames][16 positions][64 + 4]
constexpr size_t TWG_LDS = sizeof(float) * 2 * 2 * TWG_PLANE;

struct TwgradArgs {
  const float* x; const float* dy; float* part;    // part: [grid][64][3][64]
  int B, HW;
  unsigned mg; int sh;                             // multiply-shift division by HW
  const float* in_scale; const float* in_shift;    // AFF: x is the input of a BatchNorm (+ReLU) the staging threads apply (ConvArgs::in_scale)
  int in_relu;
};

template <bool AFF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void twgrad64_kernel(const TwgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int jn = wave & 1, jc = (wave >> 1) & 1, kh = wave >> 2;
  const int G = gridDim.x;
  const int slot = __builtin_amdgcn_readfirstlane((int)xcd_remap(blockIdx.x, G));
  const int HW = p.HW, NP = p.B * HW;
  const int ntiles = (NP + TWG_P - 1) / TWG_P;
  const int nmine = slot < ntiles ? (ntiles - slot + G - 1) / G : 0;
  const int frame_bytes = HW * 256;
  const long long total = (long long)NP * TC_T * 256;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)total, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)total, 0x00020000);
  // ---- loader: 128 rows (8 frames x 16 positions) x 256 B of each operand per stage: thread (row trow + 32 i, 16 B at lcol)
  const int trow = tid >> 4, lcol = (tid & 15) * 4;
  unsigned ld_voff = OOB;
  auto ld_tile = [&](int k) {
    const unsigned pg = (unsigned)((slot + k * G) * TWG_P + (trow & 15));
    const unsigned b = magic_div(pg, p.mg, p.sh);
    const unsigned hw = pg - b * HW;
    ld_voff = pg < (unsigned)NP ? (unsigned)(((b * TC_T + (trow >> 4)) * HW + hw) * 256 + lcol * 4) : OOB;
  };
  floatx4 vx[4], vy[4];
  auto issue_loads = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int so = __builtin_amdgcn_readfirstlane(2 * i * frame_bytes);
      vy[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsY, ld_voff, so, 0));
      vx[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ld_voff, so, 0));
    }
  };
  floatx4 asc = {0.f, 0.f, 0.f, 0.f}, ash = asc;           // AFF: scale / shift of this thread's four channels
  if (AFF) {
    asc = *reinterpret_cast<const floatx4*>(p.in_scale + lcol);
    ash = *reinterpret_cast<const floatx4*>(p.in_shift + lcol);
  }
  auto store_stage = [&](float* st) {                     // st: [dy plane | x plane]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      floatx4 v = vx[i];
      if (AFF) {                                          // (rows past the last position stay zero: their loads fell off the buffer)
        const bool ok = ld_voff != OOB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float o = fmaf(v[j], asc[j], ash[j]);           // bn_apply_kernel's expression, bit for bit
          if (p.in_relu) o = fmaxf(o, 0.f);
          v[j] = ok ? o : 0.f;
        }
      }
      *reinterpret_cast<floatx4*>(&st[(trow + 32 * i) * WG_LD_C + lcol]) = vy[i];
      *reinterpret_cast<floatx4*>(&st[TWG_PLANE + (trow + 32 * i) * WG_LD_C + lcol]) = v;
    }
  };
  floatx16 acc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  if (nmine > 0) {
    ld_tile(0);
    issue_loads();
    store_stage(smem);
    if (nmine > 1) { ld_tile(1); issue_loads(); }
  }
  __syncthreads();
  // a lane's fragment: its column (n or c = 32 j + l31) of the 8 pixel rows 8 h .. 8 h + 7 of a frame — 8 ds_read_b32
  // (row pitch 68 floats: the 64 lanes of one read cover the 64 banks once), split into three bf16x8
  auto frag = [&](const float* plane, int f, int col, pk_bf16x8& fh, pk_bf16x8& fm, pk_bf16x8& fl) {
    const float* q = plane + (f * TWG_P + h * 8) * WG_LD_C + col;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = q[i * WG_LD_C];
    pk_split8<true>(floatx4{v[0], v[1], v[2], v[3]}, floatx4{v[4], v[5], v[6], v[7]}, fh, fm, fl);
  };
  const int ncol = jn * 32 + l31, ccol = jc * 32 + l31;
  const int t0 = kh * 4;
  for (int k = 0; k < nmine; ++k) {
    const float* cur = smem + (k & 1) * 2 * TWG_PLANE;
    float* nxt = smem + ((k + 1) & 1) * 2 * TWG_PLANE;
    if (k + 1 < nmine) store_stage(nxt);
    if (k + 2 < nmine) { ld_tile(k + 2); issue_loads(); }
    const float* Dp = cur;
    const float* Xp = cur + TWG_PLANE;
    pk_bf16x8 xw[3][3];                                   // rolling window of split x fragments: frames t - 1, t, t + 1
    if (t0 > 0) frag(Xp, t0 - 1, ccol, xw[0][0], xw[0][1], xw[0][2]);
    frag(Xp, t0, ccol, xw[1][0], xw[1][1], xw[1][2]);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = t0 + tt;
      const int i0 = tt % 3, i1 = (tt + 1) % 3, i2 = (tt + 2) % 3;   // window slots of frames t - 1, t, t + 1
      pk_bf16x8 ah, am, al;
      frag(Dp, t, ncol, ah, am, al);
      const bool has_next = t + 1 < TC_T;                   // (wave-uniform)
      if (has_next) frag(Xp, t + 1, ccol, xw[i2][0], xw[i2][1], xw[i2][2]);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int sl = d == 0 ? i0 : (d == 1 ? i1 : i2);
        if ((d == 0 && t == 0) || (d == 2 && !has_next)) continue;     // the zero padding in time
        const pk_bf16x8 bh = xw[sl][0], bm = xw[sl][1], bl = xw[sl][2];
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[d], 0, 0, 0);
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[d], 0, 0, 0);
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[d], 0, 0, 0);
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[d], 0, 0, 0);
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[d], 0, 0, 0);
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[d], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- the two frame halves of a (jn, jc) pair: the upper half's sums through LDS, the lower half adds and writes the slab
  float* xch = smem + (wave & 3) * (3 * 16 * 64);
  if (kh == 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(d * 16 + r) * 64 + lane] = acc[d][r];
  }
  __syncthreads();
  if (kh == 0) {
    float* slab = p.part + (long long)slot * (64 * 3 * 64);
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = jn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        slab[(n * 3 + d) * 64 + ccol] = acc[d][r] + xch[(d * 16 + r) * 64 + lane];
      }
  }
}

// dst = sum_s part[s] (+ bias)(+ addend)(relu) — fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                            const float* __restrict__ addend,
                                                            const float* __restrict__ bias, long long n4, int G,
                                                            int nsplit, int relu) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    floatx4 s = reinterpret_cast<const floatx4*>(part)[i];
    for (int k = 1; k < nsplit; ++k) s += reinterpret_cast<const floatx4*>(part)[(long long)k * n4 + i];
    if (bias) s += reinterpret_cast<const floatx4*>(bias)[i % G];
    if (addend) s += reinterpret_cast<const floatx4*>(addend)[i];
    if (relu) {
      s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
    }
    reinterpret_cast<floatx4*>(dst)[i] = s;
  }
}

// splitk_reduce_kernel for a forward whose output feeds a BatchNorm: dst = sum_s part[s] (+ addend), and the
// column sums / sums of squares of the rows this block wrote go to one partial row [2][C] (fixed order).
// Block = 256 threads = (256 / G) rows x G float4 column groups, `rows_per_block` rows per block.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                                  const float* __restrict__ addend, long long rows, int C,
                                                                  int nsplit, int rows_per_block,
                                                                  float* __restrict__ stats) {
  __shared__ floatx4 sh[2][256];
  const int G = C >> 2, tid = threadIdx.x;
  const int g = tid % G, r = tid / G, rpp = 256 / G;
  const long long n4 = rows * G;
  const long long row0 = (long long)blockIdx.x * rows_per_block;
  floatx4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  for (int k = r; k < rows_per_block; k += rpp) {
    const long long row = row0 + k;
    if (row >= rows) break;
    const long long i = row * G + g;
    floatx4 v = reinterpret_cast<const floatx4*>(part)[i];
    for (int j = 1; j < nsplit; ++j) v += reinterpret_cast<const floatx4*>(part)[(long long)j * n4 + i];
    if (addend) v += reinterpret_cast<const floatx4*>(addend)[i];
    reinterpret_cast<floatx4*>(dst)[i] = v;
    s += v;
    q += v * v;
  }
  sh[0][tid] = s;
  sh[1][tid] = q;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rpp; ++k) {
      s += sh[0][k * G + g];
      q += sh[1][k * G + g];
    }
    float* o = stats + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<floatx4*>(o + g * 4) = s;
    *reinterpret_cast<floatx4*>(o + C + g * 4) = q;
  }
}

// splitk_reduce_kernel for a dgrad whose output is the gradient of a BatchNorm(+ReLU) output: dst = sum_s part[s]
// (+ addend), and the BN backward's partial sums of the rows this block wrote (sum dy_m, sum dy_m * xhat with
// dy_m the gradient masked by the recomputed ReLU) go to one partial row [2][C] — bn_bwd_partial_kernel's output
// without its pass over dy and x.  Same block shape as splitk_reduce_stats_kernel.
__global__ __launch_bounds__(256) void splitk_reduce_bnb_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                                const float* __restrict__ addend, long long rows, int C,
                                                                int nsplit, int rows_per_block,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int relu,
                                                                float* __restrict__ stats) {
  __shared__ floatx4 sh[2][256];
  const int G = C >> 2, tid = threadIdx.x;
  const int g = tid % G, r = tid / G, rpp = 256 / G;
  const long long n4 = rows * G;
  const long long row0 = (long long)blockIdx.x * rows_per_block;
  const floatx4 sc = reinterpret_cast<const floatx4*>(scale)[g], sf = reinterpret_cast<const floatx4*>(shift)[g];
  const floatx4 mu = reinterpret_cast<const floatx4*>(mean)[g], is = reinterpret_cast<const floatx4*>(invstd)[g];
  floatx4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  for (int k = r; k < rows_per_block; k += rpp) {
    const long long row = row0 + k;
    if (row >= rows) break;
    const long long i = row * G + g;
    floatx4 v = reinterpret_cast<const floatx4*>(part)[i];
    for (int j = 1; j < nsplit; ++j) v += reinterpret_cast<const floatx4*>(part)[(long long)j * n4 + i];
    if (addend) v += reinterpret_cast<const floatx4*>(addend)[i];
    reinterpret_cast<floatx4*>(dst)[i] = v;
    const floatx4 xv = reinterpret_cast<const floatx4*>(x)[i];
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(xv[j], sc[j], sf[j]) > 0.f ? v[j] : 0.f;
    }
    s += v;
    q += v * ((xv - mu) * is);
  }
  sh[0][tid] = s;
  sh[1][tid] = q;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rpp; ++k) {
      s += sh[0][k * G + g];
      q += sh[1][k * G + g];
    }
    float* o = stats + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<floatx4*>(o + g * 4) = s;
    *reinterpret_cast<floatx4*>(o + C + g * 4) = q;
  }
}

// Reduce of a strided dgrad whose stride-parity classes were K-split to different degrees (dispatch_igemm<1>):
// the slabs are destination-shaped [f][M][C]; a row belongs to the class of its (t, h, w) parities and only rows
// of classes with f > 1 pieces went through slabs (the others were written by the conv kernel directly, with
// their BatchNorm-backward sums in its workgroup rows).  Same block shape as splitk_reduce_bnb_kernel; one
// partial row [2][C] per block (zeros when the block owns no split row).
struct ClsReduce {
  int Td, Hd, Wd;
  unsigned mgW, mgH, mgT;
  int shW, shH, shT;
  int pt, ph, pw;          // parity of (coordinate + pad) along a strided axis picks the class
  int s2t, s2h, s2w;       // 1: the axis is strided
  int f_by_par[8];         // pieces of the class with parity bits (t << 2 | h << 1 | w); 0 / 1: not split
  int add_s[3], add_n[3];  // compact addend (ConvArgs::add_s / add_n); {1,1,1}: dx-shaped
};

template <bool BNB>
__global__ __launch_bounds__(256) void splitk_reduce_cls_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                                const float* __restrict__ addend, long long rows, int C,
                                                                int rows_per_block, const ClsReduce cr,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int relu,
                                                                float* __restrict__ stats) {
  __shared__ floatx4 sh[2][256];
  const int G = C >> 2, tid = threadIdx.x;
  const int g = tid % G, r = tid / G, rpp = 256 / G;
  const long long n4 = rows * G;
  const long long row0 = (long long)blockIdx.x * rows_per_block;
  floatx4 sc = {0, 0, 0, 0}, sf = sc, mu = sc, is = sc;
  if (BNB) {
    sc = reinterpret_cast<const floatx4*>(scale)[g]; sf = reinterpret_cast<const floatx4*>(shift)[g];
    mu = reinterpret_cast<const floatx4*>(mean)[g]; is = reinterpret_cast<const floatx4*>(invstd)[g];
  }
  floatx4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  for (int k = r; k < rows_per_block; k += rpp) {
    const long long row = row0 + k;
    if (row >= rows) break;
    const unsigned m = (unsigned)row;
    const unsigned q1 = magic_div(m, cr.mgW, cr.shW);
    const int wd = m - q1 * cr.Wd;
    const unsigned q2 = magic_div(q1, cr.mgH, cr.shH);
    const int hd = q1 - q2 * cr.Hd;
    const unsigned b = magic_div(q2, cr.mgT, cr.shT);
    const int td = q2 - b * cr.Td;
    const int par = (cr.s2t ? ((td + cr.pt) & 1) << 2 : 0) | (cr.s2h ? ((hd + cr.ph) & 1) << 1 : 0) |
                    (cr.s2w ? ((wd + cr.pw) & 1) : 0);
    const int f = cr.f_by_par[par];
    if (f <= 1) continue;
    const long long i = row * G + g;
    floatx4 v = reinterpret_cast<const floatx4*>(part)[i];
    for (int j = 1; j < f; ++j) v += reinterpret_cast<const floatx4*>(part)[(long long)j * n4 + i];
    if (addend) {
      if (cr.add_s[0] * cr.add_s[1] * cr.add_s[2] > 1) {
        if (((td & (cr.add_s[0] - 1)) | (hd & (cr.add_s[1] - 1)) | (wd & (cr.add_s[2] - 1))) == 0) {
          const long long ar = (((long long)b * cr.add_n[0] + (td >> (cr.add_s[0] - 1))) * cr.add_n[1] + (hd >> (cr.add_s[1] - 1))) *
                                   cr.add_n[2] + (wd >> (cr.add_s[2] - 1));
          v += reinterpret_cast<const floatx4*>(addend)[ar * G + g];
        }
      } else {
        v += reinterpret_cast<const floatx4*>(addend)[i];
      }
    }
    reinterpret_cast<floatx4*>(dst)[i] = v;
    if (BNB) {
      const floatx4 xv = reinterpret_cast<const floatx4*>(x)[i];
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(xv[j], sc[j], sf[j]) > 0.f ? v[j] : 0.f;
      }
      s += v;
      q += v * ((xv - mu) * is);
    }
  }
  if (!BNB) return;
  sh[0][tid] = s;
  sh[1][tid] = q;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rpp; ++k) {
      s += sh[0][k * G + g];
      q += sh[1][k * G + g];
    }
    float* o = stats + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<floatx4*>(o + g * 4) = s;
    *reinterpret_cast<floatx4*>(o + C + g * 4) = q;
  }
}

// ------------------------------------------------------------------------------------------------
// Gather path (the two stems: Cin = 3 / 1, channel-first source, K = 441 / 49): per-element loads
// through a k -> (tap offset, dt, dh, dw) table held in LDS.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void igemm_gather_kernel(const ConvArgs p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int PA = BM / 32, PB = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDK;
  int2* ktab = reinterpret_cast<int2*>(smem + 2 * (BM + BN) * LDK);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  const unsigned ntm = (p.M + BM - 1) / BM, ntn = p.Cd / BN;
  const unsigned tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int ntaps = p.kt * p.kh * p.kw;
  const int K = ntaps * p.Cs;
  const int nk = (K + BK - 1) / BK;

  int a_t0[PA], a_h0[PA], a_w0[PA];
  long long a_base[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int b, td, hd, wd;
    bool ok;
    decode_row(m0 + lrow + 32 * i, p.M, p.Wd, p.Hd, p.Td, b, td, hd, wd, ok);
    a_t0[i] = td * p.st - p.pt;
    a_h0[i] = hd * p.sh - p.ph;
    a_w0[i] = wd * p.sw - p.pw;
    if (!ok) a_t0[i] = -(1 << 28);
    a_base[i] = (long long)b * p.ssB + (long long)a_t0[i] * p.ssT + (long long)a_h0[i] * p.ssH +
                (long long)a_w0[i] * p.ssW;
  }
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

  floatx4 va[PA], vb[PB];
  auto load_tile = [&](int ks) {
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
    if (ks + 1 < nk) load_tile(ks + 1);
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
// wgrad.  GEMM-K runs over m (pixels) in chunks of 32 rows; a workgroup owns a
// (64*NB)(n) x (64*KC)(k-columns) tile of dw for one m-range (split).  A k-column chunk of 64 is one
// (tap, 64-channel) pair, so KC = 3 covers a kernel row of a 64-channel (1,3,3) conv and KC = 2 two
// channel chunks of one tap.  4 waves as 2(n) x 2(k): each wave NB x KC MFMA tiles, fragments are
// conflict-free ds_read_b32 (a half-wave reads 32 consecutive n / c of one m-row).
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
  int kt_tiles;                  // k tiles per n tile
  int tiles;                     // kt_tiles * n_tiles: dw tiles per split (1-D grids: tile fastest)
  long long ssB, ssT, ssH, ssW, ssC;
  unsigned mgW, mgH, mgT;        // multiply-shift division by Wd, Hd, Td
  int shW, shH, shT;
  int out_ld, out_koff;          // row pitch / first column of a tile's destination row (0 / 0: K, 0 — slabs [Cd][K])
  long long out_split;           // elements between the slabs of two splits (0: Cd * K)
};

constexpr int WG_LD = 64 + 4;

// ------------------------------------------------------------------------------------------------
// Row-table variant: the weight-gradient kernel of every vector-path layer.
//
// A GEMM of this shape with a GEMM's loader (tools/gemm_tn_lab.hip: reduction-major operands, 4-byte
// fragment reads, 4 waves, double-buffered) runs at 113-119 TFLOP/s; the first two versions of this kernel
// (8-wave ping-pong, 4-wave double-buffered; see git history) reached 80-98 because every one of the 16 threads
// that share a pixel row re-derived that row's coordinates, tap
// validity and offset for every 32-pixel chunk (~130 VALU per thread per chunk — and an MFMA only overlaps
// with OTHER waves' instructions).  Here the workgroup decodes each of its rows ONCE into an LDS table
// (byte offset of tap (0,0,0), 3x8 validity bits); per chunk a thread reads its two entries and spends
// 4 VALU per tap; the dy row step rides in the scalar soffset and rows past M fall off num_records.
// Loads of chunk c+2 and the LDS stores of chunk c+1 are issued inside the MFMA stream of chunk c.
// ------------------------------------------------------------------------------------------------
constexpr int WG_TABC = 32;   // chunks (of 32 pixel rows) covered by one fill of the row table: 1024 rows, 8 KB

// SPLIT (round 4): the products as fp32-accurate split-bf16 — a chunk of 32 pixel rows is two k-steps of 16;
// a lane reads its column of 8 consecutive rows (the two half-waves take rows 8 apart: DW and WG_LD put them on disjoint
// banks), splits the 8 values into three bf16 terms in registers and feeds six v_mfma_f32_32x32x16_bf16 per dw tile.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wg_uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wg_split8(const float (&v)[8], wg_bf16x8& fh, wg_bf16x8& fm, wg_bf16x8& fl) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2_bf16_dot(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);      // (round 5: wgrad_group_kernel 0.76 -> 0.74 ms)
  fh = __builtin_bit_cast(wg_bf16x8, wg_uintx4{h[0], h[1], h[2], h[3]});
  fm = __builtin_bit_cast(wg_bf16x8, wg_uintx4{m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(wg_bf16x8, wg_uintx4{l[0], l[1], l[2], l[3]});
}

template <int NB, int KC, bool SPLIT = false>
__device__ __forceinline__ void wgrad_tab_body(const WgradArgs& p, const unsigned lid, float* smem) {
  constexpr int DW = 64 * NB + 4;                 // dy tile row (floats)
  constexpr int BUF = 32 * (DW + KC * WG_LD);     // one stage
  uint2* tab = reinterpret_cast<uint2*>(smem + 2 * BUF);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int ntaps = p.kt * p.kh * p.kw;
  const int K = ntaps * p.Cs;
  const int cpt = p.Cs / 64, nchunks = ntaps * cpt;

  const int bx = (int)(lid % (unsigned)p.tiles), by = (int)(lid / (unsigned)p.tiles);
  const int ktile = bx % p.kt_tiles, ntile = bx / p.kt_tiles;
  const int n0 = ntile * 64 * NB;
  const int lrow = tid >> 4;         // 0..15 (+16)
  const int lcol = (tid & 15) * 4;   // 0..60
  // the KC (tap, c0) chunks of this k tile (uniform): offset from tap (0,0,0), required validity bits
  int q_tap[KC], q_c0[KC];
  unsigned q_off[KC], q_need[KC];
  bool q_ok[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    const int q = ktile * KC + j;
    q_ok[j] = q < nchunks;
    const int qq = q_ok[j] ? q : 0;
    q_tap[j] = qq / cpt;
    q_c0[j] = (qq - q_tap[j] * cpt) * 64;
    const int dw = q_tap[j] % p.kw, r = q_tap[j] / p.kw;
    const int dh = r % p.kh, dt = r / p.kh;
    q_off[j] = (unsigned)(((dt * p.Hs + dh) * p.Ws + dw) * p.Cs + q_c0[j] + lcol) * 4;
    q_need[j] = q_ok[j] ? (1u << dt) | (1u << (8 + dh)) | (1u << (16 + dw)) : 0xffffffffu;
  }

  const int total_chunks = (p.M + 31) / 32;
  const int chunk0 = by * p.chunks_per_split;
  const int chunk1 = min(chunk0 + p.chunks_per_split, total_chunks);

  // buffer descriptors (dy and x based at the first batch item / row this split touches)
  const int pix_out = p.Td * p.Hd * p.Wd, pix_in = p.Ts * p.Hs * p.Ws;
  int b_lo = (chunk0 * 32) / pix_out;
  if (b_lo >= p.B) b_lo = p.B - 1;
  const long long x_base = (long long)b_lo * pix_in * p.Cs;
  long long x_bytes = ((long long)p.B * pix_in * p.Cs - x_base) * 4;
  if (x_bytes > 0x7fffffffll) x_bytes = 0x7fffffffll;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + x_base), 0, (int)x_bytes, 0x00020000);
  const long long d_base = (long long)min(chunk0 * 32, p.M - 1) * p.Cd;
  long long d_bytes = ((long long)p.M * p.Cd - d_base) * 4;     // exact: rows >= M read as zeros
  if (d_bytes > 0x7fffffffll) d_bytes = 0x7fffffffll;
  const __amdgpu_buffer_rsrc_t rsD =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + d_base), 0, (int)d_bytes, 0x00020000);
  const int cs4 = p.Cs * 4;
  unsigned doff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) doff[i] = (unsigned)((lrow + 16 * i) * p.Cd + n0 + lcol) * 4;

  floatx4 vd[2][NB], vx[2][KC];
  int cb = chunk0;   // first chunk covered by the row table

  auto fill_table = [&](int nrows) {
    for (int r = tid; r < nrows; r += 256) {
      const int m = cb * 32 + r;
      const bool ok = m < p.M;
      const unsigned mm = ok ? (unsigned)m : 0u;
      const unsigned q1 = magic_div(mm, p.mgW, p.shW);
      const int wd = mm - q1 * p.Wd;
      const unsigned q2 = magic_div(q1, p.mgH, p.shH);
      const int hd = q1 - q2 * p.Hd;
      const int b = magic_div(q2, p.mgT, p.shT);
      const int td = q2 - b * p.Td;
      const int t0 = td * p.st - p.pt, h0 = hd * p.sh - p.ph, w0 = wd * p.sw - p.pw;
      unsigned mt = 0, mh = 0, mw = 0;
      for (int d = 0; d < p.kt; ++d) mt |= ((unsigned)(t0 + d) < (unsigned)p.Ts ? 1u : 0u) << d;
      for (int d = 0; d < p.kh; ++d) mh |= ((unsigned)(h0 + d) < (unsigned)p.Hs ? 1u : 0u) << d;
      for (int d = 0; d < p.kw; ++d) mw |= ((unsigned)(w0 + d) < (unsigned)p.Ws ? 1u : 0u) << d;
      uint2 e;
      e.x = (unsigned)((((b - b_lo) * p.Ts + t0) * p.Hs + h0) * p.Ws + w0) * cs4;
      e.y = ok ? (mt | (mh << 8) | (mw << 16)) : 0u;
      tab[r] = e;
    }
  };
  auto load_chunk = [&](int ch) {
    const int soff = (ch - chunk0) * 32 * p.Cd * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint2 e = tab[(ch - cb) * 32 + lrow + 16 * i];
#pragma unroll
      for (int t = 0; t < NB; ++t)
        vd[i][t] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsD, doff[i] + t * 256, soff, 0));
#pragma unroll
      for (int j = 0; j < KC; ++j)
        vx[i][j] = buf_load4(rsX, (e.y & q_need[j]) == q_need[j] ? e.x + q_off[j] : OOB);
    }
  };
  auto store_chunk = [&](int buf) {
    float* Ds = smem + buf * BUF;
    float* Xs = Ds + 32 * DW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int t = 0; t < NB; ++t)
        *reinterpret_cast<floatx4*>(&Ds[(lrow + 16 * i) * DW + t * 64 + lcol]) = vd[i][t];
#pragma unroll
      for (int j = 0; j < KC; ++j)
        *reinterpret_cast<floatx4*>(&Xs[j * 32 * WG_LD + (lrow + 16 * i) * WG_LD + lcol]) = vx[i][j];
    }
  };

  floatx16 acc[NB][KC];
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

  int u = 0;
  // one chunk: 8 k-step pairs of (fragment prefetch | staging in the shadow of the pair's MFMAs)
  auto chunk_body = [&](int ch, auto ST, auto LD) {
    const float* Db = smem + u * BUF + wm * 32 * NB + l31;
    const float* Xb = smem + u * BUF + 32 * DW + wn * 32 + l31;
    if (SPLIT) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        if (decltype(ST)::value && st == 0) store_chunk(u ^ 1);
        if (decltype(LD)::value && st == 1) load_chunk(ch + 2);
        const int r0 = 16 * st + 8 * h;                  // this half-wave's 8 pixel rows
        wg_bf16x8 ah[NB], am[NB], al[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = Db[(r0 + e) * DW + t * 32];                       // A[i = n][k = m]
          wg_split8(v, ah[t], am[t], al[t]);
        }
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = Xb[j * 32 * WG_LD + (r0 + e) * WG_LD];            // B[k = m][j = c]
          wg_bf16x8 bh, bm, bl;
          wg_split8(v, bh, bm, bl);
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], bm, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bm, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], bh, acc[t][j], 0, 0, 0);
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][j], 0, 0, 0);
          }
        }
      }
      return;
    }
    float a[2][2][NB], b[2][2][KC];
    auto frag = [&](int kp, int buf) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kk = 2 * kp + q;
#pragma unroll
        for (int t = 0; t < NB; ++t) a[buf][q][t] = Db[(2 * kk + h) * DW + t * 32];             // A[i = n][k = m]
#pragma unroll
        for (int j = 0; j < KC; ++j) b[buf][q][j] = Xb[j * 32 * WG_LD + (2 * kk + h) * WG_LD];  // B[k = m][j = c]
      }
    };
    frag(0, 0);
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      if (kp + 1 < 8) frag(kp + 1, (kp + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (decltype(ST)::value && kp == 0) store_chunk(u ^ 1);
      if (decltype(LD)::value && kp == 2) load_chunk(ch + 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < NB; ++t)
#pragma unroll
          for (int j = 0; j < KC; ++j)
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp & 1][q][t], b[kp & 1][q][j], acc[t][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (; cb < chunk1; cb += WG_TABC) {
    const int ce = min(cb + WG_TABC, chunk1);
    __syncthreads();                 // the previous block's table reads are done
    fill_table((ce - cb) * 32);
    __syncthreads();
    load_chunk(cb);
    store_chunk(u);
    if (cb + 1 < ce) load_chunk(cb + 1);
    __syncthreads();
    int ch = cb;
    for (; ch + 2 < ce; ++ch, u ^= 1) {          // steady state
      chunk_body(ch, std::true_type{}, std::true_type{});
      __syncthreads();
    }
    if (ch + 1 < ce) {
      chunk_body(ch, std::true_type{}, std::false_type{});
      __syncthreads();
      ++ch;
      u ^= 1;
    }
    chunk_body(ch, std::false_type{}, std::false_type{});
    u ^= 1;
  }

  const int ld = p.out_ld ? p.out_ld : K;
  float* o = p.out + (long long)by * (p.out_split ? p.out_split : (long long)p.Cd * K) + p.out_koff;
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      if (!q_ok[j]) continue;
      const int kcol = q_tap[j] * p.Cs + q_c0[j] + wn * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 32 * NB + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        o[(long long)n * ld + kcol] = acc[t][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad_pre_body: the 128 x 128 dw tile of wgrad_tab_body<2, 2, SPLIT> with every operand fragment split ONCE, by the thread that
// stages it (round 5, late; DESIGN 8g "split once at the LDS write where the loader can do the transposition").  The contraction
// index of a weight gradient is the PIXEL, so a fragment lane holds 8 consecutive pixels of one channel — across the channels-last
// rows.  In wgrad_tab_body each of the two waves that share a fragment gathers it with eight ds_read_b32 and splits it itself
// (7 vector instructions per matrix instruction).  Here a stage is ONE k-step of 16 pixel rows; wave w stages sub-tile w
// (0 / 1: dy columns 0-63 / 64-127 of the tile, 2 / 3: the two (tap, 64-channel) x chunks), thread (o = lane >> 5, cp = lane & 31)
// loads rows 8 o .. 8 o + 7 x columns 2 cp, 2 cp + 1 (eight 8-byte loads, a row = 256 contiguous bytes per half-wave), holds the
// complete content of two fragment lanes, splits them (2 x 4 pair splits) and writes six 16-byte pieces into
//     P[sub-tile 4][column half 2][hi | mid | lo][lane 64][8 bf16]            (24 KB per stage, two stages)
// at lane slot o * 32 + i * 16 + (cp & 15): consecutive threads, consecutive slots (conflict-free), and a product step reads a
// fragment with three ds_read_b128 and no vector instruction.  Slot s of a 32-column half therefore holds column
// chan(s) = 2 (s & 15) + (s >> 4): the MFMA's row / column index is a permutation of the tile's channels, undone where the
// accumulators are stored (same 128-byte segments per row).  Pixel rows map to the same (half-wave, element) positions as in
// wgrad_tab_body and the six products are issued in the same order: bit-identical to it.
// Loads run two stages ahead of the split (two register sets), the split one stage ahead of the products; one barrier per stage;
// the row table covers 64 chunks (16 KB): 64 KB of LDS, two workgroups per CU as before.
// ------------------------------------------------------------------------------------------------
constexpr int WGP_TABC = 64;                       // chunks of 32 pixel rows per fill of the row table
constexpr int WGP_STAGE = 4 * 2 * 3 * 1024;        // bytes
constexpr size_t WGP_LDS = 2 * WGP_STAGE + sizeof(unsigned) * 2 * WGP_TABC * 32;

__device__ __forceinline__ void wgrad_pre_body(const WgradArgs& p, const unsigned lid, float* smem) {
  constexpr int NB = 2, KC = 2;
  char* const sb = reinterpret_cast<char*>(smem);
  // row table: the RESOLVED byte offset of every pixel row's x element for each of the tile's two (tap, 64-channel) chunks —
  // OOB where the tap leaves the source or the row is past M — so that staging x costs what staging dy costs (two 16-byte
  // table reads per stage, fetched one stage ahead, and one add per row): tabx[chunk j][row]
  unsigned* tabx = reinterpret_cast<unsigned*>(sb + 2 * WGP_STAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int ntaps = p.kt * p.kh * p.kw;
  const int K = ntaps * p.Cs;
  const int cpt = p.Cs / 64, nchunks = ntaps * cpt;

  const int bx = (int)(lid % (unsigned)p.tiles), by = (int)(lid / (unsigned)p.tiles);
  const int ktile = bx % p.kt_tiles, ntile = bx / p.kt_tiles;
  const int n0 = ntile * 64 * NB;
  // loader role: sub-tile = wave, rows 8 o .. 8 o + 7 of the stage, columns 2 cp, 2 cp + 1 of the sub-tile
  const int o = h, cp = l31;
  int q_tap[KC], q_c0[KC];
  bool q_ok[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    const int q = ktile * KC + j;
    q_ok[j] = q < nchunks;
    const int qq = q_ok[j] ? q : 0;
    q_tap[j] = qq / cpt;
    q_c0[j] = (qq - q_tap[j] * cpt) * 64;
  }
  // the two x chunks: offset from tap (0,0,0), required validity bits
  unsigned xq_off[KC], xq_need[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    const int dw = q_tap[j] % p.kw, r = q_tap[j] / p.kw;
    const int dh = r % p.kh, dt = r / p.kh;
    xq_off[j] = (unsigned)(((dt * p.Hs + dh) * p.Ws + dw) * p.Cs + q_c0[j]) * 4;
    xq_need[j] = q_ok[j] ? (1u << dt) | (1u << (8 + dh)) | (1u << (16 + dw)) : 0xffffffffu;
  }

  const int total_chunks = (p.M + 31) / 32;
  const int chunk0 = by * p.chunks_per_split;
  const int chunk1 = min(chunk0 + p.chunks_per_split, total_chunks);

  const int pix_out = p.Td * p.Hd * p.Wd, pix_in = p.Ts * p.Hs * p.Ws;
  int b_lo = (chunk0 * 32) / pix_out;
  if (b_lo >= p.B) b_lo = p.B - 1;
  const long long x_base = (long long)b_lo * pix_in * p.Cs;
  long long x_bytes = ((long long)p.B * pix_in * p.Cs - x_base) * 4;
  if (x_bytes > 0x7fffffffll) x_bytes = 0x7fffffffll;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + x_base), 0, (int)x_bytes, 0x00020000);
  const long long d_base = (long long)min(chunk0 * 32, p.M - 1) * p.Cd;
  long long d_bytes = ((long long)p.M * p.Cd - d_base) * 4;     // exact: rows >= M read as zeros
  if (d_bytes > 0x7fffffffll) d_bytes = 0x7fffffffll;
  const __amdgpu_buffer_rsrc_t rsD =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + d_base), 0, (int)d_bytes, 0x00020000);
  const int cs4 = p.Cs * 4, cd4 = p.Cd * 4;
  const unsigned dvoff = (unsigned)((8 * o) * p.Cd + n0 + wave * 64 + 2 * cp) * 4;   // waves 0, 1

  int cb = chunk0;                                  // first chunk covered by the row table
  auto fill_table = [&](int nrows) {
    for (int r = tid; r < nrows; r += 256) {
      const int m = cb * 32 + r;
      const bool ok = m < p.M;
      const unsigned mm = ok ? (unsigned)m : 0u;
      const unsigned q1 = magic_div(mm, p.mgW, p.shW);
      const int wd = mm - q1 * p.Wd;
      const unsigned q2 = magic_div(q1, p.mgH, p.shH);
      const int hd = q1 - q2 * p.Hd;
      const int b = magic_div(q2, p.mgT, p.shT);
      const int td = q2 - b * p.Td;
      const int t0 = td * p.st - p.pt, h0 = hd * p.sh - p.ph, w0 = wd * p.sw - p.pw;
      unsigned mt = 0, mh = 0, mw = 0;
      for (int d = 0; d < p.kt; ++d) mt |= ((unsigned)(t0 + d) < (unsigned)p.Ts ? 1u : 0u) << d;
      for (int d = 0; d < p.kh; ++d) mh |= ((unsigned)(h0 + d) < (unsigned)p.Hs ? 1u : 0u) << d;
      for (int d = 0; d < p.kw; ++d) mw |= ((unsigned)(w0 + d) < (unsigned)p.Ws ? 1u : 0u) << d;
      const unsigned off = (unsigned)((((b - b_lo) * p.Ts + t0) * p.Hs + h0) * p.Ws + w0) * cs4;
      const unsigned bits = ok ? (mt | (mh << 8) | (mw << 16)) : 0u;
#pragma unroll
      for (int j = 0; j < KC; ++j) tabx[j * (WGP_TABC * 32) + r] = (bits & xq_need[j]) == xq_need[j] ? off + xq_off[j] : OOB;
    }
  };
  // x role: the eight row offsets of the NEXT stage this wave loads (stages are loaded in order), read behind the current loads
  uint4 xo[2];
  const unsigned* const my_tab = tabx + (wave & 1) * (WGP_TABC * 32) + 8 * o;
  auto fetch_offsets = [&](int hs) {
    const int r = min(hs - 2 * cb, 2 * WGP_TABC - 1) * 16;          // (the stage past the window's last: any valid entry)
    xo[0] = *reinterpret_cast<const uint4*>(my_tab + r);
    xo[1] = *reinterpret_cast<const uint4*>(my_tab + r + 4);
  };
  // stage hs (a half chunk: 16 pixel rows) -> the register set R
  const bool dy_role = __builtin_amdgcn_readfirstlane(wave) < 2;
  auto load_stage = [&](floatx2_t (&R)[8], int hs) {
    if (dy_role) {
      const int soff = __builtin_amdgcn_readfirstlane((hs - 2 * chunk0) * 16 * cd4);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        R[e] = __builtin_bit_cast(floatx2_t, __builtin_amdgcn_raw_buffer_load_b64(rsD, dvoff, soff + e * cd4, 0));
    } else {
      // (OOB + 8 cp is still beyond num_records)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        R[e] = __builtin_bit_cast(floatx2_t, __builtin_amdgcn_raw_buffer_load_b64(rsX, xo[e >> 2][e & 3] + 8u * cp, 0, 0));
      fetch_offsets(hs + 1);
    }
  };
  // the register set R, split, -> LDS stage `buf`
  char* const st_base = sb + (((wave * 2 + (cp >> 4)) * 3) * 64 + o * 32 + (cp & 15)) * 16;
  auto store_stage = [&](const floatx2_t (&R)[8], int buf) {
    char* d = st_base + buf * WGP_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v[8] = {R[0][i], R[1][i], R[2][i], R[3][i], R[4][i], R[5][i], R[6][i], R[7][i]};
      wg_bf16x8 fh, fm, fl;
      wg_split8(v, fh, fm, fl);
      *reinterpret_cast<wg_bf16x8*>(d + i * 256) = fh;
      *reinterpret_cast<wg_bf16x8*>(d + i * 256 + 1024) = fm;
      *reinterpret_cast<wg_bf16x8*>(d + i * 256 + 2048) = fl;
    }
  };

  floatx16 acc[NB][KC];
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

  const char* const a_base = sb + ((wm * 2) * 3 * 64 + lane) * 16;            // + t * 3072 + plane * 1024
  const char* const b_base = sb + (((2 * 2) + wn) * 3 * 64 + lane) * 16;      // + j * 6144 + plane * 1024
  auto products = [&](int buf) {
    const char* A = a_base + buf * WGP_STAGE;
    const char* B = b_base + buf * WGP_STAGE;
    wg_bf16x8 ah[NB], am[NB], al[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      ah[t] = *reinterpret_cast<const wg_bf16x8*>(A + t * 3072);
      am[t] = *reinterpret_cast<const wg_bf16x8*>(A + t * 3072 + 1024);
      al[t] = *reinterpret_cast<const wg_bf16x8*>(A + t * 3072 + 2048);
    }
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const wg_bf16x8 bh = *reinterpret_cast<const wg_bf16x8*>(B + j * 6144);
      const wg_bf16x8 bm = *reinterpret_cast<const wg_bf16x8*>(B + j * 6144 + 1024);
      const wg_bf16x8 bl = *reinterpret_cast<const wg_bf16x8*>(B + j * 6144 + 2048);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][j], 0, 0, 0);
        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][j], 0, 0, 0);
        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], bm, acc[t][j], 0, 0, 0);
        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bm, acc[t][j], 0, 0, 0);
        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], bh, acc[t][j], 0, 0, 0);
        acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][j], 0, 0, 0);
      }
    }
  };

  floatx2_t R0[8], R1[8];
  for (; cb < chunk1; cb += WGP_TABC) {
    const int ce = min(cb + WGP_TABC, chunk1);
    const int s0 = 2 * cb, s1 = 2 * ce;             // stages of this table window (always an even number, >= 2)
    __syncthreads();                                // the previous window's table and stage reads are done
    fill_table((ce - cb) * 32);
    __syncthreads();
    if (!dy_role) fetch_offsets(s0);
    load_stage(R0, s0);
    load_stage(R1, s0 + 1);
    store_stage(R0, 0);
    if (s0 + 2 < s1) load_stage(R0, s0 + 2);
    __syncthreads();
    // stage s is multiplied from buffer (s - s0) & 1; its successor is split into the other buffer first, then the loads of
    // stage s + 3 go into the register set that has just been written out
    for (int s = s0; s < s1; s += 2) {
      store_stage(R1, 1);                           // stage s + 1 (exists: the window has an even number of stages)
      if (s + 3 < s1) load_stage(R1, s + 3);
      products(0);
      __syncthreads();
      if (s + 2 < s1) store_stage(R0, 0);
      if (s + 4 < s1) load_stage(R0, s + 4);
      products(1);
      __syncthreads();
    }
  }

  const int ld = p.out_ld ? p.out_ld : K;
  float* op = p.out + (long long)by * (p.out_split ? p.out_split : (long long)p.Cd * K) + p.out_koff;
  const int ccol = 2 * (l31 & 15) + (l31 >> 4);     // the channel this lane's fragment slot holds
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      if (!q_ok[j]) continue;
      const int kcol = q_tap[j] * p.Cs + q_c0[j] + wn * 32 + ccol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;                       // MFMA row = fragment slot of the dy operand
        const int n = n0 + wm * 32 * NB + t * 32 + 2 * (i & 15) + (i >> 4);
        op[(long long)n * ld + kcol] = acc[t][j][r];
      }
    }
}

template <int NB, int KC, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void wgrad_tab_kernel(const WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wgrad_tab_body<NB, KC, SPLIT>(p, xcd_remap(blockIdx.x, gridDim.x), smem);   // dw tiles of one pixel range stay on one XCD
}

// ------------------------------------------------------------------------------------------------
// Grouped weight gradients: the small layers of the network (conv3x-5x temporal / strided / residual layers, the audio
// blocks, the heads: 128 x 128 dw tiles, a few dozen to a few hundred pixel chunks each) in ONE persistent launch over
// a table of layers.  A launch per layer had to cut every layer into enough K-splits to fill the chip by itself (3 - 14
// slabs per layer, 1.77x the algorithmic HBM traffic, a reduce launch each) and paid a ramp and a drain per layer; with
// the layers of a stage in one grid the planner sizes the items of ALL layers to one common number of chunks, most
// layers need no split at all (their tiles are written straight into the gradient buffer) and the few slabs left are
// summed by one grouped reduce.
// ------------------------------------------------------------------------------------------------
constexpr int WG_GROUP_MAX = 12;
struct WgradGroupArgs {
  WgradArgs layer[WG_GROUP_MAX];
  int item0[WG_GROUP_MAX + 1];       // first item of every layer; item0[n] = number of items
  int n;
  unsigned run;                      // consecutive items that stay on one XCD
};

template <bool SPLIT, bool PRE = false>
__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(const WgradGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int total = g.item0[g.n];
  // Workgroup -> item: plain round-robin over the cost-sorted items balances the load, but the hardware deals consecutive
  // workgroups to the 8 XCDs in turn, so the dw tiles of one pixel range (consecutive items, which re-read the same x and
  // dy rows) would all sit on different L2s; a contiguous eighth of every round per XCD (xcd_remap) keeps them together
  // but hands XCD 0 the 64 most expensive items of every round (1.05 -> 1.26 ms).  Runs: g.run consecutive items stay on
  // one XCD, the runs of a round go to the XCDs in turn.
  const unsigned G = gridDim.x;
  unsigned first = blockIdx.x;
  if (G % (8u * g.run) == 0) {
    const unsigned xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;          // pos-th workgroup of this XCD
    first = ((pos / g.run) * 8 + xcd) * g.run + pos % g.run;
  }
  for (int item = (int)first; item < total; item += (int)G) {
    int l = 0;
    while (l + 1 < g.n && item >= g.item0[l + 1]) ++l;
    if (PRE) wgrad_pre_body(g.layer[l], (unsigned)(item - g.item0[l]), smem);
    else wgrad_tab_body<2, 2, SPLIT>(g.layer[l], (unsigned)(item - g.item0[l]), smem);
    __syncthreads();                 // the next item refills the row table and both stages
  }
}

struct WgradGroupReduce {
  const float* part[WG_GROUP_MAX];
  float* dw[WG_GROUP_MAX];
  long long n[WG_GROUP_MAX];         // elements of one slab
  int nsplit[WG_GROUP_MAX];
  int Kp[WG_GROUP_MAX], K[WG_GROUP_MAX], koff[WG_GROUP_MAX];    // slab row [Kp] -> dw row [K] at column koff (Kp == K: plain)
  int count;
};

// the slabs of every split (or tap-trimmed) layer of a group, summed in split order (blockIdx.y = layer); the kernel
// walks dw, so the columns of dead temporal taps are written as zeros
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(const WgradGroupReduce r) {
  __shared__ floatx4 sh[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5, y = blockIdx.y;
  const long long n4 = r.n[y] >> 2;                       // float4 elements of dw
  const int nsplit = r.nsplit[y];
  const int k4 = r.K[y] >> 2, kp4 = r.Kp[y] >> 2, ko4 = r.koff[y] >> 2;
  const long long slab4 = (n4 / k4) * kp4;
  const floatx4* p4 = reinterpret_cast<const floatx4*>(r.part[y]);
  for (long long base = (long long)blockIdx.x * 32; base < n4; base += (long long)gridDim.x * 32) {
    const long long i = base + e;
    floatx4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    if (i < n4) {
      const long long row = i / k4;
      const int c = (int)(i - row * k4);
      if (c >= ko4 && c < ko4 + kp4) {
        const long long j = row * kp4 + (c - ko4);
        int k = sl;
        for (; k + 24 < nsplit; k += 32) {
          const floatx4 v0 = p4[(long long)k * slab4 + j], v1 = p4[(long long)(k + 8) * slab4 + j],
                        v2 = p4[(long long)(k + 16) * slab4 + j], v3 = p4[(long long)(k + 24) * slab4 + j];
          s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; k < nsplit; k += 8) s0 += p4[(long long)k * slab4 + j];
      }
    }
    __syncthreads();
    sh[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < n4) {
      floatx4 t = sh[0][e];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += sh[k][e];
      reinterpret_cast<floatx4*>(r.dw[y])[i] = t;
    }
  }
}

// gather-path wgrad (stems): 64(n) x 64(k) tile, k -> tap table in LDS
__global__ __launch_bounds__(256) void wgrad_gather_kernel(const WgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ds = smem;
  float* Xs = smem + 2 * 32 * WG_LD;
  int2* ktab = reinterpret_cast<int2*>(smem + 4 * 32 * WG_LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntaps = p.kt * p.kh * p.kw;
  const int K = ntaps * p.Cs;
  const int ktile = blockIdx.x % p.kt_tiles, ntile = blockIdx.x / p.kt_tiles;
  const int n0 = ntile * 64;
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
  const int total_chunks = (p.M + 31) / 32;
  const int chunk0 = blockIdx.y * p.chunks_per_split;
  const int chunk1 = min(chunk0 + p.chunks_per_split, total_chunks);
  const int lrow = tid >> 4, lcol = (tid & 15) * 4;
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
      const long long base = (long long)b * p.ssB + (long long)t0 * p.ssT + (long long)h0 * p.ssH +
                             (long long)w0 * p.ssW;
      floatx4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int2 e = ktab[lcol + j];
        const int et = e.y & 0xff, eh = (e.y >> 8) & 0xff, ew = (e.y >> 16) & 0xff;
        const int ts = t0 + et, hs = h0 + eh, ws = w0 + ew;
        const bool okx = ok & (e.y >= 0) & ((unsigned)ts < (unsigned)p.Ts) & ((unsigned)hs < (unsigned)p.Hs) &
                         ((unsigned)ws < (unsigned)p.Ws);
        const float t = p.src[okx ? base + e.x : 0];
        v[j] = okx ? t : 0.f;
      }
      vx[i] = v;
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
    for (int kk = 0; kk < 16; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Db[(2 * kk + h) * WG_LD], Xb[(2 * kk + h) * WG_LD], acc, 0, 0, 0);
    if (ch + 1 < chunk1) store_chunk(cur ^ 1);
    __syncthreads();
  }
  float* o = p.out + (long long)blockIdx.y * p.Cd * K;
  const int kcol = ktile * 64 + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (kcol < K) o[(long long)n * K + kcol] = acc[r];
  }
}

// sum partials over splits in a fixed order (deterministic): block = 32 float4 elements x 8 split slices,
// each thread with 4 independent slab streams in flight (one dependent chain of up to 64 loads per thread
// made this kernel latency-bound: 14 us per layer)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           long long n, int nsplit) {
  __shared__ floatx4 sh[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long long n4 = n >> 2;
  const long long i = (long long)blockIdx.x * 32 + e;
  const floatx4* p4 = reinterpret_cast<const floatx4*>(part);
  floatx4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
  if (i < n4) {
    int k = sl;
    for (; k + 24 < nsplit; k += 32) {
      const floatx4 v0 = p4[(long long)k * n4 + i], v1 = p4[(long long)(k + 8) * n4 + i],
                    v2 = p4[(long long)(k + 16) * n4 + i], v3 = p4[(long long)(k + 24) * n4 + i];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; k < nsplit; k += 8) s0 += p4[(long long)k * n4 + i];
  }
  sh[sl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && i < n4) {
    floatx4 t = sh[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += sh[k][e];
    reinterpret_cast<floatx4*>(dw)[i] = t;
  }
}

// wgrad of a layer whose dead temporal taps were trimmed (trim_taps): the slabs hold [Cd][Kp] (the live taps), dw is
// [Cd][K]: live columns [koff, koff + Kp) get the slab sums in split order, the dead taps' columns exact zeros.
__global__ __launch_bounds__(256) void wgrad_reduce_scatter_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ dw, int Cd, int Kp, int K,
                                                                   int koff, int nsplit) {
  const long long n4 = (long long)Cd * K / 4, slab4 = (long long)Cd * Kp / 4;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int k4 = K / 4, kp4 = Kp / 4, ko4 = koff / 4;
  const int n = (int)(i / k4), c = (int)(i - (long long)n * k4);
  floatx4 s = {0.f, 0.f, 0.f, 0.f};
  if (c >= ko4 && c < ko4 + kp4) {
    const long long j = (long long)n * kp4 + (c - ko4);
    for (int k = 0; k < nsplit; ++k) s += reinterpret_cast<const floatx4*>(part)[(long long)k * slab4 + j];
  }
  reinterpret_cast<floatx4*>(dw)[i] = s;
}

// dx of a strided 1x1x1 convolution (the blocks' residual convolution, models/network_blocks.py:49) from its COMPACT
// form: the dense input gradient over the sub-sampled grid [B][To][Ho][Wo][C] lands at the positions divisible by the
// strides, every other position is zero (+ the addend everywhere).  One 16-byte item per thread; HBM-bound on writing dx.
__global__ __launch_bounds__(256) void dx_scatter_strided_kernel(const float* __restrict__ compact, const float* __restrict__ addend,
                                                                 float* __restrict__ dx, long long n4, int c4, int Ti, int Hi, int Wi,
                                                                 int To, int Ho, int Wo, int st, int sh, int sw) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % c4);
    long long r = i / c4;
    const int w = (int)(r % Wi); r /= Wi;
    const int h = (int)(r % Hi); r /= Hi;
    const int t = (int)(r % Ti);
    const long long b = r / Ti;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    if (t % st == 0 && h % sh == 0 && w % sw == 0)
      v = reinterpret_cast<const floatx4*>(compact)[((((b * To + t / st) * Ho + h / sh) * Wo + w / sw)) * c4 + c];
    if (addend) v += reinterpret_cast<const floatx4*>(addend)[i];
    reinterpret_cast<floatx4*>(dx)[i] = v;
  }
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

// all weights of a model in one launch: blockIdx.y picks the descriptor; a block walks 32(co) x 32(ci) tiles of
// one tap through LDS so that both the read (rows of ci) and the write (rows of co) are contiguous — the
// element-wise version read with a stride of taps*Cin floats between lanes and took 0.19 ms per step
__device__ __forceinline__ void weight_transform_body(const avid_wt_desc& d, float (*tile)[33]) {
  if (d.mode >= 5) {     // three-bf16-term split in igemm_pk_kernel<.., BS>'s fragment order: 5 of w, 6 of its transpose
    const int N = d.mode == 5 ? d.Cout : d.Cin, C = d.mode == 5 ? d.Cin : d.Cout;   // operand rows, channels per tap
    const int c8 = C / 8, cpt = C / 32, ntn = N / 64;
    const long long items = (long long)N * d.ntaps * c8;     // 8 consecutive k of one row each
    char* out = reinterpret_cast<char*>(d.wt);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
      int n, tap, k8;
      float v[8];
      if (d.mode == 5) {   // k fastest: two 16-byte reads of a weight row
        k8 = (int)(i % c8);
        const long long r = i / c8;
        tap = (int)(r % d.ntaps);
        n = (int)(r / d.ntaps);
        const floatx4* src = reinterpret_cast<const floatx4*>(d.w + ((long long)n * d.ntaps + tap) * C + k8 * 8);
        const floatx4 a = src[0], b = src[1];
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
      } else {             // n fastest: neighbouring threads read neighbouring input channels of the same 8 filters
        n = (int)(i % N);
        const long long r = i / N;
        k8 = (int)(r % c8);
        tap = (int)(r / c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = d.w[((long long)(k8 * 8 + e) * d.ntaps + tap) * d.Cin + n];
      }
      pk_bf16x8 fh, fm, fl;
      pk_split8(floatx4{v[0], v[1], v[2], v[3]}, floatx4{v[4], v[5], v[6], v[7]}, fh, fm, fl);
      const int cb = k8 >> 2, st = (k8 >> 1) & 1, hh = k8 & 1;
      const long long chunk = ((long long)tap * cpt + cb) * ntn + (n >> 6);
      char* o = out + chunk * PK_BCH + ((((n >> 5) & 1) * 2 + st) * 3) * 1024 + (hh * 32 + (n & 31)) * 16;
      *reinterpret_cast<pk_bf16x8*>(o) = fh;
      *reinterpret_cast<pk_bf16x8*>(o + 1024) = fm;
      *reinterpret_cast<pk_bf16x8*>(o + 2048) = fl;
    }
    return;
  }
  if (d.mode != 0) {     // Winograd-transformed weights of a 3x3 layer (wino.hip): mode 1 / 3 forward, 2 / 4 input gradient;
    const bool fwd = d.mode & 1;       // 1, 2 in wino_kernel's fragment order, 3, 4 in wino2_kernel's
    wino_weight_elements(d.w, d.wt, fwd ? d.Cout : d.Cin, fwd ? d.Cin : d.Cout, d.Cin, !fwd,
                         (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, d.mode <= 2 ? 2 : (W2_SPLIT ? 3 : 1));
    return;
  }
  const int tco = (d.Cout + 31) / 32, tci = (d.Cin + 31) / 32;
  const long long ntile = (long long)tco * tci * d.ntaps;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int ci0 = (int)(t % tci) * 32;
    const long long r = t / tci;
    const int co0 = (int)(r % tco) * 32;
    const int tap = (int)(r / tco);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = co0 + ty + 8 * k, ci = ci0 + tx;
      tile[ty + 8 * k][tx] = (co < d.Cout && ci < d.Cin) ? d.w[((long long)co * d.ntaps + tap) * d.Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ci = ci0 + ty + 8 * k, co = co0 + tx;
      if (ci < d.Cin && co < d.Cout) d.wt[((long long)ci * d.ntaps + tap) * d.Cout + co] = tile[tx][ty + 8 * k];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const avid_wt_desc* __restrict__ descs) {
  __shared__ float tile[32][33];
  const avid_wt_desc d = descs[blockIdx.y];
  weight_transform_body(d, tile);
}

// one descriptor, passed by value (no table in device memory: usable inside a stream capture)
__global__ __launch_bounds__(256) void weight_transform_kernel(const avid_wt_desc d) {
  __shared__ float tile[32][33];
  weight_transform_body(d, tile);
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
  const int ntaps = d->kt * d->kh * d->kw;
  if (!vec) {
    AVID_REQUIRE(ntaps * d->Cin <= KTAB_MAX - 32, AVID_E_UNSUPPORTED,
                 "conv: gather path supports K <= %d (got %d)", KTAB_MAX - 32, ntaps * d->Cin);
  } else {
    AVID_REQUIRE(d->kt <= 8 && d->kh <= 8 && d->kw <= 8, AVID_E_UNSUPPORTED, "conv: vector path supports kernel extents <= 8");
    AVID_REQUIRE((long long)d->Cout * ntaps * d->Cin * 4 < (1ll << 31), AVID_E_UNSUPPORTED, "conv: weights >= 2 GiB");
  }
  const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
  const long long Mi = (long long)d->B * d->Ti * d->Hi * d->Wi;
  AVID_REQUIRE(M < (1ll << 31) && Mi < (1ll << 31), AVID_E_UNSUPPORTED, "conv: more than 2^31 pixels");
  // 32-bit byte offsets inside one group's batch span: a 128-row tile touches at most 128/pixels + 2 items
  const long long pix_in = (long long)d->Ti * d->Hi * d->Wi, pix_out = (long long)d->To * d->Ho * d->Wo;
  const long long span_in = (128 / pix_out + 2) * pix_in * d->Cin * 4;      // fwd / wgrad read x
  const long long span_out = (128 / pix_in + 2) * pix_out * d->Cout * 4;    // dgrad reads dy
  AVID_REQUIRE(span_in < (1ll << 31) && span_out < (1ll << 31), AVID_E_UNSUPPORTED,
               "conv: one batch item is too large for 32-bit tile offsets");
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

template <int WM, int WN, int TM, int TN, int MODE, bool STRIDED = false>
static int launch_igemm(const ConvArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const size_t lds = sizeof(float) * 2 * (BM + BN) * LDK + sizeof(int) * 2 * BM;   // two groups, one stage each
  static bool attr_set = false;
  auto kern = igemm_kernel<WM, WN, TM, TN, MODE, STRIDED>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const unsigned ntn = a.Cd / BN;
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_kernel<%d,%d,%d,%d,%d>%s", WM, WN, TM, TN, MODE, STRIDED ? "s2" : "");
  const double K = (double)a.kt * a.kh * a.kw * a.Cs;
  // algorithmic work: 2*M*N*K flops; bytes = one read of src + weights, one write of dst (+ addend)
  const double srcpix = (double)a.B * a.Ts * a.Hs * a.Ws;
  const double row_lo = (double)a.mt2_begin * 2 * BM, row_hi = fmin((double)a.M, row_lo + (double)a.mt2_count * 2 * BM);
  const double frac = STRIDED ? 1.0 : (row_hi - row_lo) / (double)a.M;
  // (input gradients are priced by their source pixels = the forward's output pixels: see launch_pk)
  ScopedTimer t(s, name, frac * (MODE == 1 ? 2.0 * srcpix * a.Cd * K : 2.0 * a.M * a.Cd * K),
                frac * 4.0 * (srcpix * a.Cs + (double)a.Cd * K + (double)a.M * a.Cd * (a.addend ? 2 : 1)));
  const unsigned ptiles = STRIDED ? (unsigned)a.cls_ptiles_total : (unsigned)a.mt2_count;
  hipLaunchKernelGGL(kern, dim3(ptiles * ntn * a.nsplit), dim3(512), lds, s, a);
  return check_launch("igemm");
}

// rows per block of splitk_reduce_stats_kernel (one BatchNorm partial row each): ~1000 blocks, whole passes
static int stats_rpb(long long rows, int C) {
  const int rpp = 256 / (C / 4);
  long long r = rows / 1024;
  r = (r + rpp - 1) / rpp * rpp;
  if (r < rpp) r = rpp;
  if (r > 256) r = 256;
  return (int)r;
}

// ---- persistent K-pipelined kernel: whole rounds of tiles + a K-split tail, one launch
struct PkPlan {
  int tile;          // 0: 128x128, 1: 128x64 (4 waves, 2 workgroups / CU); 2: 256x128, 3: 256x64 (8 waves, 1 / CU)
  int BM, BN, grid;
  int full;          // tiles [0, full) run whole, round-robin over the workgroups
  int tail_units;    // then tail_tiles * f (tile, K-range) pieces, at most one per workgroup
  int f, kps;        // K splits per tail tile, k-tiles per split
  int rot;           // first workgroup that takes a tail unit
  bool paired;       // 2-per-CU grid numbered so that v and v + G/2 share a CU
  long long tail_row0;   // rows [tail_row0, M) are produced by splitk_reduce_kernel from f slabs
  size_t ws_floats;
};
// AVID_PK_OVERLAP (development): the cost of two co-resident tail units relative to running them one after the other
static double pk_overlap() {
  static std::atomic<int> v{-1};
  int m = v.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("AVID_PK_OVERLAP");
    m = e ? (int)(atof(e) * 1000.0) : 1000;
    if (m < 500 || m > 1000) m = 1000;
    v.store(m, std::memory_order_relaxed);
  }
  return m / 1000.0;
}
static PkPlan plan_pk_tile(long long M, int Cd, int nk, int tile, double* cost_out, int mode) {
  PkPlan k{};
  const int cus = device_cus();
  int per_cu = 2;
  k.tile = tile;
  switch (tile) {
    case 0: k.BM = 128; k.BN = 128; break;
    default: k.BM = 128; k.BN = 64; break;
    // (256 x 128 / 256 x 64 tiles with one workgroup per CU and a two-wave 128 x 64 tile with 64 x 64 wave tiles were
    //  measured and rejected in rounds 2 / 3: DESIGN.md 8c, 9.2)
  }
  const int ntn = Cd / k.BN;
  // CUs, a whole number of M-tiles — at least one (a CU budget below the column-block count, avid_set_cu_budget / AVID_CU_RESERVE
  // with a wide Cd: one workgroup per column block is the smallest grid the deal below can express)
  const int C = cus / ntn * ntn > 0 ? cus / ntn * ntn : ntn;
  const int G = per_cu * C;
  const long long mt_all = (M + k.BM - 1) / k.BM;
  const long long T = mt_all * ntn;
  // Work is balanced per CU.  `full` tiles (a multiple of C) are dealt whole, tile i to workgroup i mod G, so
  // every CU gets the same number; the remaining tail tiles are cut into f K-ranges ("units", at most one
  // per workgroup, handed out starting with the workgroups that got one full tile less).  Cost model in
  // k-tiles per CU: a started tile or unit pays ~2 k-tiles of prologue/epilogue, a split adds the slab pass.
  const double ovh = 2.0, red = 4.0 * 128 / k.BN;
  double best = 1e300;
  const int fmax = nk / 3 < 64 ? (nk / 3 < 1 ? 1 : nk / 3) : 64;
  for (long long m = T / C; m >= 0 && m >= T / C - 1; --m) {
    const long long full = m * C, tail = T - full;
    if (tail > G) continue;
    for (int f0 = 1; f0 <= (tail ? fmax : 1); ++f0) {
      const int kps = (nk + f0 - 1) / f0, f = (nk + kps - 1) / kps;
      const long long units = tail * f;
      if (units > G) break;
      const int maxu = units == 0 ? 0 : (int)((units + C - 1) / C);       // units on the busiest CU
      // (two units on one CU run side by side, each hiding the other's load -> LDS -> barrier chains: pk_overlap() of their sum)
      const double cost = (double)m * (nk + ovh) + (maxu == 2 ? 2.0 * pk_overlap() : (double)maxu) * (kps + ovh) + (f > 1 ? red : 0.0);
      if (cost < best - 1e-9) {
        best = cost;
        k.full = (int)full; k.tail_units = (int)units; k.f = f; k.kps = kps;
      }
    }
  }
  k.rot = per_cu == 2 ? (int)((k.full / C) % 2) * C : 0;
  k.tail_row0 = k.f > 1 ? (long long)(k.full / ntn) * k.BM : M;
  k.ws_floats = k.f > 1 ? (size_t)k.f * (size_t)(M - k.tail_row0) * Cd : 0;
  k.grid = G;
  k.paired = per_cu == 2 && G % 16 == 0;
  if ((long long)k.full + k.tail_units <= C / 2) {   // small problem: a compact grid, plain numbering
    k.grid = k.full + k.tail_units;
    k.rot = 0;
    k.paired = false;
  }
  *cost_out = best * k.BN / 128.0;                  // in k-tiles of a 128-column tile
  return k;
}

// Tile shape: 128x64 when Cd is not a multiple of 128; otherwise whichever of 128x128 / 128x64 the cost model
// rates cheaper — the narrow tile quantises M x Cd better (twice the tiles to deal) but reads the activation
// rows twice and runs a few % below the square tile per flop, so it only wins where whole rounds are lost.
static PkPlan plan_pk(long long M, int Cd, int nk, int mode) {
  double c0, c1;
  if (Cd % 128 != 0) return plan_pk_tile(M, Cd, nk, 1, &c0, mode);
  const PkPlan wide = plan_pk_tile(M, Cd, nk, 0, &c0, mode);
  const PkPlan narrow = plan_pk_tile(M, Cd, nk, 1, &c1, mode);
  return c1 * 1.04 < c0 ? narrow : wide;
}

// ---- order of the tail units: tile-major or weight-stationary (ConvArgs::pk_ws).
// The hardware deals workgroup ids round-robin to the 8 XCDs; the kernel's logical slots give every XCD contiguous runs of
// units, and each XCD's L2 fetches what its units read.  Tile-major (round 2): the K pieces and column blocks of an M-tile are
// neighbours — each activation row is fetched by one XCD, and every XCD fetches ALL weights; right for the wide layers (conv2x,
// conv3x: MBs of activations against KBs of weights).  conv4x / conv5x / audio blocks 3-4 are the other way round (9.4 MB of
// weights, 14 MB pre-split, against 2 MB of activations): profiles/r05_f counted 3.5x the algorithmic bytes at the L2s' memory
// side.  Weight-stationary: M-tile fastest, then K piece, then column block — an XCD's run covers all M-tiles of a few (column
// block, K piece) weight blocks: each weight byte goes to ONE L2, the activations are what replicates.  Both orders are priced
// with the L2 fills they cause under the kernel's slot -> XCD map (distinct weight blocks + distinct (M-tile, channel block)
// activation blocks per XCD, taps assumed to re-use their rows) and the cheaper one is taken — weight-stationary only with a
// 20 % margin (the tap re-use is optimistic for temporal layers) and only for tails without full tiles (the BatchNorm partial
// row of a workgroup covers ONE column block: igemm_pk_kernel's write_stats).  Outputs do not depend on the order: a unit
// computes the same (tile, K range) and the slabs are summed in piece order.  AVID_PK_WS: 0 never, 1 (default) by the model,
// 2 whenever the tail has no full tiles.
static int pk_ws_mode() {
  static std::atomic<int> v{-1};
  int m = v.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("AVID_PK_WS");
    m = e ? atoi(e) : 1;
    if (m < 0 || m > 2) m = 1;
    v.store(m, std::memory_order_relaxed);
  }
  return m;
}
static int pk_slot_xcd(int slot, int G, bool paired) {      // mirrors igemm_pk_kernel's slot numbering (pk_paired / xcd_remap)
  if (paired) return (slot % (G >> 1)) / (G >> 4);
  if (G < 8) return slot;
  const int q = G / 8, r = G % 8;
  return slot < r * (q + 1) ? slot / (q + 1) : r + (slot - r * (q + 1)) / q;
}
static double pk_order_bytes(const PkPlan& pk, int ntn, int mtt, int nk, int cpt, double wbytes, bool ws) {
  if (mtt > 512 || ntn * pk.f > 1024) return 1e300;
  double total = 0.0;
  const int cb = cpt < 64 ? cpt : 64;
  for (int x = 0; x < 8; ++x) {
    bool wblk[1024] = {false};
    unsigned long long ablk[512] = {0};
    for (int u = 0; u < pk.tail_units; ++u) {
      if (pk_slot_xcd((u + pk.rot) % pk.grid, pk.grid, pk.paired) != x) continue;
      int mt, nt, piece;
      if (ws) { const int r = u / mtt; mt = u - r * mtt; nt = r / pk.f; piece = r - nt * pk.f; }
      else { const int t = u / pk.f; piece = u - t * pk.f; mt = t / ntn; nt = t - mt * ntn; }
      wblk[nt * pk.f + piece] = true;
      const int k0 = piece * pk.kps, k1 = k0 + pk.kps < nk ? k0 + pk.kps : nk;
      if (k1 - k0 >= cb) ablk[mt] = ~0ull;
      else for (int k = k0; k < k1; ++k) ablk[mt] |= 1ull << ((k % cpt) & 63);
    }
    int nw = 0, na = 0;
    for (int i = 0; i < ntn * pk.f; ++i) nw += wblk[i] ? 1 : 0;
    for (int i = 0; i < mtt; ++i) na += ablk[i] == ~0ull ? cb : __builtin_popcountll(ablk[i]);
    total += (double)nw * pk.BN * pk.kps * BK * wbytes + (double)na * pk.BM * BK * 4.0;
  }
  return total;
}
// ConvArgs::pk_ws of a planned launch (0: tile-major), memoised: the launch programs call this once per layer and step
static int plan_pk_order(const PkPlan& pk, long long M, int Cd, int nk, int cpt, bool presplit) {
  const int mode = pk_ws_mode();
  if (mode == 0 || pk.full != 0 || pk.tail_units == 0 || cpt <= 0) return 0;
  const int ntn = Cd / pk.BN;
  const int mtt = (int)((M + pk.BM - 1) / pk.BM);
  if (mtt * ntn * pk.f != pk.tail_units || mtt < 2) return 0;
  if (mode == 2) return mtt;
  // (an unsplit tail leaves its BatchNorm partial sums in the workgroups' rows: another order of the units would add them in
  //  another order — the model only re-orders K-split tails, whose sums come out of the reduce; outputs stay bit-identical)
  if (pk.f == 1) return 0;
  struct Entry { long long M; int Cd, nk, cpt, presplit, grid, f, rot, cus, ws; };
  static Entry cache[64];
  static std::atomic_flag lock = ATOMIC_FLAG_INIT;
  const unsigned h = (unsigned)((M * 2654435761ull + (unsigned)Cd * 40503u + (unsigned)nk * 97u + (unsigned)cpt * 7u + (presplit ? 1 : 0) +
                                 (unsigned)pk.grid * 31u + (unsigned)pk.f * 131u) & 63u);
  while (lock.test_and_set(std::memory_order_acquire)) {}
  Entry& e = cache[h];
  const int cus = device_cus();
  if (!(e.cus == cus && e.M == M && e.Cd == Cd && e.nk == nk && e.cpt == cpt && e.presplit == (presplit ? 1 : 0) && e.grid == pk.grid &&
        e.f == pk.f && e.rot == pk.rot)) {
    const double wb = presplit ? 6.0 : 4.0;
    const double tile_major = pk_order_bytes(pk, ntn, mtt, nk, cpt, wb, false), stationary = pk_order_bytes(pk, ntn, mtt, nk, cpt, wb, true);
    e = Entry{M, Cd, nk, cpt, presplit ? 1 : 0, pk.grid, pk.f, pk.rot, cus, stationary * 1.2 < tile_major ? mtt : 0};
  }
  const int ws = e.ws;
  lock.clear(std::memory_order_release);
  return ws;
}

// n / d == (n * magic) >> shift for every n < 2^31 (d >= 1)
static void magic_for(int d, unsigned& magic, int& shift) {
  int l = 0;
  while ((1ll << l) < d) ++l;
  shift = 31 + l;
  magic = (unsigned)(((1ull << shift) + (unsigned)d - 1) / (unsigned)d);
}

// the epilogue variant (igemm_pk_kernel's EPI) of a launch
template <int MODE>
static int epi_code(const ConvArgs& a) {
  if (MODE == 1) {
    if (a.bias || a.relu || a.epi_op) return EPI_ANY;
    return (a.bnb_x ? 8 : 0) | (a.addend ? 1 : 0);
  }
  if (a.addend) {
    if (a.bias || a.relu) return EPI_ANY;
    return a.epi_op == 0 ? 1 : (a.epi_op == 1 ? 4 : 5);
  }
  if (a.bias) return a.relu ? 3 : 2;
  return a.relu ? EPI_ANY : 0;
}

template <int WM, int WN, int TM, int TN, int MODE, bool STRIDED, int EPI, bool BS>
static void launch_pk_eb(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  auto kern = igemm_pk_kernel<WM, WN, TM, TN, MODE, STRIDED, EPI, BS>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, s, a);
}

// pre-split weights (ConvArgs::wsp) are consumed by the 128 x 64 tile with the epilogues convolution layers use
template <int WM, int WN, int TM, int TN, int EPI>
constexpr bool pk_takes_split() {
  return PK_SPLIT && ((WM == 4 && WN == 1 && TM == 1 && (TN == 2 || TN == 4)) || (WM == 2 && WN == 2 && TM == 2 && TN == 2)) &&
         (EPI == 0 || EPI == 1 || EPI == 8 || EPI == 9);
}
// The 128 x 128 tile with pre-split weights (round 5): half the splits (8.1 -> ~5 vector instructions per matrix instruction), but
// its stage grows from 36.9 to 43 KB — ONE workgroup per CU instead of two.  Measured per layer (tools/conv_bench.py 64, us
// forward / input gradient): where a launch deals several tiles per CU the lost occupancy costs more than the splits save
// (conv3x temporal 51.0 -> 56.1 / 56.2 -> 62.7, conv3x strided spatial forward 114.7 -> 132.7); where every workgroup has at
// most one (tile, K piece) unit anyway it wins (conv4x strided spatial forward 85.9 -> 78.9, audio block 3 32.4 -> 29.9 / 40.4 ->
// 36.4).  Round 5, second form: the input stage unpadded and XOR-swizzled (igemm_pk_kernel's ASWZ): 40 KB per stage, two workgroups per
// CU again — conv3x temporal 49.2 -> 45.2 / 54.5 -> 51.0, conv3x strided spatial forward 109.7 -> 104.5, `<2,2,2,2,*>` 0.78 -> 0.74 ms
// per step, the step 9.782 -> 9.746 ms (three alternating pairs of 300 steps, one box).
// AVID_BS_WIDE: 2 (default) every launch of the tile, 1 only plans without a full round of tiles, 0 never.
// (round 6, same box, alternating, us per layer 1 -> 0: conv3x strided temporal 60.9 / 62.1 -> 55.6 / 55.2, conv4x strided spatial
//  85.0 / 83.5 -> 78.9 / 79.2, conv4x strided temporal 52.0 / 51.1 -> 40.7 / 42.2, conv5x strided spatial 59.9 / 59.8 -> 50.9 / 50.7,
//  conv5x strided temporal 37.3 / 37.2 -> 30.7 / 30.4, audio block 3 33.2 / 33.0 -> 33.6 / 31.2; the strided kernels of the step
//  0.454 -> 0.42 ms, the step 9.638 / 9.729 -> 9.548 / 9.589 ms; the wide form with the BatchNorm-backward + addend epilogue was
//  also the last kernel of the step with spilled registers: 44)
constexpr int S2_WIDE_DEFAULT = 0;
static int bs_wide_mode() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AVID_BS_WIDE");
    v = e ? atoi(e) : 2;
    if (v < 0 || v > 2) v = 2;
  }
  return v;
}
// The 128 x 128 tile with pre-split weights as FOUR waves of 32 rows x 128 columns (igemm_pk_kernel<4,1,1,4,*>) instead of 2 x 2
// waves of 64 x 64: no input fragment is shared by two waves any more, so each is split once (one split per 24 matrix
// instructions instead of two); every wave reads all four column blocks' weight fragments from LDS (12 instead of 6 reads per
// k-step).  AVID_BS_ROWS = 1 / 0.  Same products in the same order per output element: the outputs are bit-identical.
static bool bs_rows() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AVID_BS_ROWS");
    v = e ? (atoi(e) != 0 ? 1 : 0) : 1;
  }
  return v != 0;
}
// Strided input gradients (stride-parity classes) of layers whose Cin is a multiple of 128: AVID_S2_WIDE = 1 the 128 x 128 tile
// (2 x 2 waves, both operands split in registers), 0 the 128 x 64 tile with pre-split weights (four waves of 32 x 64).
static bool s2_wide() {
  static std::atomic<int> v{-1};
  int m = v.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("AVID_S2_WIDE");
    m = e ? (atoi(e) != 0 ? 1 : 0) : S2_WIDE_DEFAULT;
    v.store(m, std::memory_order_relaxed);
  }
  return m != 0;
}
static bool bs_wide(const PkPlan& pk) { return pk.tile == 0 && (bs_wide_mode() == 2 || (bs_wide_mode() == 1 && pk.full == 0)); }

// avid_debug_presplit_launches (tests: which instruction sequence a layer ran); incremented from the forward thread and from
// autograd's backward thread
static std::atomic<long long> g_presplit_launches{0};

template <int WM, int WN, int TM, int TN, int MODE, bool STRIDED, int EPI>
static void launch_pk_e(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  if constexpr (pk_takes_split<WM, WN, TM, TN, EPI>() && !(STRIDED && WN == 2)) {
    if (a.wsp) {
      ++g_presplit_launches;
      constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
      const size_t lds3 = sizeof(float) * 2 * (BM * (BN == 128 ? BK : LDK) + (BN / 64) * PK_BCH / 4) + (STRIDED ? sizeof(int) * 2 * BM : 0);
      launch_pk_eb<WM, WN, TM, TN, MODE, STRIDED, EPI, true>(a, grid, lds3, s);
      return;
    }
  }
  launch_pk_eb<WM, WN, TM, TN, MODE, STRIDED, EPI, false>(a, grid, lds, s);
}

template <int WM, int WN, int TM, int TN, int MODE, bool STRIDED = false>
static int launch_pk(const ConvArgs& a, int grid, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const size_t lds = sizeof(float) * 2 * (BM + BN) * LDK + (STRIDED ? sizeof(int) * 2 * BM : 0);
  static char name[64] = "";
  // (every instantiation under its own template arguments, as rocprofv3 names it: the 128 x 128 tile as four waves of 32 rows x 128
  //  columns is igemm_pk_kernel<4,1,1,4,*>; the epilogue / pre-split arguments that follow MODE are pooled, here and in
  //  tools/pmc_traffic.py)
  if (!name[0]) snprintf(name, sizeof(name), "igemm_pk_kernel<%d,%d,%d,%d,%d>%s", WM, WN, TM, TN, MODE, STRIDED ? "s2" : "");
  const double K = (double)a.kt * a.kh * a.kw * a.Cs;
  const double srcpix = (double)a.B * a.Ts * a.Hs * a.Ws;
  // algorithmic work = the multiply-adds of the FORWARD convolution this launch belongs to: 2 * (output pixels) *
  // Cout * Cin * taps.  Forward: M output pixels.  Input gradient: the forward's output pixels are this kernel's
  // SOURCE pixels (dy) — for a strided layer M (= dx pixels) is prod(stride) times more, and counting 2*M*N*K
  // would price taps that the stride-parity classes never execute.  Bytes = one read of src + weights, one write
  // of dst (+ addend) (+ the BatchNorm input x when the dgrad also makes that layer's backward partial sums).
  const double flops = MODE == 1 ? 2.0 * srcpix * a.Cd * K : 2.0 * a.M * a.Cd * K;
  ScopedTimer t(s, name, flops,
                4.0 * (srcpix * a.Cs + (double)a.Cd * K + (double)a.M * a.Cd * (1 + (a.addend ? 1 : 0) + (a.bnb_x ? 1 : 0))));
  constexpr bool MAIN = (WM * WN == 4) || (WM == 2 && WN == 1);   // the tiles layers run on; the 8-wave A/B tiles keep the generic epilogue
  const int epi = MAIN ? epi_code<MODE>(a) : EPI_ANY;
  switch (epi) {
    case 0: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, MAIN ? 0 : EPI_ANY>(a, grid, lds, s); break;
    case 1: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, MAIN ? 1 : EPI_ANY>(a, grid, lds, s); break;
    case 2: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, (MAIN && MODE == 0 && !STRIDED) ? 2 : EPI_ANY>(a, grid, lds, s); break;
    case 3: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, (MAIN && MODE == 0 && !STRIDED) ? 3 : EPI_ANY>(a, grid, lds, s); break;
    case 4: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, (MAIN && MODE == 0 && !STRIDED) ? 4 : EPI_ANY>(a, grid, lds, s); break;
    case 5: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, (MAIN && MODE == 0 && !STRIDED) ? 5 : EPI_ANY>(a, grid, lds, s); break;
    case 8: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, (MAIN && MODE == 1) ? 8 : EPI_ANY>(a, grid, lds, s); break;
    case 9: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, (MAIN && MODE == 1) ? 9 : EPI_ANY>(a, grid, lds, s); break;
    default: launch_pk_e<WM, WN, TM, TN, MODE, STRIDED, EPI_ANY>(a, grid, lds, s); break;
  }
  return check_launch("igemm_pk");
}


// ---- tconv64_kernel: which launches take it, and the launch
constexpr int TCONV_PARTS_DEFAULT = 5;      // forward + weight gradient (the input gradient: see tconv_parts)
// -1: AVID_TCONV from the environment (default 1); 0 off; 1 layers with >= 3 rounds of tiles; 2 whenever it can (atomic: see g_presplit_launches)
static std::atomic<int> g_tconv_mode{-1};
static int tconv_mode() {
  int m = g_tconv_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("AVID_TCONV");
    m = e ? atoi(e) : 1;
    if (m < 0 || m > 2) m = 1;
    g_tconv_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
// Which of the three kernels the rule applies to: bit 0 forward, bit 1 input gradient, bit 2 weight gradient (AVID_TCONV_PARTS;
// avid_tconv_configure(2) = tests: all three).  Default 5: the input gradient stays on igemm_pk_kernel IN THE STEP.  Layer
// alone it is 90 -> 64 us; in the step it carries the BatchNorm-backward sums (86-93 us against ~96) and the whole step got
// SLOWER with it: same box, alternating, ms per step / average shader clock — none 9.92 / 2.28 GHz, forward only 9.89 / 2.28,
// weight gradient only 9.89 / 2.29, input gradient only 10.14 / 2.23, all three 10.08 / 2.20; second box: none 10.03 / 2.25,
// forward + weight gradient 9.99 / 2.21, all three 10.10 / 2.17.  Every other matrix kernel of the step runs 3-7 % slower when
// these kernels are in it — the clock the chip holds over the step falls with these kernels in it (they run at the lowest
// in-kernel clock of the step, 1.75 GHz) although the package power falls too (tools/power_ab.py).
static int tconv_mode();
static int tconv_parts() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AVID_TCONV_PARTS");
    v = e ? atoi(e) & 7 : TCONV_PARTS_DEFAULT;
  }
  return tconv_mode() == 2 ? 7 : v;
}
// Grid of the persistent tile walkers: the fewest workgroups that need no more rounds than all CUs would (1568 tiles on 256
// CUs are 7 rounds either way: 224 workgroups of exactly 7 tiles).  The CUs left over are not lost: in the step these kernels
// run at the lowest clock of all (1.75 GHz) and the chip's frequency management answers a full-chip launch of them by lowering
// the clock of every kernel around it — with tconv64_kernel<1> on all 256 CUs the whole step's average clock fell from 2.29 to
// 2.25 GHz and every other matrix kernel ran 4-7 % slower (step 9.88 -> 9.95-10.05 ms although the kernel itself had got
// faster); on 224 it does not.  AVID_TCONV_GRID1 caps the input gradient's grid further (experiments).
static int balanced_grid(long long ntiles) {
  const int cus = device_cus();
  if (ntiles <= cus) return (int)ntiles;
  const long long rounds = (ntiles + cus - 1) / cus;
  return (int)((ntiles + rounds - 1) / rounds);
}
static int tconv_grid(const ConvArgs& a) {
  return balanced_grid(((long long)a.B * a.Hs * a.Ws + TC_P - 1) / TC_P);
}
static bool tconv_takes(const ConvArgs& a, int mode) {
  if (!PK_SPLIT || !tconv_mode() || !a.wsp || !(tconv_parts() & (1 << mode))) return false;
  if (a.kt != 3 || a.kh != 1 || a.kw != 1 || a.st != 1 || a.sh != 1 || a.sw != 1 || a.pt != 1 || a.ph != 0 || a.pw != 0) return false;
  if (a.Cs != 64 || a.Cd != 64 || a.Ts != TC_T || a.Td != TC_T || a.Hs != a.Hd || a.Ws != a.Wd) return false;
  if (a.bias || a.relu || a.epi_op || a.add_s[0] * a.add_s[1] * a.add_s[2] != 1) return false;
  if (mode == 0 && a.bnb_x) return false;
  if ((long long)a.M * 256 >= (1ll << 31) || a.wsp_nrec < TC_B_BYTES) return false;
  const long long ntiles = ((long long)a.B * a.Hs * a.Ws + TC_P - 1) / TC_P;
  return tconv_mode() == 2 || ntiles >= 3ll * device_cus();
}

template <int MODE, int EPI, bool AFF>
static void launch_tconv_ea(const ConvArgs& a, int grid, hipStream_t s) {
  static bool attr_set = false;
  auto kern = tconv64_kernel<MODE, EPI, AFF>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TC_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), TC_LDS, s, a);
}
template <int MODE, int EPI>
static void launch_tconv_e(const ConvArgs& a, int grid, hipStream_t s) {
  if constexpr (MODE == 0) {
    if (a.in_scale) { launch_tconv_ea<MODE, EPI, true>(a, grid, s); return; }
  }
  launch_tconv_ea<MODE, EPI, false>(a, grid, s);
}

// launches of tconv64_kernel / twgrad64_kernel since the library was loaded: [0] reading their input as it is, [1] applying a
// BatchNorm to it while staging (avid_debug_in_affine_launches: tests assert that the fused form is what ran)
static std::atomic<long long> g_tconv_launches[2];
template <int MODE>
static int launch_tconv(ConvArgs& a, hipStream_t s) {
  magic_for(a.Hs * a.Ws, a.mgW, a.shW);       // position -> clip
  int grid = tconv_grid(a);
  {   // experiment: cap the input gradient's grid (AVID_TCONV_GRID1)
    static int cap = -1;
    if (cap < 0) { const char* e = getenv("AVID_TCONV_GRID1"); cap = e ? atoi(e) : 0; }
    if (MODE == 1 && cap > 0 && grid > cap) grid = cap;
  }
  const double K = 3.0 * 64;
  ++g_tconv_launches[MODE == 0 && a.in_scale ? 1 : 0];
  ScopedTimer t(s, MODE == 0 ? "tconv64_kernel<0>" : "tconv64_kernel<1>", 2.0 * a.M * 64 * K,
                4.0 * ((double)a.M * 64 + 64 * K + (double)a.M * 64 * (1 + (a.addend ? 1 : 0) + (a.bnb_x ? 1 : 0))));
  const int epi = (a.addend ? 1 : 0) | ((MODE == 1 && a.bnb_x) ? 8 : 0);
  switch (epi) {
    case 0: launch_tconv_e<MODE, 0>(a, grid, s); break;
    case 1: launch_tconv_e<MODE, 1>(a, grid, s); break;
    case 8: launch_tconv_e<MODE, MODE == 1 ? 8 : 0>(a, grid, s); break;
    default: launch_tconv_e<MODE, MODE == 1 ? 9 : 1>(a, grid, s); break;
  }
  return check_launch("tconv64");
}

// Parity-class table of a strided dgrad (strides are 1 or 2 per axis).
static void build_classes(ConvArgs& a, int BM) {
  const int S[3] = {a.st, a.sh, a.sw}, P[3] = {a.pt, a.ph, a.pw}, Kd[3] = {a.kt, a.kh, a.kw}, D[3] = {a.Td, a.Hd, a.Wd};
  int np[3];
  for (int x = 0; x < 3; ++x) np[x] = S[x];   // number of parities along axis x
  a.ncls = 0;
  int begin = 0;
  for (int ct = 0; ct < np[0]; ++ct)
    for (int ch = 0; ch < np[1]; ++ch)
      for (int cw = 0; cw < np[2]; ++cw) {
        const int c[3] = {ct, ch, cw};
        const int k = a.ncls;
        long long pixels = a.B;
        for (int x = 0; x < 3; ++x) {
          if (S[x] == 1) {
            a.cls_p0[k][x] = 0; a.cls_n[k][x] = D[x]; a.cls_d0[k][x] = 0; a.cls_nd[k][x] = Kd[x];
          } else {   // positions p with (p + pad) % 2 == c ; taps d with d % 2 == c
            const int p0 = (c[x] + P[x]) & 1;
            a.cls_p0[k][x] = p0;
            a.cls_n[k][x] = p0 < D[x] ? (D[x] - p0 + 1) / 2 : 0;
            a.cls_d0[k][x] = c[x];
            a.cls_nd[k][x] = c[x] < Kd[x] ? (Kd[x] - c[x] + 1) / 2 : 0;
          }
          pixels *= a.cls_n[k][x];
        }
        if (pixels == 0) continue;
        a.cls_begin[k] = begin;
        begin += (int)((((pixels + BM - 1) / BM) + 1) / 2);
        a.ncls++;
      }
  a.cls_begin[a.ncls] = begin;
  a.cls_ptiles_total = begin;
}

// Parity classes of a strided dgrad for the persistent kernel: tile-unit prefix sums, classes ordered by
// decreasing tap count (the round-robin deal then gives every workgroup a similar mix), class extents'
// division magics.  Classes without taps are kept: their pixels still have to be written (zeros / addend).
static int build_classes_pk(ConvArgs& a, int BM, int ntn) {
  const int S[3] = {a.st, a.sh, a.sw}, P[3] = {a.pt, a.ph, a.pw}, Kd[3] = {a.kt, a.kh, a.kw}, D[3] = {a.Td, a.Hd, a.Wd};
  struct Cls { int p0[3], n[3], d0[3], nd[3]; long long pixels; int taps; };
  Cls cl[8];
  int nc = 0;
  for (int ct = 0; ct < S[0]; ++ct)
    for (int ch = 0; ch < S[1]; ++ch)
      for (int cw = 0; cw < S[2]; ++cw) {
        const int c[3] = {ct, ch, cw};
        Cls k;
        k.pixels = a.B;
        k.taps = 1;
        for (int x = 0; x < 3; ++x) {
          if (S[x] == 1) {
            k.p0[x] = 0; k.n[x] = D[x]; k.d0[x] = 0; k.nd[x] = Kd[x];
          } else {   // positions p with (p + pad) % 2 == c ; taps d with d % 2 == c
            const int p0 = (c[x] + P[x]) & 1;
            k.p0[x] = p0;
            k.n[x] = p0 < D[x] ? (D[x] - p0 + 1) / 2 : 0;
            k.d0[x] = c[x];
            k.nd[x] = c[x] < Kd[x] ? (Kd[x] - c[x] + 1) / 2 : 0;
          }
          k.pixels *= k.n[x];
          k.taps *= k.nd[x];
        }
        if (k.pixels == 0) continue;
        if (k.taps == 0) k.nd[0] = k.nd[1] = k.nd[2] = 0;
        cl[nc++] = k;
      }
  for (int i = 1; i < nc; ++i)   // insertion sort, taps descending (stable)
    for (int j = i; j > 0 && cl[j].taps > cl[j - 1].taps; --j) { Cls t = cl[j]; cl[j] = cl[j - 1]; cl[j - 1] = t; }
  int begin = 0;
  for (int k = 0; k < nc; ++k) {
    for (int x = 0; x < 3; ++x) {
      a.cls_p0[k][x] = cl[k].p0[x]; a.cls_n[k][x] = cl[k].n[x]; a.cls_d0[k][x] = cl[k].d0[x]; a.cls_nd[k][x] = cl[k].nd[x];
      magic_for(cl[k].n[x], a.cls_mg[k][x], a.cls_shf[k][x]);
    }
    a.cls_begin[k] = begin;
    begin += (int)((cl[k].pixels + BM - 1) / BM) * ntn;
  }
  a.ncls = nc;
  a.cls_begin[nc] = begin;
  return begin;
}

// K pieces per class of a strided dgrad.  The classes differ in K (a (1,3,3) / (1,2,2) layer has classes of 4, 2, 2
// and 1 taps; a 1x1x1 / (2,2,2) residual convolution one class with its single tap and seven with none), so
// cutting every tile into the same number of pieces (the first version: up to 8, every class through destination-
// shaped slabs and one reduce over all of dx) made the tap-less classes pay slab traffic for nothing.  Here a piece
// is ~L k-tiles whatever its class: f_c = ceil(nk_c / L), classes with f_c = 1 are written directly (addend,
// BatchNorm-backward sums in the epilogue), only the rows of classes with f_c > 1 go through slabs.  L is the
// candidate that minimises (rounds of units per CU) x (piece + per-unit overhead) + the slab pass.
struct StridedPlan {
  int f[8];
  int units, grid;
  bool any_split, any_direct;
  int fmax;
};
static StridedPlan plan_strided(ConvArgs& k, int BM, int BN, size_t ws_floats) {
  const int ntn = k.Cd / BN, cus = device_cus(), cpt = k.Cs / BK;
  build_classes_pk(k, BM, ntn);
  StridedPlan pl{};
  int nk[8], tiles[8], nkmax = 0;
  long long rows[8];
  for (int c = 0; c < k.ncls; ++c) {
    nk[c] = k.cls_nd[c][0] * k.cls_nd[c][1] * k.cls_nd[c][2] * cpt;
    tiles[c] = k.cls_begin[c + 1] - k.cls_begin[c];
    rows[c] = (long long)k.B * k.cls_n[c][0] * k.cls_n[c][1] * k.cls_n[c][2];
    nkmax = nk[c] > nkmax ? nk[c] : nkmax;
  }
  const double ovh = 2.0;                                   // k-tiles of prologue / epilogue per unit
  const double ktile_us = (BN == 128 ? 1.9 : 1.0);          // one k-tile of a CU at full MFMA rate
  double best = 1e300;
  for (int div = 1; div <= 8; ++div) {
    const int L = nkmax > 0 ? (nkmax + div - 1) / div : 1;
    if (div > 1 && L < 2) break;
    int f[8], fmax = 1;
    long long units = 0;
    double piece = 0, slab_bytes = 0;
    for (int c = 0; c < k.ncls; ++c) {
      f[c] = nk[c] > 0 ? (nk[c] + L - 1) / L : 1;
      if (f[c] > 8) f[c] = 8;
      fmax = f[c] > fmax ? f[c] : fmax;
      units += (long long)tiles[c] * f[c];
      const double pc = nk[c] > 0 ? (double)((nk[c] + f[c] - 1) / f[c]) : 0.0;
      piece = pc > piece ? pc : piece;
      if (f[c] > 1) slab_bytes += 4.0 * rows[c] * k.Cd * (2.0 * f[c] + 2.0);
    }
    if (fmax > 1 && (size_t)fmax * (size_t)k.M * k.Cd > ws_floats) continue;   // no room for the slabs
    // a CU's load: its share of all the k-tiles (+ the per-unit overhead) + one piece of imbalance
    double work = 0;
    for (int c = 0; c < k.ncls; ++c) work += (double)tiles[c] * (nk[c] + f[c] * ovh);
    double cost = (work / cus + piece) * ktile_us;
    if (fmax > 1) cost += 4.0 + slab_bytes / 4.0e6;                            // reduce launch + slab traffic at ~4 TB/s (us)
    if (cost < best - 1e-9) {
      best = cost;
      for (int c = 0; c < k.ncls; ++c) pl.f[c] = f[c];
      pl.units = (int)units;
      pl.fmax = fmax;
    }
  }
  pl.any_split = pl.fmax > 1;
  pl.any_direct = false;
  int ub = 0;
  for (int c = 0; c < k.ncls; ++c) {
    k.cls_f[c] = pl.f[c];
    k.cls_ubegin[c] = ub;
    ub += tiles[c] * pl.f[c];
    if (pl.f[c] == 1) pl.any_direct = true;
  }
  k.cls_ubegin[k.ncls] = ub;
  pl.grid = pl.units < 2 * cus ? pl.units : 2 * cus;
  return pl;
}

template <int MODE>
static int dispatch_igemm(ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t s) {
  if (MODE == 1 && (a.st > 1 || a.sh > 1 || a.sw > 1) && (long long)a.M * a.Cd * 4 < (1ll << 31) &&
      a.st <= 2 && a.sh <= 2 && a.sw <= 2) {
    // strided dgrad on the persistent kernel: (class tile, K piece) units dealt round-robin, heavy classes first
    ConvArgs k = a;
    const bool wide = a.Cd % 128 == 0 && s2_wide();
    const int BN = wide ? 128 : 64;
    const StridedPlan pl = plan_strided(k, 128, BN, ws ? ws_bytes / sizeof(float) : 0);
    k.nsplit = 1;
    k.part = static_cast<float*>(ws);
    k.part_row_begin = 0;
    k.pk_full = pl.units;
    k.pk_tail_units = 0;
    k.pk_f = 1;
    k.pk_kps = 0;
    k.pk_rot = 0;
    k.pk_ws = 0;
    const int cus = device_cus();
    const int grid = pl.grid;
    k.pk_paired = (grid == 2 * cus && grid % 16 == 0) ? 1 : 0;
    // BatchNorm-backward partials: directly written tiles leave theirs in the workgroup rows [0, grid), the rows
    // of K-split classes get theirs from the reduce (rows [grid, grid + reduce blocks))
    if (!a.bnb_x) k.stats = nullptr;
    int rc = wide ? launch_pk<2, 2, 2, 2, 1, true>(k, grid, s) : launch_pk<4, 1, 1, 2, 1, true>(k, grid, s);
    if (rc || !pl.any_split) return rc;
    ClsReduce cr{};
    cr.Td = a.Td; cr.Hd = a.Hd; cr.Wd = a.Wd;
    magic_for(a.Wd, cr.mgW, cr.shW);
    magic_for(a.Hd, cr.mgH, cr.shH);
    magic_for(a.Td, cr.mgT, cr.shT);
    cr.pt = a.pt; cr.ph = a.ph; cr.pw = a.pw;
    cr.s2t = a.st == 2; cr.s2h = a.sh == 2; cr.s2w = a.sw == 2;
    for (int x = 0; x < 3; ++x) { cr.add_s[x] = a.add_s[x]; cr.add_n[x] = a.add_n[x]; }
    for (int c = 0; c < k.ncls; ++c) {   // class -> its parity bits: positions p with (p + pad) & 1 == (cls_p0 + pad) & 1
      const int par = (cr.s2t ? ((k.cls_p0[c][0] + a.pt) & 1) << 2 : 0) | (cr.s2h ? ((k.cls_p0[c][1] + a.ph) & 1) << 1 : 0) |
                      (cr.s2w ? ((k.cls_p0[c][2] + a.pw) & 1) : 0);
      cr.f_by_par[par] = k.cls_f[c];
    }
    double split_rows = 0;
    for (int c = 0; c < k.ncls; ++c)
      if (k.cls_f[c] > 1) split_rows += (double)a.B * k.cls_n[c][0] * k.cls_n[c][1] * k.cls_n[c][2];
    const int rpb = stats_rpb(a.M, a.Cd);
    const unsigned rgrid = (unsigned)ceil_div((long long)a.M, rpb);
    if (a.bnb_x && a.stats) {
      ScopedTimer t(s, "splitk_reduce_bnb_kernel", 0.0, 4.0 * split_rows * a.Cd * (pl.fmax + 2 + (a.addend ? 1 : 0)));
      hipLaunchKernelGGL(splitk_reduce_cls_kernel<true>, dim3(rgrid), dim3(256), 0, s, k.part, a.dst, a.addend,
                         (long long)a.M, a.Cd, rpb, cr, a.bnb_x, a.bnb_scale, a.bnb_shift, a.bnb_mean, a.bnb_invstd,
                         a.bnb_relu, a.stats + (long long)grid * 2 * a.Cd);
      return check_launch("splitk_reduce_cls");
    }
    ScopedTimer t(s, "splitk_reduce_kernel", 0.0, 4.0 * split_rows * a.Cd * (pl.fmax + 1 + (a.addend ? 1 : 0)));
    hipLaunchKernelGGL(splitk_reduce_cls_kernel<false>, dim3(rgrid), dim3(256), 0, s, k.part, a.dst, a.addend,
                       (long long)a.M, a.Cd, rpb, cr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    return check_launch("splitk_reduce_cls");
  }
  if (MODE == 1 && (a.st > 1 || a.sh > 1 || a.sw > 1)) {   // strided dgrad: per-parity-class dense sub-problems
    a.nsplit = 1;
    a.ksteps_per_split = 1 << 30;
    a.part = nullptr;
    build_classes(a, 128);
    if (a.Cd % 128 == 0 && (long long)a.cls_ptiles_total * (a.Cd / 128) >= 2 * 256)
      return launch_igemm<2, 2, 2, 2, 1, true>(a, s);
    return launch_igemm<4, 1, 1, 2, 1, true>(a, s);
  }
  const int nk_total = a.kt * a.kh * a.kw * (a.Cs / BK);
  // Everything dense goes to the persistent kernel (whole rounds + K-split tail in one launch).
  {
    PkPlan pk = plan_pk(a.M, a.Cd, nk_total, MODE);
    if (tconv_takes(a, MODE)) {   // conv2x's temporal layers: taps staged once, weights resident in LDS
      const int rows = pk.grid + (pk.f > 1 ? (int)ceil_div(a.M - pk.tail_row0, stats_rpb(a.M - pk.tail_row0, a.Cd)) : 0);
      if (!(a.stats && (MODE == 0 || a.bnb_x)) || rows >= tconv_grid(a)) {
        ConvArgs k = a;
        k.stats = (MODE == 0 || a.bnb_x) ? a.stats : nullptr;
        k.stats_rows = rows;       // (the rows avid_conv_fwd_stats_rows / avid_conv_dgrad_bn_rows promised: the kernel zero-fills beyond its own)
        return launch_tconv<MODE>(k, s);
      }
    }
    AVID_REQUIRE(!a.in_scale, AVID_E_UNSUPPORTED,
                 "conv_fwd_in: this layer does not run on a kernel that applies the input's BatchNorm (avid_conv_takes_in_affine)");
    if (pk.f > 1 && (ws == nullptr || ws_bytes < sizeof(float) * pk.ws_floats)) {   // no scratch: unsplit
      AVID_REQUIRE(!a.stats, AVID_E_BADARG, "conv: BatchNorm partials need the planned workspace");
      pk.tail_units /= pk.f;
      pk.f = 1;
      pk.kps = nk_total;
      pk.tail_row0 = a.M;
    }
    ConvArgs k = a;
    k.mt2_begin = 0;
    k.mt2_count = 0;
    k.nsplit = 1;
    k.pk_full = pk.full;
    k.pk_tail_units = pk.tail_units;
    k.pk_f = pk.f;
    k.pk_kps = pk.kps;
    k.pk_rot = pk.rot;
    k.pk_paired = pk.paired ? 1 : 0;
    k.pk_ws = plan_pk_order(pk, a.M, a.Cd, nk_total, a.Cs / BK, a.wsp != nullptr);
    k.stats = (MODE == 0 || a.bnb_x) ? a.stats : nullptr;
    k.part = static_cast<float*>(ws);
    k.part_row_begin = (int)pk.tail_row0;
    magic_for(k.Wd, k.mgW, k.shW);
    magic_for(k.Hd, k.mgH, k.shH);
    magic_for(k.Td, k.mgT, k.shT);
    int rc;
    switch (pk.tile) {
      case 0: {
        const int e = epi_code<MODE>(k);      // (the epilogues the pre-split form is instantiated for: pk_takes_split)
        const bool rows = k.wsp && bs_rows() && (e == 0 || e == 1 || e == 8 || e == 9);
        rc = rows ? launch_pk<4, 1, 1, 4, MODE>(k, pk.grid, s) : launch_pk<2, 2, 2, 2, MODE>(k, pk.grid, s);
        break;
      }
      default: rc = launch_pk<4, 1, 1, 2, MODE>(k, pk.grid, s); break;
    }
    if (rc || pk.f == 1) return rc;
    const long long rows = a.M - pk.tail_row0, n4 = rows * a.Cd / 4, off = pk.tail_row0 * a.Cd;
    if (MODE == 1 && a.bnb_x && a.stats) {   // the tail rows' BatchNorm-backward partials come out of the reduce
      ScopedTimer t(s, "splitk_reduce_bnb_kernel", 0.0, 4.0 * rows * a.Cd * (pk.f + 2 + (a.addend ? 1 : 0)));
      const int rpb = stats_rpb(rows, a.Cd);
      hipLaunchKernelGGL(splitk_reduce_bnb_kernel, dim3((unsigned)ceil_div(rows, rpb)), dim3(256), 0, s, k.part,
                         a.dst + off, a.addend ? a.addend + off : nullptr, rows, a.Cd, pk.f, rpb, a.bnb_x + off,
                         a.bnb_scale, a.bnb_shift, a.bnb_mean, a.bnb_invstd, a.bnb_relu,
                         a.stats + (long long)pk.grid * 2 * a.Cd);
      return check_launch("splitk_reduce_bnb");
    }
    if (MODE == 0 && a.stats) {   // the tail rows' BatchNorm partials come out of the reduce
      ScopedTimer t(s, "splitk_reduce_stats_kernel", 0.0, 4.0 * rows * a.Cd * (pk.f + 1 + (a.addend ? 1 : 0)));
      const int rpb = stats_rpb(rows, a.Cd);
      hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((unsigned)ceil_div(rows, rpb)), dim3(256), 0, s, k.part,
                         a.dst + off, a.addend ? a.addend + off : nullptr, rows, a.Cd, pk.f, rpb,
                         a.stats + (long long)pk.grid * 2 * a.Cd);
      return check_launch("splitk_reduce_stats");
    }
    long long grid = ceil_div(n4, 256);
    if (grid > 2048) grid = 2048;
    ScopedTimer t(s, "splitk_reduce_kernel", 0.0, 4.0 * rows * a.Cd * (pk.f + 1 + (a.addend ? 1 : 0)));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, s, k.part, a.dst + off,
                       a.addend ? a.addend + off : nullptr, a.bias, n4, a.Cd / 4, pk.f, a.relu);
    return check_launch("splitk_reduce");
  }
}

// scratch floats the non-strided igemm dispatch wants for an M x Cd problem (mirrors dispatch_igemm)
static size_t igemm_ws_floats(long long M, int Cd, int nk, int mode) { return plan_pk(M, Cd, nk, mode).ws_floats; }

static int launch_gather(const ConvArgs& a, hipStream_t s) {
  const long long M = a.M;
  const bool big = ((M + 127) / 128) * (a.Cd / 64) >= 256;
  const double K = (double)a.kt * a.kh * a.kw * a.Cs;
  const double srcpix = (double)a.B * a.Ts * a.Hs * a.Ws;
  if (big) {
    const size_t lds = sizeof(float) * 2 * (128 + 64) * LDK + sizeof(int2) * KTAB_MAX;
    ScopedTimer t(s, "igemm_gather_kernel<4,1,1,2>", 2.0 * a.M * a.Cd * K,
                  4.0 * (srcpix * a.Cs + (double)a.Cd * K + (double)a.M * a.Cd));
    hipLaunchKernelGGL((igemm_gather_kernel<4, 1, 1, 2>), dim3((unsigned)(((M + 127) / 128) * (a.Cd / 64))), dim3(256),
                       lds, s, a);
  } else {
    const size_t lds = sizeof(float) * 2 * (64 + 64) * LDK + sizeof(int2) * KTAB_MAX;
    ScopedTimer t(s, "igemm_gather_kernel<2,2,1,1>", 2.0 * a.M * a.Cd * K,
                  4.0 * (srcpix * a.Cs + (double)a.Cd * K + (double)a.M * a.Cd));
    hipLaunchKernelGGL((igemm_gather_kernel<2, 2, 1, 1>), dim3((unsigned)(((M + 63) / 64) * (a.Cd / 64))), dim3(256),
                       lds, s, a);
  }
  return check_launch("igemm_gather");
}

// Dead temporal taps.  conv5x of R(2+1)D-18 sees T = 1: of its (3,1,1) kernels' three taps two only ever meet the
// zero padding (67 % of those layers' MFMAs multiplied zeros).  A tap dt is live if some output frame reads a real
// input frame through it; the live range [dt0, dt0 + kt') is computed on the host and the layer runs as a
// (kt',kh,kw) convolution with pad pt - dt0 whose weight rows are a window of the full rows (pitch = full row,
// base offset = dt0 taps).  Vector (channels-last) layers on the persistent kernel only.
struct Trim {
  avid_conv_desc d;   // kt / pt replaced; everything downstream plans and launches from this
  int dt0;            // first live temporal tap
  int kt_full;        // the real kernel depth
  bool on;
};
static Trim trim_taps(const avid_conv_desc* d) {
  Trim t{*d, 0, d->kt, false};
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("AVID_TRIM_TAPS");
    on = e ? atoi(e) != 0 : 1;
  }
  if (!on || d->x_channel_first || d->kt <= 1 || d->Cin % 64 || d->Cout % 64) return t;
  int lo = d->kt, hi = -1;
  for (int dt = 0; dt < d->kt; ++dt)
    for (int to = 0; to < d->To; ++to) {
      const int ti = to * d->st + dt - d->pt;
      if (ti >= 0 && ti < d->Ti) { lo = dt < lo ? dt : lo; hi = dt > hi ? dt : hi; break; }
    }
  if (hi < 0 || (lo == 0 && hi == d->kt - 1)) return t;
  t.dt0 = lo;
  t.d.kt = hi - lo + 1;
  t.d.pt = d->pt - lo;
  t.on = true;
  return t;
}

static void fill_common(ConvArgs& a, const avid_conv_desc* d) {
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw;
  a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.B = d->B;
  a.nsplit = 1;
  a.ksteps_per_split = 1 << 30;
  a.part = nullptr;
  a.epi_op = 0;
  a.ncls = 1;
  a.mgW = a.mgH = a.mgT = 0; a.shW = a.shH = a.shT = 0;
  a.stats = nullptr;
  a.bnb_x = a.bnb_scale = a.bnb_shift = a.bnb_mean = a.bnb_invstd = nullptr;
  a.bnb_relu = 0;
  a.cls_ptiles_total = 0;
  a.mt2_begin = 0;
  a.mt2_count = 0;
  a.part_row_begin = 0;
  a.add_s[0] = a.add_s[1] = a.add_s[2] = 1;
  a.add_n[0] = a.add_n[1] = a.add_n[2] = 0;
  a.wsp = nullptr;
  a.wsp_nrec = a.wsp_kstep = 0;
  a.stats_rows = 0;
}

// C[M][N] = A[M][K] . Bq[N][K]^T, optionally combined with Cin by min / max — the similarity GEMMs of
// the CMA search run on the same MFMA kernel as a 1x1x1 convolution over M "pixels".
int sim_gemm_nt(const float* A, const float* Bq, float* Cout_, const float* Cin, int op, long long M, int N, int K,
                hipStream_t s) {
  AVID_REQUIRE(N % 64 == 0 && K % 32 == 0 && M > 0 && M < (1ll << 31), AVID_E_UNSUPPORTED,
               "sim_gemm: need N %% 64 == 0, K %% 32 == 0 (N=%d K=%d)", N, K);
  ConvArgs a{};
  a.kt = a.kh = a.kw = 1; a.st = a.sh = a.sw = 1; a.pt = a.ph = a.pw = 0;
  a.B = 1; a.Ts = 1; a.Hs = 1; a.Ws = (int)M; a.Cs = K;
  a.Td = 1; a.Hd = 1; a.Wd = (int)M; a.Cd = N;
  a.M = (int)M;
  a.src = A; a.wk = Bq; a.addend = Cin; a.bias = nullptr; a.dst = Cout_;
  a.w_row = K; a.w_nrec = (int)((long long)N * K * 4);
  a.mode = 0; a.relu = 0; a.epi_op = op;
  a.nsplit = 1; a.ksteps_per_split = 1 << 30; a.part = nullptr; a.ncls = 1; a.cls_ptiles_total = 0;
  a.mt2_begin = 0; a.mt2_count = (int)((((long long)M + 127) / 128 + 1) / 2); a.part_row_begin = 0;
  a.ssB = a.ssT = a.ssH = a.ssW = a.ssC = 0;
  a.stats = nullptr;
  a.bnb_x = a.bnb_scale = a.bnb_shift = a.bnb_mean = a.bnb_invstd = nullptr;
  a.bnb_relu = 0;
  a.mgW = a.mgH = a.mgT = 0; a.shW = a.shH = a.shT = 0;
  a.add_s[0] = a.add_s[1] = a.add_s[2] = 1;
  a.add_n[0] = a.add_n[1] = a.add_n[2] = 0;
  a.wsp = nullptr;
  a.wsp_nrec = a.wsp_kstep = 0;
  a.stats_rows = 0;
  // persistent kernel, no K-split (K = 128: 4 k-tiles per tile, thousands of tiles): 54 -> ~90 TFLOP/s
  return dispatch_igemm<0>(a, nullptr, 0, s);
}

}  // namespace avid

using namespace avid;

// Does this layer's forward (which 0) / input gradient (1) run on the 128 x 64 tile of igemm_pk_kernel, which takes its
// weights pre-split (avid_wt_desc mode 5 / 6) as `u`?  Mirrors avid_conv_fwd / avid_conv_dgrad / dispatch_igemm.
static bool conv_takes_split(const avid_conv_desc* d, int which) {
#ifdef AVID_NO_PRESPLIT   // (tools/build_variant.sh: A/B against the fp32 instruction on one box)
  return false;
#endif
  if (!PK_SPLIT || d->x_channel_first) return false;
  if (which == 0) {
    if (d->Cin % 32 || d->Cout % 64 || wino_supported(d, 0)) return false;
    const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
    const PkPlan pk = plan_pk(M, d->Cout, trim_taps(d).d.kt * d->kh * d->kw * (d->Cin / BK), 0);
    return pk.tile == 1 || bs_wide(pk);
  }
  if (d->Cin % 64 || d->Cout % 32 || d->st > 2 || d->sh > 2 || d->sw > 2 || wino_supported(d, 1)) return false;
  const long long M = (long long)d->B * d->Ti * d->Hi * d->Wi;
  if (d->st > 1 || d->sh > 1 || d->sw > 1) return M * d->Cin * 4 < (1ll << 31) && (d->Cin % 128 != 0 || !s2_wide());
  const PkPlan pk = plan_pk(M, d->Cin, trim_taps(d).d.kt * d->kh * d->kw * (d->Cout / BK), 1);
  return pk.tile == 1 || bs_wide(pk);
}

extern "C" long long avid_debug_presplit_launches(void) { return g_presplit_launches.load(std::memory_order_relaxed); }

extern "C" int avid_tconv_configure(int mode) {
  g_tconv_mode.store((mode >= 0 && mode <= 2) ? mode : -1, std::memory_order_relaxed);
  return tconv_mode();
}

extern "C" int avid_conv_uses_split(const avid_conv_desc* d, int which) {
  if (!d || validate(d) || which < 0 || which > 1) return 0;
  return conv_takes_split(d, which) ? 1 : 0;
}

extern "C" size_t avid_conv_split_bytes(const avid_conv_desc* d) {
  if (!d || validate(d)) return 0;
  return (size_t)6 * d->Cout * d->kt * d->kh * d->kw * d->Cin;
}

// ConvArgs::wsp for an operand of N rows x (taps x C) pre-split into chunks, the first dt0 temporal taps skipped
static void set_split_operand(ConvArgs& a, const void* planes, int N, int C, int ntaps_full, int khw, int dt0) {
  const int kstep = (N / 64) * PK_BCH;
  const long long off = (long long)dt0 * khw * (C / BK) * kstep, total = (long long)ntaps_full * (C / BK) * kstep;
  a.wsp = static_cast<const char*>(planes) + off;
  a.wsp_nrec = (int)(total - off);
  a.wsp_kstep = kstep;
}

extern "C" size_t avid_conv_fwd_workspace_bytes(const avid_conv_desc* d) {
  if (!d || validate(d)) return 0;
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  if (!vec) return stem_fwd_supported(d) ? stem_fwd_ws_bytes(d) : 0;
  const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
  const Trim tr = trim_taps(d);
  const int nk = tr.d.kt * d->kh * d->kw * (d->Cin / BK);
  size_t need = sizeof(float) * igemm_ws_floats(M, d->Cout, nk, 0);
  if (wino_supported(d, 0) && wino_ws_bytes(d, 0) > need) need = wino_ws_bytes(d, 0);
  return need;
}

extern "C" int avid_conv_fwd_stats_rows(const avid_conv_desc* d) {
  if (!d || validate(d)) return 0;
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  if (!vec) return stem_fwd_supported(d) ? stem_fwd_grid(d) : 0;     // LDS-patch stems: one row per workgroup
  if (wino_supported(d, 0)) return wino_grid(d, 0);
  if (d->Cout > 1024 || (256 % (d->Cout / 4)) != 0) return 0;
  const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
  const PkPlan pk = plan_pk(M, d->Cout, trim_taps(d).d.kt * d->kh * d->kw * (d->Cin / BK), 0);
  return pk.grid + (pk.f > 1 ? (int)ceil_div(M - pk.tail_row0, stats_rpb(M - pk.tail_row0, d->Cout)) : 0);
}

// conv2x's temporal layers at the descriptor level (the twin of tconv_takes, which sees the call's arguments): forward (which 0)
// / input gradient (1) on tconv64_kernel — given the layer's pre-split weights
static bool tconv_layer(const avid_conv_desc* d, int which) {
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  const long long tc_tiles = ((long long)d->B * d->Hi * d->Wi + TC_P - 1) / TC_P;
  return PK_SPLIT && tconv_mode() && vec && which >= 0 && which < 2 && (tconv_parts() & (1 << which)) && d->kt == 3 && d->kh == 1 && d->kw == 1 &&
         d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 1 && d->ph == 0 && d->pw == 0 && d->Cin == 64 && d->Cout == 64 && d->Ti == TC_T &&
         d->To == TC_T && (long long)d->B * d->Ti * d->Hi * d->Wi * 256 < (1ll << 31) &&
         (tconv_mode() == 2 || tc_tiles >= 3ll * device_cus());
}
static bool twgrad_takes(const avid_conv_desc* d);

// AVID_IN_AFFINE (default 1): 0 = no layer offers to apply its input's BatchNorm (every BatchNorm writes its output)
static bool in_affine_on() {
  static std::atomic<int> v{-1};
  int m = v.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("AVID_IN_AFFINE");
    m = e ? (atoi(e) != 0 ? 1 : 0) : 1;
    v.store(m, std::memory_order_relaxed);
  }
  return m != 0;
}

extern "C" int avid_conv_takes_in_affine(const avid_conv_desc* d) {
  if (!d || validate(d) || !in_affine_on()) return 0;
  // both readers of the layer's input have to apply the map: the forward (tconv64_kernel) and the weight gradient (twgrad64_kernel)
  return tconv_layer(d, 0) && twgrad_takes(d) && conv_takes_split(d, 0) ? 1 : 0;
}

extern "C" long long avid_debug_in_affine_launches(int fused) { return g_tconv_launches[fused ? 1 : 0].load(std::memory_order_relaxed); }

extern "C" int avid_conv_fwd(const avid_conv_desc* d, const float* x, const float* w, const float* u, const float* addend,
                             const float* bias, int relu, float* y, float* bn_partials, void* ws, size_t ws_bytes,
                             avid_stream_t stream) {
  return avid_conv_fwd_in(d, x, nullptr, w, u, addend, bias, relu, y, bn_partials, ws, ws_bytes, stream);
}

extern "C" int avid_conv_fwd_in(const avid_conv_desc* d, const float* x, const avid_in_affine* in, const float* w, const float* u,
                                const float* addend, const float* bias, int relu, float* y, float* bn_partials, void* ws,
                                size_t ws_bytes, avid_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(x && w && y, AVID_E_BADARG, "conv_fwd: null pointer");
  if (in) {
    AVID_REQUIRE(in->scale && in->shift, AVID_E_BADARG, "conv_fwd_in: null scale / shift");
    AVID_REQUIRE(avid_conv_takes_in_affine(d) && u, AVID_E_UNSUPPORTED,
                 "conv_fwd_in: this layer does not apply its input's BatchNorm (avid_conv_takes_in_affine; needs its pre-split weights)");
  }
  if (stem_fwd_supported(d) && !addend && !bias && !relu && ws && ws_bytes >= stem_fwd_ws_bytes(d))
    return stem_fwd(d, x, w, y, bn_partials, ws, (hipStream_t)stream);
  if (wino_supported(d, 0) && !bias && !relu) {
    // (avid_conv_fwd_stats_rows promised this kernel's rows: without its workspace the partials would not match)
    AVID_REQUIRE(!bn_partials || u || (ws && ws_bytes >= wino_ws_bytes(d, 0)), AVID_E_BADARG,
                 "conv_fwd: BatchNorm partials of this layer need the planned workspace (avid_conv_fwd_workspace_bytes)");
    if (u || (ws && ws_bytes >= wino_ws_bytes(d, 0)))
      return wino_conv(d, 0, x, w, u, y, addend, bn_partials, nullptr, ws, (hipStream_t)stream);
  }
  const Trim tr = trim_taps(d);
  ConvArgs a{};
  fill_common(a, &tr.d);
  a.src = x; a.addend = addend; a.bias = bias; a.dst = y;
  {   // weights: the live temporal taps of every [kt][kh][kw][Cin] row
    const long long tapsz = (long long)d->kh * d->kw * d->Cin, off = tr.dt0 * tapsz;
    a.wk = w + off;
    a.w_row = (int)(tr.kt_full * tapsz);
    a.w_nrec = (int)(((long long)d->Cout * a.w_row - off) * 4);
  }
  if (u && conv_takes_split(d, 0)) set_split_operand(a, u, d->Cout, d->Cin, d->kt * d->kh * d->kw, d->kh * d->kw, tr.dt0);
  a.Ts = d->Ti; a.Hs = d->Hi; a.Ws = d->Wi; a.Cs = d->Cin;
  a.Td = d->To; a.Hd = d->Ho; a.Wd = d->Wo; a.Cd = d->Cout;
  a.M = d->B * d->To * d->Ho * d->Wo;
  a.mode = 0;
  a.relu = relu;
  if (in) { a.in_scale = in->scale; a.in_shift = in->shift; a.in_relu = in->relu; }
  fill_src_strides(d, a.ssB, a.ssT, a.ssH, a.ssW, a.ssC);
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  if (bn_partials) {
    AVID_REQUIRE(vec, AVID_E_BADARG, "conv_fwd: BatchNorm partials of a stem need its workspace (and no addend / bias / ReLU)");
    AVID_REQUIRE(!bias && !relu && avid_conv_fwd_stats_rows(d) > 0, AVID_E_UNSUPPORTED,
                 "conv_fwd: BatchNorm partials need the persistent kernel, no bias and no ReLU");
    a.stats = bn_partials;
  }
  return vec ? dispatch_igemm<0>(a, ws, ws_bytes, (hipStream_t)stream) : launch_gather(a, (hipStream_t)stream);
}

static size_t dgrad_wt_bytes(const avid_conv_desc* d) {
  return (sizeof(float) * (size_t)d->Cout * d->kt * d->kh * d->kw * d->Cin + 255) / 256 * 256;
}

// A strided 1x1x1 layer's input gradient as a dense problem over the sub-sampled grid + a scatter (standalone calls only:
// inside a block the residual convolution's gradient rides compact in spt_conv1's input gradient, see add_s)
static bool dgrad_compactable(const avid_conv_desc* d) {
  return d->kt == 1 && d->kh == 1 && d->kw == 1 && (d->st > 1 || d->sh > 1 || d->sw > 1) && d->pt == 0 && d->ph == 0 &&
         d->pw == 0 && !d->x_channel_first && d->Cin % 4 == 0;
}
static avid_conv_desc dgrad_compact_desc(const avid_conv_desc* d) {
  avid_conv_desc c = *d;
  c.Ti = d->To; c.Hi = d->Ho; c.Wi = d->Wo;
  c.st = c.sh = c.sw = 1;
  return c;
}
static size_t dgrad_compact_bytes(const avid_conv_desc* d) {
  return (sizeof(float) * (size_t)d->B * d->To * d->Ho * d->Wo * d->Cin + 255) / 256 * 256;
}

extern "C" size_t avid_conv_dgrad_workspace_bytes(const avid_conv_desc* d) {
  if (!d || validate(d)) return 0;
  if (dgrad_compactable(d)) {
    const avid_conv_desc c = dgrad_compact_desc(d);
    return dgrad_compact_bytes(d) + avid_conv_dgrad_workspace_bytes(&c);
  }
  const long long M = (long long)d->B * d->Ti * d->Hi * d->Wi;
  const int nk = trim_taps(d).d.kt * d->kh * d->kw * (d->Cout / BK);
  size_t fl = igemm_ws_floats(M, d->Cin, nk, 1);
  if (d->st > 1 || d->sh > 1 || d->sw > 1) {   // strided: destination-shaped slabs of the K-split classes (<= 8 pieces)
    const size_t want = (size_t)8 * (size_t)M * d->Cin;
    if (want > fl) fl = want;
  }
  size_t need = dgrad_wt_bytes(d) + sizeof(float) * fl;
  if (wino_supported(d, 1) && wino_ws_bytes(d, 1) > need) need = wino_ws_bytes(d, 1);
  return need;
}

extern "C" int avid_weight_transpose_batched(int n, const avid_wt_desc* descs_dev, int64_t max_elems,
                                             avid_stream_t stream) {
  AVID_REQUIRE(n > 0 && descs_dev && max_elems > 0, AVID_E_BADARG, "weight_transpose_batched: bad argument");
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t(s, "weight_transpose_batched_kernel", 0.0, 0.0);
  long long gx = ceil_div(max_elems, 1024);   // 32 x 32 tiles of the largest tensor, capped: blocks stride over tiles
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, s, descs_dev);
  return check_launch("weight_transpose_batched");
}

extern "C" int avid_weight_transform(const avid_wt_desc* desc, avid_stream_t stream) {
  AVID_REQUIRE(desc && desc->w && desc->wt && desc->Cout > 0 && desc->Cin > 0 && desc->ntaps > 0 && desc->mode >= 0 && desc->mode <= 6,
               AVID_E_BADARG, "weight_transform: bad descriptor");
  AVID_REQUIRE(desc->mode < 5 || (desc->mode == 5 ? desc->Cout % 64 == 0 && desc->Cin % 32 == 0 : desc->Cin % 64 == 0 && desc->Cout % 32 == 0),
               AVID_E_UNSUPPORTED, "weight_transform: mode %d needs 64 | rows and 32 | channels", desc->mode);
  hipStream_t s = (hipStream_t)stream;
  ScopedTimer t(s, "weight_transform_kernel", 0.0, 0.0);
  long long gx = ceil_div((long long)desc->Cout * desc->ntaps * desc->Cin, 1024);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(weight_transform_kernel, dim3((unsigned)gx), dim3(256), 0, s, *desc);
  return check_launch("weight_transform");
}

// rows of BatchNorm-backward partial sums a dgrad of this layer writes (0: this layer cannot — strided, or not on
// the persistent kernel); dense layers: one row per workgroup + the K-split tail's reduce blocks
extern "C" int avid_conv_dgrad_bn_rows(const avid_conv_desc* d) {
  if (!d || validate(d)) return 0;
  if (d->x_channel_first || d->Cin % 64 || d->Cout % 32 || d->st > 2 || d->sh > 2 || d->sw > 2) return 0;
  if (wino_supported(d, 1)) return wino_grid(d, 1);
  if (d->Cin > 1024 || (256 % (d->Cin / 4)) != 0) return 0;
  const long long M = (long long)d->B * d->Ti * d->Hi * d->Wi;
  if (d->st > 1 || d->sh > 1 || d->sw > 1) {   // strided: the plan of dispatch_igemm<1>'s parity-class branch
    if (M * d->Cin * 4 >= (1ll << 31)) return 0;
    const Trim tr = trim_taps(d);
    ConvArgs a{};
    fill_common(a, &tr.d);
    a.Td = d->Ti; a.Hd = d->Hi; a.Wd = d->Wi; a.Cd = d->Cin;
    const int BN = d->Cin % 128 == 0 && s2_wide() ? 128 : 64;
    a.Cs = d->Cout;
    a.M = (int)M;
    const StridedPlan pl = plan_strided(a, 128, BN, (size_t)8 * (size_t)M * d->Cin);
    return pl.grid + (pl.any_split ? (int)ceil_div(M, stats_rpb(M, d->Cin)) : 0);
  }
  const PkPlan pk = plan_pk(M, d->Cin, trim_taps(d).d.kt * d->kh * d->kw * (d->Cout / BK), 1);
  return pk.grid + (pk.f > 1 ? (int)ceil_div(M - pk.tail_row0, stats_rpb(M - pk.tail_row0, d->Cin)) : 0);
}

extern "C" int avid_conv_dgrad(const avid_conv_desc* d, const float* dy, const float* w, const float* wt_in,
                               const float* u, const float* addend, const int32_t* addend_stride, float* dx, const avid_bn_bwd_fuse* bn,
                               void* ws, size_t ws_bytes, avid_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(dy && w && dx && ws, AVID_E_BADARG, "conv_dgrad: null pointer");
  bool sparse_add = false;
  if (addend && addend_stride) {
    for (int x = 0; x < 3; ++x) {
      AVID_REQUIRE(addend_stride[x] == 1 || addend_stride[x] == 2, AVID_E_UNSUPPORTED, "conv_dgrad: addend strides must be 1 or 2");
      sparse_add |= addend_stride[x] == 2;
    }
    AVID_REQUIRE(!sparse_add || ((d->st > 1 || d->sh > 1 || d->sw > 1) &&
                                 (long long)d->B * d->Ti * d->Hi * d->Wi * d->Cin * 4 < (1ll << 31)),
                 AVID_E_UNSUPPORTED, "conv_dgrad: a compact addend needs a strided layer on the persistent kernel");
  }
  if (bn) {
    AVID_REQUIRE(bn->x && bn->scale && bn->shift && bn->mean && bn->invstd && bn->partials, AVID_E_BADARG,
                 "conv_dgrad: incomplete BatchNorm descriptor");
    AVID_REQUIRE(avid_conv_dgrad_bn_rows(d) > 0, AVID_E_UNSUPPORTED,
                 "conv_dgrad: this layer cannot produce BatchNorm-backward partial sums (avid_conv_dgrad_bn_rows == 0)");
    AVID_REQUIRE(ws_bytes >= avid_conv_dgrad_workspace_bytes(d), AVID_E_BADARG,
                 "conv_dgrad: BatchNorm partials need the planned workspace");
  }
  AVID_REQUIRE(!d->x_channel_first && d->Cin % 64 == 0 && d->Cout % 32 == 0, AVID_E_UNSUPPORTED,
               "conv_dgrad: needs channels-last x, Cin %% 64 == 0 and Cout %% 32 == 0 (Cin=%d Cout=%d)", d->Cin,
               d->Cout);
  AVID_REQUIRE(d->st <= 2 && d->sh <= 2 && d->sw <= 2, AVID_E_UNSUPPORTED, "conv_dgrad: stride > 2");
  AVID_REQUIRE(ws_bytes >= dgrad_wt_bytes(d), AVID_E_BADARG, "conv_dgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (dgrad_compactable(d) && !bn && !sparse_add && ws_bytes >= avid_conv_dgrad_workspace_bytes(d)) {
    const avid_conv_desc c = dgrad_compact_desc(d);
    float* compact = static_cast<float*>(ws);
    const size_t cb = dgrad_compact_bytes(d);
    rc = avid_conv_dgrad(&c, dy, w, wt_in, u, nullptr, nullptr, compact, nullptr, static_cast<char*>(ws) + cb, ws_bytes - cb, stream);
    if (rc) return rc;
    const long long n4 = (long long)d->B * d->Ti * d->Hi * d->Wi * (d->Cin / 4);
    long long grid = ceil_div(n4, 256 * 4);
    if (grid > 4096) grid = 4096;
    ScopedTimer t(s, "dx_scatter_strided_kernel", 0.0, 16.0 * n4 * (1 + (addend ? 1 : 0)) + 16.0 * n4 / (d->st * d->sh * d->sw));
    hipLaunchKernelGGL(dx_scatter_strided_kernel, dim3((unsigned)grid), dim3(256), 0, s, compact, addend, dx, n4, d->Cin / 4, d->Ti,
                       d->Hi, d->Wi, d->To, d->Ho, d->Wo, d->st, d->sh, d->sw);
    return check_launch("dx_scatter_strided");
  }
  if (wino_supported(d, 1) && !sparse_add && (u || ws_bytes >= wino_ws_bytes(d, 1)))
    return wino_conv(d, 1, dy, w, u, dx, addend, nullptr, bn, ws, s);   // (u: this layer's pre-transformed weights)
  const int ntaps = d->kt * d->kh * d->kw;
  const float* wt = wt_in;
  const bool split = u && conv_takes_split(d, 1);      // (u: this layer's pre-split transposed weights; wt is not read)
  if (split) wt = w;
  if (!wt) {
    float* wt_ws = static_cast<float*>(ws);
    const long long nw = (long long)d->Cout * ntaps * d->Cin;
    {
      ScopedTimer t(s, "weight_transpose_kernel", 0.0, 8.0 * nw);
      hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, s, w, wt_ws, d->Cout,
                         ntaps, d->Cin);
    }
    rc = check_launch("weight_transpose");
    if (rc) return rc;
    wt = wt_ws;
  }
  const Trim tr = trim_taps(d);
  ConvArgs a{};
  fill_common(a, &tr.d);
  a.src = dy; a.addend = addend; a.bias = nullptr; a.dst = dx;
  {   // transposed weights [Cin][kt][kh][kw][Cout]: the live temporal taps of every row
    const long long tapsz = (long long)d->kh * d->kw * d->Cout, off = tr.dt0 * tapsz;
    a.wk = wt + off;
    a.w_row = (int)(tr.kt_full * tapsz);
    a.w_nrec = (int)(((long long)d->Cin * a.w_row - off) * 4);
  }
  if (split) set_split_operand(a, u, d->Cin, d->Cout, ntaps, d->kh * d->kw, tr.dt0);
  a.Ts = d->To; a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Cout;
  a.Td = d->Ti; a.Hd = d->Hi; a.Wd = d->Wi; a.Cd = d->Cin;
  a.M = d->B * d->Ti * d->Hi * d->Wi;
  a.mode = 1;
  a.relu = 0;
  a.ssB = a.ssT = a.ssH = a.ssW = a.ssC = 0;
  if (sparse_add) {
    const int D[3] = {d->Ti, d->Hi, d->Wi};
    for (int x = 0; x < 3; ++x) {
      a.add_s[x] = addend_stride[x];
      a.add_n[x] = (D[x] + addend_stride[x] - 1) / addend_stride[x];
    }
  }
  if (bn) {
    a.bnb_x = bn->x; a.bnb_scale = bn->scale; a.bnb_shift = bn->shift; a.bnb_mean = bn->mean;
    a.bnb_invstd = bn->invstd; a.bnb_relu = bn->relu;
    a.stats = bn->partials;
  }
  return dispatch_igemm<1>(a, static_cast<char*>(ws) + dgrad_wt_bytes(d), ws_bytes - dgrad_wt_bytes(d), s);
}

// weight gradients of the vector-path layers as split-bf16 products (wgrad_tab_body<.., SPLIT>); AVID_WGRAD_BF16X3=0: fp32 MFMA
static bool wgrad_split() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("AVID_WGRAD_BF16X3");
    on = e ? atoi(e) != 0 : 1;
  }
  return on != 0;
}

// grouped weight gradients with the fragments split once at the LDS write (wgrad_pre_body); AVID_WGRAD_PRE=0: split per use
static int g_wgrad_pre = -1;
static bool wgrad_pre() {
  if (g_wgrad_pre < 0) {
    const char* e = getenv("AVID_WGRAD_PRE");
    g_wgrad_pre = e ? (atoi(e) != 0 ? 1 : 0) : 1;
  }
  return g_wgrad_pre != 0 && wgrad_split();
}
extern "C" int avid_wgrad_pre_configure(int on) {
  g_wgrad_pre = on < 0 ? -1 : (on ? 1 : 0);
  return AVID_OK;
}

// ---- wgrad plan
struct WgradPlan {
  bool vec;
  int NB, KC;
  int kt_tiles, n_tiles, nsplit, cps;
};

static WgradPlan wgrad_plan(const avid_conv_desc* d) {
  WgradPlan pl;
  const int ntaps = d->kt * d->kh * d->kw;
  const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
  pl.vec = (d->Cin % 64 == 0) && !d->x_channel_first;
  if (pl.vec) {
    pl.NB = d->Cout % 128 == 0 ? 2 : 1;
    pl.KC = pl.NB == 1 ? 3 : 2;
    const int nchunks = ntaps * (d->Cin / 64);
    pl.kt_tiles = (nchunks + pl.KC - 1) / pl.KC;
    pl.n_tiles = d->Cout / (64 * pl.NB);
  } else {
    pl.NB = pl.KC = 1;
    pl.kt_tiles = (int)ceil_div((long long)ntaps * d->Cin, 64);
    pl.n_tiles = d->Cout / 64;
  }
  const long long tiles = (long long)pl.kt_tiles * pl.n_tiles;
  const long long chunks = ceil_div(M, 32);
  // 4-wave workgroups, two per CU: one launch aims to fill 512 of them = the whole chip (fewer splits measured: no gain inside the step)
  const int slots = 512;
  long long want = slots / tiles;
  long long max_split = chunks / 4 > 0 ? chunks / 4 : 1;  // >= 4 chunks (128 rows) per split (conv4x temporal: 56 -> 42 us vs 8)
  long long ns = want < 1 ? 1 : (want > max_split ? max_split : want);
  if (ns > 512) ns = 512;
  // 32-bit byte offsets inside one split: (batch items touched) * bytes per input item < 2 GiB
  const long long pix_out = (long long)d->To * d->Ho * d->Wo;
  const long long per_b_in = (long long)d->Ti * d->Hi * d->Wi * d->Cin * 4;
  while (ns < chunks && (ceil_div(chunks, ns) * 32 / pix_out + 2) * per_b_in >= (1ll << 31)) ns *= 2;
  if (ns > chunks) ns = chunks;
  pl.cps = (int)ceil_div(chunks, ns);
  pl.nsplit = (int)ceil_div(chunks, pl.cps);
  return pl;
}

// kernel arguments of one layer (d: the layer with its dead temporal taps already trimmed)
static void fill_wgrad_args(WgradArgs& a, const avid_conv_desc* d, const float* x, const float* dy, float* out,
                            const WgradPlan& pl) {
  memset(&a, 0, sizeof(a));
  a.src = x; a.dy = dy; a.out = out;
  a.B = d->B; a.Ts = d->Ti; a.Hs = d->Hi; a.Ws = d->Wi; a.Cs = d->Cin;
  a.Td = d->To; a.Hd = d->Ho; a.Wd = d->Wo; a.Cd = d->Cout;
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw;
  a.st = d->st; a.sh = d->sh; a.sw = d->sw;
  a.pt = d->pt; a.ph = d->ph; a.pw = d->pw;
  a.M = d->B * d->To * d->Ho * d->Wo;
  a.nsplit = pl.nsplit; a.chunks_per_split = pl.cps; a.kt_tiles = pl.kt_tiles;
  a.tiles = pl.kt_tiles * pl.n_tiles;
  magic_for(a.Wd, a.mgW, a.shW);
  magic_for(a.Hd, a.mgH, a.shH);
  magic_for(a.Td, a.mgT, a.shT);
  fill_src_strides(d, a.ssB, a.ssT, a.ssH, a.ssW, a.ssC);
}

// ---- twgrad64_kernel: the layers it takes (the descriptor-level twin of tconv_takes; same switch: avid_tconv_configure)
static bool twgrad_layer(const avid_conv_desc* d) {
  return PK_SPLIT && wgrad_split() && !d->x_channel_first && d->kt == 3 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 &&
         d->sw == 1 && d->pt == 1 && d->ph == 0 && d->pw == 0 && d->Cin == 64 && d->Cout == 64 && d->Ti == TC_T && d->To == TC_T &&
         (long long)d->B * d->Ti * d->Hi * d->Wi * 256 < (1ll << 31);
}
static int twgrad_grid(const avid_conv_desc* d) {
  return balanced_grid(((long long)d->B * d->Hi * d->Wi + TWG_P - 1) / TWG_P);
}
static size_t twgrad_ws_bytes(const avid_conv_desc* d) { return sizeof(float) * (size_t)twgrad_grid(d) * 64 * 3 * 64; }
static bool twgrad_takes(const avid_conv_desc* d) {
  if (!tconv_mode() || !(tconv_parts() & 4) || !twgrad_layer(d)) return false;
  const long long ntiles = ((long long)d->B * d->Hi * d->Wi + TWG_P - 1) / TWG_P;
  return tconv_mode() == 2 || ntiles >= 6ll * device_cus();
}

extern "C" size_t avid_conv_wgrad_workspace_bytes(const avid_conv_desc* d) {
  if (!d || validate(d)) return 0;
  const Trim tr = trim_taps(d);      // (a trimmed layer always goes through slabs: the reduce scatters them into dw)
  WgradPlan pl = wgrad_plan(&tr.d);
  size_t nb = sizeof(float) * (size_t)pl.nsplit * d->Cout * tr.d.kt * d->kh * d->kw * d->Cin;
  if (stem_wgrad_supported(d) && stem_wgrad_ws_bytes(d) > nb) nb = stem_wgrad_ws_bytes(d);
  if (wino_wgrad_supported(d) && wino_wgrad_ws_bytes(d) > nb) nb = wino_wgrad_ws_bytes(d);
  if (twgrad_layer(d) && twgrad_ws_bytes(d) > nb) nb = twgrad_ws_bytes(d);
  return nb;
}

extern "C" int avid_conv_wgrad(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws,
                               size_t ws_bytes, avid_stream_t stream) {
  return avid_conv_wgrad_in(d, x, nullptr, dy, dw, ws, ws_bytes, stream);
}

extern "C" int avid_conv_wgrad_in(const avid_conv_desc* d, const float* x, const avid_in_affine* in, const float* dy, float* dw,
                                  void* ws, size_t ws_bytes, avid_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(x && dy && dw, AVID_E_BADARG, "conv_wgrad: null pointer");
  if (in) {
    AVID_REQUIRE(in->scale && in->shift, AVID_E_BADARG, "conv_wgrad_in: null scale / shift");
    AVID_REQUIRE(avid_conv_takes_in_affine(d) && ws && ws_bytes >= twgrad_ws_bytes(d), AVID_E_UNSUPPORTED,
                 "conv_wgrad_in: this layer does not apply its input's BatchNorm (avid_conv_takes_in_affine; needs its workspace)");
  }
  if (stem_wgrad_supported(d) && ws && ws_bytes >= stem_wgrad_ws_bytes(d))
    return stem_wgrad(d, x, dy, dw, ws, (hipStream_t)stream);
  if (wino_wgrad_supported(d) && ws && ws_bytes >= wino_wgrad_ws_bytes(d)) {
    hipStream_t s = (hipStream_t)stream;
    int nsplit = 1;
    rc = wino_wgrad(d, x, dy, dw, ws, &nsplit, s);
    if (rc || nsplit == 1) return rc;
    const long long n = (long long)d->Cout * 9 * d->Cin;
    ScopedTimer t(s, "wgrad_reduce_kernel", 0.0, 4.0 * n * (nsplit + 1));
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(n, 128)), dim3(256), 0, s,
                       static_cast<const float*>(ws), dw, n, nsplit);
    return check_launch("wgrad_reduce");
  }
  if (twgrad_takes(d) && ws && ws_bytes >= twgrad_ws_bytes(d)) {   // conv2x's temporal layers: the taps share their split fragments
    hipStream_t s = (hipStream_t)stream;
    TwgradArgs a{};
    a.x = x; a.dy = dy; a.part = static_cast<float*>(ws);
    a.B = d->B; a.HW = d->Hi * d->Wi;
    if (in) { a.in_scale = in->scale; a.in_shift = in->shift; a.in_relu = in->relu; }
    magic_for(a.HW, a.mg, a.sh);
    const int grid = twgrad_grid(d);
    static bool set = false;
    if (!set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(twgrad64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TWG_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(twgrad64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TWG_LDS);
      set = true;
    }
    const double M = (double)d->B * TC_T * a.HW, K = 3.0 * 64;
    {
      ++g_tconv_launches[in ? 1 : 0];
      ScopedTimer t(s, "twgrad64_kernel", 2.0 * M * 64 * K, 4.0 * (M * 64 + M * 64 + 64 * K));
      if (in) hipLaunchKernelGGL(twgrad64_kernel<true>, dim3(grid), dim3(512), TWG_LDS, s, a);
      else hipLaunchKernelGGL(twgrad64_kernel<false>, dim3(grid), dim3(512), TWG_LDS, s, a);
    }
    rc = check_launch("twgrad64");
    if (rc) return rc;
    const long long n = 64ll * 3 * 64;
    ScopedTimer t(s, "wgrad_reduce_kernel", 0.0, 4.0 * n * (grid + 1));
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(n, 128)), dim3(256), 0, s, static_cast<const float*>(ws), dw, n, grid);
    return check_launch("wgrad_reduce");
  }
  Trim tr = trim_taps(d);
  if (tr.on && !(ws && ws_bytes >= avid_conv_wgrad_workspace_bytes(d))) tr = Trim{*d, 0, d->kt, false};   // no scratch
  const avid_conv_desc* dfull = d;
  d = &tr.d;                         // from here on: the (possibly trimmed) layer
  WgradPlan pl = wgrad_plan(d);
  AVID_REQUIRE(pl.nsplit == 1 || (ws && ws_bytes >= avid_conv_wgrad_workspace_bytes(dfull)), AVID_E_BADARG,
               "conv_wgrad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  WgradArgs a;
  fill_wgrad_args(a, d, x, dy, (pl.nsplit == 1 && !tr.on) ? dw : static_cast<float*>(ws), pl);
  dim3 grid((unsigned)(pl.kt_tiles * pl.n_tiles), (unsigned)pl.nsplit);
  dim3 grid_pp((unsigned)(pl.kt_tiles * pl.n_tiles), (unsigned)((pl.nsplit + 1) / 2));   // two splits per workgroup
  {
    const double K = (double)a.kt * a.kh * a.kw * a.Cs;
    const double srcpix = (double)a.B * a.Ts * a.Hs * a.Ws;
    const char* name = !pl.vec ? "wgrad_gather_kernel" : (pl.NB == 1 ? "wgrad_tab_kernel<1,3>" : "wgrad_tab_kernel<2,2>");
    ScopedTimer t(s, name, 2.0 * a.M * a.Cd * K, 4.0 * (srcpix * a.Cs + (double)a.M * a.Cd + (double)a.Cd * K));
    if (!pl.vec) {
      const size_t lds = sizeof(float) * 4 * 32 * WG_LD + sizeof(int2) * 64;
      hipLaunchKernelGGL(wgrad_gather_kernel, grid, dim3(256), lds, s, a);
    } else if (pl.NB == 1) {
      const size_t lds = sizeof(float) * 2 * 32 * (64 * 1 + 4 + 3 * WG_LD) + sizeof(uint2) * WG_TABC * 32;
      static bool set = false;
      if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tab_kernel<1, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tab_kernel<1, 3, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        set = true;
      }
      if (wgrad_split()) hipLaunchKernelGGL((wgrad_tab_kernel<1, 3, true>), dim3(grid.x * grid.y), dim3(256), lds, s, a);
      else hipLaunchKernelGGL((wgrad_tab_kernel<1, 3>), dim3(grid.x * grid.y), dim3(256), lds, s, a);
    } else {
      const size_t lds = sizeof(float) * 2 * 32 * (64 * 2 + 4 + 2 * WG_LD) + sizeof(uint2) * WG_TABC * 32;
      static bool set = false;
      if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tab_kernel<2, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tab_kernel<2, 2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        set = true;
      }
      if (wgrad_split()) hipLaunchKernelGGL((wgrad_tab_kernel<2, 2, true>), dim3(grid.x * grid.y), dim3(256), lds, s, a);
      else hipLaunchKernelGGL((wgrad_tab_kernel<2, 2>), dim3(grid.x * grid.y), dim3(256), lds, s, a);
    }
  }
  rc = check_launch("wgrad");
  if (rc) return rc;
  if (tr.on) {   // live taps' slabs -> their columns of dw, zeros for the dead taps
    const int tap = d->kh * d->kw * d->Cin, Kp = d->kt * tap, K = tr.kt_full * tap, koff = tr.dt0 * tap;
    const long long n4 = (long long)d->Cout * K / 4;
    ScopedTimer t(s, "wgrad_reduce_kernel", 0.0, 4.0 * ((double)d->Cout * Kp * pl.nsplit + (double)d->Cout * K));
    hipLaunchKernelGGL(wgrad_reduce_scatter_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, s,
                       static_cast<const float*>(ws), dw, d->Cout, Kp, K, koff, pl.nsplit);
    return check_launch("wgrad_reduce_scatter");
  }
  if (pl.nsplit > 1) {
    const long long n = (long long)d->Cout * d->kt * d->kh * d->kw * d->Cin;
    ScopedTimer t(s, "wgrad_reduce_kernel", 0.0, 4.0 * n * (pl.nsplit + 1));
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(n, 128)), dim3(256), 0, s,
                       static_cast<const float*>(ws), dw, n, pl.nsplit);
    rc = check_launch("wgrad_reduce");
  }
  return rc;
}

// ---- grouped weight gradients (wgrad_group_kernel)
extern "C" int avid_conv_wgrad_groupable(const avid_conv_desc* d) {
  if (!d || validate(d)) return 0;
  if (stem_wgrad_supported(d) || wino_wgrad_supported(d)) return 0;
  const Trim tr = trim_taps(d);
  const WgradPlan pl = wgrad_plan(&tr.d);
  return pl.vec && pl.NB == 2 && pl.KC == 2;
}

struct GroupLayer {
  Trim tr;
  WgradPlan pl;       // kt_tiles / n_tiles of the trimmed layer; nsplit / cps chosen by the group planner
  long long chunks;
  bool slab;          // through a slab + the grouped reduce (K-split, or dead taps whose dw columns must be zeroed)
  size_t slab_off;    // floats into ws
};

// One common item size for all layers of the group: C pixel chunks (of 32 rows) per item.  Estimated time in chunk units
// (a chunk of a 128 x 128 tile = 64 MFMAs per wave, ~2 us): rounds of items over the chip's 512 workgroup slots, each
// item C chunks + ~1.5 for its set-up and its stores, + the slab traffic of the split layers at ~4 TB/s.
static int plan_group(int n, const avid_wgrad_item* items, GroupLayer* L, int* item0, size_t* ws_floats, int* order) {
  const int slots = 2 * device_cus();
  long long maxc = 1;
  for (int l = 0; l < n; ++l) {
    L[l].tr = trim_taps(&items[l].d);
    L[l].pl = wgrad_plan(&L[l].tr.d);
    const avid_conv_desc* d = &L[l].tr.d;
    L[l].chunks = ceil_div((long long)d->B * d->To * d->Ho * d->Wo, 32);
    if (L[l].chunks > maxc) maxc = L[l].chunks;
  }
  auto splits_for = [&](int l, long long C) {
    const avid_conv_desc* d = &L[l].tr.d;
    long long ns = ceil_div(L[l].chunks, C);
    // 32-bit byte offsets inside one split (as wgrad_plan): (batch items touched) * bytes per input item < 2 GiB
    const long long pix_out = (long long)d->To * d->Ho * d->Wo, per_b_in = (long long)d->Ti * d->Hi * d->Wi * d->Cin * 4;
    while (ns < L[l].chunks && (ceil_div(L[l].chunks, ns) * 32 / pix_out + 2) * per_b_in >= (1ll << 31)) ns *= 2;
    if (ns > L[l].chunks) ns = L[l].chunks;
    if (ns > 512) ns = 512;
    return ns;
  };
  double best = 1e300;
  long long bestC = maxc;
  for (long long C = maxc; C >= 2; C = C > 48 ? C - C / 12 : C - 1) {
    double nitems = 0, slab_bytes = 0, work = 0, biggest = 0;
    for (int l = 0; l < n; ++l) {
      const long long ns = splits_for(l, C), tiles = (long long)L[l].pl.kt_tiles * L[l].pl.n_tiles;
      const double cost = (double)ceil_div(L[l].chunks, ns) + 1.5;
      nitems += (double)tiles * ns;
      work += (double)tiles * ns * cost;
      if (cost > biggest) biggest = cost;
      if (ns > 1) slab_bytes += 2.0 * 4.0 * ns * items[l].d.Cout * L[l].tr.d.kt * items[l].d.kh * items[l].d.kw * items[l].d.Cin;
    }
    // items are dealt round-robin in order of decreasing cost: the last workgroup finishes about one average item after
    // the mean load, and never before the biggest item
    double t = work / slots + 0.5 * work / nitems;
    if (t < biggest) t = biggest;
    t += slab_bytes / 4e12 / 2e-6;
    if (t < best) { best = t; bestC = C; }
  }
  size_t off = 0;
  for (int l = 0; l < n; ++l) {
    const long long ns = splits_for(l, bestC);
    L[l].pl.cps = (int)ceil_div(L[l].chunks, ns);
    L[l].pl.nsplit = (int)ceil_div(L[l].chunks, L[l].pl.cps);
    L[l].slab = L[l].pl.nsplit > 1 || L[l].tr.on;
    L[l].slab_off = off;
    if (L[l].slab)
      off += (size_t)L[l].pl.nsplit * items[l].d.Cout * L[l].tr.d.kt * items[l].d.kh * items[l].d.kw * items[l].d.Cin;
    order[l] = l;
  }
  for (int a = 1; a < n; ++a)                      // table order: decreasing item cost (insertion sort, stable)
    for (int b = a; b > 0 && L[order[b]].pl.cps > L[order[b - 1]].pl.cps; --b) {
      const int tmp = order[b]; order[b] = order[b - 1]; order[b - 1] = tmp;
    }
  int it = 0;
  for (int k = 0; k < n; ++k) {
    const int l = order[k];
    item0[k] = it;
    it += L[l].pl.kt_tiles * L[l].pl.n_tiles * L[l].pl.nsplit;
  }
  item0[n] = it;
  *ws_floats = off;
  return AVID_OK;
}

extern "C" size_t avid_conv_wgrad_group_workspace_bytes(int n, const avid_wgrad_item* items) {
  if (n <= 0 || n > WG_GROUP_MAX || !items) return 0;
  GroupLayer L[WG_GROUP_MAX];
  int item0[WG_GROUP_MAX + 1], order[WG_GROUP_MAX];
  size_t fl = 0;
  for (int l = 0; l < n; ++l)
    if (!avid_conv_wgrad_groupable(&items[l].d)) return 0;
  plan_group(n, items, L, item0, &fl, order);
  return sizeof(float) * fl + 256;
}

extern "C" int avid_conv_wgrad_group(int n, const avid_wgrad_item* items, void* ws, size_t ws_bytes,
                                     avid_stream_t stream) {
  AVID_REQUIRE(n > 0 && n <= WG_GROUP_MAX && items, AVID_E_BADARG, "conv_wgrad_group: 1..%d layers", WG_GROUP_MAX);
  for (int l = 0; l < n; ++l) {
    AVID_REQUIRE(items[l].x && items[l].dy && items[l].dw, AVID_E_BADARG, "conv_wgrad_group: null pointer (layer %d)", l);
    AVID_REQUIRE(avid_conv_wgrad_groupable(&items[l].d), AVID_E_UNSUPPORTED,
                 "conv_wgrad_group: layer %d does not run on wgrad_tab_kernel<2,2> (avid_conv_wgrad_groupable)", l);
  }
  GroupLayer L[WG_GROUP_MAX];
  WgradGroupArgs g;
  size_t fl = 0;
  int order[WG_GROUP_MAX];
  plan_group(n, items, L, g.item0, &fl, order);
  AVID_REQUIRE(fl == 0 || (ws && ws_bytes >= sizeof(float) * fl), AVID_E_BADARG, "conv_wgrad_group: workspace too small");
  g.n = n;
  {
    // runs of 32 consecutive items per XCD — measured (3 launches per step): 8: 494 MB per launch 1.076 ms; 16: 428 MB
    // 1.070 ms; 32: 389 MB 1.080 ms; 64: 377 MB 1.254 ms (278 MB algorithmic)
    g.run = 32u;
  }
  WgradGroupReduce r;
  memset(&r, 0, sizeof(r));
  double flops = 0, bytes = 0, red_bytes = 0;
  long long max_n4 = 0;
  for (int k = 0; k < n; ++k) {
    const int l = order[k];                            // table slot k holds layer l
    const avid_conv_desc* d = &L[l].tr.d;
    const int tap = d->kh * d->kw * d->Cin, Kp = d->kt * tap, K = L[l].tr.kt_full * tap, koff = L[l].tr.dt0 * tap;
    float* slab = static_cast<float*>(ws) + L[l].slab_off;
    fill_wgrad_args(g.layer[k], d, items[l].x, items[l].dy, L[l].slab ? slab : items[l].dw, L[l].pl);
    if (L[l].slab) {
      const int y = r.count++;
      r.part[y] = slab; r.dw[y] = items[l].dw;
      r.n[y] = (long long)d->Cout * K;                 // dw elements (the reduce walks dw: dead-tap columns get zeros)
      r.nsplit[y] = L[l].pl.nsplit;
      r.Kp[y] = Kp; r.K[y] = K; r.koff[y] = koff;
      if (r.n[y] / 4 > max_n4) max_n4 = r.n[y] / 4;
      red_bytes += 4.0 * ((double)d->Cout * Kp * L[l].pl.nsplit + (double)d->Cout * K);
    }
    const double M = (double)d->B * d->To * d->Ho * d->Wo;
    flops += 2.0 * M * d->Cout * Kp;
    bytes += 4.0 * ((double)d->B * d->Ti * d->Hi * d->Wi * d->Cin + M * d->Cout + (double)d->Cout * Kp);
  }
  hipStream_t s = (hipStream_t)stream;
  const bool pre = wgrad_pre();
  const size_t lds = pre ? WGP_LDS : sizeof(float) * 2 * 32 * (64 * 2 + 4 + 2 * WG_LD) + sizeof(uint2) * WG_TABC * 32;
  static bool set = false;
  if (!set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_group_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)WGP_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_group_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_group_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    set = true;
  }
  int grid = g.item0[n];
  const int grid_cap = 2 * device_cus();      // two workgroups per CU
  if (grid > grid_cap) grid = grid_cap;
  {
    ScopedTimer t(s, "wgrad_group_kernel", flops, bytes);
    if (pre) hipLaunchKernelGGL((wgrad_group_kernel<true, true>), dim3((unsigned)grid), dim3(256), lds, s, g);
    else if (wgrad_split()) hipLaunchKernelGGL(wgrad_group_kernel<true>, dim3((unsigned)grid), dim3(256), lds, s, g);
    else hipLaunchKernelGGL(wgrad_group_kernel<false>, dim3((unsigned)grid), dim3(256), lds, s, g);
  }
  int rc = check_launch("wgrad_group");
  if (rc || r.count == 0) return rc;
  long long gx = ceil_div(max_n4, 32);
  if (gx > 512) gx = 512;
  ScopedTimer t(s, "wgrad_group_reduce_kernel", 0.0, red_bytes);
  hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3((unsigned)gx, (unsigned)r.count), dim3(256), 0, s, r);
  return check_launch("wgrad_group_reduce");
}

extern "C" int avid_conv_uses_wino(const avid_conv_desc* d, int which) {
  if (!d || validate(d)) return 0;
  if (which == 2) return wino_wgrad_supported(d) ? 1 : 0;
  return (which == 0 || which == 1) && wino_supported(d, which) ? wino_variant(d, which) : 0;
}

extern "C" int avid_conv_kernel_name(const avid_conv_desc* d, int which, char* buf, int len) {
  int rc = validate(d);
  if (rc) return rc;
  AVID_REQUIRE(buf && len > 0 && which >= 0 && which <= 2, AVID_E_BADARG, "conv_kernel_name: bad argument");
  const bool vec = (d->Cin % 32 == 0) && !d->x_channel_first;
  const Trim tr = trim_taps(d);
  const int ktl = tr.d.kt;           // live temporal taps
  auto pk_name = [&](long long M, int Cd, int nk, int mode) {
    const PkPlan pk = plan_pk(M, Cd, nk, mode);
    static const char* kPk[] = {"2,2,2,2", "4,1,1,2", "4,2,2,2", "4,2,2,1", "2,1,2,2"};
    // (the 128 x 128 tile of a layer that is handed its pre-split weights runs as four waves of 32 x 128 with the epilogues
    //  convolution layers use — dispatch_igemm's `rows`; a caller that passes no table or a bias / ReLU epilogue gets <2,2,2,2>)
    const char* tile = pk.tile == 0 && bs_rows() && conv_takes_split(d, mode) ? "4,1,1,4" : kPk[pk.tile];
    const int cs = mode == 0 ? d->Cin : d->Cout;
    const int ws = plan_pk_order(pk, M, Cd, nk, cs / BK, conv_takes_split(d, mode));
    snprintf(buf, len, "igemm_pk_kernel<%s,%d> full=%d tail_units=%d f=%d%s", tile, mode, pk.full, pk.tail_units, pk.f,
             ws ? " weight-stationary" : "");
  };
  // conv2x's temporal layers (given their pre-split weights): tconv64_kernel — the descriptor-level form of tconv_takes
  const long long tc_tiles = ((long long)d->B * d->Hi * d->Wi + TC_P - 1) / TC_P;
  const bool tc = tconv_layer(d, which);
  if (tc) {
    snprintf(buf, len, "tconv64_kernel<%d> grid=%d", which, balanced_grid(tc_tiles));
    return AVID_OK;
  }
  if (which == 0) {
    const long long M = (long long)d->B * d->To * d->Ho * d->Wo;
    if (!vec) {
      if (stem_fwd_supported(d))
        snprintf(buf, len, stem_fwd_is_split(d) ? (stem_fwd_is_presplit(d) ? "stem_fwd3p_kernel<%d,%d>" : "stem_fwd3_kernel<%d,%d>")
                                                : "stem_fwd_kernel<%d,%d>", d->Cin, d->kt);
      else
        snprintf(buf, len, "igemm_gather_kernel<%s>", ((M + 127) / 128) * (d->Cout / 64) >= 256 ? "4,1,1,2" : "2,2,1,1");
    } else if (wino_supported(d, 0)) {
      snprintf(buf, len, "wino_kernel<0> grid=%d", wino_grid(d, 0));
    } else {
      pk_name(M, d->Cout, ktl * d->kh * d->kw * (d->Cin / BK), 0);
    }
  } else if (which == 1) {
    const long long M = (long long)d->B * d->Ti * d->Hi * d->Wi;
    const bool strided = d->st > 1 || d->sh > 1 || d->sw > 1;
    if (wino_supported(d, 1)) {
      snprintf(buf, len, "wino_kernel<1> grid=%d", wino_grid(d, 1));
    } else if (strided && dgrad_compactable(d)) {
      snprintf(buf, len, "igemm_pk_kernel over the sub-sampled grid + dx_scatter_strided_kernel");
    } else if (strided) {
      snprintf(buf, len, "igemm_pk_kernel<%s,1>s2 (stride-parity classes)", d->Cin % 128 == 0 && s2_wide() ? "2,2,2,2" : "4,1,1,2");
    } else {
      pk_name(M, d->Cin, ktl * d->kh * d->kw * (d->Cout / BK), 1);
    }
  } else {
    WgradPlan pl = wgrad_plan(&tr.d);
    if (twgrad_takes(d)) {
      snprintf(buf, len, "twgrad64_kernel grid=%d", twgrad_grid(d));
      return AVID_OK;
    }
    if (wino_wgrad_supported(d))
      snprintf(buf, len, "wino_wgrad_kernel");
    else if (!pl.vec)
      snprintf(buf, len, stem_wgrad_supported(d) ? "stem_wgrad_kernel splits=%d" : "wgrad_gather_kernel splits=%d", pl.nsplit);
    else
      snprintf(buf, len, "wgrad_tab_kernel<%d,%d> splits=%d", pl.NB, pl.KC, pl.nsplit);
  }
  return AVID_OK;
}
