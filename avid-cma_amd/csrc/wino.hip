// Winograd F(2x2, 3x3) for the (1,3,3) stride-1 convolutions of the residual blocks (models/network_blocks.py:35,40:
// spt_conv1 / spt_conv2, and Basic2DBlock's 3x3 layers) — forward and input gradient — as ONE fused kernel on the fp32
// MFMA: input transform -> 16 [32 tiles x Cr] x [Cr x 64] products -> output transform; only the source tensor, the
// transformed weights U and the destination touch memory.  2.25x fewer multiply-adds than the direct form
// (csrc/conv.hip), which is what counts once the matrix pipe's clock is the limit.
//
//   source   [F][H][W][Cr]   (F = B*T frames; forward: x, Cr = Cin; input gradient: dy, Cr = Cout)
//   U        16 x Cn x Cr    = G g G^T per (output channel n, reduction channel k), stored in the ORDER THE KERNEL'S
//                              LANES READ IT (common.h: wino_weight_elements — one contiguous KB per wave load
//                              instruction; as U[xi][n][k] the texture addresser was as busy as the matrix pipe:
//                              DESIGN.md 3.1e); the input gradient uses the taps flipped and the channel roles
//                              swapped (a stride-1 pad-1 3x3 correlation again)
//   dest     [F][H][W][Cn]
//   workgroup = 4 waves: 32 tiles (2x2 outputs each) x 64 output channels; wave w owns the transform points
//   xi = 4w .. 4w+3 (row w of the 4x4), i.e. 8 accumulator blocks of 32 x 32.
//   LDS: V[16][32 tiles][32 + 4]: one 32-channel chunk of the transformed input at a time (73.7 KB, two workgroups
//   per CU).  One MFMA operand: V[xi][tile = lane & 31][16 * (lane >> 5) + s]; the other straight from U (L2-resident):
//   element (xi, n = lane & 31, k = 16 * (lane >> 5) + s) — both are 16 consecutive k per lane, the MFMA's k index runs
//   (s, lane >> 5).  (wino2_kernel, below: the same computation as one instruction stream per SIMD, for layers with
//   many tiles.)
//   Output transform: columns in registers, rows across the four waves through LDS (V is dead by then); wave w then
//   stores output pixel (w >> 1, w & 1) of every tile: 128 contiguous bytes per 32 lanes.
//   Epilogues: BatchNorm partial sums of the output (forward), addend and the BatchNorm-backward sums of the
//   gradient being written (input gradient) — the same contracts as igemm_pk_kernel's.
// Rounding differs from the direct form (transforms add before the products): measured 3.5e-7 of the output scale
// against fp64 on conv2x, inside every tolerance the direct kernels are tested to.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace avid {

constexpr int W_TB = 32, W_CK = 32, W_VLD = W_CK + 4;
constexpr int W_LDS_FLOATS = 16 * W_TB * W_VLD;
#ifndef AVID_W_LATE
#define AVID_W_LATE -1     // (>= 0: issue the next chunk's patch loads after product slot N instead of before the products: no difference, 174-175 us on conv2x either way)
#endif
constexpr int W_LATE = AVID_W_LATE;
constexpr int W_TAB_INTS = 2 * W_TB * 4;          // double-buffered tile table: (frame or -1, ti, tj, -) per tile of a unit

struct WinoArgs {
  const float* __restrict__ src;
  const float* __restrict__ U;
  float* __restrict__ dst;
  const float* __restrict__ addend;
  float* __restrict__ stats;            // one row [2][Cn] per workgroup, or null
  const float* __restrict__ bnb_x;      // EPI & 4: BatchNorm input at the destination positions + saved coefficients
  const float* __restrict__ bnb_scale;
  const float* __restrict__ bnb_shift;
  const float* __restrict__ bnb_mean;
  const float* __restrict__ bnb_invstd;
  int bnb_relu;
  int F, H, W, TH, TW, Cr, Cn, ncb;
  long long ntiles;
  int units;                            // tile blocks x column blocks, column block fastest
  int xcd_local;                        // XCD-contiguous unit order
  int v2;                               // host only: launch wino2_kernel (units are 64-tile blocks then)
};

// U(xi, n, k) = (G g G^T)[xi / 4][xi % 4],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], written in fragment order frag
// flip = 0: g[a][b] = w[n][a][b][k] (forward);  flip = 1: g[a][b] = w[k][2-a][2-b][n] (input gradient)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cn,
                                                          int Cr, int Cin, int flip, int frag) {
  wino_weight_elements(w, U, Cn, Cr, Cin, flip, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                       (long long)gridDim.x * blockDim.x, frag);
}

#ifdef AVID_WINO_TRACE
__device__ long long g_wino_trace[1024 * 8];
extern "C" int avid_debug_wino_trace(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_trace), sizeof(long long) * 1024 * 8);
}
#define W_STAMP(i) do { if (threadIdx.x == 0) { const long long now_ = wall_clock64(); g_wino_trace[blockIdx.x * 8 + (i)] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define W_STAMP(i) do {} while (0)
#endif
// Packed fp32 adds as inline assembly: every VALU instruction of a wave costs ~2.2 ns of its SIMD's MFMA time
// (tools/mfma_shadow: the fp32 MFMA runs at the vector rate, on the same lanes), a packed add costs the same as a
// scalar one — and the compiler UNPACKS v_pk_add_f32 wherever it sits behind an MFMA (two instructions for one).
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4_t __attribute__((__vector_size__(4 * sizeof(unsigned))));
__device__ __forceinline__ floatx2 pk_add(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ floatx2 pk_sub(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ floatx4 pk4_add(floatx4 a, floatx4 b) {
  const floatx2 lo = pk_add(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
  const floatx2 hi = pk_add(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ floatx4 pk4_sub(floatx4 a, floatx4 b) {
  const floatx2 lo = pk_sub(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
  const floatx2 hi = pk_sub(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// EPI bits: 1 BatchNorm partial sums of the output, 2 addend, 4 BatchNorm-backward sums (needs bnb_*)
template <int EPI>
__global__ __launch_bounds__(256, 2) void wino_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int H = p.H, W = p.W, TW = p.TW, TPF = p.TH * p.TW, Cr = p.Cr, Cn = p.Cn;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((long long)p.F * H * W * Cr * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, 16 * Cn * Cr * 4, 0x00020000);
  const int nchunks = Cr / W_CK;
  // BatchNorm partial sums (EPI & 5): a lane's accumulators hold ONE tile and 32 channels, so the sums over tiles go
  // through a small LDS staging area per wave (below); this lane then owns channel  16 r + (lane & 15)  of round r =
  // 2 j + (g >> 1) for the tiles 8 (lane >> 4) .. + 7 of every unit
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  // Workgroup (q, cb): column block cb = lid % ncb, tile blocks q, q + G', q + 2 G', ... (G' = grid / ncb).  The logical id
  // lid gives XCD k a contiguous eighth of every round: consecutive tile blocks are neighbouring tiles, whose 4x4 patches
  // overlap by two rows / columns — dealt in hardware order they sit on eight different L2s and every input pixel is
  // fetched from memory up to four times (counter traffic 2.0x the algorithmic bytes).  The blocks of the last, partial
  // round are spread over the XCDs one by one (block j of it -> XCD j % 8), or XCD 0 would do them all.
  const bool xl = p.xcd_local && gridDim.x % (8 * p.ncb) == 0;
  const int lid = xl ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cb = lid % p.ncb, q = lid / p.ncb, Gq = (int)gridDim.x / p.ncb;
  const int nblk = p.units / p.ncb;
  const int fullq = xl ? nblk / Gq * Gq : nblk;      // first tile block of the partial round
  const int qlast = fullq + (q % (Gq / 8)) * 8 + q / (Gq / 8);
  auto blk_after = [&](int blk) {                    // this workgroup's next tile block (>= nblk: none)
    if (blk >= fullq) return nblk;
    const int nx = blk + Gq;
    return nx < fullq ? nx : qlast;
  };
  const int blk0 = q < fullq ? q : (xl ? qlast : q);
#ifdef AVID_WINO_TRACE
  long long tprev_ = wall_clock64();
  if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) g_wino_trace[blockIdx.x * 8 + i] = 0;
#endif
  // tile table of a unit (tile index -> frame, ti, tj; frame = -1 past the end), double-buffered: the two integer
  // divisions per tile are done once, by 32 lanes, instead of by every thread of the transform and 16 times per lane
  // of the epilogue (~1000 VALU instructions per wave and unit, on the matrix pipe's time)
  int4* tabs = reinterpret_cast<int4*>(sm + W_LDS_FLOATS);
  auto fill_tab = [&](int blk, int buf) {
    int tid_t = tid, tpf = TPF, tw = TW;             // (opaque copies: the divisions' reciprocals and the table address are
    asm volatile("" : "+v"(tid_t), "+s"(tpf), "+s"(tw));   //  recomputed per unit instead of being spilled as loop invariants)
    if (tid_t < W_TB) {
      const long long t = (long long)blk * W_TB + tid_t;
      int4 e = {-1, 0, 0, 0};
      if (blk < nblk && t < p.ntiles) {
        const int f = (int)(t / tpf), rem = (int)(t - (long long)f * tpf);
        e.x = f; e.y = rem / tw; e.z = rem - e.y * tw;
      }
      tabs[buf * W_TB + tid_t] = e;
    }
  };
  // this thread's item of the input transform: (4 channels = tid & 7, tile = tid >> 3).  The 4x4 patch of the NEXT
  // chunk travels in registers while the current chunk is multiplied: its 16 loads are issued right before the
  // products and land under them (the first version loaded, transformed and multiplied one chunk after the other:
  // 47 % products / 27 % input transform / 26 % output transform, nothing overlapping — DESIGN.md 3.1c).
  const int c4 = (tid & 7) * 4, ttl = tid >> 3;
  floatx4 raw[4][4];
  auto issue_loads = [&](const int4 e, int ck) {
    const bool t_ok = e.x >= 0;
    const int y1 = 2 * e.y, x1 = 2 * e.z;
    // pixel (2 ti, 2 tj) = patch element (1, 1) is always inside the frame; the range check of a buffer load covers
    // the vector offset only, so the columns to its right ride in the scalar offset and column 0 / row 0 are masked
    const unsigned o11 = (unsigned)(((e.x * H + y1) * W + x1) * Cr + ck * W_CK + c4) * 4u;
    const unsigned px_b = (unsigned)Cr * 4u, row_b = (unsigned)W * px_b;
    const bool oky[4] = {t_ok && e.y > 0, t_ok, t_ok && y1 + 1 < H, t_ok && y1 + 2 < H};
    const bool okx[4] = {e.z > 0, true, x1 + 1 < W, x1 + 2 < W};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const unsigned rowoff = o11 + (unsigned)(a - 1) * row_b;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        raw[a][b] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                    rsX, (oky[a] && okx[b]) ? (b == 0 ? rowoff - px_b : rowoff) : 0x80000000u,
                                                    b == 0 ? 0 : (b - 1) * (int)px_b, 0));
    }
  };
  fill_tab(blk0, 0);
  __syncthreads();
  int4 e_cur = tabs[ttl];
  issue_loads(e_cur, 0);
  int uidx = 0;
  for (int blk = blk0; blk < nblk; blk = blk_after(blk)) {
    W_STAMP(0);
    floatx16 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
    const int tbuf = uidx & 1;
    const int4* tab = tabs + tbuf * W_TB;
    ++uidx;
    for (int ck = 0; ck < nchunks; ++ck) {
      __syncthreads();                         // the previous chunk's / unit's LDS reads are done
      if (ck == 0) fill_tab(blk_after(blk), tbuf ^ 1);      // (visible behind the next barrier; nobody reads that buffer now)
      {
        // V = B^T d B,  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: column by column, then row by row
        floatx4 w_[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          w_[0][b] = pk4_sub(raw[0][b], raw[2][b]);
          w_[1][b] = pk4_add(raw[1][b], raw[2][b]);
          w_[2][b] = pk4_sub(raw[2][b], raw[1][b]);
          w_[3][b] = pk4_sub(raw[1][b], raw[3][b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float* dst = sm + ((a * 4) * W_TB + ttl) * W_VLD + c4;
          *reinterpret_cast<floatx4*>(dst + 0 * W_TB * W_VLD) = pk4_sub(w_[a][0], w_[a][2]);
          *reinterpret_cast<floatx4*>(dst + 1 * W_TB * W_VLD) = pk4_add(w_[a][1], w_[a][2]);
          *reinterpret_cast<floatx4*>(dst + 2 * W_TB * W_VLD) = pk4_sub(w_[a][2], w_[a][1]);
          *reinterpret_cast<floatx4*>(dst + 3 * W_TB * W_VLD) = pk4_sub(w_[a][1], w_[a][3]);
        }
      }
      __syncthreads();
      W_STAMP(1);
      // the next chunk's patch (of this unit, or chunk 0 of this workgroup's next unit): in flight under the products
      auto issue_next = [&]() {
        if (ck + 1 < nchunks) {
          issue_loads(e_cur, ck + 1);
        } else {
          e_cur = tabs[(tbuf ^ 1) * W_TB + ttl];
          issue_loads(e_cur, 0);
        }
      };
      if (W_LATE < 0) issue_next();
      // ---- the wave's four products, computed TRANSPOSED (D[n][tile] = sum_k U[n][k] V[tile][k]: U is the MFMA's A
      // operand): a lane then holds ONE tile and, in every group of four accumulator registers, four consecutive
      // output channels — the output transform stores 16 contiguous bytes per lane and needs one destination offset
      // per lane instead of sixteen.  U (from L2) sits in three rotating slots of (transform point, 32-column
      // half): the loads run two slots ahead of the MFMAs that consume them
      floatx4 bv[3][4];
      // U in fragment order (wino_weight_elements, frag = 2): a wave's load instruction reads ONE contiguous KB.  As
      // U[xi][n][k] it read 2 x 16 bytes of 32 rows, which the texture addresser serves at one lane per cycle — 107 ns
      // per instruction beside the MFMAs against 13-27 ns (tools/mfma_shadow): 256 such loads per CU and chunk kept the
      // addresser as busy as the matrix pipe.
      auto load_b = [&](int s_) {
        const int xi = wave * 4 + (s_ >> 1), j = s_ & 1;
        const unsigned vo = (unsigned)(lane * 16 + ((((xi * (Cn >> 5) + cb * 2 + j) * nchunks + ck) * 4) * 64) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bv[s_ % 3][q] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsU, vo + q * 1024u, 0, 0));
      };
      load_b(0);
      load_b(1);
      // (s_setprio 3 around the products — so that the co-resident workgroup's transform only fills gaps — made the
      // kernel 22 % SLOWER: 177 -> 217 us on conv2x; the transforming workgroup then never gets to its own products)
      floatx4 av[4];
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const int c = s_ >> 1, j = s_ & 1;
        if (s_ + 2 < 8) load_b(s_ + 2);
        if (s_ == W_LATE) issue_next();
        if (j == 0) {
          const float* Ap = sm + ((wave * 4 + c) * W_TB + l31) * W_VLD + 16 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const floatx4*>(Ap + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[s_ % 3][q][e], av[q][e], acc[c][j], 0, 0, 0);
      }
      W_STAMP(2);
    }
    // ---- output transform.  Columns in registers: T[r][0] = M0 + M1 + M2, T[r][1] = M1 - M2 - M3 (r = wave); rows
    // across the four waves through LDS (V is dead by then; lane-private 16-byte slots, conflict-free).  Accumulator
    // layout (transposed products): lane = (tile l31, h), register r <-> channel (r & 3) + 8 (r >> 2) + 4 h of the
    // 32-column half j.  (With the MFMA's usual orientation — one channel and 16 tiles per lane — a unit took 32
    // four-byte store instructions per lane, and their issue made the output phase as long as the products.)
    W_STAMP(3);
    floatx16 T0[2], T1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      T0[j] = acc[0][j] + acc[1][j] + acc[2][j];
      T1[j] = acc[1][j] - acc[2][j] - acc[3][j];
    }
    // the output phase's lane constants, recomputed per unit from a thread id the compiler cannot see through: as loop invariants
    // they were hoisted in front of the unit loop and spilled there (wino_kernel<1> / <4> / <6>: 14 / 22 / 24 registers stored once
    // and re-read per unit, tools/kernel_resources.py)
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int lane_o = tid_o & 63, wave_o = __builtin_amdgcn_readfirstlane(tid_o >> 6), l31o = lane_o & 31, h_o = lane_o >> 5;
    __syncthreads();                           // V is dead: the LDS becomes the exchange buffer T[r][q][j][reg / 4][lane_o]
    float* ex = sm;                             // ex[row][q][j][r / 4][lane_o][r % 4]: 16-byte accesses, conflict-free
    auto ex_at = [&](int row, int q_, int j, int r4) { return ex + (((((row * 2 + q_) * 2 + j) * 4 + r4) * 64 + lane_o) << 2); };
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        *reinterpret_cast<floatx4*>(ex_at(wave_o, 0, j, r4)) = floatx4{T0[j][4 * r4], T0[j][4 * r4 + 1], T0[j][4 * r4 + 2], T0[j][4 * r4 + 3]};
        *reinterpret_cast<floatx4*>(ex_at(wave_o, 1, j, r4)) = floatx4{T1[j][4 * r4], T1[j][4 * r4 + 1], T1[j][4 * r4 + 2], T1[j][4 * r4 + 3]};
      }
    __syncthreads();
    // rows across waves: this wave_o writes output pixel (po, qo) of every tile:
    //   Y[0][q] = T[0][q] + T[1][q] + T[2][q],  Y[1][q] = T[1][q] - T[2][q] - T[3][q]
    const int po = wave_o >> 1, qo = wave_o & 1;
    int ooff;                                   // destination offset (in floats, < 2^29) of the lane_o's tile, or -1
    {
      const int4 e = tab[l31o];
      const int yy = 2 * e.y + po, xx = 2 * e.z + qo;
      ooff = (e.x >= 0 && yy < H && xx < W) ? ((e.x * H + yy) * W + xx) * Cn + cb * 64 + 4 * h_o : -1;
    }
    const bool ok = ooff >= 0;
    const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};
    // statistics staging: st[wave_o][tile 32][16 channels] (2 KB per wave_o, behind the exchange buffer), 16-byte slots
    // XOR-swizzled by the tile so that the eight lanes of a 16-byte store group and the 32 lanes of a 4-byte load group
    // hit distinct banks.  One term at a time; LDS operations of a wave_o execute in order, so no barrier is involved.
    float* st = sm + 4 * 2 * 2 * 4 * 64 * 4 + wave_o * (W_TB * 16);
    const int st_row = l31o * 16, st_sw = ((l31o >> 1) & 3) * 4;
    const int st_ch = lane_o & 15, st_tg = lane_o >> 4;
    auto col_sums = [&]() {                      // sum over this lane_o's 8 tiles of channel st_ch of the staged round
      float s_ = 0.f;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int tile = 8 * st_tg + ((m + st_tg) & 7);
        s_ += st[tile * 16 + (st_ch ^ (((tile >> 1) & 3) * 4))];
      }
      return s_;
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      floatx4 ad[4], xb[4];
      if (EPI & 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) ad[g] = ok ? *reinterpret_cast<const floatx4*>(p.addend + ooff + j * 32 + 8 * g) : z4;
      }
      if (EPI & 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) xb[g] = ok ? *reinterpret_cast<const floatx4*>(p.bnb_x + ooff + j * 32 + 8 * g) : z4;
      }
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        floatx4 t0[2], t1[2];
#pragma unroll
        for (int gl = 0; gl < 2; ++gl) {
          const int g = 2 * gp + gl;
          const floatx4 a = *reinterpret_cast<const floatx4*>(ex_at(po, qo, j, g));        // rows po, po + 1, po + 2
          const floatx4 b = *reinterpret_cast<const floatx4*>(ex_at(po + 1, qo, j, g));
          const floatx4 c = *reinterpret_cast<const floatx4*>(ex_at(po + 2, qo, j, g));
          floatx4 v = po == 0 ? a + b + c : a - b - c;
          if (EPI & 2) v += ad[g];
          if (ok) *reinterpret_cast<floatx4*>(p.dst + ooff + j * 32 + 8 * g) = v;
          if (EPI & 5) {
            t0[gl] = ok ? v : z4;                 // the two terms whose column sums are wanted
            if (EPI & 1) {
              t1[gl] = t0[gl] * t0[gl];
            } else {
              const int col = cb * 64 + j * 32 + 8 * g + 4 * h_o;
              const floatx4 bsc = *reinterpret_cast<const floatx4*>(p.bnb_scale + col), bsh = *reinterpret_cast<const floatx4*>(p.bnb_shift + col);
              const floatx4 bmu = *reinterpret_cast<const floatx4*>(p.bnb_mean + col), bis = *reinterpret_cast<const floatx4*>(p.bnb_invstd + col);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                t0[gl][i] = (!p.bnb_relu || fmaf(xb[g][i], bsc[i], bsh[i]) > 0.f) ? t0[gl][i] : 0.f;
                t1[gl][i] = t0[gl][i] * ((xb[g][i] - bmu[i]) * bis[i]);
              }
            }
          }
        }
        if (EPI & 5) {            // round r = 2 j + gp: channels 16 r .. 16 r + 15 of this column block
          const int r = 2 * j + gp;
          *reinterpret_cast<floatx4*>(st + st_row + ((4 * h_o) ^ st_sw)) = t0[0];
          *reinterpret_cast<floatx4*>(st + st_row + ((8 + 4 * h_o) ^ st_sw)) = t0[1];
          cs[r] += col_sums();
          *reinterpret_cast<floatx4*>(st + st_row + ((4 * h_o) ^ st_sw)) = t1[0];
          *reinterpret_cast<floatx4*>(st + st_row + ((8 + 4 * h_o) ^ st_sw)) = t1[1];
          cq[r] += col_sums();
        }
      }
    }
  }
  W_STAMP(4);
  if ((EPI & 5) && p.stats) {   // one partial row [2][Cn] per workgroup: the four tile groups of a wave, then the four waves
    __syncthreads();
    float* red = sm;            // [2][4][64]
    int lane_r = lane;
    asm volatile("" : "+v"(lane_r));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = cs[r], b = cq[r];
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      if (lane_r < 16) {
        red[wave * 64 + 16 * r + lane_r] = a;
        red[256 + wave * 64 + 16 * r + lane_r] = b;
      }
    }
    __syncthreads();
    float* row = p.stats + (long long)blockIdx.x * 2 * Cn;
    for (int c = tid; c < Cn; c += 256) {
      float a = 0.f, b = 0.f;
      if (c / 64 == cb) {
        const int cc = c % 64;
        a = (red[cc] + red[64 + cc]) + (red[128 + cc] + red[192 + cc]);
        b = (red[256 + cc] + red[320 + cc]) + (red[384 + cc] + red[448 + cc]);
      }
      row[c] = a;
      row[Cn + c] = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// wino2_kernel: the same computation as ONE instruction stream per SIMD (round 3).  In wino_kernel a workgroup's input
// transform, products and output transform are phases separated by barriers, and the co-resident workgroup is meant to
// fill the matrix pipe meanwhile; the phase trace shows what that costs: a transforming wave that competes with another
// wave's back-to-back MFMAs gets about one vector instruction issued per MFMA (64 cycles), so ~100 instructions of
// transform take 3 us beside 3.4 us of products, and the pipe idles whenever both workgroups are outside their products.
// Inside ONE wave a vector instruction between two MFMAs costs ~5 cycles (tools/mfma_shadow).  So here:
//   * one workgroup of 4 waves per CU (one wave per SIMD, up to 512 registers), unit = 64 tiles x 64 output channels;
//     wave (th, nh) owns ALL 16 transform points of the 32 tiles x 32 channels block (tile half th, channel half nh):
//     16 accumulators of 32 x 32, and the output transform is lane-local — no exchange through LDS, no barrier;
//   * the reduction runs in chunks of 16 channels: V[16][64 tiles][16] = 64 KB per stage, two stages; while chunk k is
//     multiplied out of one stage, every thread transforms its 4x4 patch x 4 channels of chunk k + 1 (registers, loaded
//     during chunk k - 1) into the other stage and then issues the loads of chunk k + 2 — all of it between the MFMAs
//     of the products, ONE barrier per chunk;
//   * U comes straight from L2 into four rotating register slots, three transform points ahead.  Vector memory loads
//     return in order, so a wait for U also waits for every patch load issued before it: the patch loads go out two per
//     transform point, U first, and are not needed before the next chunk;
//   * a unit's first MFMA per transform point takes a constant-zero C operand (no 256-instruction clear);
//   * BatchNorm sums: the four output pixels of a tile are summed in the lane, then one pass per 16 channels through
//     the per-wave LDS staging of wino_kernel.
// Units are 64-tile blocks dealt as in wino_kernel; it is used where a layer has enough of them to fill the CUs.
#ifndef AVID_W2_DBG
#define AVID_W2_DBG 0      // development: 1 skip transform, 2 skip patch loads, 4 skip U loads, 8 skip output phase, 16 skip V reads
#endif
constexpr int W2_DBG = AVID_W2_DBG;
#ifndef AVID_W2_T0
#define AVID_W2_T0 8
#endif
constexpr int W2_T0 = AVID_W2_T0;
#ifndef AVID_W2_STAGGER
#define AVID_W2_STAGGER 1
#endif
constexpr int W2_TB = 64, W2_CK = 16;
typedef __bf16 w2_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w2_uintx4 __attribute__((ext_vector_type(4)));
// 8 fp32 values -> three bf16x8 fragments (hi, mid, lo: common.h split2_bf16)
__device__ __forceinline__ void w2_split8(const floatx4& v0, const floatx4& v1, w2_bf16x8& fh, w2_bf16x8& fm, w2_bf16x8& fl) {
  const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2_bf16(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
  fh = __builtin_bit_cast(w2_bf16x8, w2_uintx4{h[0], h[1], h[2], h[3]});
  fm = __builtin_bit_cast(w2_bf16x8, w2_uintx4{m[0], m[1], m[2], m[3]});
  fl = __builtin_bit_cast(w2_bf16x8, w2_uintx4{l[0], l[1], l[2], l[3]});
}
typedef unsigned uintx2_t __attribute__((ext_vector_type(2)));
// (the split of wino2p_kernel's transform threads; -DAVID_W2P_DOT: the v_dot2c form)
__device__ __forceinline__ void w2p_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
#ifdef AVID_W2P_DOT
  split2_bf16_dot(x0, x1, h, m, l);
#else
  split2_bf16(x0, x1, h, m, l);
#endif
}
constexpr int W2_STAGE = 16 * W2_TB * W2_CK;                 // floats per stage
constexpr int W2_TAB_OFF = 2 * W2_STAGE;                     // three tile tables of 64 x int4
constexpr int W2_LDS_FLOATS = W2_TAB_OFF + 3 * W2_TB * 4;

template <int EPI>
__global__ __launch_bounds__(256) void wino2_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int th = wave & 1, nh = wave >> 1;
  const int H = p.H, W = p.W, TW = p.TW, TPF = p.TH * p.TW, Cr = p.Cr, Cn = p.Cn;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((long long)p.F * H * W * Cr * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, 16 * Cn * Cr * (W2_SPLIT ? 6 : 4), 0x00020000);
  const int dbytes = (int)((long long)p.F * H * W * Cn * 4);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 2) ? p.addend : p.src), 0, (EPI & 2) ? dbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 4) ? p.bnb_x : p.src), 0, (EPI & 4) ? dbytes : 0, 0x00020000);
  const int nchunks = Cr / W2_CK;
  // unit order: see wino_kernel
  const bool xl = p.xcd_local && gridDim.x % (8 * p.ncb) == 0;
  const int lid = xl ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cb = lid % p.ncb, q = lid / p.ncb, Gq = (int)gridDim.x / p.ncb;
  const int nblk = p.units / p.ncb;
  const int fullq = xl ? nblk / Gq * Gq : nblk;
  const int qlast = fullq + (q % (Gq / 8)) * 8 + q / (Gq / 8);
  auto blk_after = [&](int blk) {
    if (blk >= fullq) return nblk;
    const int nx = blk + Gq;
    return nx < fullq ? nx : qlast;
  };
  const int blk0 = q < fullq ? q : (xl ? qlast : q);
  int4* tabs = reinterpret_cast<int4*>(sm + W2_TAB_OFF);
  auto fill_tab = [&](int blk, int buf) {
    if (tid < W2_TB) {
      const long long t = (long long)blk * W2_TB + tid;
      int4 e = {-1, 0, 0, 0};
      if (blk < nblk && t < p.ntiles) {
        const int f = (int)(t / TPF), rem = (int)(t - (long long)f * TPF);
        e.x = f; e.y = rem / TW; e.z = rem - e.y * TW;
      }
      tabs[buf * W2_TB + tid] = e;
    }
  };
  // ---- this thread's item of the input transform: tile tt, channels c4 .. c4 + 3 of the chunk
  const int tt = tid >> 2, c4 = (tid & 3) * 4;
  const unsigned px_b = (unsigned)Cr * 4u, row_b = (unsigned)W * px_b;
  floatx4 raw[4][4];
  // load cursor: the chunk whose patch is loaded next (two chunks ahead of the products)
  unsigned ld_o11 = 0;
  bool ld_oky[4] = {false, false, false, false}, ld_okx[4] = {false, true, false, false};
  int ld_ck = 0, ld_buf = 0;
  auto ld_unit = [&](int buf) {                    // the cursor enters the unit whose table sits in buffer buf
    const int4 e = tabs[buf * W2_TB + tt];
    const bool t_ok = e.x >= 0;
    const int y1 = 2 * e.y, x1 = 2 * e.z;
    // based at patch element (1, 1), which is always inside the frame (see wino_kernel)
    ld_o11 = (unsigned)(((e.x * H + y1) * W + x1) * Cr + c4) * 4u;
    ld_oky[0] = t_ok && e.y > 0; ld_oky[1] = t_ok; ld_oky[2] = t_ok && y1 + 1 < H; ld_oky[3] = t_ok && y1 + 2 < H;
    ld_okx[0] = e.z > 0; ld_okx[1] = true; ld_okx[2] = x1 + 1 < W; ld_okx[3] = x1 + 2 < W;
  };
  auto ld_issue = [&](int a, int b) {
    const unsigned rowoff = ld_o11 + (unsigned)(a - 1) * row_b;
    raw[a][b] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                rsX, (ld_oky[a] && ld_okx[b]) ? (b == 0 ? rowoff - px_b : rowoff) : 0x80000000u,
                                                ld_ck * (W2_CK * 4) + (b == 0 ? 0 : (b - 1) * (int)px_b), 0));
  };
  auto ld_advance = [&]() {
    if (++ld_ck == nchunks) {
      ld_ck = 0;
      ld_buf = ld_buf == 2 ? 0 : ld_buf + 1;
      ld_unit(ld_buf);
    }
  };
  // transform: column pass (raw -> w_), row pass in eight slices of two transform points (w_ -> LDS)
  floatx4 w_[4][4];
  auto col_pass = [&]() {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      w_[0][b] = pk4_sub(raw[0][b], raw[2][b]);
      w_[1][b] = pk4_add(raw[1][b], raw[2][b]);
      w_[2][b] = pk4_sub(raw[2][b], raw[1][b]);
      w_[3][b] = pk4_sub(raw[1][b], raw[3][b]);
    }
  };
  // V[xi][tile][16]: the four 16-byte slots of a row are XOR-swizzled by (tile >> 2) & 3 — 16 lanes of a 16-byte
  // access then cover all 64 banks once, for the stores (4 threads per tile) and for the fragment loads (lane = tile)
  const int wr_off = tt * W2_CK + (((tid & 3) ^ ((tt >> 2) & 3)) << 2);
  auto row_slice = [&](float* Wst, int i) {
    const int a = i >> 1;
    float* dst = Wst + (a * 4) * (W2_TB * W2_CK) + wr_off;
    if ((i & 1) == 0) {
      *reinterpret_cast<floatx4*>(dst + 0 * (W2_TB * W2_CK)) = pk4_sub(w_[a][0], w_[a][2]);
      *reinterpret_cast<floatx4*>(dst + 1 * (W2_TB * W2_CK)) = pk4_add(w_[a][1], w_[a][2]);
    } else {
      *reinterpret_cast<floatx4*>(dst + 2 * (W2_TB * W2_CK)) = pk4_sub(w_[a][2], w_[a][1]);
      *reinterpret_cast<floatx4*>(dst + 3 * (W2_TB * W2_CK)) = pk4_sub(w_[a][1], w_[a][3]);
    }
  };
  // ---- products: operands
  const int ptile = 32 * th + l31;
  const int psw = (l31 >> 2) & 3;
  const int rd_off0 = ptile * W2_CK + (((2 * h) ^ psw) << 2), rd_off1 = ptile * W2_CK + (((2 * h + 1) ^ psw) << 2);
  // U in fragment order (wino_weight_elements, frag = 1; split-bf16: frag = 3, three terms): one contiguous KB per load
  // instruction, UP of them per (transform point, 32 output channels, chunk)
  constexpr int UP = W2_SPLIT ? 3 : 2;
  const unsigned u_voff = (unsigned)(lane * 16);
  const int u_xi = Cn * Cr * (W2_SPLIT ? 6 : 4);
  // (nh is wave-uniform, which the compiler cannot see: without readfirstlane every U load sits in a waterfall loop)
  const int u_ck = UP * 1024;
  const int u_base = __builtin_amdgcn_readfirstlane((cb * 2 + nh) * nchunks * u_ck);
  floatx4 bv[4][UP], av[2][2];
  auto load_u = [&](int slot, int xi, int ck) {
#pragma unroll
    for (int q = 0; q < UP; ++q)
      bv[slot][q] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsU, u_voff + 1024u * q, u_base + xi * u_xi + ck * u_ck, 0));
  };
  // split-bf16: the lane's 8 V values of a transform point (channels 8 h .. 8 h + 7 of its tile) as three bf16x8 fragments
  w2_bf16x8 vs[2][3];
  auto split_v = [&](int slot) { w2_split8(av[slot][0], av[slot][1], vs[slot][0], vs[slot][1], vs[slot][2]); };
  auto read_v = [&](const float* Vst, int slot, int xi) {
    av[slot][0] = *reinterpret_cast<const floatx4*>(Vst + xi * (W2_TB * W2_CK) + rd_off0);
    av[slot][1] = *reinterpret_cast<const floatx4*>(Vst + xi * (W2_TB * W2_CK) + rd_off1);
  };
  floatx16 acc[16];

  // ---- stagger: all workgroups start together and do identical work, so the whole chip would load, multiply and
  // store in lockstep — the 16 stores per lane of a unit's output then hit memory as one 17 MB burst while the matrix
  // pipes wait for it (vector memory operations complete in order: the next operand load is behind the stores).
  // Workgroups that have one unit less than the longest delay their start by up to most of a unit's time.
#if AVID_W2_STAGGER
  {
    int mine = 0;
    for (int b = blk0; b < nblk; b = blk_after(b)) ++mine;
    const int most = (nblk + Gq - 1) / Gq;
    if (mine < most) {
      const int steps = ((lid * 5) & 7) * nchunks / 8 * AVID_W2_STAGGER;
      for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(127);
    }
  }
#endif
  // ---- prologue
  fill_tab(blk0, 0);
  fill_tab(blk_after(blk0), 1);
  __syncthreads();
  ld_unit(0);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) ld_issue(a, b);
  ld_advance();
  col_pass();
#pragma unroll
  for (int i = 0; i < 8; ++i) row_slice(sm, i);
  load_u(0, 0, 0);
  load_u(1, 1, 0);
  load_u(2, 2, 0);

  // Schedule of a chunk (xi = transform point whose 8 MFMAs the step carries):
  //   first chunk of a unit only: the 16 patch loads of the NEXT chunk at xi = 0 .. 3 (nothing is held in registers
  //   across the output transform, which needs them: with the patch live there the allocator spilled it everywhere);
  //   xi = 8: column pass; xi = 8 .. 15: the row pass, two transform points per step, into the other stage;
  //   xi = 10 .. 15 (not in a unit's last chunk): the patch loads of the chunk after the next, rows as they die.
  auto chunk = [&](auto first_tag, int ck, int stage) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int ck_n = ck + 1 < nchunks ? ck + 1 : 0;
    const bool last = !FIRST && ck + 1 == nchunks;
    const float* Vst = sm + stage * W2_STAGE;
    float* Wst = sm + (stage ^ 1) * W2_STAGE;
    read_v(Vst, 0, 0);
    if (W2_SPLIT) {
      read_v(Vst, 1, 1);
      split_v(0);
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      if (!(W2_DBG & 4)) { if (xi + 3 < 16) load_u((xi + 3) & 3, xi + 3, ck); else load_u((xi + 3) & 3, xi + 3 - 16, ck_n); }
      if (W2_SPLIT) {     // V of point xi + 1 is split beside the products of xi, V of xi + 2 is on its way from LDS
        if (xi + 1 < 16) split_v((xi + 1) & 1);
        if (xi + 2 < 16) read_v(Vst, xi & 1, xi + 2);
      } else if (!(W2_DBG & 16) && xi + 1 < 16) {
        read_v(Vst, (xi + 1) & 1, xi + 1);
      }
      if (!(W2_DBG & 2) && FIRST && xi < 4) {
#pragma unroll
        for (int b = 0; b < 4; ++b) ld_issue(xi, b);
        if (xi == 3) ld_advance();
      }
      constexpr int T0 = FIRST ? 8 : W2_T0;                 // step of the column pass
      if (!(W2_DBG & 1) && xi == T0) col_pass();
      if (!(W2_DBG & 1) && xi >= T0 && xi < T0 + 8) row_slice(Wst, xi - T0);
      if (!(W2_DBG & 2) && xi >= T0 + 2 && xi < T0 + 8 && !last) {
        const int j = xi - (T0 + 2);           // 3, 3, 3, 3, 2, 2 loads
        const int n = j < 4 ? 3 : 2, base = j < 4 ? j * 3 : 12 + (j - 4) * 2;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < n) ld_issue((base + k) >> 2, (base + k) & 3);
        if (j == 5) ld_advance();
      }
      if (W2_SPLIT) {
        const w2_bf16x8 uh = __builtin_bit_cast(w2_bf16x8, bv[xi & 3][0]), um = __builtin_bit_cast(w2_bf16x8, bv[xi & 3][1]),
                        ul = __builtin_bit_cast(w2_bf16x8, bv[xi & 3][UP - 1]);
        const w2_bf16x8 vh = vs[xi & 1][0], vm = vs[xi & 1][1], vl = vs[xi & 1][2];
        if (FIRST) {
          const floatx16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vl, z16, 0, 0, 0);
        } else {
          acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vl, acc[xi], 0, 0, 0);
        }
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ul, vh, acc[xi], 0, 0, 0);
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vm, acc[xi], 0, 0, 0);
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vm, acc[xi], 0, 0, 0);
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vh, acc[xi], 0, 0, 0);
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vh, acc[xi], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (FIRST && e == 0) {
            const floatx16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[xi & 3][0][0], av[xi & 1][0][0], z16, 0, 0, 0);
          } else {
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[xi & 3][e >> 2][e & 3], av[xi & 1][e >> 2][e & 3], acc[xi], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- output phase.  The output transform is lane-local in the MFMA layout (lane = tile, a group of four accumulator
  // registers = four channels), but in that layout every global access of the epilogue — the stores, the addend, the
  // BatchNorm input of the backward sums — is 32 pieces of 32 bytes per instruction, which the texture addresser serves
  // at one lane per cycle (107 ns per load beside the MFMAs against 27: tools/mfma_shadow; the loads alone cost
  // wino2_kernel<4> 9 % and <6> 16 %).  So Y goes through LDS once — this wave's 16 KB of stage 1, which is dead between
  // the unit's last products and the next unit's first transform (one more barrier per unit) — into a layout in which a
  // lane holds four channels (slot = lane & 7) of the tiles 8 k + (lane >> 3), k = 0 .. 3: a wave instruction then
  // touches 8 tiles x 128 contiguous bytes.  The epilogue runs in that layout: the BatchNorm coefficients of a lane's
  // four channels are loaded once per kernel, the sums over tiles accumulate in eight registers per lane across all units
  // and are reduced over lanes once, at the end (the MFMA layout needed an LDS staging pass per unit for them).
  //   xch[pq][tile 32][32 channels]: the eight 16-byte slots of a row XOR-swizzled by tile & 7 — conflict-free for the
  //   stores (8 consecutive lanes = 8 tiles, one slot) and for the loads' lane groups.
  float* xch = sm + W2_STAGE + wave * 4096;
  const int xw_row = l31 * 32, xw_sw = l31 & 7;
  const int t8 = lane >> 3, slot = lane & 7;
  const int xr_off = t8 * 32 + ((slot ^ t8) << 2);               // + (pq * 32 + 8 k) * 32
  const int ccol = cb * 64 + nh * 32 + 4 * slot;                 // this lane's four channels in the coalesced layout
  floatx4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, bis = bsc;
  if (EPI & 4) {
    bsc = *reinterpret_cast<const floatx4*>(p.bnb_scale + ccol); bsh = *reinterpret_cast<const floatx4*>(p.bnb_shift + ccol);
    bmu = *reinterpret_cast<const floatx4*>(p.bnb_mean + ccol); bis = *reinterpret_cast<const floatx4*>(p.bnb_invstd + ccol);
  }
  floatx4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;                     // BatchNorm partial sums of this lane's tiles, all units
  const int pix_b[4] = {0, Cn * 4, W * Cn * 4, (W + 1) * Cn * 4};   // byte offset of output pixel pq inside a tile

  int ubuf = 0;                                  // table buffer of the current unit
  for (int blk = blk0; blk < nblk; blk = blk_after(blk)) {
    __syncthreads();
    fill_tab(blk_after(blk_after(blk)), ubuf == 0 ? 2 : ubuf - 1);     // unit u + 2 -> buffer (u + 2) % 3
    chunk(std::true_type{}, 0, 0);
    for (int ck = 1; ck < nchunks; ++ck) {
      if (!(W2_DBG & 256)) __syncthreads();
      chunk(std::false_type{}, ck, ck & 1);
    }
    // destination offsets of the coalesced layout: tile 32 th + 8 k + t8 of the unit, k = 0 .. 3
    unsigned voff[4];
    int okm = 0;                                 // bit 4 k + pq: output pixel pq of tile k exists
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int4 te = tabs[ubuf * W2_TB + 32 * th + 8 * k + t8];
      const int y0 = 2 * te.y, x0 = 2 * te.z;
      const bool tok = te.x >= 0, okx = x0 + 1 < W, oky = y0 + 1 < H;
      voff[k] = (unsigned)((((te.x * H + y0) * W + x0) * Cn + ccol) * 4);
      okm |= ((tok ? 1 : 0) | (tok && okx ? 2 : 0) | (tok && oky ? 4 : 0) | (tok && okx && oky ? 8 : 0)) << (4 * k);
    }
    ubuf = ubuf == 2 ? 0 : ubuf + 1;
    if (W2_DBG & 8) {                 // (timing experiment: no output transform, every accumulator stored once)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const floatx4 v = floatx4{acc[x][4 * (x & 3)], acc[x][4 * (x & 3) + 1], acc[x][4 * (x & 3) + 2], acc[x][4 * (x & 3) + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, v), rsD, (okm & 1) ? voff[0] : 0x80000000u, (x & 3) * 32, 0);
        asm volatile("s_nop 1" : : "v"(v));
      }
      continue;
    }
    // the epilogue's inputs are requested before the output transform: their latency hides under it
    floatx4 xb[4][4], ad[4][4];
    auto epi_off = [&](int k, int pq) { return ((okm >> (4 * k + pq)) & 1) ? voff[k] : 0x80000000u; };
    if (EPI & 4) {
#pragma unroll
      for (int pq = 0; pq < 4; ++pq)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          xb[pq][k] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, epi_off(k, pq), pix_b[pq], 0));
    }
    __syncthreads();                  // every wave is done with stage 1 (the last chunk's V): it becomes the exchange buffer
    // ---- output transform, lane-local: Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]; the lane holds tile ptile and, per
    // group g of four accumulator registers, the channels 8 g + 4 h .. + 3 of the wave's 32
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      floatx4 T[4][2];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        // (read from the accumulation registers HERE: left to the compiler, the 256 reads of a unit are hoisted into
        // the block behind the chunk loop and ~150 values wait in vector registers — the allocator then spills the
        // per-lane offsets the chunk loop needs and reloads them with s_waitcnt vmcnt(0) at the top of every chunk)
        floatx4 M[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float t_;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t_) : "a"(acc[4 * a + b][4 * g + i]));
            M[b][i] = t_;
          }
        T[a][0] = pk4_add(pk4_add(M[0], M[1]), M[2]);
        T[a][1] = pk4_sub(pk4_sub(M[1], M[2]), M[3]);
      }
      float* wp = xch + xw_row + (((2 * g + h) ^ xw_sw) << 2);
#pragma unroll
      for (int qo = 0; qo < 2; ++qo) {
        *reinterpret_cast<floatx4*>(wp + qo * 1024) = pk4_add(pk4_add(T[0][qo], T[1][qo]), T[2][qo]);
        *reinterpret_cast<floatx4*>(wp + (2 + qo) * 1024) = pk4_sub(pk4_sub(T[1][qo], T[2][qo]), T[3][qo]);
      }
      __builtin_amdgcn_sched_barrier(0);      // one group at a time: hoisting the accumulator reads of all four costs 150 registers
    }
    if (EPI & 2) {
#pragma unroll
      for (int pq = 0; pq < 4; ++pq)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          ad[pq][k] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, epi_off(k, pq), pix_b[pq], 0));
    }
    // ---- epilogue in the coalesced layout (a wave's LDS operations execute in order: no barrier between its own stores
    // above and these loads)
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        floatx4 v = *reinterpret_cast<const floatx4*>(xch + (pq * 32 + 8 * k) * 32 + xr_off);
        if (EPI & 2) v = pk4_add(v, ad[pq][k]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, v), rsD, (W2_DBG & 128) ? 0x80000000u : epi_off(k, pq), pix_b[pq], 0);
        // The store reads its four data registers over several cycles.  The compiler's hazard recogniser assumes a
        // buffer store whose soffset is an SGPR is exempt from the ">64-bit store data, then VALU write" wait state;
        // measured on gfx950 it is not: a v_pk_add_f32 scheduled right behind the last store of the unit overwrote the
        // fourth dword in flight (wrong values in one channel of some tiles, run to run).  Keep v alive for two more
        // wait states.
        asm volatile("s_nop 1" : : "v"(v));
        if (EPI & 5) {
          const bool okv = (okm >> (4 * k + pq)) & 1;
          const floatx4 t0 = okv ? v : floatx4{0.f, 0.f, 0.f, 0.f};
          if (EPI & 1) {
            s0 += t0;
            s1 += t0 * t0;
          } else {
            const floatx4 x_ = xb[pq][k];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = (!p.bnb_relu || fmaf(x_[i], bsc[i], bsh[i]) > 0.f) ? t0[i] : 0.f;
              s0[i] += t;
              s1[i] += t * ((x_[i] - bmu[i]) * bis[i]);
            }
          }
        }
      }
    }
  }
  if ((EPI & 5) && p.stats) {   // one partial row [2][Cn] per workgroup: the eight tile groups of a wave, then its two tile halves
    __syncthreads();
    float* red = sm;            // [2 terms][4 waves][32]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = s0[i], b = s1[i];
      a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      if (lane < 8) {
        red[wave * 32 + 4 * lane + i] = a;
        red[128 + wave * 32 + 4 * lane + i] = b;
      }
    }
    __syncthreads();
    float* row = p.stats + (long long)blockIdx.x * 2 * Cn;
    for (int c = tid; c < Cn; c += 256) {
      float a = 0.f, b = 0.f;
      if (c / 64 == cb) {
        const int cc = c % 64, w0 = (cc >> 5) * 2, c32 = cc & 31;       // waves w0 (th = 0) and w0 + 1 (th = 1)
        a = red[w0 * 32 + c32] + red[(w0 + 1) * 32 + c32];
        b = red[128 + w0 * 32 + c32] + red[128 + (w0 + 1) * 32 + c32];
      }
      row[c] = a;
      row[Cn + c] = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// wino2p_kernel (round 5): wino2_kernel with V split ONCE, by the thread that transforms it, and kept in LDS as the three bf16
// planes the products read.  In wino2_kernel every wave splits the V fragment of every transform point it multiplies — the two
// waves of a tile half split the same 8 values per lane: 36 vector instructions beside 6 matrix instructions, 16 times per chunk,
// as long as the products themselves.  Here the transform thread (tile, 4 channels) splits its two channel pairs per
// transform point (18 instructions, 288 per chunk instead of 576) and a product step reads its fragment with three ds_read_b128.
//   LDS: V as bf16 planes is 6 B per element: a chunk (16 points x 64 tiles x 16 channels) takes 96 KB, two do not fit.  The
//   stage is cut in HALVES of 8 transform points (48 KB) in a ring of three: while half g is multiplied the transform writes
//   half g + 2 (the same half of the next chunk: rows 0-1 of the 4 x 4 transform with the first half, rows 2-3 with the second)
//   into the third buffer; one barrier per half says "everybody is done reading g" (g + 3 will overwrite it) — reading g + 1,
//   complete since the barrier before, may start in front of it, so the fragment prefetch runs across the barriers.
//     half-stage [point 8][hi | mid | lo][tile 64][16 channels bf16 = 32 B], the two 16-byte halves of a row swapped on tiles
//     with bit 3 set: the fragment reads (16 lanes = 16 tiles x 16 B, row pitch 32 B) and the transform's 8-byte stores
//     (4 threads per tile) each cover the 64 banks once.
//   Schedule of a chunk: column pass at point 0, ONE point transformed, split and stored per product step, the patch loads of
//   the chunk after the next at points 2-7 (the raw rows are dead after the column pass); a unit's first chunk loads the next
//   chunk's patch at points 0-3 and transforms two points per step in its second half, as wino2_kernel does.
//   Output phase: the exchange buffer is the ring's dead buffer (48 KB; the other two hold the next unit's first chunk), so
//   Y goes through it in two passes of two output pixels (8 KB per wave), the second pair waiting in registers.
constexpr int W2P_HS = 8 * 3 * W2_TB * 32;                 // bytes of a half-stage
constexpr int W2P_TAB_OFF = 3 * W2P_HS;                    // bytes: three tile tables of 64 x int4 behind the ring
constexpr int W2P_LDS_BYTES = W2P_TAB_OFF + 3 * W2_TB * 16;

template <int EPI>
__global__ __launch_bounds__(256) void wino2p_kernel(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  char* smb = reinterpret_cast<char*>(sm);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int th = wave & 1, nh = wave >> 1;
  const int H = p.H, W = p.W, TW = p.TW, TPF = p.TH * p.TW, Cr = p.Cr, Cn = p.Cn;
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((long long)p.F * H * W * Cr * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, 0, 16 * Cn * Cr * 6, 0x00020000);
  const int dbytes = (int)((long long)p.F * H * W * Cn * 4);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, dbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 2) ? p.addend : p.src), 0, (EPI & 2) ? dbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 4) ? p.bnb_x : p.src), 0, (EPI & 4) ? dbytes : 0, 0x00020000);
  const int nchunks = Cr / W2_CK;
  // unit order: see wino_kernel
  const bool xl = p.xcd_local && gridDim.x % (8 * p.ncb) == 0;
  const int lid = xl ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cb = lid % p.ncb, q = lid / p.ncb, Gq = (int)gridDim.x / p.ncb;
  const int nblk = p.units / p.ncb;
  const int fullq = xl ? nblk / Gq * Gq : nblk;
  const int qlast = fullq + (q % (Gq / 8)) * 8 + q / (Gq / 8);
  auto blk_after = [&](int blk) {
    if (blk >= fullq) return nblk;
    const int nx = blk + Gq;
    return nx < fullq ? nx : qlast;
  };
  const int blk0 = q < fullq ? q : (xl ? qlast : q);
  int4* tabs = reinterpret_cast<int4*>(smb + W2P_TAB_OFF);
  auto fill_tab = [&](int blk, int buf) {
    int tid_t = tid;                                 // (per unit, from an opaque copy: as loop invariants the table address and the
    asm volatile("" : "+v"(tid_t));                  //  divisions' reciprocals were spilled in front of the unit loop)
    if (tid_t < W2_TB) {
      const long long t = (long long)blk * W2_TB + tid_t;
      int4 e = {-1, 0, 0, 0};
      if (blk < nblk && t < p.ntiles) {
        int tpf = TPF, tw = TW;
        asm volatile("" : "+s"(tpf), "+s"(tw));
        const int f = (int)(t / tpf), rem = (int)(t - (long long)f * tpf);
        e.x = f; e.y = rem / tw; e.z = rem - e.y * tw;
      }
      tabs[buf * W2_TB + tid_t] = e;
    }
  };
  // ---- this thread's item of the input transform: tile tt, channels c4 .. c4 + 3 of the chunk
  const int tt = tid >> 2, cq = tid & 3, c4 = cq * 4;
  const unsigned px_b = (unsigned)Cr * 4u, row_b = (unsigned)W * px_b;
  floatx4 raw[4][4];
  unsigned ld_o11 = 0;
  bool ld_oky[4] = {false, false, false, false}, ld_okx[4] = {false, true, false, false};
  int ld_ck = 0, ld_buf = 0;
  auto ld_unit = [&](int buf) {
    const int4 e = tabs[buf * W2_TB + tt];
    const bool t_ok = e.x >= 0;
    const int y1 = 2 * e.y, x1 = 2 * e.z;
    ld_o11 = (unsigned)(((e.x * H + y1) * W + x1) * Cr + c4) * 4u;
    ld_oky[0] = t_ok && e.y > 0; ld_oky[1] = t_ok; ld_oky[2] = t_ok && y1 + 1 < H; ld_oky[3] = t_ok && y1 + 2 < H;
    ld_okx[0] = e.z > 0; ld_okx[1] = true; ld_okx[2] = x1 + 1 < W; ld_okx[3] = x1 + 2 < W;
  };
  auto ld_issue = [&](int a, int b) {
    const unsigned rowoff = ld_o11 + (unsigned)(a - 1) * row_b;
    raw[a][b] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                rsX, (ld_oky[a] && ld_okx[b]) ? (b == 0 ? rowoff - px_b : rowoff) : 0x80000000u,
                                                ld_ck * (W2_CK * 4) + (b == 0 ? 0 : (b - 1) * (int)px_b), 0));
  };
  auto ld_advance = [&]() {
    if (++ld_ck == nchunks) {
      ld_ck = 0;
      ld_buf = ld_buf == 2 ? 0 : ld_buf + 1;
      ld_unit(ld_buf);
    }
  };
  floatx4 w_[4][4];
  auto col_pass = [&]() {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      w_[0][b] = pk4_sub(raw[0][b], raw[2][b]);
      w_[1][b] = pk4_add(raw[1][b], raw[2][b]);
      w_[2][b] = pk4_sub(raw[2][b], raw[1][b]);
      w_[3][b] = pk4_sub(raw[1][b], raw[3][b]);
    }
  };
  // transform point xi (row a = xi >> 2 of the column pass, column j = xi & 3), split, three 8-byte stores
  const int wr_off = tt * 32 + (((cq >> 1) ^ ((tt >> 3) & 1)) << 4) + ((cq & 1) << 3);
  auto point = [&](char* Wb, int xi) {
    const int a = xi >> 2, j = xi & 3;
    const floatx4 v = j == 0 ? pk4_sub(w_[a][0], w_[a][2]) : j == 1 ? pk4_add(w_[a][1], w_[a][2])
                    : j == 2 ? pk4_sub(w_[a][2], w_[a][1]) : pk4_sub(w_[a][1], w_[a][3]);
    unsigned h0, m0, l0, h1, m1, l1;
    w2p_split2(v[0], v[1], h0, m0, l0);
    w2p_split2(v[2], v[3], h1, m1, l1);
    char* dst = Wb + wr_off + (xi & 7) * (3 * W2_TB * 32);
    *reinterpret_cast<uintx2_t*>(dst) = uintx2_t{h0, h1};
    *reinterpret_cast<uintx2_t*>(dst + W2_TB * 32) = uintx2_t{m0, m1};
    *reinterpret_cast<uintx2_t*>(dst + 2 * W2_TB * 32) = uintx2_t{l0, l1};
  };
  // ---- products: operands
  const int ptile = 32 * th + l31;
  const int rd_off = ptile * 32 + ((h ^ ((ptile >> 3) & 1)) << 4);
  const unsigned u_voff = (unsigned)(lane * 16);
  const int u_xi = Cn * Cr * 6;
  const int u_ck = 3 * 1024;
  const int u_base = __builtin_amdgcn_readfirstlane((cb * 2 + nh) * nchunks * u_ck);
  floatx4 bv[4][3];
  auto load_u = [&](int slot, int xi, int ck) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
      bv[slot][q] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsU, u_voff + 1024u * q, u_base + xi * u_xi + ck * u_ck, 0));
  };
  w2_bf16x8 vs[2][3];
  auto read_v = [&](const char* Vb, int slot, int xi8) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
      vs[slot][t] = *reinterpret_cast<const w2_bf16x8*>(Vb + rd_off + (xi8 * 3 + t) * (W2_TB * 32));
  };
  floatx16 acc[16];

#if AVID_W2_STAGGER
  {
    int mine = 0;
    for (int b = blk0; b < nblk; b = blk_after(b)) ++mine;
    const int most = (nblk + Gq - 1) / Gq;
    if (mine < most) {
      const int steps = ((lid * 5) & 7) * nchunks / 8 * AVID_W2_STAGGER;
      for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(127);
    }
  }
#endif
  // ---- prologue: the first unit's first chunk into ring buffers 0 and 1
  fill_tab(blk0, 0);
  fill_tab(blk_after(blk0), 1);
  __syncthreads();
  ld_unit(0);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) ld_issue(a, b);
  ld_advance();
  col_pass();
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) point(smb + (xi >> 3) * W2P_HS, xi);
  load_u(0, 0, 0);
  load_u(1, 1, 0);
  load_u(2, 2, 0);
  int rb = 0;                                     // ring buffer of the current chunk's first half

  auto chunk = [&](auto first_tag, int ck) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int ck_n = ck + 1 < nchunks ? ck + 1 : 0;
    const bool last = !FIRST && ck + 1 == nchunks;
    const int b1 = rb == 2 ? 0 : rb + 1, b2 = b1 == 2 ? 0 : b1 + 1;
    const char* R0 = smb + rb * W2P_HS;           // halves of this chunk
    const char* R1 = smb + b1 * W2P_HS;
    char* W0 = smb + b2 * W2P_HS;                 // halves of the next chunk: the free buffer, then this chunk's first half
    char* W1 = smb + rb * W2P_HS;
    if (FIRST) read_v(R0, 0, 0);                  // (otherwise the previous chunk's last step has asked for it)
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      if (!(W2_DBG & 4)) { if (xi + 3 < 16) load_u((xi + 3) & 3, xi + 3, ck); else load_u((xi + 3) & 3, xi + 3 - 16, ck_n); }
      // the fragment of the next point: this chunk's, or — complete since the barrier in the middle of this chunk — the next chunk's
      // first (not across the output phase)
      if (!(W2_DBG & 16)) {
        if (xi + 1 < 16) read_v(xi + 1 < 8 ? R0 : R1, (xi + 1) & 1, (xi + 1) & 7);
        else if (!last) read_v(W0, 0, 0);
      }
      if (FIRST) {
        if (xi < 4 && !(W2_DBG & 2)) {
#pragma unroll
          for (int b = 0; b < 4; ++b) ld_issue(xi, b);
          if (xi == 3) ld_advance();
        }
        if (xi == 8 && !(W2_DBG & 1)) col_pass();
        if (xi >= 8 && !(W2_DBG & 1)) {
          point(xi < 12 ? W0 : W1, 2 * (xi - 8));
          point(xi < 12 ? W0 : W1, 2 * (xi - 8) + 1);
        }
        if (xi >= 10 && !(W2_DBG & 2)) {
          const int j = xi - 10;                 // 3, 3, 3, 3, 2, 2 loads
          const int n = j < 4 ? 3 : 2, base = j < 4 ? j * 3 : 12 + (j - 4) * 2;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < n) ld_issue((base + k) >> 2, (base + k) & 3);
          if (j == 5) ld_advance();
        }
      } else {
        if (xi == 0 && !(W2_DBG & 1)) col_pass();
        if (!(W2_DBG & 1)) point(xi < 8 ? W0 : W1, xi);
        if (xi >= 2 && xi < 8 && !last && !(W2_DBG & 2)) {
          const int j = xi - 2;
          const int n = j < 4 ? 3 : 2, base = j < 4 ? j * 3 : 12 + (j - 4) * 2;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < n) ld_issue((base + k) >> 2, (base + k) & 3);
          if (j == 5) ld_advance();
        }
      }
      const w2_bf16x8 uh = __builtin_bit_cast(w2_bf16x8, bv[xi & 3][0]), um = __builtin_bit_cast(w2_bf16x8, bv[xi & 3][1]),
                      ul = __builtin_bit_cast(w2_bf16x8, bv[xi & 3][2]);
      const w2_bf16x8 vh = vs[xi & 1][0], vm = vs[xi & 1][1], vl = vs[xi & 1][2];
      if (FIRST) {
        const floatx16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vl, z16, 0, 0, 0);
      } else {
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vl, acc[xi], 0, 0, 0);
      }
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ul, vh, acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vm, acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vm, acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(um, vh, acc[xi], 0, 0, 0);
      acc[xi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, vh, acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if ((xi == 7 || xi == 15) && !(W2_DBG & 32)) __syncthreads();    // everybody is done reading this half: the next half's transform writes over it
    }
    rb = b2;
  };

  // ---- output phase (see wino2_kernel), the exchange buffer in the ring's dead buffer, two output pixels per pass
  // The BatchNorm's per-channel vectors of this thread's four channels are fetched per UNIT, behind the output transform (buffer
  // loads next to the loop's stores: the compiler cannot hoist them).  Loaded once in front of the unit loop they were 16
  // registers that lived through every chunk loop: the 13 / 23 spilled registers of the two BatchNorm-backward forms
  // (tools/kernel_resources.py) were exactly such loop-invariants, stored once and re-read per unit.
  const __amdgpu_buffer_rsrc_t rsSc = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 4) ? p.bnb_scale : p.src), 0, (EPI & 4) ? Cn * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsSh = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 4) ? p.bnb_shift : p.src), 0, (EPI & 4) ? Cn * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsMu = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 4) ? p.bnb_mean : p.src), 0, (EPI & 4) ? Cn * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsIs = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI & 4) ? p.bnb_invstd : p.src), 0, (EPI & 4) ? Cn * 4 : 0, 0x00020000);
  floatx4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  const int pix_b[4] = {0, Cn * 4, W * Cn * 4, (W + 1) * Cn * 4};

  int ubuf = 0;
  for (int blk = blk0; blk < nblk; blk = blk_after(blk)) {
    __syncthreads();
    fill_tab(blk_after(blk_after(blk)), ubuf == 0 ? 2 : ubuf - 1);
    chunk(std::true_type{}, 0);
    for (int ck = 1; ck < nchunks; ++ck) chunk(std::false_type{}, ck);
    // (the barrier behind the last half: every wave is done with it — rb now names the buffer the NEXT chunk's first half sits in,
    //  the one before it in the ring is the dead one)
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int wave_o = __builtin_amdgcn_readfirstlane(tid_o >> 6), th_o = wave_o & 1;
    float* xch = reinterpret_cast<float*>(smb + (rb == 0 ? 2 : rb - 1) * W2P_HS) + wave_o * 2048;
    // the output phase's lane constants, recomputed per unit from a lane id the compiler cannot see through: as loop invariants
    // they were hoisted in front of the unit loop and — the chunk loops use every register — spilled there (nine scratch stores
    // and reloads per unit where eight vector instructions do)
    const int lane_o = tid_o & 63, h_o = lane_o >> 5;
    const int l31o = lane_o & 31;
    const int xw_row = l31o * 32, xw_sw = l31o & 7;
    const int t8 = lane_o >> 3, slot = lane_o & 7;
    const int xr_off = t8 * 32 + ((slot ^ t8) << 2);
    const int ccol = cb * 64 + nh * 32 + 4 * slot;
    unsigned voff[4];
    int okm = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int4 te = tabs[ubuf * W2_TB + 32 * th_o + 8 * k + t8];
      const int y0 = 2 * te.y, x0 = 2 * te.z;
      const bool tok = te.x >= 0, okx = x0 + 1 < W, oky = y0 + 1 < H;
      voff[k] = (unsigned)((((te.x * H + y0) * W + x0) * Cn + ccol) * 4);
      okm |= ((tok ? 1 : 0) | (tok && okx ? 2 : 0) | (tok && oky ? 4 : 0) | (tok && okx && oky ? 8 : 0)) << (4 * k);
    }
    ubuf = ubuf == 2 ? 0 : ubuf + 1;
    if (W2_DBG & 8) {                 // (timing experiment: no output transform, every accumulator stored once)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const floatx4 v = floatx4{acc[x][4 * (x & 3)], acc[x][4 * (x & 3) + 1], acc[x][4 * (x & 3) + 2], acc[x][4 * (x & 3) + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, v), rsD, (okm & 1) ? voff[0] : 0x80000000u, (x & 3) * 32, 0);
        asm volatile("s_nop 1" : : "v"(v));
      }
      continue;
    }
    // the epilogue's inputs (BatchNorm input of the backward sums, addend), two output pixels at a time like the exchange: the
    // first pair is requested before / behind the output transform, the second pair as the first pair's registers come free
    floatx4 xb[2][4], ad[2][4];
    auto epi_off = [&](int k, int pq) { return ((okm >> (4 * k + pq)) & 1) ? voff[k] : 0x80000000u; };
    auto load_xb = [&](int pq) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        xb[pq & 1][k] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsB, epi_off(k, pq), pix_b[pq], 0));
    };
    auto load_ad = [&](int pq) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        ad[pq & 1][k] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsA, epi_off(k, pq), pix_b[pq], 0));
    };
    // (the first pair only: 16 more registers held across the output transform — its peak — were 13 / 23 spilled registers in the
    //  two BatchNorm-backward forms; the second pixel's rows are requested behind the transform, with the addend's)
    if (EPI & 4) load_xb(0);
    floatx4 low[4][2];                 // output row 1 (pixels 2, 3) of every channel group: the second pass
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      floatx4 T[4][2];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        floatx4 M[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float t_;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t_) : "a"(acc[4 * a + b][4 * g + i]));
            M[b][i] = t_;
          }
        T[a][0] = pk4_add(pk4_add(M[0], M[1]), M[2]);
        T[a][1] = pk4_sub(pk4_sub(M[1], M[2]), M[3]);
      }
      float* wp = xch + xw_row + (((2 * g + h_o) ^ xw_sw) << 2);
#pragma unroll
      for (int qo = 0; qo < 2; ++qo) {
        *reinterpret_cast<floatx4*>(wp + qo * 1024) = pk4_add(pk4_add(T[0][qo], T[1][qo]), T[2][qo]);
        low[g][qo] = pk4_sub(pk4_sub(T[1][qo], T[2][qo]), T[3][qo]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI & 4) load_xb(1);
    if (EPI & 2) { load_ad(0); load_ad(1); }
    floatx4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, bis = bsc;
    if (EPI & 4) {
      bsc = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsSc, ccol * 4, 0, 0));
      bsh = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsSh, ccol * 4, 0, 0));
      bmu = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsMu, ccol * 4, 0, 0));
      bis = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsIs, ccol * 4, 0, 0));
    }
#pragma unroll
    for (int pq = 0; pq < 4; ++pq) {
      if (pq == 2) {      // second pass (a wave's LDS operations execute in order: its loads of the first pass are behind it)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float* wp = xch + xw_row + (((2 * g + h_o) ^ xw_sw) << 2);
          *reinterpret_cast<floatx4*>(wp) = low[g][0];
          *reinterpret_cast<floatx4*>(wp + 1024) = low[g][1];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        floatx4 v = *reinterpret_cast<const floatx4*>(xch + ((pq & 1) * 32 + 8 * k) * 32 + xr_off);
        if (EPI & 2) v = pk4_add(v, ad[pq & 1][k]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, v), rsD, epi_off(k, pq), pix_b[pq], 0);
        asm volatile("s_nop 1" : : "v"(v));      // (see wino2_kernel: the store's data registers are read over several cycles)
        if (EPI & 5) {
          const bool okv = (okm >> (4 * k + pq)) & 1;
          const floatx4 t0 = okv ? v : floatx4{0.f, 0.f, 0.f, 0.f};
          if (EPI & 1) {
            s0 += t0;
            s1 += t0 * t0;
          } else {
            const floatx4 x_ = xb[pq & 1][k];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = (!p.bnb_relu || fmaf(x_[i], bsc[i], bsh[i]) > 0.f) ? t0[i] : 0.f;
              s0[i] += t;
              s1[i] += t * ((x_[i] - bmu[i]) * bis[i]);
            }
          }
        }
      }
      if (pq < 2) {
        __builtin_amdgcn_sched_barrier(0);
        if (EPI & 4) load_xb(pq + 2);
        if (EPI & 2) load_ad(pq + 2);
      }
    }
  }
  if ((EPI & 5) && p.stats) {
    __syncthreads();
    float* red = sm;            // [2 terms][4 waves][32]
    int lane_r = lane;
    asm volatile("" : "+v"(lane_r));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = s0[i], b = s1[i];
      a += __shfl_xor(a, 8, 64); b += __shfl_xor(b, 8, 64);
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      if (lane_r < 8) {
        red[wave * 32 + 4 * lane_r + i] = a;
        red[128 + wave * 32 + 4 * lane_r + i] = b;
      }
    }
    __syncthreads();
    float* row = p.stats + (long long)blockIdx.x * 2 * Cn;
    for (int c = tid; c < Cn; c += 256) {
      float a = 0.f, b = 0.f;
      if (c / 64 == cb) {
        const int cc = c % 64, w0 = (cc >> 5) * 2, c32 = cc & 31;
        a = red[w0 * 32 + c32] + red[(w0 + 1) * 32 + c32];
        b = red[128 + w0 * 32 + c32] + red[128 + (w0 + 1) * 32 + c32];
      }
      row[c] = a;
      row[Cn + c] = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers through the Winograd identity (backward of models/network_blocks.py:35,40):
//     dU[xi][n][c] = sum over tiles of (A dY A^T)[xi][tile][n] * (B^T d B)[xi][tile][c],     dw = G^T dU G
// 16 instead of 36 multiply-adds per (tile, n, c) — the 2.25x of the forward.  One workgroup = 8 waves = one
// (64 output channels) x (64 input channels) block of dw for a contiguous range of tiles; GEMM-K is the tile index.
//   per chunk of 8 tiles: the threads of waves 0-3 (tile = tid >> 5, channel pair = tid & 31) load a 4x4 input patch as
//   8-byte vectors and transform it (V = B^T d B), those of waves 4-7 a 2x2 gradient tile (dM = A dY A^T); out-of-frame
//   taps / pixels read 0.  LDS holds two stages of V | dM as [tile][xi][64 channels] (131 KB: one workgroup per CU,
//   two waves per SIMD).  While chunk k is multiplied, the registers holding chunk k + 1 are transformed and written
//   to the other stage and the loads of chunk k + 2 are issued, all inside the MFMA stream: one barrier per chunk.
//   products: wave w owns the quadrant (n half, c half) = w & 3 of the block for the transform rows i in {2e, 2e+1},
//   e = w >> 2, all four columns j: 8 accumulators of 32 x 32; both MFMA operands are ds_read_b32 fragments
//   (lane = channel, half-wave = tile of the k pair).
//   epilogue: G^T . G in registers — columns j inside a wave, rows i as two partial sums; the e = 1 waves hand theirs
//   over through LDS, the e = 0 waves write the block as dw[n][3][3][c] into this split's slab; wgrad_reduce_kernel
//   (csrc/conv.hip) sums the slabs in fixed order.
// Development switches of wino_wgrad_kernel (tools/build_variant.sh NAME wino "-DAVID_WW_DBG=n"; results are WRONG with any of
// them: timing only).  1: no global loads; 2: no transform / LDS stores of the next chunk; 4: no barrier per chunk; 8: the products
// as one vector multiply-add each (fragment reads stay); 16: no fragment reads (the products on registers).
#ifndef AVID_WW_DBG
#define AVID_WW_DBG 0
#endif
constexpr int WW_DBG = AVID_WW_DBG;
// -DAVID_WW_TRACE: shader cycles per wave, summed over its chunks — [0] k-step 0 (products + transform of the next chunk),
// [1] k-step 1 (+ the loads of the chunk after), [2] k-steps 2-3, [3] at the barrier, [4] first half of k-step 0 (the left factor),
// [6] / [7] shader cycles / 10 ns ticks over the loop; tools/wino_wgrad_trace.py
#ifdef AVID_WW_TRACE
__device__ long long g_ww_trace[1024 * 8 * 8];
extern "C" int avid_debug_ww_trace(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ww_trace), sizeof(long long) * 1024 * 8 * 8);
}
#define WW_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = clock64(); ww_t[i] += now_ - ww_prev; ww_prev = now_; \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WW_STAMP(i) do {} while (0)
#endif
constexpr int WW_TK = 8;                               // tiles per LDS stage (GEMM-K of a stage)
constexpr int WW_HALF = WW_TK * 16 * 64;               // floats of V (or dM) in one stage: [tile][xi][64 channels]
constexpr int WW_STAGE = 2 * WW_HALF;                  // V | dM
constexpr int WW_LDS_FLOATS = 2 * WW_STAGE;            // two stages: 131072 B

struct WinoWgradArgs {
  const float* __restrict__ x;
  const float* __restrict__ dy;
  float* __restrict__ out;              // [nsplit][Cn][9][Cr]
  int F, H, W, TH, TW, Cr, Cn, ncb;     // Cr = Cin, Cn = Cout, ncb = Cr / 64
  int pairs;                            // (Cn / 64) * ncb blocks of dw
  int nchunks, cps;                     // chunks of WW_TK tiles; chunks per split
  long long ntiles;
  unsigned mgTPF, mgTW;                 // multiply-shift division by TH * TW and TW
  int shTPF, shTW;
};

// ROLE 0 (waves 0-3): input-transform threads, V = B^T d B of (tile, channel pair); ROLE 1 (waves 4-7): gradient-
// transform threads, dM' = |A| dY |A|^T of (tile, output-channel pair) — the signs of A's last row (A[3] = [0,-1]) are
// applied once, in the epilogue (dU[i][j] = s_i s_j dU'[i][j], s = [1,1,1,-1]).  Every wave multiplies.
// Measured alternatives (conv2x, 401408 pixels; this form: 152 us): 16-tile single stage with the transform between
// two barriers 173 us; this pipeline with compiler-scheduled (unpacked) adds 158 us; one wave per SIMD holding all 16
// transform points of a quadrant (no exchange in the epilogue) 168 us; the barrier moved behind k-step 2 with the next
// chunk's first fragments fetched under k-step 3: 168 us.
template <int ROLE>
__device__ __forceinline__ void wino_wgrad_body(const WinoWgradArgs& p, float* sm, floatx16 (&acc)[8], int cbk, int nbk,
                                                int ch0, int ch1) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int H = p.H, W = p.W;
  const int C = ROLE == 0 ? p.Cr : p.Cn;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(ROLE == 0 ? p.x : p.dy), 0, (int)((long long)p.F * H * W * C * 4), 0x00020000);
  const int tl = (tid >> 5) & 7, c2 = (tid & 31) * 2;           // this thread's tile of a chunk and its channel pair
  const unsigned chan_b = (unsigned)((ROLE == 0 ? cbk : nbk) * 64 + c2) * 4u;
  const int px_b = C * 4;
  const unsigned row_b = (unsigned)W * (unsigned)px_b;
  constexpr int NP = ROLE == 0 ? 4 : 2;                         // patch extent: 4x4 input patch / 2x2 gradient tile
  floatx2 raw[NP][NP];                                          // the chunk after the one in LDS

  auto load_chunk = [&](int ch) {
    const long long t = (long long)ch * WW_TK + tl;
    const bool t_ok = ch < ch1 && t < p.ntiles;
    const unsigned tt = t_ok ? (unsigned)t : 0u;
    const unsigned f = (unsigned)(((unsigned long long)tt * p.mgTPF) >> p.shTPF);
    const unsigned rem = tt - f * (unsigned)(p.TH * p.TW);
    const unsigned ti = (unsigned)(((unsigned long long)rem * p.mgTW) >> p.shTW);
    const unsigned tj = rem - ti * (unsigned)p.TW;
    // pixel (2 ti, 2 tj) — element (1, 1) of the input patch, (0, 0) of the gradient tile — is always inside the frame.
    // The range check of a buffer load covers the vector offset only, so every offset is based there: the columns to
    // its right ride in the scalar offset, column 0 / row 0 of the patch lie one pixel / row below it and are masked
    // when they leave the frame.
    const unsigned o11 = (unsigned)(((int)f * H + 2 * (int)ti) * W + 2 * (int)tj) * (unsigned)px_b + chan_b;
    const int y1 = 2 * (int)ti, x1 = 2 * (int)tj;
    const bool oky4[4] = {t_ok && ti > 0, t_ok, t_ok && y1 + 1 < H, t_ok && y1 + 2 < H};
    const bool okx4[4] = {tj > 0, true, x1 + 1 < W, x1 + 2 < W};
    constexpr int O = ROLE == 0 ? 0 : 1;                        // patch index of local index 0
#pragma unroll
    for (int a = 0; a < NP; ++a) {
      const unsigned rowoff = o11 + (unsigned)(a + O - 1) * row_b;
#pragma unroll
      for (int b = 0; b < NP; ++b) {
        const int pb = b + O;                                    // patch column 0..3
        const unsigned vo = pb == 0 ? rowoff - (unsigned)px_b : rowoff;
        if (WW_DBG & 1) { raw[a][b] = floatx2{(float)vo, 1.f}; continue; }
        raw[a][b] = __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(
                                                    rs, (oky4[a + O] && okx4[pb]) ? vo : 0x80000000u, pb == 0 ? 0 : (pb - 1) * px_b, 0));
      }
    }
  };
  floatx2 w_[4][NP];
  auto col = [&](int b) {     // one column of the patch through the left factor
    if (ROLE == 0) {          // B^T d:  rows of B^T = [1,0,-1,0], [0,1,1,0], [0,-1,1,0], [0,1,0,-1]
      w_[0][b] = pk_sub(raw[0][b], raw[2 % NP][b]);
      w_[1][b] = pk_add(raw[1][b], raw[2 % NP][b]);
      w_[2][b] = pk_sub(raw[2 % NP][b], raw[1][b]);
      w_[3][b] = pk_sub(raw[1][b], raw[3 % NP][b]);
    } else {                  // |A| dY:  |A| = [[1,0],[1,1],[1,-1],[0,1]]
      w_[0][b] = raw[0][b];
      w_[1][b] = pk_add(raw[0][b], raw[1][b]);
      w_[2][b] = pk_sub(raw[0][b], raw[1][b]);
      w_[3][b] = raw[1][b];
    }
  };
  auto row_store = [&](float* stage, int a) {     // row a through the right factor -> LDS stage ([tile][xi][64])
    float* dst = stage + (ROLE == 0 ? 0 : WW_HALF) + (tl * 16 + a * 4) * 64 + c2;
    floatx2 o0, o1, o2, o3;
    if (ROLE == 0) {
      o0 = pk_sub(w_[a][0], w_[a][2 % NP]); o1 = pk_add(w_[a][1], w_[a][2 % NP]);
      o2 = pk_sub(w_[a][2 % NP], w_[a][1]); o3 = pk_sub(w_[a][1], w_[a][3 % NP]);
    } else {
      o0 = w_[a][0]; o1 = pk_add(w_[a][0], w_[a][1]); o2 = pk_sub(w_[a][0], w_[a][1]); o3 = w_[a][1];
    }
    *reinterpret_cast<floatx2*>(dst + 0 * 64) = o0;
    *reinterpret_cast<floatx2*>(dst + 1 * 64) = o1;
    *reinterpret_cast<floatx2*>(dst + 2 * 64) = o2;
    *reinterpret_cast<floatx2*>(dst + 3 * 64) = o3;
  };

  const int quad = wave & 3, e = wave >> 2, nh = quad >> 1, chh = quad & 1;
  // fragment bases inside a stage: dM (A operand, lane = output channel) and V (B operand, lane = input channel)
  const int a_frag = WW_HALF + (h * 16 + e * 8) * 64 + nh * 32 + l31;
  const int b_frag = (h * 16 + e * 8) * 64 + chh * 32 + l31;
  load_chunk(ch0);
#pragma unroll
  for (int b = 0; b < NP; ++b) col(b);
#pragma unroll
  for (int a = 0; a < 4; ++a) row_store(sm, a);
  load_chunk(ch0 + 1);
  __syncthreads();
#ifdef AVID_WW_TRACE
  long long ww_t[6] = {0, 0, 0, 0, 0, 0}, ww_prev = clock64();
  const long long ww_wall0 = wall_clock64(), ww_c0 = ww_prev;
#endif
  auto product = [&](int x8, float a, float b) {
    if (WW_DBG & 8) acc[x8][0] = fmaf(a, b, acc[x8][0]);
    else acc[x8] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[x8], 0, 0, 0);
  };
  for (int ch = ch0; ch < ch1; ++ch) {
    const int u = (ch - ch0) & 1;
    const float* cur = sm + u * WW_STAGE;
    float* nxt = sm + (u ^ 1) * WW_STAGE;
    const float* Ab = cur + a_frag;
    const float* Bb = cur + b_frag;
    float af[2][8], bf[2][8];
    auto frag = [&](int kk, int x8) {
      if (WW_DBG & 16) { af[kk & 1][x8] = (float)(kk + x8); bf[kk & 1][x8] = (float)(ch + x8); return; }
      af[kk & 1][x8] = Ab[(kk * 2 * 16 + x8) * 64];
      bf[kk & 1][x8] = Bb[(kk * 2 * 16 + x8) * 64];
    };
#pragma unroll
    for (int x8 = 0; x8 < 8; ++x8) frag(0, x8);
    // k-step 0: the registers (chunk ch + 1) are transformed and written to the other stage, one slice per MFMA
#pragma unroll
    for (int x8 = 0; x8 < 8; ++x8) {
      frag(1, x8);
      product(x8, af[0][x8], bf[0][x8]);
      if (!(WW_DBG & 2)) {
        if (x8 < 4) {
          if (x8 < NP) col(x8);
        } else {
          row_store(nxt, x8 - 4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef AVID_WW_TRACE
      if (x8 == 3) WW_STAMP(4);
#endif
    }
    WW_STAMP(0);
    // k-step 1: the loads of chunk ch + 2 are issued (address arithmetic and VMEM spread over the MFMAs)
#pragma unroll
    for (int x8 = 0; x8 < 8; ++x8) frag(2, x8);
    load_chunk(ch + 2);
#pragma unroll
    for (int x8 = 0; x8 < 8; ++x8) product(x8, af[1][x8], bf[1][x8]);
#pragma unroll
    for (int x8 = 0; x8 < 8; ++x8) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                     // DS read
      __builtin_amdgcn_sched_group_barrier(0x002, ROLE == 0 ? 6 : 4, 0);     // VALU (address arithmetic)
      __builtin_amdgcn_sched_group_barrier(0x020, ROLE == 0 ? 2 : 1, 0);     // VMEM read
    }
    __builtin_amdgcn_sched_barrier(0);
    WW_STAMP(1);
#pragma unroll
    for (int kk = 2; kk < WW_TK / 2; ++kk) {
#pragma unroll
      for (int x8 = 0; x8 < 8; ++x8) {
        if (kk + 1 < WW_TK / 2) frag(kk + 1, x8);
        product(x8, af[kk & 1][x8], bf[kk & 1][x8]);
      }
    }
    WW_STAMP(2);
    if (!(WW_DBG & 4)) __syncthreads();       // the other stage is written; everyone is done reading this one
    WW_STAMP(3);
  }
#ifdef AVID_WW_TRACE
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) g_ww_trace[((blockIdx.x & 1023) * 8 + wave) * 8 + i] = ww_t[i];
    g_ww_trace[((blockIdx.x & 1023) * 8 + wave) * 8 + 6] = clock64() - ww_c0;            // shader cycles ...
    g_ww_trace[((blockIdx.x & 1023) * 8 + wave) * 8 + 7] = wall_clock64() - ww_wall0;    // ... per 10 ns ticks: the clock in the kernel
  }
#endif
}

__global__ __launch_bounds__(512) void wino_wgrad_kernel(const WinoWgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int Cr = p.Cr, Cn = p.Cn;
  // the (Cout block, Cin block) pairs of one tile range on ONE XCD: they read the same dy / x halves (in hardware order,
  // id % 8 = XCD, the four pairs of a 128-channel layer sat on four different L2s: every operand fetched twice)
  const int lid = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int pair = lid % p.pairs, split = lid / p.pairs;
  const int cbk = pair % p.ncb, nbk = pair / p.ncb;
  const int ch0 = split * p.cps, ch1 = min(ch0 + p.cps, p.nchunks);
  const int quad = wave & 3, e = wave >> 2, nh = quad >> 1, chh = quad & 1;
  floatx16 acc[8];
#pragma unroll
  for (int x8 = 0; x8 < 8; ++x8)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x8][r] = 0.f;
  if (wave < 4)
    wino_wgrad_body<0>(p, sm, acc, cbk, nbk, ch0, ch1);
  else
    wino_wgrad_body<1>(p, sm, acc, cbk, nbk, ch0, ch1);

  // ---- dw block = G^T dU G.  Columns (j) in registers: T[i][b],  G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
  // (the accumulators hold dU' = s_i s_j dU, s = [1,1,1,-1]: column 3 and, for the e = 1 waves, row 3 change sign)
  floatx16 T[2][3];
#pragma unroll
  for (int il = 0; il < 2; ++il) {
    const float si = (e == 1 && il == 1) ? -1.f : 1.f;
    const floatx16 d0 = si * acc[il * 4 + 0], d1 = si * acc[il * 4 + 1], d2 = si * acc[il * 4 + 2], d3 = -si * acc[il * 4 + 3];
    const floatx16 hs = 0.5f * (d1 + d2);
    T[il][0] = d0 + hs;
    T[il][1] = 0.5f * (d1 - d2);
    T[il][2] = hs + d3;
  }
  // rows (i): dw[0][b] = T0 + .5 T1 + .5 T2,  dw[1][b] = .5 T1 - .5 T2,  dw[2][b] = .5 T1 + .5 T2 + T3.
  // The e = 1 waves (rows 2, 3) hand X[b] = .5 T2[b] and Y[b] = .5 T2[b] + T3[b] to their e = 0 partner through LDS.
  float* ex = sm;                                 // ex[quad][plane 6][r / 4][lane][4]
  auto ex_at = [&](int plane, int r4) { return ex + ((((quad * 6 + plane) * 4 + r4) * 64 + lane) << 2); };
  if (e == 1) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const floatx16 X = 0.5f * T[0][b], Y = X + T[1][b];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        *reinterpret_cast<floatx4*>(ex_at(b, r4)) = floatx4{X[4 * r4], X[4 * r4 + 1], X[4 * r4 + 2], X[4 * r4 + 3]};
        *reinterpret_cast<floatx4*>(ex_at(3 + b, r4)) = floatx4{Y[4 * r4], Y[4 * r4 + 1], Y[4 * r4 + 2], Y[4 * r4 + 3]};
      }
    }
  }
  __syncthreads();
  if (e == 0) {
    float* o = p.out + (long long)split * Cn * 9 * Cr;
    const int c = cbk * 64 + chh * 32 + l31;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const floatx16 hT1 = 0.5f * T[1][b];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const floatx4 X = *reinterpret_cast<const floatx4*>(ex_at(b, r4));
        const floatx4 Y = *reinterpret_cast<const floatx4*>(ex_at(3 + b, r4));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * r4 + q;
          const int n = nbk * 64 + nh * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          float* row = o + ((long long)n * 9 + b) * Cr + c;
          row[0] = T[0][b][r] + hT1[r] + X[q];                 // tap (a = 0, b)
          row[3 * Cr] = hT1[r] - X[q];                         // tap (1, b)
          row[6 * Cr] = hT1[r] + Y[q];                         // tap (2, b)
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
static int wino_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

static int wino_cus() { return device_cus(); }

// The switches of this path: AVID_WINO / AVID_WINO_MIN_M / AVID_WINO_MAXC from the environment, unless
// avid_wino_configure() (include/avid_hip.h) has overridden them — tests force small fixtures through the kernel.
struct WinoCfg { int on; long long min_m; int max_c; bool loaded; int wg_on; long long wg_min_m; int wg_max_c; long long min_work, wg_min_work; };
static WinoCfg g_wino_cfg = {1, 6000, 256, false, 1, 6000, 256, 1000000, 1000000};
static int g_wino_override[3] = {-1, -1, -1};

static const WinoCfg& wino_cfg() {
  WinoCfg& c = g_wino_cfg;
  if (!c.loaded) {
    c.on = g_wino_override[0] >= 0 ? g_wino_override[0] : wino_env("AVID_WINO", 1);
    // (25088 pixels: 32 vs 33 us forward, 31 vs 39 us input gradient; 16000: 33.5 vs 28.6)
    // Where the fused kernel beats the direct one (tools/conv_bench.py, batch 64, round 3): conv2x / conv3x (64 / 128
    // channels, 401408 / 50176 pixels) 1.6x, conv4x (256 channels, 6272 pixels: 196 of 512 workgroup slots) 84 -> 69 us
    // forward and 91 -> 70 us input gradient, audio block 1 (64 channels, 16000 pixels) 28 -> 26 and 34 -> 27 us; it
    // loses on audio block 2 (128 channels, 4160 pixels: 31 -> 39 us), block 3 (256, 1344: 36 -> 66) and conv5x (512 channels,
    // 1024 pixels: 59 -> 120).  Rule: >= min_m pixels, <= max_c output channels, pixels x max(channels) >= min_work.
    c.min_m = g_wino_override[1] >= 0 ? g_wino_override[1] : 6000;
    c.max_c = g_wino_override[2] >= 0 ? g_wino_override[2] : 256;
    c.min_work = g_wino_override[1] >= 0 ? 0 : 1000000;
    // the weight-gradient kernel follows the same switches unless its own are set
    c.wg_on = c.on && wino_env("AVID_WINO_WGRAD", 1);
    // (weight gradient, same rule: conv4x 89 -> 67 us, audio block 1 35 -> 29 us per layer; the smaller layers ride in
    // the grouped launch at the same ~100 TFLOP/s the Winograd kernel would give them)
    c.wg_min_m = g_wino_override[1] >= 0 ? g_wino_override[1] : 6000;
    c.wg_max_c = g_wino_override[2] >= 0 ? g_wino_override[2] : 256;
    c.wg_min_work = g_wino_override[1] >= 0 ? 0 : 1000000;
    c.loaded = true;
  }
  return c;
}

// mode 0: forward (x -> y), mode 1: input gradient (dy -> dx)
bool wino_supported(const avid_conv_desc* d, int mode) {
  const WinoCfg& c = wino_cfg();
  if (!c.on || d->x_channel_first) return false;
  if (d->kt != 1 || d->kh != 3 || d->kw != 3 || d->st != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pt != 0 || d->ph != 1 || d->pw != 1) return false;
  const int Cr = mode ? d->Cout : d->Cin, Cn = mode ? d->Cin : d->Cout;
  if (Cr % W_CK || Cn % 64 || Cn > c.max_c) return false;
  const long long M = (long long)d->B * d->Ti * d->Hi * d->Wi;
  if (M < c.min_m || M * (Cr > Cn ? Cr : Cn) < c.min_work) return false;
  const long long big = M * (Cr > Cn ? Cr : Cn) * 4;
  return big < (1ll << 31);
}

size_t wino_ws_bytes(const avid_conv_desc* d, int mode) {   // U: 16 transform points x 4 bytes, or x 6 (three bf16 terms)
  (void)mode;
  return (size_t)96 * d->Cin * d->Cout;
}

// wino2_kernel (one workgroup per CU, 64-tile units) where the layer has at least 1.5 rounds of units for the CUs
// (avid_wino2_configure changes the bound)
static int g_wino2_override = -1;      // avid_wino2_configure

static bool wino_use_v2(const avid_conv_desc* d, int mode) {
  const int min_r10 = g_wino2_override >= 0 ? g_wino2_override : 15;
  const int Cn = mode ? d->Cin : d->Cout, ncb = Cn / 64;
  const long long TH = (d->Hi + 1) / 2, TW = (d->Wi + 1) / 2;
  const long long units = ceil_div((long long)d->B * d->Ti * TH * TW, W2_TB) * ncb;
  return units * 10 >= (long long)min_r10 * wino_cus();
}

int wino_variant(const avid_conv_desc* d, int mode) { return wino_use_v2(d, mode) ? 2 : 1; }

int wino_grid(const avid_conv_desc* d, int mode) {
  const int Cn = mode ? d->Cin : d->Cout, ncb = Cn / 64;
  // (the fewest workgroups with the same number of rounds — 224 instead of 256 for conv2x at batch 64 — was measured in round 5:
  //  every workgroup then has the full count, the start stagger of wino2_kernel has nobody to delay, the chip loads and stores in
  //  lockstep: layer alone 136 -> 142 us, the step unchanged (10.07 vs 10.06 ms).  Not kept.)
  if (wino_use_v2(d, mode)) return wino_cus() / ncb * ncb;
  int g = 2 * wino_cus();
  g = g / ncb * ncb;
  const long long TH = (d->Hi + 1) / 2, TW = (d->Wi + 1) / 2;
  const long long units = ceil_div((long long)d->B * d->Ti * TH * TW, W_TB) * ncb;
  if (units < g) g = (int)(ceil_div(units, ncb) * ncb);
  return g;
}

// AVID_WINO2_PRE (default 1): wino2p_kernel — V split once, by the transform (split-bf16 builds only)
static int g_wino2_pre_override = -1;   // avid_wino2_pre_configure
static bool wino2_pre() {
  static int on = -1;
  if (on < 0) on = wino_env("AVID_WINO2_PRE", 1) != 0;
  return W2_SPLIT && (g_wino2_pre_override >= 0 ? g_wino2_pre_override != 0 : on != 0);
}

template <int EPI>
static void wino2_launch(const WinoArgs& a, int grid, hipStream_t s) {
  if (wino2_pre()) {
    auto kp = wino2p_kernel<EPI>;
    static bool setp[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!setp[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, W2P_LDS_BYTES);
      setp[dev] = true;
    }
    hipLaunchKernelGGL(kp, dim3(grid), dim3(256), W2P_LDS_BYTES, s, a);
    return;
  }
  auto kern = wino2_kernel<EPI>;
  const size_t lds = sizeof(float) * W2_LDS_FLOATS;
  static bool set[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
}

template <int EPI>
static void wino_launch(const WinoArgs& a, int grid, hipStream_t s) {
  if (a.v2) { wino2_launch<EPI>(a, grid, s); return; }
  auto kern = wino_kernel<EPI>;
  const size_t lds = sizeof(float) * W_LDS_FLOATS + sizeof(int) * W_TAB_INTS;
  static bool set[64] = {false};           // the attribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
}

// src / dst are x / y (mode 0) or dy / dx (mode 1); ws holds U (wino_ws_bytes)
int wino_conv(const avid_conv_desc* d, int mode, const float* src, const float* w, const float* u_pre, float* dst,
              const float* addend, float* stats, const avid_bn_bwd_fuse* bn, void* ws, hipStream_t s) {
  WinoArgs a;
  memset(&a, 0, sizeof(a));
  a.src = src; a.dst = dst; a.addend = addend;
  a.F = d->B * d->Ti; a.H = d->Hi; a.W = d->Wi;
  a.TH = (d->Hi + 1) / 2; a.TW = (d->Wi + 1) / 2;
  a.Cr = mode ? d->Cout : d->Cin;
  a.Cn = mode ? d->Cin : d->Cout;
  a.ncb = a.Cn / 64;
  a.ntiles = (long long)a.F * a.TH * a.TW;
  a.v2 = wino_use_v2(d, mode) ? 1 : 0;
  a.units = (int)(ceil_div(a.ntiles, a.v2 ? W2_TB : W_TB) * a.ncb);
  {
    a.xcd_local = 1;
  }
  int rc = AVID_OK;
  if (u_pre) {            // transformed once per step by avid_weight_transpose_batched (mode 1 / 2 descriptors)
    a.U = u_pre;
  } else {
    float* U = static_cast<float*>(ws);
    a.U = U;
    const long long n = (long long)a.Cn * a.Cr;
    ScopedTimer t(s, "wino_weight_kernel", 0.0, 4.0 * n * 25);
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, w, U, a.Cn, a.Cr, d->Cin, mode,
                       a.v2 ? (W2_SPLIT ? 3 : 1) : 2);
    rc = check_launch("wino_weight");
    if (rc) return rc;
  }
  int epi = 0;
  if (mode == 0 && stats) { epi = 1; a.stats = stats; }
  if (addend) epi |= 2;
  if (mode == 1 && bn) {
    epi |= 4;
    a.bnb_x = bn->x; a.bnb_scale = bn->scale; a.bnb_shift = bn->shift; a.bnb_mean = bn->mean; a.bnb_invstd = bn->invstd;
    a.bnb_relu = bn->relu;
    a.stats = bn->partials;
  }
  const int grid = wino_grid(d, mode);
  const double M = (double)d->B * d->Ti * d->Hi * d->Wi;
  // flops: the multiply-adds of the in-image part of the tiles (M / 4 tiles of 16 products of [Cr] x [Cr x Cn]) — not the
  // direct form's 2.25x larger count, and not the padding of the 2 x 2 tiles that hang over an odd extent (7 x 7 frames
  // run 4 x 4 tiles: 31 % more matrix instructions than pixels / 4; round 4 counted those as achieved work);
  // bytes: source + destination (+ addend, + x of the BatchNorm-backward sums)
  static const char* kNames[24] = {"wino_kernel<0>", "wino_kernel<1>", "wino_kernel<2>", "wino_kernel<3>",
                                   "wino_kernel<4>", "wino_kernel<5>", "wino_kernel<6>", "wino_kernel<7>",
                                   "wino2_kernel<0>", "wino2_kernel<1>", "wino2_kernel<2>", "wino2_kernel<3>",
                                   "wino2_kernel<4>", "wino2_kernel<5>", "wino2_kernel<6>", "wino2_kernel<7>",
                                   "wino2p_kernel<0>", "wino2p_kernel<1>", "wino2p_kernel<2>", "wino2p_kernel<3>",
                                   "wino2p_kernel<4>", "wino2p_kernel<5>", "wino2p_kernel<6>", "wino2p_kernel<7>"};
  ScopedTimer t(s, kNames[(epi & 7) + 8 * (a.v2 ? (wino2_pre() ? 2 : 1) : 0)], 2.0 * 16.0 * (M / 4.0) * a.Cr * a.Cn,
                4.0 * M * (a.Cr + a.Cn * (1 + (addend ? 1 : 0) + ((epi & 4) ? 1 : 0))));
  switch (epi) {
    case 0: wino_launch<0>(a, grid, s); break;
    case 1: wino_launch<1>(a, grid, s); break;
    case 2: wino_launch<2>(a, grid, s); break;
    case 3: wino_launch<3>(a, grid, s); break;
    case 4: wino_launch<4>(a, grid, s); break;
    case 6: wino_launch<6>(a, grid, s); break;
    default: set_error("wino_conv: unsupported epilogue %d", epi); return AVID_E_UNSUPPORTED;
  }
  return check_launch("wino_conv");
}

// ---- weight gradient
bool wino_wgrad_supported(const avid_conv_desc* d) {
  const WinoCfg& c = wino_cfg();
  if (!c.wg_on || d->x_channel_first) return false;
  if (d->kt != 1 || d->kh != 3 || d->kw != 3 || d->st != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pt != 0 || d->ph != 1 || d->pw != 1) return false;
  if (d->Cin % 64 || d->Cout % 64 || d->Cin > c.wg_max_c || d->Cout > c.wg_max_c) return false;
  const long long M = (long long)d->B * d->Ti * d->Hi * d->Wi;
  if (M < c.wg_min_m || M * (d->Cin > d->Cout ? d->Cin : d->Cout) < c.wg_min_work) return false;
  return M * (d->Cin > d->Cout ? d->Cin : d->Cout) * 4 < (1ll << 31);
}

struct WinoWgradPlan { int pairs, nchunks, cps, nsplit; long long ntiles; };

static WinoWgradPlan wino_wgrad_plan(const avid_conv_desc* d) {
  WinoWgradPlan pl;
  const long long TH = (d->Hi + 1) / 2, TW = (d->Wi + 1) / 2;
  pl.ntiles = (long long)d->B * d->Ti * TH * TW;
  pl.nchunks = (int)ceil_div(pl.ntiles, WW_TK);
  pl.pairs = (d->Cin / 64) * (d->Cout / 64);
  int want = wino_cus() / pl.pairs;            // one workgroup per CU (139 KB of LDS each)
  if (want < 1) want = 1;
  if (want > pl.nchunks) want = pl.nchunks;
  pl.cps = (int)ceil_div(pl.nchunks, want);
  pl.nsplit = (int)ceil_div(pl.nchunks, pl.cps);
  return pl;
}

size_t wino_wgrad_ws_bytes(const avid_conv_desc* d) {
  const WinoWgradPlan pl = wino_wgrad_plan(d);
  return sizeof(float) * (size_t)pl.nsplit * d->Cout * 9 * d->Cin;
}

static void wino_magic(int dv, unsigned& magic, int& shift) {     // n / dv == (n * magic) >> shift for n < 2^31
  int l = 0;
  while ((1ll << l) < dv) ++l;
  shift = 31 + l;
  magic = (unsigned)(((1ull << shift) + (unsigned)dv - 1) / (unsigned)dv);
}

// slabs[nsplit][Cout][9][Cin] -> ws; *nsplit_out tells the caller how many to reduce (1: written straight into dw)
int wino_wgrad(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, int* nsplit_out,
               hipStream_t s) {
  const WinoWgradPlan pl = wino_wgrad_plan(d);
  WinoWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.dy = dy;
  a.out = pl.nsplit == 1 ? dw : static_cast<float*>(ws);
  a.F = d->B * d->Ti; a.H = d->Hi; a.W = d->Wi;
  a.TH = (d->Hi + 1) / 2; a.TW = (d->Wi + 1) / 2;
  a.Cr = d->Cin; a.Cn = d->Cout; a.ncb = d->Cin / 64;
  a.pairs = pl.pairs; a.nchunks = pl.nchunks; a.cps = pl.cps; a.ntiles = pl.ntiles;
  wino_magic(a.TH * a.TW, a.mgTPF, a.shTPF);
  wino_magic(a.TW, a.mgTW, a.shTW);
  const size_t lds = sizeof(float) * WW_LDS_FLOATS;
  static bool set[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    set[dev] = true;
  }
  const double M = (double)d->B * d->Ti * d->Hi * d->Wi;
  {
    // flops: the multiply-adds of the in-image part of the tiles (16 per tile and channel pair, M / 4 tiles: the tiles that
    // hang over an odd extent are executed, not counted); bytes: x + dy + dw once
    ScopedTimer t(s, "wino_wgrad_kernel", 2.0 * 16.0 * (M / 4.0) * d->Cin * d->Cout,
                  4.0 * (M * (d->Cin + d->Cout) + 9.0 * d->Cin * d->Cout));
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3((unsigned)(pl.pairs * pl.nsplit)), dim3(512), lds, s, a);
  }
  *nsplit_out = pl.nsplit;
  return check_launch("wino_wgrad");
}

}  // namespace avid

extern "C" int avid_wino2_pre_configure(int on) {
  avid::g_wino2_pre_override = on < 0 ? -1 : (on ? 1 : 0);
  return AVID_OK;
}

extern "C" int avid_wino2_configure(int min_rounds_x10) {
  avid::g_wino2_override = min_rounds_x10;
  return AVID_OK;
}

extern "C" int avid_wino_configure(int enabled, int64_t min_pixels, int max_channels) {
  avid::g_wino_override[0] = enabled < 0 ? -1 : (enabled ? 1 : 0);
  avid::g_wino_override[1] = min_pixels < 0 ? -1 : (int)(min_pixels > 0x7fffffff ? 0x7fffffff : min_pixels);
  avid::g_wino_override[2] = max_channels < 0 ? -1 : max_channels;
  avid::g_wino_cfg.loaded = false;
  return AVID_OK;
}
