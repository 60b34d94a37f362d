// Stem convolutions (models/video.py:20  Conv3d(3,64,(3,7,7),s(1,2,2),p(1,3,3));
//                    models/audio.py:22  Conv2d(1,64,7,s2,p3)) as a DIRECT convolution from an LDS patch.
//
// With Cin = 3 / 1 an im2col row is 441 / 49 scattered scalars, so the generic gather kernel was
// load-issue bound (61 TF fwd / 46 TF wgrad).  Here a workgroup (8 waves) owns 256 consecutive output
// pixels of one (b, t) frame, stages the input patch they touch (<= 19 rows x (W + 6) cols x Cin*kt
// planes, zero padded, read once with coalesced row loads from the reference's channel-first tensor)
// in LDS, and feeds the fp32 MFMA straight from it:
//     A[pixel i][k'] = patch[plane(c,dt)][2*(ho_i - ho_lo) + dh][2*wo_i + dw]
// so a lane reads patch[base_i + dh*PW + dw] — one ds_read_b32 per MFMA pair, no address tables.
// wgrad:   k' = (c*kt + dt)*49 + dh*7 + dw   (dense: k' is the GEMM-N dimension; 441 of 448; the audio stem keeps
//          the older ((c*kt + dt)*7 + dh)*8 + dw with dw padded 7 -> 8)
// forward: k' = (c*kt + dt)*50 + dh*7 + dw    (the 49 taps of a plane flattened, padded to 50: 441 of 450 useful);
//          weights repacked once per call to Wt[k'][64] and streamed through a double-buffered LDS stage, one
//          plane (25 k-steps) per chunk.
#include "common.h"

#include <stdlib.h>

namespace avid {

struct StemArgs {
  const float* __restrict__ x;    // [B][CIN][Ti][Hi][Wi]
  const float* __restrict__ wt;   // [R*8][64]  repacked weights
  float* __restrict__ y;          // [B][To][Ho][Wo][64]
  const float* __restrict__ w;    // [64][kt][7][7][CIN]  (for wgrad / repack)
  const float* __restrict__ dy;   // wgrad: [B][To][Ho][Wo][64]
  float* __restrict__ part;       // wgrad partials [G][64][R*8]
  float* __restrict__ stats;      // forward: BatchNorm partial sums of y, one row [2][64] per workgroup, or null
  int B, Ti, Hi, Wi, Ho, Wo;
  int PW, rows_in_max, tiles_per_frame;
  int patch_plane_bytes;        // stem_fwd3p_kernel: bytes of one bf16 plane of the patch
  int tile_px;                    // wgrad: output pixels per tile (whole rows when a row fits, <= STEM_TILE)
  int ntiles;                     // B*Ti*tiles_per_frame
  int xcd_local;                  // XCD-contiguous tile order
};

constexpr int STEM_TILE = 256;    // output pixels per workgroup tile
// LDS rows of 64 floats read as MFMA operands by (lane & 31 = column, lane >> 5 = row parity): the two 32-column
// halves of ODD rows are stored swapped (column ^ 32), so the half-waves land on disjoint banks without padding
// (with a 68-float pitch the odd-row half-wave overlapped 28 of the even row's 32 banks: 21-28 % of the LDS
// cycles of the stem kernels were bank conflicts).
constexpr int WS_LD = 64;

// Wt[k'][n] = w[n][dt][dh][dw][c]   (k' = ((c*KT+dt)*7+dh)*8+dw ; dw == 7 -> 0)
// Forward weight layout: Wt[plane = c*KT+dt][f][n] with f = dh*7 + dw flattened to 50 per plane (f = 49 -> 0):
// 49 of 50 k' are useful (the 8-per-kernel-row layout of the wgrad kernel wastes 1 in 8).
constexpr int FK = 50;
template <int CIN, int KT>
__global__ void stem_repack_kernel(const float* __restrict__ w, float* __restrict__ wt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CIN * KT * FK * 64) return;
  const int n = i & 63, kp = i >> 6;
  const int f = kp % FK, pl = kp / FK;
  const int dh = f / 7, dw = f - dh * 7;
  const int dt = pl % KT, c = pl / KT;
  wt[i] = f < 49 ? w[(((n * KT + dt) * 7 + dh) * 7 + dw) * CIN + c] : 0.f;
}

// patch[plane = c*KT+dt][row][col] ; row 0 <-> hi = 2*ho_lo - 3 ; col 0 <-> wi = -4 (4-float left pad keeps
// the 16-B row copies aligned on both sides; PW % 4 == 0).  One float4 per (row, 4 cols) work item.
template <int CIN, int KT>
__device__ __forceinline__ void stem_load_patch(const StemArgs& p, float* P, int b, int to, int ho_lo, int nrows_in) {
  const int q4 = p.PW >> 2;                         // float4 slots per patch row
  const int total = CIN * KT * nrows_in * q4;
  const bool vec = (p.Wi & 3) == 0;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int cq = e % q4;
    int r = e / q4;
    const int row = r % nrows_in;
    const int pl = r / nrows_in;
    const int dt = pl % KT, c = pl / KT;
    const int ti = to + dt - KT / 2, hi = 2 * ho_lo - 3 + row, wi = cq * 4 - 4;
    const bool rok = ((unsigned)ti < (unsigned)p.Ti) & ((unsigned)hi < (unsigned)p.Hi);
    const long long rbase = ((((long long)b * CIN + c) * p.Ti + (rok ? ti : 0)) * p.Hi + (rok ? hi : 0)) * p.Wi;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    if (rok && wi >= 0 && wi + 3 < p.Wi && vec) {
      v = *reinterpret_cast<const floatx4*>(p.x + rbase + wi);
    } else if (rok) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((unsigned)(wi + j) < (unsigned)p.Wi) v[j] = p.x[rbase + wi + j];
    }
    *reinterpret_cast<floatx4*>(P + (long long)r * p.PW + cq * 4) = v;
  }
}


// Persistent: 2 workgroups per CU walk the 128-pixel tiles; the next tile's patch is fetched into registers
// under the current tile's MFMAs and written to LDS between tiles (with the loads between the tiles the
// kernel took 1.10 ms, of which 0.28 ms were the loads and only ~0.1 ms of that hidden by the co-resident
// workgroup).  The weight chunks cycle through their two LDS stages across tile boundaries.
// WAVES = 4: 128-pixel tiles, two workgroups per CU (<= 80 KB of LDS each); WAVES = 8: 256-pixel tiles, one
// workgroup per CU — the form the 224 x 224 inputs of the shipped configs need (their 13-row patch is 109 KB).
template <int CIN, int KT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 8 / WAVES) void stem_fwd_kernel(const StemArgs p) {
  constexpr int NT = WAVES * 64, TILE = WAVES * 32;
  constexpr int NCH = CIN * KT;               // weight chunks = input planes (c, dt): 50 k' (25 k-steps) each
  constexpr int PIT = WAVES == 4 ? 14 : 16;   // float4 patch items per thread (the LDS left beside the weight stages bounds it)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                           // [2][FK][WS_LD]
  float* P = smem + 2 * FK * WS_LD;           // patch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int npix = p.Ho * p.Wo;
  const int q4 = p.PW >> 2;
  const long long item_floats = (long long)CIN * p.Ti * p.Hi * p.Wi;

  struct TileGeo { int frame, to, b, p0, p1, ho_lo, nrows_in; };
  auto geo = [&](int tile) {
    TileGeo g;
    const int tf = tile % p.tiles_per_frame;
    g.frame = tile / p.tiles_per_frame;
    g.to = g.frame % p.Ti;
    g.b = g.frame / p.Ti;
    g.p0 = tf * TILE;
    g.p1 = min(g.p0 + TILE, npix);
    g.ho_lo = g.p0 / p.Wo;
    g.nrows_in = 2 * ((g.p1 - 1) / p.Wo - g.ho_lo) + 7;
    return g;
  };
  // patch[plane = c*KT+dt][row][col]: row 0 <-> hi = 2*ho_lo - 3, col 0 <-> wi = -4; PW = 4*q4, so work item e
  // (one float4, entirely inside or outside its row since Wi % 4 == 0) lands at LDS float 4*e
  floatx4 pre_p[PIT];
  const unsigned mgq = 0xffffffffu / (unsigned)q4 + 1u;     // e / q4 == mulhi(e, mgq) for the e < 2^12 used here
  auto prefetch = [&](const TileGeo& g) {
    const int total = CIN * KT * g.nrows_in * q4;
    const unsigned mgn = 0xffffffffu / (unsigned)g.nrows_in + 1u;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (long long)g.b * item_floats), 0, (int)(item_floats * 4), 0x00020000);
    // The item -> (plane, row, column) decode is recomputed per tile from an opaque copy of tid (~20 VALU per
    // item): left to the compiler its tile-invariant parts are hoisted into 30+ long-lived registers and spill.
    int t0 = tid;
    asm volatile("" : "+v"(t0));
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = t0 + it * NT;
      const int r = (int)__umulhi((unsigned)e, mgq), cq = e - r * q4;
      const int pl = (int)__umulhi((unsigned)r, mgn), row = r - pl * g.nrows_in;
      const int dt = pl % KT, c = pl / KT;
      const int ti = g.to + dt - KT / 2, hi = 2 * g.ho_lo - 3 + row, wi = cq * 4 - 4;
      const bool ok = (e < total) & ((unsigned)ti < (unsigned)p.Ti) & ((unsigned)hi < (unsigned)p.Hi) &
                      ((unsigned)wi < (unsigned)p.Wi);
      const unsigned off = (unsigned)(((c * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * 4u;
      pre_p[it] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : 0xfffffff0u, 0, 0));
    }
  };
  auto commit = [&](const TileGeo& g) {
    const int total = CIN * KT * g.nrows_in * q4;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + it * NT;
      if (e < total) *reinterpret_cast<floatx4*>(P + 4 * e) = pre_p[it];
    }
  };

  // weight chunk: FK rows x 16 float4 = 800 float4 -> up to WPT per thread
  constexpr int WPT = (FK * 16 + NT - 1) / NT;
  floatx4 wv[WPT];
  auto load_w = [&](int ch) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + NT * i;
      const floatx4 z = {0.f, 0.f, 0.f, 0.f};
      wv[i] = idx < FK * 16 ? *reinterpret_cast<const floatx4*>(p.wt + ((long long)ch * FK * 16 + idx) * 4) : z;
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + NT * i, row = idx >> 4, col = (idx & 15) * 4;
      if (idx < FK * 16)
        *reinterpret_cast<floatx4*>(&Ws[buf * FK * WS_LD + row * WS_LD + (col ^ ((row & 1) << 5))]) = wv[i];
    }
  };

  // tiles in XCD-contiguous order (round 3: time unchanged, weight-gradient traffic -1 %): consecutive tiles are neighbouring row bands of a frame and
  // neighbouring frames, whose input patches overlap (6 of 19 rows, two of three planes) — dealt in hardware order (id %
  // 8 = XCD) the overlap is fetched by eight different L2s
  int tile = p.xcd_local ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (tile >= p.ntiles) return;
  int u = 0;                                  // LDS stage of the weight chunk about to be multiplied
  float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};   // this lane's running column sums / sums of squares of y
  load_w(0);
  prefetch(geo(tile));
  store_w(0);
  for (; tile < p.ntiles; tile += gridDim.x) {
    const TileGeo g = geo(tile);
    const int p0 = g.p0, p1 = g.p1, ho_lo = g.ho_lo;
    __syncthreads();                          // the previous tile's patch reads are done
    commit(g);
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) prefetch(geo(tile + gridDim.x));

    // this lane's pixel; patch col of tap dw = 2*wo + dw + 1 (4-float left pad, conv pad 3)
    const int pi = p0 + wave * 32 + l31;
    const bool pok = pi < p1;
    const int ho = (pok ? pi : p0) / p.Wo, wo = (pok ? pi : p0) - ho * p.Wo;
    const int base = (2 * (ho - ho_lo)) * p.PW + 2 * wo + 1 + h;
    const int plane = g.nrows_in * p.PW;

    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Per plane: 25 k-steps over the 7 x 7 taps flattened (f = dh*7 + dw; step q takes f = 2q + h).  The lane's
    // address is base0 + dh*PW + dw: both halves share dh except when f = 2q is the last tap of a row (dw = 6),
    // where the h = 1 half starts the next row — two per-lane offsets (lo / lw) cover the two cases, dh*PW + dw
    // of the even tap is uniform.  The last step pairs tap 48 with the zero weight row 49 (both halves read tap 48).
    const int base0 = base - h;
    const int lo = h, lw = h ? p.PW - 6 : 0;
    for (int ch = 0; ch < NCH; ++ch, u ^= 1) {
      load_w(ch + 1 < NCH ? ch + 1 : 0);      // cyclic: the last chunk of a tile fetches chunk 0 of the next
      const float* Wb = Ws + u * FK * WS_LD + l31;
      const float* Pp = P + ch * plane + base0;
      // operands of the next group of 5 k-steps are fetched before the 10 MFMAs of the current one issue (pinned
      // with sched_barrier: the compiler otherwise puts every ds_read right in front of its MFMA)
      float af[2][5], b0f[2][5], b1f[2][5];
      auto frag = [&](int grp, int buf) {
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
          const int q = grp * 5 + s5, f0 = 2 * q;
          const int dh = f0 / 7, dw = f0 - dh * 7;          // compile-time after unrolling
          const int lane_off = q == 24 ? 0 : (dw == 6 ? lw : lo);
          af[buf][s5] = Pp[dh * p.PW + dw + lane_off];
          b0f[buf][s5] = Wb[(2 * q + h) * WS_LD + 32 * h];          // row parity == h: swapped halves
          b1f[buf][s5] = Wb[(2 * q + h) * WS_LD + 32 - 32 * h];
        }
      };
      frag(0, 0);
#pragma unroll
      for (int grp = 0; grp < 5; ++grp) {
        if (grp + 1 < 5) frag(grp + 1, (grp + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[grp & 1][s5], b0f[grp & 1][s5], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[grp & 1][s5], b1f[grp & 1][s5], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      store_w(u ^ 1);
      __syncthreads();
    }

    // epilogue: row = (r&3) + 8*(r>>2) + 4*h within the wave's 32 pixels, col = l31 (+32)
    const long long m_base = (long long)g.frame * npix + p0 + wave * 32;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (p0 + wave * 32 + rr < p1) {
          const float v = acc[j][r];
          p.y[(m_base + rr) * 64 + j * 32 + l31] = v;
          cs[j] += v;                       // BatchNorm statistics of the output ride along (p.stats)
          cq[j] = fmaf(v, v, cq[j]);
        }
      }
  }
  if (p.stats) {   // one partial row [2][64] per workgroup: fold the half-waves, then the waves in fixed order
    float* red = smem;                      // [2][WAVES][64]; the weight stages are dead
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a = cs[j] + __shfl_xor(cs[j], 32, 64), b = cq[j] + __shfl_xor(cq[j], 32, 64);
      if (h == 0) {
        red[wave * 64 + j * 32 + l31] = a;
        red[WAVES * 64 + wave * 64 + j * 32 + l31] = b;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, q = tid >> 6;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) t += red[q * WAVES * 64 + w * 64 + c];
      p.stats[(long long)blockIdx.x * 128 + q * 64 + c] = t;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Round 4: the video stem's forward on the bf16 matrix instruction with fp32 accuracy ("bf16x3").
// Every fp32 operand is split into three bf16 terms (round-to-nearest each: together the fp32 value to its last bit) and
// a product is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid;
// the dropped terms are <= 2^-24 of the product; measured error against fp64 BELOW the fp32 instruction's, whose
// accumulator rounds once per term instead of once per sixteen) — 6/16 of the issue time of v_mfma_f32_32x32x2_f32.
// The stem is the layer where that pays at once: K = 441 per output from a patch that sits in LDS, so there is no
// operand traffic to speak of per matrix instruction (tools/bf16x3_lab.hip: on the K = 192 layers the loads, not the
// products, set the time).
//
// Same tile geometry, patch loader (fp32 patch in LDS), tile order and epilogue as stem_fwd_kernel; different inner loop:
//   k' rows r = (c*KT + dt)*7 + dh, r < R = CIN*KT*7; one k-step of 16 = rows (2s, 2s + 1) x 8 tap slots (dw = 7 -> zero
//   weight); the half-wave g = lane >> 5 takes the taps of ITS parity, dw = g + 2j: consecutive pixels read patch columns
//   two apart and the halves read neighbouring columns — every LDS bank once, as in the fp32 kernel;
//   lane (pixel i, g): 8 patch floats (4 of row 2s, 4 of row 2s + 1) -> split in registers (44 VALU) -> 3 fragments;
//   weights: split ONCE per call into fragment order Wf[step][column tile][hi | mid | lo][lane][8 bf16] and streamed
//   through a double-buffered LDS stage four steps (24 KB) at a time.
// 8 waves, 256-pixel tiles, one workgroup per CU (two waves per SIMD: the split of one hides in the other's products).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned uintx4_t __attribute__((ext_vector_type(4)));
constexpr int S3_STEP_BYTES = 2 * 3 * 1024;     // weight fragments of one k-step: 2 column tiles x (hi | mid | lo) x 1 KB

__device__ __forceinline__ void s3_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  split2_bf16_asm(x0, x1, h, m, l);
}

// Wf[s][tile][plane][lane][e]: element e of lane (j = lane & 31, g = lane >> 5) = w[n = 32 tile + j][row 2s + (e >> 2)][dw = g + 2 (e & 3)]
template <int CIN, int KT>
__global__ void stem_split_weights_kernel(const float* __restrict__ w, uintx4_t* __restrict__ Wf) {
  constexpr int R = CIN * KT * 7;
  const int frag = blockIdx.x, lane = threadIdx.x;
  const int s = frag >> 1, tile = frag & 1, n = 32 * tile + (lane & 31), g = lane >> 5;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = 2 * s + (e >> 2), dw = g + 2 * (e & 3);
    const int pl = r / 7, dh = r - pl * 7, dt = pl % KT, c = pl / KT;
    v[e] = (r < R && dw < 7) ? w[(((n * KT + dt) * 7 + dh) * 7 + dw) * CIN + c] : 0.f;
  }
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s3_split2(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
  uintx4_t* dst = Wf + ((size_t)frag * 3) * 64 + lane;
  dst[0] = uintx4_t{h[0], h[1], h[2], h[3]};
  dst[64] = uintx4_t{m[0], m[1], m[2], m[3]};
  dst[128] = uintx4_t{l[0], l[1], l[2], l[3]};
}

// (Splitting the patch ONCE at commit time into three 16-bit planes and assembling fragments from 16-bit LDS loads was
// measured too: 24 ds_read_u16 + 12 v_perm per k-step instead of 4 ds_read2_b32 + 44 VALU — 735 us against 661: the LDS
// instruction rate binds sooner than the vector ALU.)
// Measured around this kernel (64 clips of 8 x 112 x 112: 661-700 us whatever the instruction mix — three ways of getting
// the split terms into registers, two workgroup shapes, six-deep or three-deep accumulation chains): (a) the patch split
// ONCE at commit time into three 16-bit planes, fragments from 24 ds_read_u16 + 12 v_perm per k-step: 735 us; (b) a dword
// plane (hi << 16 | mid) + a 16-bit plane of lo, 4 ds_read2_b32 + 8 ds_read_u16 + 12 v_perm: 667; (c) this form, fp32 patch,
// 4 ds_read2_b32 + 44 VALU per k-step: 661-680; (d) all chunks unrolled (compile-time patch offsets): 689; (e) 4 waves x
// 128-pixel tiles x two workgroups per CU: 701.  One product instead of six: 318 us, three: 470 — 242 us of tile
// overhead + 71 us per product where the pipe needs 51 at the 1.99 GHz the chip holds under this kernel (2.29 under the
// fp32 one).  What does not move is the LDS traffic of the weight fragments: every wave reads all 64 columns' three planes,
// 6 KB per k-step and wave = 16 bytes per cycle and wave at the pipe's full rate — eight waves ask for the LDS's whole
// 128 bytes per cycle.  A 64 x 64 wave tile would halve that and needs a 512-pixel patch (117 KB) beside the weight stages.
template <int CIN, int KT>
__global__ __launch_bounds__(512, 1) void stem_fwd3_kernel(const StemArgs p) {
  constexpr int WAVES = 8, NT = WAVES * 64, TILE = WAVES * 32;
  constexpr int S3_Q = 4;                     // k-steps per weight chunk (24 KB: 3 b128 items per thread)
  constexpr int S3_CHUNK = S3_Q * S3_STEP_BYTES;
  constexpr int R = CIN * KT * 7, NS = (R + 1) / 2, NCH = (NS + S3_Q - 1) / S3_Q;
  constexpr int PIT = 16;                     // float4 patch items per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Wl = reinterpret_cast<char*>(smem);                 // [2][S3_CHUNK]
  float* P = smem + 2 * S3_CHUNK / 4;                       // patch (fp32)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, l31 = lane & 31;
  const int npix = p.Ho * p.Wo;
  const int q4 = p.PW >> 2;
  const long long item_floats = (long long)CIN * p.Ti * p.Hi * p.Wi;

  struct TileGeo { int frame, to, b, p0, p1, ho_lo, nrows_in; };
  auto geo = [&](int tile) {
    TileGeo t;
    const int tf = tile % p.tiles_per_frame;
    t.frame = tile / p.tiles_per_frame;
    t.to = t.frame % p.Ti;
    t.b = t.frame / p.Ti;
    t.p0 = tf * TILE;
    t.p1 = min(t.p0 + TILE, npix);
    t.ho_lo = t.p0 / p.Wo;
    t.nrows_in = 2 * ((t.p1 - 1) / p.Wo - t.ho_lo) + 7;
    return t;
  };
  floatx4 pre_p[PIT];
  const unsigned mgq = 0xffffffffu / (unsigned)q4 + 1u;
  auto prefetch = [&](const TileGeo& t) {
    const int total = CIN * KT * t.nrows_in * q4;
    const unsigned mgn = 0xffffffffu / (unsigned)t.nrows_in + 1u;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (long long)t.b * item_floats), 0, (int)(item_floats * 4), 0x00020000);
    int t0 = tid;
    asm volatile("" : "+v"(t0));
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = t0 + it * NT;
      const int r = (int)__umulhi((unsigned)e, mgq), cq = e - r * q4;
      const int pl = (int)__umulhi((unsigned)r, mgn), row = r - pl * t.nrows_in;
      const int dt = pl % KT, c = pl / KT;
      const int ti = t.to + dt - KT / 2, hi = 2 * t.ho_lo - 3 + row, wi = cq * 4 - 4;
      const bool ok = (e < total) & ((unsigned)ti < (unsigned)p.Ti) & ((unsigned)hi < (unsigned)p.Hi) &
                      ((unsigned)wi < (unsigned)p.Wi);
      const unsigned off = (unsigned)(((c * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * 4u;
      pre_p[it] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : 0xfffffff0u, 0, 0));
    }
  };
  auto commit = [&](const TileGeo& t) {
    const int total = CIN * KT * t.nrows_in * q4;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + it * NT;
      if (e < total) *reinterpret_cast<floatx4*>(P + 4 * e) = pre_p[it];
    }
  };
  // weight chunk: S3_CHUNK / 16 = 1536 b128 items -> 3 per thread
  uintx4_t wv[3];
  const uintx4_t* Wf = reinterpret_cast<const uintx4_t*>(p.wt);
  auto load_w = [&](int ch) {
#pragma unroll
    for (int i = 0; i < 3; ++i) wv[i] = Wf[(size_t)ch * (S3_CHUNK / 16) + tid + NT * i];
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<uintx4_t*>(Wl + buf * S3_CHUNK + (tid + NT * i) * 16) = wv[i];
  };

  int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
  if (tile >= p.ntiles) return;
  int u = 0;
  float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};
  load_w(0);
  prefetch(geo(tile));
  store_w(0);
  for (; tile < p.ntiles; tile += gridDim.x) {
    const TileGeo t = geo(tile);
    const int p0 = t.p0, p1 = t.p1, ho_lo = t.ho_lo;
    __syncthreads();
    commit(t);
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) prefetch(geo(tile + gridDim.x));

    const int pi = p0 + wave * 32 + l31;
    const bool pok = pi < p1;
    const int ho = (pok ? pi : p0) / p.Wo, wo = (pok ? pi : p0) - ho * p.Wo;
    const int pb = (2 * (ho - ho_lo)) * p.PW + 2 * wo + 1 + g;             // tap dw = g + 2 j of row offset 0
    const float* Pb = P + pb;
    const int plane = t.nrows_in * p.PW;

    floatx16 acc[2], cor[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; cor[j][r] = 0.f; }

    // (unrolling all NCH chunks — compile-time (plane, row) per k-step instead of 3.5 scalar instructions per matrix
    //  instruction — was measured slower: 689 us against 661)
    for (int ch = 0; ch < NCH; ++ch, u ^= 1) {
      load_w(ch + 1 < NCH ? ch + 1 : 0);
      const char* Wb = Wl + u * S3_CHUNK + lane * 16;
#pragma unroll
      for (int q = 0; q < S3_Q; ++q) {
        const int s = ch * S3_Q + q;
        if (s < NS) {
          // rows 2s, 2s + 1 (the pad row past R reads row R - 1: its weights are zero, its data finite)
          const int r0 = 2 * s, r1 = (2 * s + 1 < R) ? 2 * s + 1 : R - 1;
          const int pl0 = r0 / 7, pl1 = r1 / 7;
          const int o0 = pl0 * plane + (r0 - pl0 * 7) * p.PW, o1 = pl1 * plane + (r1 - pl1 * 7) * p.PW;
          unsigned h[4], m[4], l[4];
          const float* A0 = Pb + o0;
          const float* A1 = Pb + o1;
          s3_split2(A0[0], A0[2], h[0], m[0], l[0]);
          s3_split2(A0[4], A0[6], h[1], m[1], l[1]);
          s3_split2(A1[0], A1[2], h[2], m[2], l[2]);
          s3_split2(A1[4], A1[6], h[3], m[3], l[3]);
          const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, uintx4_t{h[0], h[1], h[2], h[3]});
          const bf16x8_t am = __builtin_bit_cast(bf16x8_t, uintx4_t{m[0], m[1], m[2], m[3]});
          const bf16x8_t al = __builtin_bit_cast(bf16x8_t, uintx4_t{l[0], l[1], l[2], l[3]});
          // six products per column tile; the five correction products go to their own accumulators: four independent
          // accumulation chains per wave (a chain of six back-to-back dependent instructions ran at 2/3 of the pipe's
          // rate), and the small terms are summed among themselves before they meet the large one
          bf16x8_t bh[2], bm[2], bl[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const char* bp = Wb + ((q * 2 + j) * 3) * 1024;
            bh[j] = *reinterpret_cast<const bf16x8_t*>(bp);
            bm[j] = *reinterpret_cast<const bf16x8_t*>(bp + 1024);
            bl[j] = *reinterpret_cast<const bf16x8_t*>(bp + 2048);
          }
          cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[0], cor[0], 0, 0, 0);
          cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[1], cor[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[1], acc[1], 0, 0, 0);
          cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[0], cor[0], 0, 0, 0);
          cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[1], cor[1], 0, 0, 0);
          cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[0], cor[0], 0, 0, 0);
          cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[1], cor[1], 0, 0, 0);
          cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[0], cor[0], 0, 0, 0);
          cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[1], cor[1], 0, 0, 0);
          cor[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[0], cor[0], 0, 0, 0);
          cor[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[1], cor[1], 0, 0, 0);
        }
      }
      store_w(u ^ 1);
      __syncthreads();
    }

    const long long m_base = (long long)t.frame * npix + p0 + wave * 32;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * g;
        if (p0 + wave * 32 + rr < p1) {
          const float v = acc[j][r] + cor[j][r];
          p.y[(m_base + rr) * 64 + j * 32 + l31] = v;
          cs[j] += v;
          cq[j] = fmaf(v, v, cq[j]);
        }
      }
  }
  if (p.stats) {
    float* red = smem;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a = cs[j] + __shfl_xor(cs[j], 32, 64), b = cq[j] + __shfl_xor(cq[j], 32, 64);
      if (g == 0) {
        red[wave * 64 + j * 32 + l31] = a;
        red[WAVES * 64 + wave * 64 + j * 32 + l31] = b;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, q = tid >> 6;
      float tt = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tt += red[q * WAVES * 64 + w * 64 + c];
      p.stats[(long long)blockIdx.x * 128 + q * 64 + c] = tt;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// stem_fwd3p_kernel (round 5, late): stem_fwd3_kernel with the patch split ONCE, when it is committed to LDS, and a k-step laid
// out so that a lane's fragment is CONTIGUOUS in the split planes.  stem_fwd3_kernel's lane takes the taps of its parity from two
// rows (stride-2 gathers: fine for 4-byte elements, and the reason variant (a) above needed 24 ds_read_u16 per k-step); here the
// half-wave g takes ROW 2s + g and its eight k are the eight consecutive patch columns 2 wo .. 2 wo + 7 of pixel wo = taps
// dw = -1 .. 6, the first one a pad slot with zero weight (the window starts at an EVEN column: 4-byte aligned in a 2-byte
// plane).  The patch lives in LDS as three bf16 planes [hi | mid | lo][(c, dt)][row][PW]; a fragment is 16 contiguous bytes per
// plane = two ds_read2_b32, the term's operand register as it is: 6 LDS reads and NO vector instruction per k-step where
// stem_fwd3_kernel has 4 reads + 44 vector instructions (1408 per tile and wave: the "tile overhead" that did not scale with the
// number of products).  The commit splits: a thread's float4 item = two pair splits + three 8-byte stores (14 vector instructions
// per item, 16 items per tile).  Every patch element used to be split by each of its ~12 users.  LDS: 6 bytes per patch element
// (110 KB at 112 x 112 input) + the two 24 KB weight stages = 156 KB; larger inputs keep stem_fwd3_kernel.
// Same products in the same order per (row, tap); the assignment of (row, tap) pairs to the matrix instruction's k index differs
// from stem_fwd3_kernel's, so sums agree to rounding, not bit for bit.
// ------------------------------------------------------------------------------------------------
typedef unsigned uintx2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// Wf[s][tile][plane][lane][e]: element e of lane (j = lane & 31, g = lane >> 5) = w[n = 32 tile + j][row 2s + g][dw = e - 1]  (e = 0: 0)
template <int CIN, int KT>
__global__ void stem_split_weights_rows_kernel(const float* __restrict__ w, uintx4_t* __restrict__ Wf) {
  constexpr int R = CIN * KT * 7;
  const int frag = blockIdx.x, lane = threadIdx.x;
  const int s = frag >> 1, tile = frag & 1, n = 32 * tile + (lane & 31), g = lane >> 5;
  const int r = 2 * s + g;
  const int pl = r / 7, dh = r - pl * 7, dt = pl % KT, c = pl / KT;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (r < R && e > 0) ? w[(((n * KT + dt) * 7 + dh) * 7 + (e - 1)) * CIN + c] : 0.f;
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s3_split2(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
  uintx4_t* dst = Wf + ((size_t)frag * 3) * 64 + lane;
  dst[0] = uintx4_t{h[0], h[1], h[2], h[3]};
  dst[64] = uintx4_t{m[0], m[1], m[2], m[3]};
  dst[128] = uintx4_t{l[0], l[1], l[2], l[3]};
}

// TM = 32-pixel row blocks per wave: 1 (default) = eight waves of 32 pixels (every wave reads all weight fragments: 6 KB of LDS
// reads per 12 matrix instructions), 2 = FOUR waves of 64 pixels, one per SIMD: a weight fragment feeds two row blocks (6 KB per
// 24), eight independent accumulation chains per wave.  Same products, same order per accumulator: the outputs are bit-identical;
// measured slower (stem_fwd3p_tm below), kept as a switch.
template <int CIN, int KT, int TM>
__global__ __launch_bounds__(512 / TM, 1) void stem_fwd3p_kernel(const StemArgs p) {
  constexpr int WAVES = 8 / TM, NT = WAVES * 64, TILE = 256;
  constexpr int S3_Q = 4;                     // k-steps per weight chunk (24 KB: 3 b128 items per thread)
  constexpr int S3_CHUNK = S3_Q * S3_STEP_BYTES;
  constexpr int R = CIN * KT * 7, NS = (R + 1) / 2, NCH = (NS + S3_Q - 1) / S3_Q;
  constexpr int PIT = TM == 1 ? 10 : 19;      // float4 patch items per thread: >= 4864 items = the 19 456 floats the host admits
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Wl = reinterpret_cast<char*>(smem);                 // [2][S3_CHUNK]
  char* P = Wl + 2 * S3_CHUNK;                              // patch: [hi | mid | lo][plane_bytes] of bf16
  const int plane_bytes = p.patch_plane_bytes;              // 2 bytes per patch element, rounded up to 16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, l31 = lane & 31;
  const int npix = p.Ho * p.Wo;
  const int q4 = p.PW >> 2;
  const long long item_floats = (long long)CIN * p.Ti * p.Hi * p.Wi;

  struct TileGeo { int frame, to, b, p0, p1, ho_lo, nrows_in; };
  auto geo = [&](int tile) {
    TileGeo t;
    const int tf = tile % p.tiles_per_frame;
    t.frame = tile / p.tiles_per_frame;
    t.to = t.frame % p.Ti;
    t.b = t.frame / p.Ti;
    t.p0 = tf * TILE;
    t.p1 = min(t.p0 + TILE, npix);
    t.ho_lo = t.p0 / p.Wo;
    t.nrows_in = 2 * ((t.p1 - 1) / p.Wo - t.ho_lo) + 7;
    return t;
  };
  floatx4 pre_p[PIT];
  const unsigned mgq = 0xffffffffu / (unsigned)q4 + 1u;
  auto prefetch = [&](const TileGeo& t) {
    const int total = CIN * KT * t.nrows_in * q4;
    const unsigned mgn = 0xffffffffu / (unsigned)t.nrows_in + 1u;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (long long)t.b * item_floats), 0, (int)(item_floats * 4), 0x00020000);
    int t0 = tid;
    asm volatile("" : "+v"(t0));
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = t0 + it * NT;
      const int r = (int)__umulhi((unsigned)e, mgq), cq = e - r * q4;
      const int pl = (int)__umulhi((unsigned)r, mgn), row = r - pl * t.nrows_in;
      const int dt = pl % KT, c = pl / KT;
      const int ti = t.to + dt - KT / 2, hi = 2 * t.ho_lo - 3 + row, wi = cq * 4 - 4;
      const bool ok = (e < total) & ((unsigned)ti < (unsigned)p.Ti) & ((unsigned)hi < (unsigned)p.Hi) &
                      ((unsigned)wi < (unsigned)p.Wi);
      const unsigned off = (unsigned)(((c * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * 4u;
      pre_p[it] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : 0xfffffff0u, 0, 0));
    }
  };
  // item e = patch floats 4 e .. 4 e + 3 -> bytes 8 e .. 8 e + 7 of each plane
  auto commit = [&](const TileGeo& t) {
    const int total = CIN * KT * t.nrows_in * q4;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + it * NT;
      if (e < total) {
        unsigned h[2], m[2], l[2];
        s3_split2(pre_p[it][0], pre_p[it][1], h[0], m[0], l[0]);
        s3_split2(pre_p[it][2], pre_p[it][3], h[1], m[1], l[1]);
        char* d = P + 8 * e;
        *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2*>(d + plane_bytes) = make_uint2(m[0], m[1]);
        *reinterpret_cast<uint2*>(d + 2 * plane_bytes) = make_uint2(l[0], l[1]);
      }
    }
  };
  constexpr int WIT = S3_CHUNK / 16 / NT;      // b128 items of a weight chunk per thread
  uintx4_t wv[WIT];
  const uintx4_t* Wf = reinterpret_cast<const uintx4_t*>(p.wt);
  auto load_w = [&](int ch) {
#pragma unroll
    for (int i = 0; i < WIT; ++i) wv[i] = Wf[(size_t)ch * (S3_CHUNK / 16) + tid + NT * i];
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WIT; ++i) *reinterpret_cast<uintx4_t*>(Wl + buf * S3_CHUNK + (tid + NT * i) * 16) = wv[i];
  };

  int tile = (int)xcd_remap(blockIdx.x, gridDim.x);
  if (tile >= p.ntiles) return;
  int u = 0;
  float cs[2] = {0.f, 0.f}, cq[2] = {0.f, 0.f};
  load_w(0);
  prefetch(geo(tile));
  store_w(0);
  for (; tile < p.ntiles; tile += gridDim.x) {
    const TileGeo t = geo(tile);
    const int p0 = t.p0, p1 = t.p1, ho_lo = t.ho_lo;
    __syncthreads();
    commit(t);
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) prefetch(geo(tile + gridDim.x));

    // this lane's windows of row offset 0: patch columns 2 wo .. 2 wo + 7 (taps dw = -1 .. 6), hi plane
    const char* Pb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int pi = p0 + (wave * TM + i) * 32 + l31;
      const bool pok = pi < p1;
      const int ho = (pok ? pi : p0) / p.Wo, wo = (pok ? pi : p0) - ho * p.Wo;
      Pb[i] = P + 2 * ((2 * (ho - ho_lo)) * p.PW + 2 * wo);
    }
    const int plane = t.nrows_in * p.PW;

    floatx16 acc[TM][2], cor[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }

    for (int ch = 0; ch < NCH; ++ch, u ^= 1) {
      load_w(ch + 1 < NCH ? ch + 1 : 0);
      const char* Wb = Wl + u * S3_CHUNK + lane * 16;
#pragma unroll
      for (int q = 0; q < S3_Q; ++q) {
        const int s = ch * S3_Q + q;
        if (s < NS) {
          // rows 2s (g = 0), 2s + 1 (g = 1; the pad row past R reads row R - 1: its weights are zero, its data finite)
          const int r0 = 2 * s, r1 = (2 * s + 1 < R) ? 2 * s + 1 : R - 1;
          const int pl0 = r0 / 7, pl1 = r1 / 7;
          const int o0 = pl0 * plane + (r0 - pl0 * 7) * p.PW, o1 = pl1 * plane + (r1 - pl1 * 7) * p.PW;
          const int og = 2 * (g ? o1 : o0);
          auto frag = [&](const char* a) {
            const uintx2_a4 lo = *reinterpret_cast<const uintx2_a4*>(a);
            const uintx2_a4 hi = *reinterpret_cast<const uintx2_a4*>(a + 8);
            return __builtin_bit_cast(bf16x8_t, uintx4_t{lo[0], lo[1], hi[0], hi[1]});
          };
          bf16x8_t ah[TM], am[TM], al[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const char* A = Pb[i] + og;
            ah[i] = frag(A); am[i] = frag(A + plane_bytes); al[i] = frag(A + 2 * plane_bytes);
          }
          bf16x8_t bh[2], bm[2], bl[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const char* bp = Wb + ((q * 2 + j) * 3) * 1024;
            bh[j] = *reinterpret_cast<const bf16x8_t*>(bp);
            bm[j] = *reinterpret_cast<const bf16x8_t*>(bp + 1024);
            bl[j] = *reinterpret_cast<const bf16x8_t*>(bp + 2048);
          }
          // six products per (row block, column tile); the five correction products go to their own accumulators (2 TM
          // x 2 independent chains per wave), in the same order per accumulator as stem_fwd3_kernel
#define S3P_ALL(DST, AF, BF)                                                                              \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)             \
      DST[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i], BF[j], DST[i][j], 0, 0, 0);
          S3P_ALL(cor, ah, bl)
          S3P_ALL(acc, ah, bh)
          S3P_ALL(cor, al, bh)
          S3P_ALL(cor, am, bm)
          S3P_ALL(cor, ah, bm)
          S3P_ALL(cor, am, bh)
#undef S3P_ALL
        }
      }
      store_w(u ^ 1);
      __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int blk = p0 + (wave * TM + i) * 32;
      const long long m_base = (long long)t.frame * npix + blk;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * g;
          if (blk + rr < p1) {
            const float v = acc[i][j][r] + cor[i][j][r];
            p.y[(m_base + rr) * 64 + j * 32 + l31] = v;
            cs[j] += v;
            cq[j] = fmaf(v, v, cq[j]);
          }
        }
    }
  }
  if (p.stats) {
    float* red = smem;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a = cs[j] + __shfl_xor(cs[j], 32, 64), b = cq[j] + __shfl_xor(cq[j], 32, 64);
      if (g == 0) {
        red[wave * 64 + j * 32 + l31] = a;
        red[WAVES * 64 + wave * 64 + j * 32 + l31] = b;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, q = tid >> 6;
      float tt = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tt += red[q * WAVES * 64 + w * 64 + c];
      p.stats[(long long)blockIdx.x * 128 + q * 64 + c] = tt;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stem wgrad: dW[n][k'] = sum_pixels dy[pixel][n] * patch(pixel, k').  Persistent workgroups (one per
// CU) walk tiles of up to 256 pixels (whole output rows), keep the 64 x (R*8) accumulator in registers (each of the 8 waves owns
// 2 n-tiles x NKT k'-tiles), and write one partial slab each; a fixed-order reduce finishes.
//   A[i = n][k = pixel] = dyS[pixel][n]         (LDS, [256][64+4])
//   B[k = pixel][j = k'] = patch[pixbase[pixel] + koff(k'_j)]
// ------------------------------------------------------------------------------------------------
// DENSE (the video stem, 441 taps): k' = (c*kt+dt)*49 + dh*7 + dw without the 7 -> 8 padding: 14 k' tiles instead of
// 16.  14 x 2 (n-tiles) = 28 accumulator tiles do not split evenly over 8 identical waves, so the waves take two
// roles: waves 0-3 own 2 n-tiles x 2 k'-tiles (k' tiles 0-7), waves 4-7 own 1 n-tile x 3 k'-tiles (k' tiles 8-13);
// waves w and w + 4 share a SIMD, which then issues 7 MFMAs per k-step instead of 8 (12.5 % fewer).
template <int CIN, int KT, bool DENSE>
__global__ __launch_bounds__(512) void stem_wgrad_kernel(const StemArgs p) {
  constexpr int R = CIN * KT * 7;
  constexpr int KP = DENSE ? (CIN * KT * 49 + 31) / 32 * 32 : R * 8;   // k' columns of a slab
  constexpr int NKT_ALL = (KP + 31) / 32;         // k' tiles of 32
  constexpr int NKT = DENSE ? 3 : (NKT_ALL + 7) / 8;   // k' tiles per wave (DENSE: 2 or 3 by role)
  static_assert(!DENSE || NKT_ALL == 14, "the two-role split is laid out for 14 k' tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ds = smem;                               // [256][WS_LD]
  int* pixbase = reinterpret_cast<int*>(smem + STEM_TILE * WS_LD);   // [256]
  float* P = smem + STEM_TILE * WS_LD + STEM_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int npix = p.Ho * p.Wo;

  // this lane's k' columns: k' = tile*32 + l31 ; koff = patch offset of that tap
  const bool big = !DENSE || wave < 4;            // role (wave-uniform)
  const int nb = DENSE ? (big ? 2 : 3) : NKT;     // k' tiles of this wave
  const int nsel = DENSE && !big ? ((wave - 4) & 1) : 0;   // the one n-tile of a small wave
  auto tile_of = [&](int j) {
    if (!DENSE) return wave + 8 * j;
    return big ? 2 * wave + j : 8 + 3 * ((wave - 4) >> 1) + j;
  };
  int koff[NKT];
  bool kok[NKT];
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    const int kp = tile_of(j) * 32 + l31;
    kok[j] = j < nb && (DENSE ? kp < CIN * KT * 49 : (kp < KP && (kp & 7) < 7));
    koff[j] = 0;
  }

  constexpr int NACC = DENSE ? 4 : 2 * NKT;       // accumulator tiles per wave: [n-tile][k'-tile] (big) / [k'-tile] (small)
  floatx16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // The next tile's patch and dy rows are fetched into registers while the current tile is multiplied and
  // written to LDS between the two barriers that separate tiles: the global-load phase (a quarter of this
  // kernel's time when it ran between the tiles: 0.29 of 1.10 ms) disappears behind the MFMAs.  The patch
  // row pitch is PW = 4 * q4 floats, so work item e (one float4) lands at LDS float 4 * e.
  constexpr int PIT = 11;                          // float4 patch items per thread (160 KB of LDS bounds it)
  floatx4 pre_p[PIT], pre_d[STEM_TILE * 16 / 512];
  struct TileGeo { int frame, to, b, p0, p1, ho_lo, nrows_in; };
  auto geo = [&](int tile) {
    TileGeo g;
    const int tf = tile % p.tiles_per_frame;
    g.frame = tile / p.tiles_per_frame;
    g.to = g.frame % p.Ti;
    g.b = g.frame / p.Ti;
    g.p0 = tf * p.tile_px;
    g.p1 = min(g.p0 + p.tile_px, npix);
    g.ho_lo = g.p0 / p.Wo;
    g.nrows_in = 2 * ((g.p1 - 1) / p.Wo - g.ho_lo) + 7;
    return g;
  };
  // (Wi % 4 == 0, checked by the host: a patch float4 is then entirely inside or entirely outside its row.)
  const int q4 = p.PW >> 2;
  const long long item_floats = (long long)CIN * p.Ti * p.Hi * p.Wi;
  auto prefetch = [&](const TileGeo& g) {
    const int total = CIN * KT * g.nrows_in * q4;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (long long)g.b * item_floats), 0, (int)(item_floats * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + it * 512;
      const int r = e / q4, cq = e - r * q4;
      const int pl = r / g.nrows_in, row = r - pl * g.nrows_in;
      const int dt = pl % KT, c = pl / KT;
      const int ti = g.to + dt - KT / 2, hi = 2 * g.ho_lo - 3 + row, wi = cq * 4 - 4;
      const bool ok = (e < total) & ((unsigned)ti < (unsigned)p.Ti) & ((unsigned)hi < (unsigned)p.Hi) &
                      ((unsigned)wi < (unsigned)p.Wi);
      const unsigned off = (unsigned)(((c * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * 4u;
      pre_p[it] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : 0xfffffff0u, 0, 0));
    }
    // dy rows of the tile; rows past the frame end fall off num_records and read as zeros
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ((long long)g.frame * npix + g.p0) * 64), 0, (g.p1 - g.p0) * 256, 0x00020000);
#pragma unroll
    for (int it = 0; it < STEM_TILE * 16 / 512; ++it) {
      const int e = tid + it * 512;
      pre_d[it] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsD, (unsigned)e * 16u, 0, 0));
    }
  };
  auto commit = [&](const TileGeo& g) {
    const int total = CIN * KT * g.nrows_in * q4;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + it * 512;
      if (e < total) *reinterpret_cast<floatx4*>(P + 4 * e) = pre_p[it];
    }
#pragma unroll
    for (int it = 0; it < STEM_TILE * 16 / 512; ++it) {
      const int e = tid + it * 512;
      *reinterpret_cast<floatx4*>(&Ds[(e >> 4) * WS_LD + (((e & 15) * 4) ^ (((e >> 4) & 1) << 5))]) = pre_d[it];
    }
  };

  const int tile0 = p.xcd_local ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (tile0 < p.ntiles) prefetch(geo(tile0));
  for (int tile = tile0; tile < p.ntiles; tile += gridDim.x) {
    const TileGeo g = geo(tile);
    const int p0 = g.p0, p1 = g.p1, ho_lo = g.ho_lo;
    const int plane = g.nrows_in * p.PW;
    __syncthreads();                               // previous tile fully consumed
    commit(g);
    if (tid < STEM_TILE) {
      const int pi = min(p0 + tid, p1 - 1);
      const int ho = pi / p.Wo, wo = pi - ho * p.Wo;
      pixbase[tid] = (2 * (ho - ho_lo)) * p.PW + 2 * wo + 1;
    }
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int kp = tile_of(j) * 32 + l31;
      int dw, dh, pl;
      if (DENSE) {
        pl = kp / 49;
        const int f = kp - pl * 49;
        dh = f / 7;
        dw = f - dh * 7;
      } else {
        dw = kp & 7;
        const int row = kp >> 3;
        dh = row % 7;
        pl = row / 7;
      }
      koff[j] = kok[j] ? pl * plane + dh * p.PW + dw : 0;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) prefetch(geo(tile + gridDim.x));
    // The patch gather is a dependent LDS chain (pixbase -> patch address): its base is fetched two k-steps
    // ahead and the operands one k-step ahead of the MFMAs that use them (order pinned with sched_barrier;
    // two k-steps per trip so the register sets ping-pong without copies).  Lanes of the padded / out-of-range
    // k' columns read patch word pb + 0: their accumulator columns are never read (stem_wgrad_reduce_kernel
    // skips dw == 7, the slab write skips k' >= KP), so no select is spent on them.
    const float* dsp = Ds + h * WS_LD + l31;        // this lane's dy column, row 2*kk + h (odd rows: halves swapped)
    const int c0 = 32 * h, c1 = 32 - 32 * h;
    const int* pbp = pixbase + h;
    const int nkk = (p.tile_px + 3) / 4 * 2;      // k-steps (pixel pairs), even; rows past the tile are zero dy rows
    auto kloop = [&](auto BIG) {
      constexpr bool B_ = decltype(BIG)::value;
      constexpr int NBv = DENSE ? (B_ ? 2 : 3) : NKT;    // patch operands per k-step
      const int cs = (DENSE && !B_) ? (nsel ? c1 : c0) : c0;   // small wave: its one dy column block
      float a0[2], a1[2], bv[2][NBv];
      int pbn;
      {
        const int pb0 = pbp[0];
        a0[0] = dsp[cs];
        if (B_) a1[0] = dsp[c1];
#pragma unroll
        for (int j = 0; j < NBv; ++j) bv[0][j] = P[pb0 + koff[j]];
        pbn = pbp[2];
      }
      auto step = [&](int kk, int cur) {   // MFMAs of k-step kk from set `cur`; operands of kk+1 into the other set
        const int nx = cur ^ 1;
        const int r1 = 2 * min(kk + 1, nkk - 1), r2 = 2 * min(kk + 2, nkk - 1);
        a0[nx] = dsp[r1 * WS_LD + cs];
        if (B_) a1[nx] = dsp[r1 * WS_LD + c1];
#pragma unroll
        for (int j = 0; j < NBv; ++j) bv[nx][j] = P[pbn + koff[j]];
        pbn = pbp[r2];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NBv; ++j) {
          if (B_) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], bv[cur][j], acc[j], 0, 0, 0);
            acc[NBv + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], bv[cur][j], acc[NBv + j], 0, 0, 0);
          } else {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], bv[cur][j], acc[j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int kk = 0; kk < nkk; kk += 2) {
        step(kk, 0);
        step(kk + 1, 1);
      }
    };
    if (big) kloop(std::true_type{}); else kloop(std::false_type{});
  }
  // partial slab [blockIdx.x][n][k']   (accumulator tile a: big wave -> n-tile a / nb, k' tile a % nb; small -> nsel, a)
  float* o = p.part + (long long)blockIdx.x * 64 * KP;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int t = big ? a / (DENSE ? 2 : NKT) : nsel, j = big ? a % (DENSE ? 2 : NKT) : a;
    if (!big && a >= 3) continue;
    const int kp = tile_of(j) * 32 + l31;
    if (kp < KP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        o[(long long)n * KP + kp] = acc[a][r];
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Round 4: the video stem's weight gradient as fp32-accurate split-bf16 products (as stem_fwd3_kernel).
// Same tiles (whole output rows), patch / dy prefetch, two-role wave layout, partial slabs and reduce as stem_wgrad_kernel;
// the contraction runs over 16 pixels per k-step:
//   A[i = n][k = pixel]  : lane (n, g = lane >> 5) reads dy of 8 consecutive pixels p0 + 8 g + e of its channel
//   B[k = pixel][j = k'] : lane (k', g) reads the patch at pixbase[p0 + 8 g] + 2 e + koff(k') — a group of 8 pixels lies in
//                          one output row (8 | Wo, whole-row tiles), so one table entry serves the group;
// both fragments are split in registers (44 VALU each) and feed six matrix instructions per (n-tile, k'-tile).
// dy rows are stored with their 32-column halves swapped on every second GROUP of 8 pixels (the two half-waves read groups
// 8 pixels apart: disjoint banks).
// ------------------------------------------------------------------------------------------------
// PRE (round 5): dy is split ONCE, when its rows are committed to LDS, into the fragments the products read — DyF[k-step][n-tile]
// [hi | mid | lo][lane][8 bf16], 6 KB per 16 pixels.  In the form above every wave splits the dy fragments it uses: the 2 n-tile
// fragments of a k-step are split by 12 (wave, n-tile) users — 11 vector instructions per matrix instruction, the most of any
// kernel in the step.  The transposition a pixel-contraction fragment needs (a lane = 8 consecutive pixels of ONE channel) is
// done by the loader's thread mapping instead of by 16-bit LDS traffic: a thread fetches the same 4 channels of 8 CONSECUTIVE
// pixels (8 row-coalesced 16-byte loads), i.e. the complete fragment content of 4 lanes — 16 pair splits and 12 ds_write_b128
// per thread and tile; a product step then reads a dy fragment with 3 ds_read_b128 and no vector instruction.
template <int CIN, int KT, bool PRE = false>
__global__ __launch_bounds__(512, 1) void stem_wgrad3_kernel(const StemArgs p) {
  constexpr int KP = (CIN * KT * 49 + 31) / 32 * 32;   // k' columns of a slab (dense layout)
  constexpr int NKT_ALL = KP / 32, NKT = 3;
  static_assert(NKT_ALL == 14, "the two-role split is laid out for 14 k' tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // PRE: DyF takes (tile_px / 16) x 6 KB where Ds took 64 KB
  const int dy_floats = PRE ? (p.tile_px / 16) * (S3_STEP_BYTES / 4) : STEM_TILE * WS_LD;
  float* Ds = smem;                               // [256][WS_LD]   (PRE: DyF)
  char* DyF = reinterpret_cast<char*>(smem);
  int* pixbase = reinterpret_cast<int*>(smem + dy_floats);   // [256]
  float* P = smem + dy_floats + STEM_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, l31 = lane & 31;
  const int npix = p.Ho * p.Wo;
  const bool big = wave < 4;                      // role (wave-uniform): 2 n-tiles x 2 k'-tiles, or 1 n-tile x 3 k'-tiles
  const int nb = big ? 2 : 3;
  const int nsel = big ? 0 : ((wave - 4) & 1);
  auto tile_of = [&](int j) { return big ? 2 * wave + j : 8 + 3 * ((wave - 4) >> 1) + j; };
  int koff[NKT];
  bool kok[NKT];
#pragma unroll
  for (int j = 0; j < NKT; ++j) {
    const int kp = tile_of(j) * 32 + l31;
    kok[j] = j < nb && kp < CIN * KT * 49;
    koff[j] = 0;
  }
  floatx16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr int PIT = 11;
  floatx4 pre_p[PIT], pre_d[STEM_TILE * 16 / 512];
  struct TileGeo { int frame, to, b, p0, p1, ho_lo, nrows_in; };
  auto geo = [&](int tile) {
    TileGeo t;
    const int tf = tile % p.tiles_per_frame;
    t.frame = tile / p.tiles_per_frame;
    t.to = t.frame % p.Ti;
    t.b = t.frame / p.Ti;
    t.p0 = tf * p.tile_px;
    t.p1 = min(t.p0 + p.tile_px, npix);
    t.ho_lo = t.p0 / p.Wo;
    t.nrows_in = 2 * ((t.p1 - 1) / p.Wo - t.ho_lo) + 7;
    return t;
  };
  const int q4 = p.PW >> 2;
  const long long item_floats = (long long)CIN * p.Ti * p.Hi * p.Wi;
  auto prefetch = [&](const TileGeo& t) {
    const int total = CIN * KT * t.nrows_in * q4;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (long long)t.b * item_floats), 0, (int)(item_floats * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + it * 512;
      const int r = e / q4, cq = e - r * q4;
      const int pl = r / t.nrows_in, row = r - pl * t.nrows_in;
      const int dt = pl % KT, c = pl / KT;
      const int ti = t.to + dt - KT / 2, hi = 2 * t.ho_lo - 3 + row, wi = cq * 4 - 4;
      const bool ok = (e < total) & ((unsigned)ti < (unsigned)p.Ti) & ((unsigned)hi < (unsigned)p.Hi) &
                      ((unsigned)wi < (unsigned)p.Wi);
      const unsigned off = (unsigned)(((c * p.Ti + ti) * p.Hi + hi) * p.Wi + wi) * 4u;
      pre_p[it] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? off : 0xfffffff0u, 0, 0));
    }
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ((long long)t.frame * npix + t.p0) * 64), 0, (t.p1 - t.p0) * 256, 0x00020000);
#pragma unroll
    for (int it = 0; it < STEM_TILE * 16 / 512; ++it) {
      // PRE: pixel 8 (tid >> 4) + it, channels 4 (tid & 15) ..: a thread holds 8 consecutive pixels of its 4 channels
      const int e = PRE ? (((tid >> 4) * 8 + it) * 16 + (tid & 15)) : tid + it * 512;
      pre_d[it] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsD, (unsigned)e * 16u, 0, 0));
    }
  };
  auto commit = [&](const TileGeo& t) {
    const int total = CIN * KT * t.nrows_in * q4;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int e = tid + it * 512;
      if (e < total) *reinterpret_cast<floatx4*>(P + 4 * e) = pre_p[it];
    }
    if (PRE) {
      const int pgrp = tid >> 4, ks = pgrp >> 1, gg = pgrp & 1;
      if (ks < p.tile_px / 16) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int n = 4 * (tid & 15) + c;
          unsigned hh[4], mm[4], ll[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) s3_split2(pre_d[2 * i][c], pre_d[2 * i + 1][c], hh[i], mm[i], ll[i]);
          char* q = DyF + ((ks * 2 + (n >> 5)) * 3) * 1024 + ((n & 31) + 32 * gg) * 16;
          *reinterpret_cast<uintx4_t*>(q) = uintx4_t{hh[0], hh[1], hh[2], hh[3]};
          *reinterpret_cast<uintx4_t*>(q + 1024) = uintx4_t{mm[0], mm[1], mm[2], mm[3]};
          *reinterpret_cast<uintx4_t*>(q + 2048) = uintx4_t{ll[0], ll[1], ll[2], ll[3]};
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < STEM_TILE * 16 / 512; ++it) {
        const int e = tid + it * 512;                  // pixel e >> 4, columns 4 (e & 15) ..: halves swapped on odd pixel groups
        *reinterpret_cast<floatx4*>(&Ds[(e >> 4) * WS_LD + (((e & 15) * 4) ^ (((e >> 7) & 1) << 5))]) = pre_d[it];
      }
    }
  };

  const int tile0 = (int)xcd_remap(blockIdx.x, gridDim.x);
  if (tile0 < p.ntiles) prefetch(geo(tile0));
  for (int tile = tile0; tile < p.ntiles; tile += gridDim.x) {
    const TileGeo t = geo(tile);
    const int p0 = t.p0, p1 = t.p1, ho_lo = t.ho_lo;
    const int plane = t.nrows_in * p.PW;
    __syncthreads();
    commit(t);
    if (tid < STEM_TILE) {
      const int pi = min(p0 + tid, p1 - 1);
      const int ho = pi / p.Wo, wo = pi - ho * p.Wo;
      pixbase[tid] = (2 * (ho - ho_lo)) * p.PW + 2 * wo + 1;
    }
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
      const int kp = tile_of(j) * 32 + l31;
      const int pl = kp / 49, f = kp - pl * 49, dh = f / 7, dw = f - dh * 7;
      koff[j] = kok[j] ? pl * plane + dh * p.PW + dw : 0;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) prefetch(geo(tile + gridDim.x));
    const int nsteps = p.tile_px / 16;             // (the host admits only tiles of whole 16-pixel steps)
    // this lane's dy column for n-tile 0 / 1 (its own n-tile for a small wave), pixel group g of the step
    const int colA0 = (big ? 0 : 32 * nsel) + l31, colA1 = 32 + l31;
    auto frag3 = [&](const float (&v)[8], bf16x8_t& fh, bf16x8_t& fm, bf16x8_t& fl) {
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) s3_split2(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
      fh = __builtin_bit_cast(bf16x8_t, uintx4_t{h[0], h[1], h[2], h[3]});
      fm = __builtin_bit_cast(bf16x8_t, uintx4_t{m[0], m[1], m[2], m[3]});
      fl = __builtin_bit_cast(bf16x8_t, uintx4_t{l[0], l[1], l[2], l[3]});
    };
    auto six = [&](floatx16& c, const bf16x8_t& ah, const bf16x8_t& am, const bf16x8_t& al, const bf16x8_t& bh,
                   const bf16x8_t& bm, const bf16x8_t& bl) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    };
    auto kloop = [&](auto BIG) {
      constexpr bool B_ = decltype(BIG)::value;
      constexpr int NBv = B_ ? 2 : 3;
      for (int s = 0; s < nsteps; ++s) {
        const int px = 16 * s + 8 * g;               // first pixel of this half-wave's group
        const int swz = (px >> 3 & 1) << 5;          // the group's column swizzle
        const float* dr = Ds + px * WS_LD;
        const int pb = pixbase[px];
        bf16x8_t ah[2], am[2], al[2];
        if (PRE) {
          const char* q0 = DyF + ((s * 2 + (B_ ? 0 : nsel)) * 3) * 1024 + lane * 16;
          ah[0] = *reinterpret_cast<const bf16x8_t*>(q0);
          am[0] = *reinterpret_cast<const bf16x8_t*>(q0 + 1024);
          al[0] = *reinterpret_cast<const bf16x8_t*>(q0 + 2048);
          if (B_) {
            ah[1] = *reinterpret_cast<const bf16x8_t*>(q0 + 3072);
            am[1] = *reinterpret_cast<const bf16x8_t*>(q0 + 4096);
            al[1] = *reinterpret_cast<const bf16x8_t*>(q0 + 5120);
          }
        } else {
          float va[2][8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            va[0][e] = dr[e * WS_LD + (colA0 ^ swz)];
            if (B_) va[1][e] = dr[e * WS_LD + (colA1 ^ swz)];
          }
          frag3(va[0], ah[0], am[0], al[0]);
          if (B_) frag3(va[1], ah[1], am[1], al[1]);
        }
#pragma unroll
        for (int j = 0; j < NBv; ++j) {
          float vb[8];
          const float* pp = P + pb + koff[j];
#pragma unroll
          for (int e = 0; e < 8; ++e) vb[e] = pp[2 * e];
          bf16x8_t bh, bm, bl;
          frag3(vb, bh, bm, bl);
          if (B_) {
            six(acc[j], ah[0], am[0], al[0], bh, bm, bl);
            six(acc[NBv + j], ah[1], am[1], al[1], bh, bm, bl);
          } else {
            six(acc[j], ah[0], am[0], al[0], bh, bm, bl);
          }
        }
      }
    };
    if (big) kloop(std::true_type{}); else kloop(std::false_type{});
  }
  // partial slab [blockIdx.x][n][k']   (accumulator tile a: big wave -> n-tile a / 2, k' tile a % 2; small -> nsel, a)
  float* o = p.part + (long long)blockIdx.x * 64 * KP;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int t = big ? a / 2 : nsel, j = big ? a % 2 : a;
    if (!big && a >= 3) continue;
    const int kp = tile_of(j) * 32 + l31;
    if (kp < KP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        o[(long long)n * KP + kp] = acc[a][r];
      }
    }
  }
}

// dw[n][dt][dh][dw][c] = sum_g part[g][n][k'(c,dt,dh,dw)] — block = 32 elements x 8 slices of the G partials,
// 4 independent streams per thread (a single chain of G = 256 loads per thread took 111 us); fixed order.
template <int CIN, int KT, bool DENSE>
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                int G) {
  constexpr int R = CIN * KT * 7, KP = DENSE ? (CIN * KT * 49 + 31) / 32 * 32 : R * 8;
  __shared__ float sh[8][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;     // over n * K (K = R*7)
  const bool ok = i < 64 * R * 7;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (ok) {
    const int c = i % CIN;
    int r = i / CIN;
    const int dwi = r % 7; r /= 7;
    const int dh = r % 7; r /= 7;
    const int dt = r % KT;
    const int n = r / KT;
    const int kp = DENSE ? (c * KT + dt) * 49 + dh * 7 + dwi : (((c * KT + dt) * 7 + dh) * 8) + dwi;
    const float* q = part + (long long)n * KP + kp;
    const long long st = 64ll * KP;
    int g = sl;
    for (; g + 24 < G; g += 32) {
      const float v0 = q[g * st], v1 = q[(g + 8) * st], v2 = q[(g + 16) * st], v3 = q[(g + 24) * st];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; g < G; g += 8) s0 += q[g * st];
  }
  sh[sl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && ok) {
    float t = sh[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += sh[k][e];
    dw[i] = t;
  }
}

static bool stem_match(const avid_conv_desc* d) {
  return d->x_channel_first && d->Cout == 64 && d->kh == 7 && d->kw == 7 && d->sh == 2 && d->sw == 2 && d->st == 1 &&
         d->ph == 3 && d->pw == 3 && ((d->Cin == 3 && d->kt == 3 && d->pt == 1) || (d->Cin == 1 && d->kt == 1 && d->pt == 0)) &&
         d->To == d->Ti;
}

// Pixels per wgrad tile: whole output rows (no ragged last tile per frame: 4 x 56 = 224 of 256 pixel slots
// instead of 13 tiles x 256 slots for 3136 pixels) when at least one row fits.
static int stem_wgrad_tile_px(const avid_conv_desc* d) {
  const int rows = STEM_TILE / d->Wo;
  return rows >= 1 ? rows * d->Wo : STEM_TILE;
}

static void stem_geometry(const avid_conv_desc* d, StemArgs& a, int tile) {
  a.B = d->B; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo;
  a.PW = (d->Wi + 4 + 4 + 3) / 4 * 4;                       // 4 left pad + >= 4 right pad (taps reach wi = W+3)
  const int npix = d->Ho * d->Wo;
  a.tile_px = tile;
  a.tiles_per_frame = (npix + tile - 1) / tile;
  const int rows_out = tile % d->Wo == 0 ? tile / d->Wo : (tile - 1 + d->Wo - 1) / d->Wo + 1;
  a.rows_in_max = 2 * (rows_out - 1) + 7;
  a.ntiles = d->B * d->Ti * a.tiles_per_frame;
  a.xcd_local = 1;
}

size_t stem_patch_floats(const avid_conv_desc* d, int tile) {
  StemArgs a;
  stem_geometry(d, a, tile);
  return (size_t)d->Cin * d->kt * a.rows_in_max * a.PW + 16;
}

static size_t stem_fwd_lds(const avid_conv_desc* d, int tile) {
  return sizeof(float) * (2 * FK * WS_LD + stem_patch_floats(d, tile));
}
// forward tile: 128 pixels x 2 workgroups per CU when that fits, else 256 pixels x 1; 0 = neither
static int stem_fwd_tile(const avid_conv_desc* d) {
  // 14 / 16 float4 patch items per thread (the kernel's register prefetch)
  if (stem_fwd_lds(d, 128) <= 80 * 1024 && stem_patch_floats(d, 128) <= 14 * 256 * 4) return 128;
  if (stem_fwd_lds(d, 256) <= 160 * 1024 && stem_patch_floats(d, 256) <= 16 * 512 * 4) return 256;
  return 0;
}
// the bf16x3 forward (stem_fwd3_kernel): the video stem, 256-pixel tiles, patch + two 24 KB weight stages in 160 KB
// k-steps of the split weights, padded to whole chunks of either kernel shape (4 steps)
static int stem_fwd3_steps(const avid_conv_desc* d) { return ((d->Cin * d->kt * 7 + 1) / 2 + 3) / 4 * 4; }
static size_t stem_fwd3_lds(const avid_conv_desc* d) { return 2 * 4 * (size_t)S3_STEP_BYTES + sizeof(float) * stem_patch_floats(d, 256); }
static bool stem_fwd3_ok(const avid_conv_desc* d) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("AVID_STEM_BF16X3");
    on = e ? atoi(e) != 0 : 1;
  }
  return on && d->Cin == 3 && d->kt == 3 && stem_fwd3_lds(d) <= 160 * 1024 && stem_patch_floats(d, 256) <= 16 * 512 * 4;
}
static size_t stem_wgrad_lds(const avid_conv_desc* d) {
  return sizeof(float) * (STEM_TILE * WS_LD + STEM_TILE + stem_patch_floats(d, stem_wgrad_tile_px(d)));
}

bool stem_fwd_supported(const avid_conv_desc* d) {
  return stem_match(d) && stem_fwd_tile(d) != 0 && d->Wi % 4 == 0 &&
         (long long)d->Cin * d->Ti * d->Hi * d->Wi * 4 < (1ll << 31);
}
bool stem_wgrad_supported(const avid_conv_desc* d) {
  // 11 float4 patch items per thread of 512 (the kernel's register prefetch) cover any patch that fits the LDS
  return stem_match(d) && stem_wgrad_lds(d) <= 160 * 1024 && stem_patch_floats(d, stem_wgrad_tile_px(d)) <= 11 * 512 * 4 &&
         d->Wi % 4 == 0 && (long long)d->Cin * d->Ti * d->Hi * d->Wi * 4 < (1ll << 31);
}

size_t stem_fwd_ws_bytes(const avid_conv_desc* d) {
  const size_t fp32_path = sizeof(float) * (size_t)d->Cin * d->kt * FK * 64;
  const size_t split_path = stem_fwd3_ok(d) ? (size_t)stem_fwd3_steps(d) * S3_STEP_BYTES : 0;
  return fp32_path > split_path ? fp32_path : split_path;
}
int stem_wgrad_groups() { return device_cus(); }
size_t stem_wgrad_ws_bytes(const avid_conv_desc* d) {
  return sizeof(float) * (size_t)stem_wgrad_groups() * 64 * d->Cin * d->kt * 7 * 8;
}

template <int CIN, int KT, int WAVES>
static int stem_fwd_launch(const avid_conv_desc* d, const float* x, const float* w, float* y, float* stats, void* ws,
                           hipStream_t s) {
  StemArgs a{};
  stem_geometry(d, a, WAVES * 32);
  a.x = x; a.w = w; a.y = y; a.wt = static_cast<float*>(ws); a.stats = stats;
  hipLaunchKernelGGL((stem_repack_kernel<CIN, KT>), dim3((CIN * KT * FK * 64 + 255) / 256), dim3(256), 0, s, w,
                     static_cast<float*>(ws));
  const size_t lds = stem_fwd_lds(d, WAVES * 32);
  static bool set = false;
  if (!set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd_kernel<CIN, KT, WAVES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    set = true;
  }
  const double M = (double)d->B * d->To * d->Ho * d->Wo, K = (double)CIN * KT * 49;
  ScopedTimer t(s, CIN == 3 ? "stem_fwd_kernel<3,3>" : "stem_fwd_kernel<1,1>", 2.0 * M * 64 * K,
                4.0 * ((double)d->B * CIN * d->Ti * d->Hi * d->Wi + 64 * K + M * 64));
  const int slots = (8 / WAVES) * device_cus();           // workgroups resident on the CUs
  const int grid = a.ntiles < slots ? a.ntiles : slots;
  hipLaunchKernelGGL((stem_fwd_kernel<CIN, KT, WAVES>), dim3(grid), dim3(WAVES * 64), lds, s, a);
  return check_launch("stem_fwd");
}

// the bf16x3 weight gradient (stem_wgrad3_kernel): the video stem, tiles of whole 16-pixel steps in whole output rows
static bool stem_wgrad3_ok(const avid_conv_desc* d) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("AVID_STEM_BF16X3");
    on = e ? atoi(e) != 0 : 1;
  }
  const int tile = stem_wgrad_tile_px(d);
  return on && d->Cin == 3 && d->kt == 3 && d->Wo % 8 == 0 && tile % d->Wo == 0 && tile % 16 == 0 && (d->Ho * d->Wo) % tile == 0;
}

template <int CIN, int KT>
static int stem_wgrad_launch(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws,
                             hipStream_t s) {
  StemArgs a{};
  stem_geometry(d, a, stem_wgrad_tile_px(d));
  a.x = x; a.dy = dy; a.part = static_cast<float*>(ws);
  int G = stem_wgrad_groups();
  if (G > a.ntiles) G = a.ntiles;
  const size_t lds = stem_wgrad_lds(d);
  if (CIN == 3 && stem_wgrad3_ok(d)) {
    static bool set3 = false;
    if (!set3) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad3_kernel<3, 3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad3_kernel<3, 3, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      set3 = true;
    }
    // dy pre-split into fragment order (PRE) where its 6 KB per 16 pixels fit beside the patch; AVID_STEM_WGRAD_PRE=0: never
    static int pre_on = -1;
    if (pre_on < 0) { const char* e = getenv("AVID_STEM_WGRAD_PRE"); pre_on = e ? atoi(e) != 0 : 1; }
    const size_t lds_pre = (size_t)(a.tile_px / 16) * S3_STEP_BYTES + sizeof(float) * (STEM_TILE + stem_patch_floats(d, stem_wgrad_tile_px(d)));
    const bool pre = pre_on && lds_pre <= 160 * 1024;
    const double M = (double)d->B * d->To * d->Ho * d->Wo, K = 3.0 * 3 * 49;
    {
      ScopedTimer t(s, "stem_wgrad3_kernel<3,3>", 2.0 * M * 64 * K,
                    4.0 * ((double)d->B * 3 * d->Ti * d->Hi * d->Wi + 64 * K + M * 64));
      if (pre) hipLaunchKernelGGL((stem_wgrad3_kernel<3, 3, true>), dim3(G), dim3(512), lds_pre, s, a);
      else hipLaunchKernelGGL((stem_wgrad3_kernel<3, 3>), dim3(G), dim3(512), lds, s, a);
    }
    int rc3 = check_launch("stem_wgrad3");
    if (rc3) return rc3;
    const int n3 = 64 * 3 * 3 * 49;
    hipLaunchKernelGGL((stem_wgrad_reduce_kernel<3, 3, true>), dim3((n3 + 31) / 32), dim3(256), 0, s,
                       static_cast<const float*>(ws), dw, G);
    return check_launch("stem_wgrad_reduce");
  }
  static bool set = false;
  if (!set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_kernel<CIN, KT, (CIN * KT * 49 > 64)>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    set = true;
  }
  const double M = (double)d->B * d->To * d->Ho * d->Wo, K = (double)CIN * KT * 49;
  {
    ScopedTimer t(s, CIN == 3 ? "stem_wgrad_kernel<3,3>" : "stem_wgrad_kernel<1,1>", 2.0 * M * 64 * K,
                  4.0 * ((double)d->B * CIN * d->Ti * d->Hi * d->Wi + 64 * K + M * 64));
    hipLaunchKernelGGL((stem_wgrad_kernel<CIN, KT, (CIN * KT * 49 > 64)>), dim3(G), dim3(512), lds, s, a);
  }
  int rc = check_launch("stem_wgrad");
  if (rc) return rc;
  const int n = 64 * CIN * KT * 49;
  hipLaunchKernelGGL((stem_wgrad_reduce_kernel<CIN, KT, (CIN * KT * 49 > 64)>), dim3((n + 31) / 32), dim3(256), 0, s,
                     static_cast<const float*>(ws), dw, G);
  return check_launch("stem_wgrad_reduce");
}

// stem_fwd3p_kernel (the patch split once at commit time, three bf16 planes): where 6 bytes per patch element fit beside the weight
// stages; AVID_STEM_FWD_PRE=0 / avid_stem_fwd_pre_configure(0): stem_fwd3_kernel everywhere
static int g_stem_fwd_pre = -1;
static size_t stem_patch_plane_bytes(const avid_conv_desc* d) { return (2 * stem_patch_floats(d, 256) + 15) / 16 * 16; }
static size_t stem_fwd3p_lds(const avid_conv_desc* d) { return 2 * 4 * (size_t)S3_STEP_BYTES + 3 * stem_patch_plane_bytes(d); }
static bool stem_fwd3p_ok(const avid_conv_desc* d) {
  if (g_stem_fwd_pre < 0) {
    const char* e = getenv("AVID_STEM_FWD_PRE");
    g_stem_fwd_pre = e ? (atoi(e) != 0 ? 1 : 0) : 1;
  }
  return g_stem_fwd_pre != 0 && stem_fwd3_ok(d) && stem_fwd3p_lds(d) <= 160 * 1024 && d->Wi % 4 == 0 &&
         stem_patch_floats(d, 256) <= 19 * 256 * 4;       // (the register prefetch of either wave shape: 10 x 512 / 19 x 256 items)
}
void stem_fwd_pre_configure(int on) { g_stem_fwd_pre = on < 0 ? -1 : (on ? 1 : 0); }
// wave shape of stem_fwd3p_kernel: AVID_STEM_FWD_TM = 1 (default: eight waves of 32 pixels) or 2 (four waves of 64 pixels, one per
// SIMD: half the weight-fragment reads and eight accumulation chains per wave — and 0.625-0.649 against 0.564-0.568 ms: with one
// wave on a SIMD nothing runs while it waits for its fragments.  Same outputs bit for bit; the BatchNorm partial sums add the
// pixels of a wave in another order)
static int stem_fwd3p_tm() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AVID_STEM_FWD_TM");
    v = e ? atoi(e) : 1;
    if (v != 1 && v != 2) v = 1;
  }
  return v;
}

template <int CIN, int KT>
static int stem_fwd3_launch(const avid_conv_desc* d, const float* x, const float* w, float* y, float* stats, void* ws,
                            hipStream_t s) {
  StemArgs a{};
  stem_geometry(d, a, 256);
  a.x = x; a.w = w; a.y = y; a.wt = static_cast<float*>(ws); a.stats = stats;
  const bool pre = stem_fwd3p_ok(d);
  if (pre) {
    a.patch_plane_bytes = (int)stem_patch_plane_bytes(d);
    hipLaunchKernelGGL((stem_split_weights_rows_kernel<CIN, KT>), dim3(stem_fwd3_steps(d) * 2), dim3(64), 0, s, w, static_cast<uintx4_t*>(ws));
  } else {
    hipLaunchKernelGGL((stem_split_weights_kernel<CIN, KT>), dim3(stem_fwd3_steps(d) * 2), dim3(64), 0, s, w, static_cast<uintx4_t*>(ws));
  }
  static bool set = false;
  if (!set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd3_kernel<CIN, KT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd3p_kernel<CIN, KT, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd3p_kernel<CIN, KT, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    set = true;
  }
  const double M = (double)d->B * d->To * d->Ho * d->Wo, K = (double)CIN * KT * 49;
  ScopedTimer t(s, pre ? "stem_fwd3p_kernel<3,3>" : "stem_fwd3_kernel<3,3>", 2.0 * M * 64 * K,
                4.0 * ((double)d->B * CIN * d->Ti * d->Hi * d->Wi + 64 * K + M * 64));
  const int grid = a.ntiles < device_cus() ? a.ntiles : device_cus();
  if (pre && stem_fwd3p_tm() == 2) hipLaunchKernelGGL((stem_fwd3p_kernel<CIN, KT, 2>), dim3(grid), dim3(256), stem_fwd3p_lds(d), s, a);
  else if (pre) hipLaunchKernelGGL((stem_fwd3p_kernel<CIN, KT, 1>), dim3(grid), dim3(512), stem_fwd3p_lds(d), s, a);
  else hipLaunchKernelGGL((stem_fwd3_kernel<CIN, KT>), dim3(grid), dim3(512), stem_fwd3_lds(d), s, a);
  return check_launch("stem_fwd3");
}

// workgroups of the forward launch == rows of its BatchNorm partial sums
int stem_fwd_grid(const avid_conv_desc* d) {
  if (stem_fwd3_ok(d)) {
    StemArgs a{};
    stem_geometry(d, a, 256);
    return a.ntiles < device_cus() ? a.ntiles : device_cus();
  }
  const int tile = stem_fwd_tile(d);
  if (!tile) return 0;
  StemArgs a{};
  stem_geometry(d, a, tile);
  const int slots = (tile == 128 ? 2 : 1) * device_cus();
  return a.ntiles < slots ? a.ntiles : slots;
}

bool stem_fwd_is_split(const avid_conv_desc* d) { return stem_fwd_supported(d) && stem_fwd3_ok(d); }
bool stem_fwd_is_presplit(const avid_conv_desc* d) { return stem_fwd_is_split(d) && stem_fwd3p_ok(d); }

int stem_fwd(const avid_conv_desc* d, const float* x, const float* w, float* y, float* stats, void* ws, hipStream_t s) {
  if (stem_fwd3_ok(d)) return stem_fwd3_launch<3, 3>(d, x, w, y, stats, ws, s);
  if (stem_fwd_tile(d) == 128)
    return d->Cin == 3 ? stem_fwd_launch<3, 3, 4>(d, x, w, y, stats, ws, s) : stem_fwd_launch<1, 1, 4>(d, x, w, y, stats, ws, s);
  return d->Cin == 3 ? stem_fwd_launch<3, 3, 8>(d, x, w, y, stats, ws, s) : stem_fwd_launch<1, 1, 8>(d, x, w, y, stats, ws, s);
}
int stem_wgrad(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, hipStream_t s) {
  return d->Cin == 3 ? stem_wgrad_launch<3, 3>(d, x, dy, dw, ws, s) : stem_wgrad_launch<1, 1>(d, x, dy, dw, ws, s);
}

}  // namespace avid

extern "C" int avid_stem_fwd_pre_configure(int on) {
  avid::stem_fwd_pre_configure(on);
  return AVID_OK;
}
