/*
 * avid_hip.h — C-ABI of libavid_hip.so: the MI355X (gfx950) kernels behind the AVID / AVID-CMA
 * training step.  Plain C types only; no torch / C++ types cross this boundary.
 *
 * The reference (facebookresearch/AVID-CMA) is pure Python and has no FFI of its own: every entry
 * point below replaces the ATen/cuDNN op that a reference line dispatches to.  Each declaration
 * cites that line (paths relative to the reference root).  The Python side that binds these symbols
 * with ctypes lives in avid-cma_amd/avid_hip/lib.py; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - All device pointers are fp32 unless stated; indices are int64 (reference: default collate of
 *     Python ints, datasets/video_db.py:263) except positive_set (int32, criterions/avid_cma.py:223).
 *   - Activations are channels-last: [B, T, H, W, C] (2-D audio uses T = 1).  Conv weights are
 *     [Cout][kt][kh][kw][Cin] — the memory of a torch tensor of logical shape [Cout,Cin,kt,kh,kw]
 *     held in torch.channels_last_3d format, so state_dict keys *and shapes* equal the reference's.
 *   - The stem convs read the reference's NCDHW / NCHW input directly (x_channel_first = 1).
 *   - Every call is asynchronous on `stream` (a hipStream_t); nothing here synchronises the host.
 *   - The library never allocates persistent device memory: scratch is passed in by the caller
 *     (sizes from the *_workspace_bytes queries); torch owns every buffer.
 *   - Return value: 0 = AVID_OK, negative = error; avid_last_error() gives the message
 *     (thread-local).  Kernels never abort.
 */
#ifndef AVID_HIP_H
#define AVID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVID_OK 0
#define AVID_E_BADARG (-1)
#define AVID_E_SHAPE (-2)
#define AVID_E_HIP (-3)
#define AVID_E_UNSUPPORTED (-4)

typedef void* avid_stream_t; /* hipStream_t */

const char* avid_last_error(void);
int avid_version(void);
/* CU count / LDS per CU / arch name of `device` ("gfx950" expected). */
int avid_device_info(int device, int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* Per-kernel timing with HIP events recorded on the launch stream.  enable(1) clears old records and
 * starts bracketing every hot kernel launch with two events; report() synchronises them and writes
 * one line per kernel: "name;launches;total_ms;algorithmic_flops;algorithmic_bytes\n". */
int avid_timing_enable(int on);
int avid_timing_report(char* buf, size_t len);

/* ------------------------------------------------------------------------------------------------
 * Convolution: fp32 in, fp32 out, fp32 accumulation, fp32-accurate products.  Which kernel a layer runs on is the library's
 * business (avid_conv_kernel_name tells): implicit GEMM, Winograd F(2x2,3x3) for the stride-1 3x3 layers, LDS-patch kernels for
 * the stems, a tap-sharing kernel for conv2x's (3,1,1) layers.  Since version 110 most of them assemble every fp32 product
 * from SIX bf16 matrix instructions (v_mfma_f32_32x32x16_bf16 on operands split into three bf16 terms: error against float64
 * at or below the fp32 instruction's, tests/test_gpu_precision.py; a non-finite input becomes NaN where the fp32 instruction
 * would keep an infinity); wino_kernel, wino_wgrad_kernel and the bias / ReLU epilogues of the linear layers issue the fp32
 * instruction v_mfma_f32_32x32x2_f32 itself.
 * Replaces nn.Conv3d / nn.Conv2d / nn.Linear forward+backward:
 *   models/video.py:20, models/network_blocks.py:18,20,35,37,40,42,49, models/audio.py:22,
 *   models/av_wrapper.py:25 (Linear == 1x1x1 conv over a [B,1,1,1,C] tensor).
 * ---------------------------------------------------------------------------------------------- */
typedef struct avid_conv_desc {
  int32_t B, Ti, Hi, Wi, Cin; /* input  [B,Ti,Hi,Wi,Cin] */
  int32_t To, Ho, Wo, Cout;   /* output [B,To,Ho,Wo,Cout] */
  int32_t kt, kh, kw;         /* kernel extent */
  int32_t st, sh, sw;         /* stride */
  int32_t pt, ph, pw;         /* zero padding */
  int32_t x_channel_first;    /* 1: x is [B,Cin,Ti,Hi,Wi] (stems only; Cin in {1,3}) */
} avid_conv_desc;

/* y = conv(x, w) [+ addend] [+ bias] [relu].  addend: [B,To,Ho,Wo,Cout] or NULL (residual add of
 * models/network_blocks.py:59 fused into tmp_conv2/res_conv); bias: [Cout] or NULL.
 * ws: scratch for the split-K partial slabs of small-M layers (may be NULL: single pass, slower).
 * Large (1,3,3) stride-1 layers with <= 256 output channels run as a fused Winograd F(2x2,3x3) kernel (forward and
 * input gradient; csrc/wino.hip, AVID_WINO=0 to switch it off): same contract, results within 1e-6 of the direct
 * form; its transformed weights live in ws, so BatchNorm partials / the fused BatchNorm-backward sums of such a
 * layer need the planned workspace. */
size_t avid_conv_fwd_workspace_bytes(const avid_conv_desc* d);
/* bn_partials (or NULL): BatchNorm partial sums of the output y, [rows][2][Cout] with rows =
 * avid_conv_fwd_stats_rows(d) (0 = this layer cannot produce them), written by the conv epilogue so that
 * avid_bn_fwd_train can skip its statistics pass over y.  Needs bias == NULL and relu == 0. */
int avid_conv_fwd_stats_rows(const avid_conv_desc* d);
/* u (or NULL): for a layer on the Winograd path (v = avid_conv_uses_wino(d, 0) != 0), its weights already transformed by
 * avid_weight_transpose_batched (a descriptor of mode 1 if v == 1, mode 3 if v == 2; current for this w): the call then
 * skips its own transform launch.  For a layer with avid_conv_uses_split(d, 0) != 0 (version >= 120): its mode-5 table
 * (the weights pre-split into bf16 terms).  Ignored by every other layer. */
int avid_conv_fwd(const avid_conv_desc* d, const float* x, const float* w, const float* u, const float* addend,
                  const float* bias, int relu, float* y, float* bn_partials, void* ws, size_t ws_bytes,
                  avid_stream_t stream);

/* The same convolution reading the INPUT of a BatchNorm (+ReLU) instead of its output (version >= 130): x is the tensor the
 * BatchNorm normalises (the previous convolution's output), `in` its saved scale / shift ([Cin] each, avid_bn_fwd_train's
 * save_scale / save_shift) — the kernel applies fma(x, scale[c], shift[c]) (+ max(., 0) if relu) to every element it stages,
 * the expression of avid_bn_fwd_train's own apply pass, bit for bit; padding stays zero.  The normalised tensor is then never
 * written: avid_bn_fwd_train(y = NULL) makes the statistics only.  Replaces the `x = ReLU(bn(x))` round trip through memory
 * between models/network_blocks.py:36 / 41 and :37 / 42 (spt_bn1 -> tmp_conv1, spt_bn2 -> tmp_conv2).  in = NULL: avid_conv_fwd.
 * Only layers for which avid_conv_takes_in_affine() answers 1 (conv2x's temporal layers: tconv64_kernel / twgrad64_kernel, given
 * their pre-split weights as u); anything else returns AVID_E_UNSUPPORTED.  A layer that takes it must be given the same pair
 * in its weight gradient (avid_conv_wgrad_in); its input gradient does not read x. */
typedef struct avid_in_affine {
  const float* scale;
  const float* shift;
  int32_t relu;
} avid_in_affine;
int avid_conv_takes_in_affine(const avid_conv_desc* d);
int avid_conv_fwd_in(const avid_conv_desc* d, const float* x, const avid_in_affine* in, const float* w, const float* u,
                     const float* addend, const float* bias, int relu, float* y, float* bn_partials, void* ws, size_t ws_bytes,
                     avid_stream_t stream);
/* Launches of tconv64_kernel's forward / twgrad64_kernel since the library was loaded that read their input as it is
 * (fused = 0) or applied a BatchNorm to it while staging (fused = 1): tests assert which form ran. */
long long avid_debug_in_affine_launches(int fused);

/* dx = conv_transpose(dy, w) [+ addend].  ws: scratch for the transposed weights (+ split-K slabs).
 * wt (nullable): the weights already repacked as [Cin][taps][Cout] by avid_weight_transpose_batched (mode 0, current
 * for this w); NULL = repack inside the call (one extra small launch per layer).
 * u (nullable, version >= 110: its own argument): for a layer whose input gradient runs on the Winograd path
 * (v = avid_conv_uses_wino(d, 1) != 0) its mode-2 (v == 1) or mode-4 (v == 2) transform from
 * avid_weight_transpose_batched; NULL = transform inside the call.  Each pointer is read only by the path it belongs
 * to: a Winograd-eligible layer that falls through to the implicit GEMM (compact addend) reads wt, never u.
 * A layer with avid_conv_uses_split(d, 1) != 0 (version >= 120) takes its mode-6 table as u and then reads neither wt
 * nor the repack scratch. */
size_t avid_conv_dgrad_workspace_bytes(const avid_conv_desc* d);
/* bn (or NULL): dx is the COMPLETE gradient of the output of a training-mode BatchNorm(+ReLU) whose input was
 * bn->x (shape of dx) — the usual conv <- ReLU <- BN chain of models/network_blocks.py:30-60.  The dgrad epilogue
 * then also writes that BatchNorm's backward partial sums ([rows][2][Cin]: sum dy_m, sum dy_m * xhat with dy_m the
 * gradient masked by the recomputed ReLU; rows = avid_conv_dgrad_bn_rows(d), 0 = this layer cannot), which
 * avid_bn_bwd takes instead of making its own pass over dy and x.  scale / shift / mean / invstd: the tensors
 * avid_bn_fwd_train saved. */
typedef struct {
  const float* x;
  const float* scale;
  const float* shift;
  const float* mean;
  const float* invstd;
  int32_t relu;
  float* partials;
} avid_bn_bwd_fuse;
int avid_conv_dgrad_bn_rows(const avid_conv_desc* d);
/* addend_stride (nullable = {1,1,1}): strides (1 or 2 per axis) of a COMPACT addend — the gradient of the
 * sub-sampled view x[:, ::st, ::sh, ::sw] that a block's 1x1x1 strided residual convolution reads
 * (models/network_blocks.py:47-51,58), shape [B][ceil(Ti/st)][ceil(Hi/sh)][ceil(Wi/sw)][Cin]: it is added at the
 * positions divisible by the strides only, instead of being scattered into a dx-shaped tensor of mostly zeros
 * first.  Strided layers on the persistent kernel only. */
int avid_conv_dgrad(const avid_conv_desc* d, const float* dy, const float* w, const float* wt, const float* u,
                    const float* addend, const int32_t* addend_stride, float* dx, const avid_bn_bwd_fuse* bn,
                    void* ws, size_t ws_bytes, avid_stream_t stream);

/* One launch that repacks every conv / linear weight of a model for its input-gradient pass:
 * w[Cout][taps][Cin] -> wt[Cin][taps][Cout] for each descriptor.  descs_dev: n descriptors in DEVICE
 * memory (they do not change between steps); max_elems = max over descriptors of Cout*taps*Cin.
 * Call it once after each optimizer step (the autograd.Function of torch's conv does this per call). */
typedef struct avid_wt_desc {
  const float* w;
  float* wt;
  int32_t Cout, ntaps, Cin;
  int32_t mode; /* 0: wt[Cin][taps][Cout]; 3x3 layers on the Winograd path (ntaps = 9): wt = the 16 x Cout x Cin transformed
                   weights U in the operand-fragment order (and element format: fp32, or three bf16 terms) of the kernel that
                   will read them, a buffer of 96 * Cout * Cin bytes whatever the format — 1 / 3: for the forward
                   (pass it as `u` to avid_conv_fwd), 2 / 4: with flipped taps and swapped channel roles for the input
                   gradient (pass it as `u` to avid_conv_dgrad); 1, 2 for wino_kernel, 3, 4 for wino2_kernel
                   (avid_conv_uses_wino says which of the two a layer runs on);
                   5 / 6: the weights (5) / their [Cin][taps][Cout] transpose (6) split into three bf16 terms (hi, mid, lo:
                   w = hi + mid + lo to fp32 accuracy) in the operand-fragment order of igemm_pk_kernel's 128 x 64 tile,
                   6 bytes per weight (avid_conv_split_bytes) — pass it as `u` to avid_conv_fwd (5) / avid_conv_dgrad (6)
                   of a layer for which avid_conv_uses_split answers 1; needs Cout % 64 == 0 and Cin % 32 == 0 (5) or
                   Cin % 64 == 0 and Cout % 32 == 0 (6) */
} avid_wt_desc;
int avid_weight_transpose_batched(int n, const avid_wt_desc* descs_dev, int64_t max_elems, avid_stream_t stream);
/* The same for ONE descriptor in HOST memory (passed to the kernel by value: nothing to upload, legal inside a stream
 * capture) — what a per-layer caller without a per-step table uses. */
int avid_weight_transform(const avid_wt_desc* desc, avid_stream_t stream);

/* Which kernel instantiation a descriptor dispatches to (which: 0 fwd, 1 dgrad, 2 wgrad), e.g.
 * "igemm_kernel<4,1,1,2,1>" — lets bench.py attribute HIP-event timings to rocprofv3 kernel names. */
int avid_conv_kernel_name(const avid_conv_desc* d, int which, char* buf, int len);
/* Nonzero if this layer's forward (which 0) / input gradient (1) / weight gradient (2) runs on the Winograd kernels;
 * for which 0 / 1: 1 = wino_kernel, 2 = wino2_kernel (layers with >= 1.5 rounds of 64-tile units for the CUs). */
int avid_conv_uses_wino(const avid_conv_desc* d, int which);
/* Nonzero if this layer's forward (which 0) / input gradient (1) runs on the implicit-GEMM tile that reads its weights
 * pre-split into bf16 terms: `u` of avid_conv_fwd / avid_conv_dgrad is then the avid_wt_desc mode 5 / 6 table of this
 * layer (avid_conv_split_bytes bytes).  Without it (u = NULL) the same layer runs with the fp32 matrix instruction on the
 * weights as they are (and avid_conv_dgrad on wt): both forms are fp32-accurate, their roundings differ. */
int avid_conv_uses_split(const avid_conv_desc* d, int which);
size_t avid_conv_split_bytes(const avid_conv_desc* d);
/* Launches of igemm_pk_kernel that consumed a pre-split weight table since the library was loaded (tests assert that
 * avid_conv_uses_split and the kernel that ran agree; the HIP-event log names both forms igemm_pk_kernel<...>). */
long long avid_debug_presplit_launches(void);

/* Dispatch switches of the Winograd path (defaults: on, layers of >= 6000 output pixels, <= 256 output channels and
 * pixels x max(Cin, Cout) >= 1e6; environment AVID_WINO / AVID_WINO_MIN_M / AVID_WINO_MAXC / AVID_WINO_MIN_WORK — an
 * explicit min_pixels here drops the pixels x channels rule).  A negative argument returns that switch to its
 * environment / default value.  Changes what avid_conv_fwd / avid_conv_dgrad / avid_conv_wgrad dispatch to and what
 * the *_workspace_bytes / *_rows queries answer from the next call on: callers that cache those answers per layer
 * must drop them (ops.wino_configure does).  Parity tests use it to send the reference-generated small fixtures
 * (tests/golden) through the Winograd kernels (models/network_blocks.py:35,40 at 2 clips). */
int avid_wino_configure(int enabled, int64_t min_pixels, int max_channels);
/* Which of the two forward / input-gradient kernels a Winograd layer takes: wino2_kernel (one workgroup per CU, 64-tile
 * units, one instruction stream per SIMD) where the layer has at least min_rounds_x10 / 10 rounds of units for the CUs
 * (default 15, environment AVID_WINO2_MIN_ROUNDS; AVID_WINO2=0: never), wino_kernel otherwise.  0 sends every Winograd
 * layer through wino2_kernel (tests), negative restores the environment / default. */
int avid_wino2_configure(int min_rounds_x10);
/* wino2_kernel's two forms: 1 (default; environment AVID_WINO2_PRE) = wino2p_kernel — the transformed input V is split into its
 * three bf16 terms ONCE, by the thread that transforms it, and kept in LDS as the fragments the products read (a ring of three
 * half-stages of 8 transform points); 0 = every product wave splits the fragments it multiplies.  Same split, same products,
 * same order: bit-identical results (tests/test_gpu_ops.py::test_wino2_presplit_is_bit_identical).  Negative: environment /
 * default.  Takes effect from the next launch. */
int avid_wino2_pre_configure(int on);
/* wgrad_group_kernel's two forms (the grouped weight gradients of the 128-wide layers, backward of models/network_blocks.py:37-49,
 * models/audio.py:27-30): 1 (default; environment AVID_WGRAD_PRE) = every dy / x fragment is split into its three bf16 terms ONCE, by
 * the thread that stages it (one k-step of 16 pixel rows per LDS stage, fragments kept as three planes); 0 = each of the two waves
 * that share a fragment gathers and splits it itself.  Same split, same products, same order: bit-identical results
 * (tests/test_gpu_ops.py::test_grouped_weight_gradients_presplit_is_bit_identical).  Negative: environment / default. */
int avid_wgrad_pre_configure(int on);
/* The video stem's forward (models/video.py:20) in its split-bf16 form: 1 (default; environment AVID_STEM_FWD_PRE) =
 * stem_fwd3p_kernel — the input patch is split into its three bf16 terms ONCE, when it is committed to LDS (three planes), and a
 * lane's operand fragment is eight consecutive patch columns of one row, read as it is; 0 = stem_fwd3_kernel (fp32 patch, every lane
 * gathers and splits its taps per k-step).  Same products per (row, tap), assigned to other k indices of the matrix instruction:
 * results agree to rounding (tests/test_gpu_ops.py::test_stem_fwd_presplit_patch).  Inputs whose split patch does not fit in LDS
 * (224 x 224) keep stem_fwd3_kernel.  Negative: environment / default. */
int avid_stem_fwd_pre_configure(int on);

/* Which launches take tconv64_kernel — the (3,1,1) stride-1 pad-1 layers with 64 -> 64 channels and 8 frames
 * (models/network_blocks.py:37,42 in conv2x), forward and input gradient, when the layer's pre-split weights are passed
 * (`u`): every input row staged once for its three taps, the weights resident in LDS.  0: never (igemm_pk_kernel as
 * before); 1: layers with at least three rounds of 32-position tiles for the CUs (the default; environment AVID_TCONV);
 * 2: every layer it can run (tests send small fixtures through it); anything else: back to the environment / default.
 * Returns the mode in force.  Does not change any *_workspace_bytes / *_rows answer. */
int avid_tconv_configure(int mode);

/* CU budget of the persistent kernels.  igemm_pk_kernel, the stem kernels, the Winograd kernels and the grouped weight
 * gradient size their grids for — and deal their tiles over — every CU of the device; a workgroup that cannot be placed
 * (another kernel's long-lived workgroups hold the CU: RCCL's, when the gradient all-reduce of
 * utils/main_utils.py:112 runs beside the backward pass) costs such a kernel a whole extra round.  cus > 0: plan for that
 * many CUs (rounded down to a multiple of 8 = whole CUs per XCD, at least 8); cus <= 0: every CU (the default; the
 * environment variable AVID_CU_RESERVE = n starts the process at device CUs - n).  Returns the effective count, as does
 * avid_cu_budget().  Like the Winograd switches it changes what the *_workspace_bytes / *_rows queries answer: callers
 * that cache them must drop the cache (ops.set_cu_budget does).  Results stay deterministic at any budget; they are
 * bit-identical across budgets only for layers whose K-split / slab plan does not depend on the CU count. */
int avid_set_cu_budget(int cus);
int avid_cu_budget(void);

/* dw[Cout][kt][kh][kw][Cin] = sum_m dy[m][:]^T x_col[m][:]  (deterministic split-M + tree reduce). */
size_t avid_conv_wgrad_workspace_bytes(const avid_conv_desc* d);
int avid_conv_wgrad(const avid_conv_desc* d, const float* x, const float* dy, float* dw, void* ws,
                    size_t ws_bytes, avid_stream_t stream);
/* ... with x the input of the BatchNorm (+ReLU) whose output the layer convolved (avid_conv_fwd_in's twin; in = NULL:
 * avid_conv_wgrad).  Needs the layer's workspace. */
int avid_conv_wgrad_in(const avid_conv_desc* d, const float* x, const avid_in_affine* in, const float* dy, float* dw, void* ws,
                       size_t ws_bytes, avid_stream_t stream);

/* The weight gradients of up to 12 layers in ONE persistent launch (+ one grouped reduce): the small layers of a stage
 * (conv3x-5x temporal / strided / residual convolutions, audio blocks, heads — backward of models/network_blocks.py:
 * 18,20,37,42,49 and models/av_wrapper.py:25).  A launch per layer has to K-split every layer until it fills the chip
 * alone; in a group the items of all layers share one size and most layers need no split (their dw is written
 * directly).  Only layers for which avid_conv_wgrad_groupable() is 1 (128-wide dw tiles, not a stem, not on the
 * Winograd path).  Every dw is fully written (dead temporal taps as zeros).  Results equal avid_conv_wgrad's up to the
 * summation order over pixel chunks. */
typedef struct avid_wgrad_item {
  avid_conv_desc d;
  const float* x;
  const float* dy;
  float* dw;
} avid_wgrad_item;
int avid_conv_wgrad_groupable(const avid_conv_desc* d);
size_t avid_conv_wgrad_group_workspace_bytes(int n, const avid_wgrad_item* items);
int avid_conv_wgrad_group(int n, const avid_wgrad_item* items, void* ws, size_t ws_bytes, avid_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm (train / eval) over a channels-last [M, C] view, fused ReLU.
 * Replaces nn.BatchNorm3d/2d + nn.ReLU: models/network_blocks.py:19,21,36,38,41,43,54-59,
 * models/video.py:21-22, models/audio.py:23-24.
 * ---------------------------------------------------------------------------------------------- */
size_t avid_bn_workspace_bytes(int64_t M, int C);
/* Train: batch mean / biased var -> save_mean, save_invstd [C]; running stats updated in place
 * (momentum, unbiased var); y = [relu](fma(x, scale, shift)) with scale = gamma * invstd,
 * shift = beta - mean * scale, both also saved ([C]) so backward can recompute the ReLU mask bit-exactly.
 * num_batches_tracked: device int64 counter bumped by one (nn.BatchNorm's buffer), or NULL.
 * partials / nparts: [nparts][2][C] partial sums of x from avid_conv_fwd (skips the statistics pass), or NULL / 0.
 * y = NULL (version >= 130): statistics only — the saved vectors and the running statistics are made, the normalised tensor
 * is not; its consumer applies the map while it stages x (avid_conv_fwd_in / avid_conv_wgrad_in). */
int avid_bn_fwd_train(int64_t M, int C, const float* x, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, int relu,
                      float* y, float* save_mean, float* save_invstd, float* save_scale,
                      float* save_shift, int64_t* num_batches_tracked, const float* partials, int nparts,
                      void* ws, size_t ws_bytes, avid_stream_t stream);
/* Eval: uses running stats.  save4 (nullable, [4][C]): a backward will follow (frozen-BatchNorm fine-tuning) —
 * mean (= running_mean) / invstd / scale / shift are written there and y = fma(x, scale, shift), the expression
 * avid_bn_bwd(frozen = 1) recomputes the ReLU mask with. */
int avid_bn_fwd_eval(int64_t M, int C, const float* x, const float* gamma, const float* beta,
                     const float* running_mean, const float* running_var, float eps, int relu,
                     float* y, float* save4, avid_stream_t stream);
/* Backward of train-mode BN(+ReLU).  The ReLU mask is fma(x, scale, shift) > 0 recomputed from the conv
 * output x (the forward's exact expression), so the saved activation is not re-read: 2 + 3 passes.
 * partials / nparts: the partial sums the dgrad that produced dy already made (avid_conv_dgrad's `bn`), or
 * NULL / 0 — then the 2-read statistics pass runs here.
 * frozen != 0: backward of an EVAL-mode BatchNorm (statistics are constants): dx = gamma * invstd * dy_m,
 * dgamma = sum dy_m * xhat, dbeta = sum dy_m. */
int avid_bn_bwd(int64_t M, int C, const float* x, const float* dy, const float* gamma,
                const float* save_mean, const float* save_invstd, const float* save_scale,
                const float* save_shift, int relu, float* dx, float* dgamma, float* dbeta,
                const float* partials, int nparts, int frozen, void* ws, size_t ws_bytes,
                avid_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling.  MaxPool3d((1,3,3),(1,2,2),(0,1,1)) — models/video.py:23;  AdaptiveMaxPool{3,2}d(1) —
 * models/video.py:41, models/audio.py:31.  Ties: first maximum in (t,h,w) scan order (ATen CPU).
 * ---------------------------------------------------------------------------------------------- */
/* Stem tail fused (models/video.py:21-23): y = maxpool(relu(bn_train(x))) with x [B,T,H,W,C] the stem conv
 * output, y / argmax [B,T,Ho,Wo,C]; the normalised activation is never materialised.  Backward rebuilds the
 * un-pooled gradient from (dy, argmax) on the fly.  Same saved tensors / workspace / `partials` (the stem
 * convolution's own BatchNorm partial sums, avid_conv_fwd) as avid_bn_fwd_train. */
int avid_bn_relu_maxpool_fwd(int B, int T, int H, int W, int C, const float* x, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                             float* y, uint8_t* argmax, float* save_mean, float* save_invstd, float* save_scale,
                             float* save_shift, int64_t* num_batches_tracked, const float* partials, int nparts,
                             void* ws, size_t ws_bytes, avid_stream_t stream);
int avid_bn_relu_maxpool_bwd(int B, int T, int H, int W, int C, const float* x, const float* dy,
                             const uint8_t* argmax, const float* gamma, const float* save_mean,
                             const float* save_invstd, const float* save_scale, const float* save_shift, float* dx,
                             float* dgamma, float* dbeta, void* ws, size_t ws_bytes, avid_stream_t stream);
int avid_maxpool_hw3s2_fwd(int B, int T, int H, int W, int C, const float* x, float* y,
                           uint8_t* argmax, avid_stream_t stream);
int avid_maxpool_hw3s2_bwd(int B, int T, int H, int W, int C, const float* dy,
                           const uint8_t* argmax, float* dx, avid_stream_t stream);
/* x [B, S, C] -> y [B, C], argmax [B, C] (int32 position in S). */
int avid_global_maxpool_fwd(int B, int S, int C, const float* x, float* y, int32_t* argmax,
                            avid_stream_t stream);
int avid_global_maxpool_bwd(int B, int S, int C, const float* dy, const int32_t* argmax, float* dx,
                            avid_stream_t stream);

/* Small helpers for the projection heads (models/av_wrapper.py:23-29). */
int avid_relu_bwd(int64_t n, const float* y, const float* dy, float* dx, avid_stream_t stream);
int avid_colsum(int64_t M, int C, const float* x, float* out, avid_stream_t stream); /* bias grad */

/* ------------------------------------------------------------------------------------------------
 * Criterion path (criterions/avid.py, criterions/nce.py, utils/alias_method.py).
 * ---------------------------------------------------------------------------------------------- */
/* F.normalize(x, p=2, dim=1) — criterions/avid.py:52-53.  norm_out [bs] = max(||x||, 1e-12). */
int avid_l2norm_fwd(int bs, int D, const float* x, float* y, float* norm_out, avid_stream_t stream);
int avid_l2norm_bwd(int bs, int D, const float* y, const float* norm, const float* dy, float* dx,
                    avid_stream_t stream);

/* AliasMethod.draw + "avoid self" — utils/alias_method.py:56-71, criterions/avid.py:82-86.
 * out[i] = alias_select(prob, alias, kk_i, u_i) with (kk_i, u_i) from Philox4x32-10
 * (counter = (i, offset), key = seed); if y != NULL: out[i] += (out[i] >= y[i / per_row]).
 * uniform != 0 short-circuits the table lookup for the all-ones table (prob == 1, alias == 0).
 * offset_dev != NULL: the offset is read from device memory instead of the by-value argument. */
int avid_alias_draw(int64_t n, int64_t K, const float* prob, const int64_t* alias, int uniform,
                    uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const int64_t* y,
                    int64_t per_row, int64_t* out, avid_stream_t stream);
/* *counter += inc on the stream.  offset_dev (above) / step_dev (avid_adam_flat) point at such device
 * counters so a captured hipGraph of the whole step advances its RNG stream / Adam step on replay. */
int avid_counter_add(uint64_t* counter, uint64_t inc, avid_stream_t stream);

/* Device-side index errors.  The reference's gathers / index_copy_ raise an IndexError for an id outside
 * [0, N) (criterions/avid.py:57-62,124; avid_cma.py:199).  A kernel cannot raise: the three entry points below
 * take `err` (nullable), one int32 word in device memory that they OR a code into while staying inside the
 * table themselves (gathers clamp, the update skips the sample).  The host polls the word without a
 * synchronisation (avid_hip/ops.py: DeviceErrors) and raises IndexError at most one step late. */
#define AVID_DEVERR_BANK_INDEX 1   /* avid_bank_scores_fwd: idx outside [0, N) */
#define AVID_DEVERR_UPDATE_INDEX 2 /* avid_bank_update: y outside [0, N) */
#define AVID_DEVERR_CMA_INDEX 4    /* avid_cma_negatives: y outside [0, N) */

/* scores[b][j] = <bank[idx[b][j]], emb[b]> * inv_T — the gather + bmm of criterions/avid.py:57-71.
 * idx [bs][R] int64, bank [N][D], emb [bs][D], D in {64,128,256,512}.  rows_out (nullable,
 * [bs][R][D]) receives a snapshot of the gathered rows: the reference's autograd keeps the
 * PRE-update rows for backward (the bank is updated inside forward, avid.py:78). */
int avid_bank_scores_fwd(int bs, int R, int D, int64_t N, const int64_t* idx, const float* bank,
                         const float* emb, float inv_T, float* scores, float* rows_out, int32_t* err,
                         avid_stream_t stream);
/* demb[b] (+)= inv_T * sum_j dscores[b][j] * row(b,j)  (autograd of torch.bmm, avid.py:66), where
 * row(b,j) = rows[b][j] if rows != NULL (the snapshot) else bank[idx[b][j]]. */
int avid_bank_scores_bwd(int bs, int R, int D, int64_t N, const float* rows, const int64_t* idx,
                         const float* bank, const float* dscores, float inv_T, int accumulate,
                         float* demb, avid_stream_t stream);

/* NCE — criterions/nce.py:38-58.  Score matrices are row-major with leading dimension ld_* so the
 * positive / negative column blocks of one [bs][P+K] score buffer can be passed without a copy.
 * mean_exp: out[0] = mean(exp(s[rows][cols])) (first-call Z, nce.py:27).
 * fwd: loss[0] (+)= scale * mean_b( -mean_p log Pmt - sum_k log Pon ); Z is read from device memory.
 * bwd: dpos [bs][P], dneg [bs][K] (dense) = dloss[0] * scale * dL/ds. */
int avid_mean_exp(int rows, int cols, int ld, const float* s, float* out, avid_stream_t stream);
size_t avid_nce_workspace_bytes(void);
/* ws: optional scratch of avid_nce_workspace_bytes(), ZERO-FILLED ONCE by the caller and then left to this op
 * (per-block partial sums + a ticket counter the kernel re-arms itself); NULL -> single-block kernel. */
int avid_nce_fwd(int bs, int P, int K, const float* spos, int ld_pos, const float* sneg, int ld_neg,
                 const float* Z, float scale, int accumulate, float* loss, void* ws, size_t ws_bytes,
                 avid_stream_t stream);
int avid_nce_bwd(int bs, int P, int K, const float* spos, int ld_pos, const float* sneg, int ld_neg,
                 const float* Z, const float* dloss, float scale, float* dpos, float* dneg,
                 avid_stream_t stream);

/* update_memory — criterions/avid.py:118-129: bank[y[i]] = normalize(m*bank[y[i]] + (1-m)*emb[i]).
 * Duplicate ids: the LAST occurrence wins (the reference's index_copy_ order is unspecified). */
int avid_bank_update(int B, int D, int64_t N, float* bank, const int64_t* y, const float* emb,
                     float momentum, int32_t* err, avid_stream_t stream);

/* Fused cross-modal criterion for the steady state (partition constant Z already frozen, criterions/nce.py:22-24):
 * criterions/avid.py:52-71 (F.normalize of both embeddings; gather of row y and of the K negative rows idx from BOTH
 * banks; bmm / T: v2a = video embedding . audio bank, a2v = audio embedding . video bank) + criterions/nce.py:38-58 for
 * both score sets + the backward of all of it with respect to the two raw embeddings, formed in the same pass while a
 * gathered row is in registers (the banks are updated before backward runs, avid.py:78 — the unfused ops keep a
 * snapshot of the gathered rows for that; here none is needed).
 *   v_emb, a_emb [bs][128] raw embeddings; y [bs], idx [bs][K] int64; bank_v = view1_mem, bank_a = view2_mem;
 *   v_hat, a_hat [bs][128]: the normalised embeddings (input of the bank update / its all-gather);
 *   losses [4]: L_v2a, L_a2v, L_v2a / 2 + L_a2v / 2 (avid.py:221-222), coeff * that (the total loss);
 *   dv, da [bs][128]: d(total) / d(v_emb), d(total) / d(a_emb) for an upstream gradient of 1.
 * ws: avid_xmodal_fused_workspace_bytes(bs, K) bytes, ZERO-FILLED ONCE by the caller and then owned by this op (device
 * tickets that re-arm themselves); one per stream that may run the op concurrently.  Fixed summation order, no
 * floating-point atomics.  D must be 128.  err: device error word (AVID_DEVERR_BANK_INDEX), may be NULL. */
size_t avid_xmodal_fused_workspace_bytes(int bs, int K);
int avid_xmodal_fused(int bs, int K, int D, int64_t N, const float* v_emb, const float* a_emb, const int64_t* y,
                      const int64_t* idx, const float* bank_v, const float* bank_a, float inv_T, const float* Z,
                      float coeff, float* v_hat, float* a_hat, float* losses, float* dv, float* da, void* ws,
                      size_t ws_bytes, int32_t* err, avid_stream_t stream);
/* The same for the AVID+CMA criterion in its stock form — cross-modal instance terms + within-modal positive terms
 * (criterions/avid_cma.py:150-194, 338-358 with xModalInst and wModalPos on; Z frozen): rows gathered once from both
 * banks = the sample's own row y | its P positives pos[bs][P] (avid_cma_negatives) | the K negatives idx[bs][K]; four
 * score sets per row pair (inst-v2a / inst-a2v over self + K negatives, pos-v2v / pos-a2a over the P positives — mean
 * over P, criterions/nce.py:44 — and the first Kw negatives, avid_cma.py:184-190), their NCE terms, and the gradient of
 *   total = coeff_inst (L_inst-v2a + L_inst-a2v) / 2 + coeff_pos (L_pos-v2v + L_pos-a2a) / 2
 * with respect to both raw embeddings (each receives gradient through rows of BOTH banks).
 *   losses [8]: L_inst-v2a, L_inst-a2v, L_pos-v2v, L_pos-a2a, the two group means, the total, (unused).
 * Everything else as avid_xmodal_fused (ws: avid_cma_fused_workspace_bytes(bs, P, K), zero-filled once). */
size_t avid_cma_fused_workspace_bytes(int bs, int P, int K);
int avid_cma_fused(int bs, int P, int K, int Kw, int D, int64_t N, const float* v_emb, const float* a_emb, const int64_t* y,
                   const int64_t* pos, const int64_t* idx, const float* bank_v, const float* bank_a, float inv_T,
                   const float* Z, float coeff_inst, float coeff_pos, float* v_hat, float* a_hat, float* losses, float* dv,
                   float* da, void* ws, size_t ws_bytes, int32_t* err, avid_stream_t stream);
/* avid_bank_update for both banks in one launch (criterions/avid.py:118-129): bank0 <- emb0 (momentum0), bank1 <- emb1; bit-identical to two avid_bank_update calls. */
int avid_bank_update2(int B, int D, int64_t N, float* bank0, float* bank1, const int64_t* y, const float* emb0,
                      const float* emb1, float momentum0, float momentum1, int32_t* err, avid_stream_t stream);

/* memory_sampling remap — criterions/avid_cma.py:196-209.  positive_set int32 [N][P] (rows sorted);
 * pos_out [bs][P] = positive_set[y];  neg_out[b][k] = r + #{j : r >= pos_j - j}, r = rand_idx[b][k]. */
int avid_cma_negatives(int bs, int K, int P, int64_t N, const int32_t* positive_set, const int64_t* y,
                       const int64_t* rand_idx, int64_t* pos_out, int64_t* neg_out, int32_t* err,
                       avid_stream_t stream);

/* CMA correspondence search — criterions/avid_cma.py:42-73: for queries q in [q0, q0+nq):
 * sim = combine(V V[q]^T, A A[q]^T) (kind 0 consensus=min, 1 union=max, 2 video, 3 audio);
 * top-(pos_k+1) by similarity, drop the best (self), sort ascending -> out [nq][pos_k] int32. */
size_t avid_cma_topk_workspace_bytes(int64_t N, int nq, int pos_k);
/* fallbacks (nullable): device int32 counter, += 1 when this batch overflowed the threshold filter's candidate
 * lists (heavy score ties) and was redone by the exact scan — diagnostics, the result is exact either way. */
int avid_cma_topk(int64_t N, int D, const float* view1, const float* view2, int64_t q0, int nq,
                  int pos_k, int kind, int32_t* out, int32_t* fallbacks, void* ws, size_t ws_bytes,
                  avid_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam semantics (L2 weight decay folded into the gradient; bias-corrected)
 * over one flat fp32 buffer — utils/main_utils.py:250-261.
 * ---------------------------------------------------------------------------------------------- */
/* lr_dev (nullable): the learning rate is read from this device float instead of `lr` — a captured hipGraph
 * freezes by-value arguments, a scheduler then writes the device word (utils/main_utils.py:258 MultiStepLR). */
int avid_adam_flat(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int64_t step, const int64_t* step_dev,
                   const float* lr_dev, float grad_scale, avid_stream_t stream);

/* Video clip front end: frames [B][T][H][W][3] uint8 (what the decoder / augmentation hands over)
 * -> out [B][3][T][H][W] fp32 = ((u / 255) - mean[c]) / std[c], the reference's ClipToTensor + Normalize
 * (utils/videotransforms/volume_transforms.py:14-66, tensor_transforms.py:13-37; datasets/preprocessing.py:45-48)
 * in the same fp32 operation order (bit-identical).  mean3 / std3: HOST pointers to 3 floats. */
int avid_clip_normalize(int B, int T, int H, int W, const uint8_t* frames, const float* mean3, const float* std3,
                        float* out, avid_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Audio front end on the GPU — datasets/preprocessing.py:158-186 LogSpectrogram:
 * out[b][0][t][f] = z-score( top_db-floored dB( bin-pair mean( |STFT(sig[b])|^2 ) ) ), STFT = librosa.stft
 * defaults (centred / reflect-padded frames, periodic Hann, n_stft = 2 * n_fft, hop samples).
 * sig [B][L] mono fp32; out [B][1][T][n_stft/4 + 1], T <= 1 + L/hop (the reference truncates to
 * int(duration * rate) frames before the dB floor).  basis: avid_logspec_basis_floats(n_stft) floats filled
 * once by avid_logspec_basis.  mean / std: [n_stft/4 + 1] or both NULL.
 * ---------------------------------------------------------------------------------------------- */
size_t avid_logspec_basis_floats(int n_stft);
int avid_logspec_basis(int n_stft, float* basis, avid_stream_t stream);
size_t avid_logspec_workspace_bytes(int B, int n_stft, int T);
int avid_logspec(int B, int L, const float* sig, int n_stft, int hop, int T, const float* basis,
                 const float* mean, const float* std, float top_db, float* out, void* ws, size_t ws_bytes,
                 avid_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Collectives: NOT part of this library.  SURVEY.md 8(b) sketched avid_comm_init / avid_allreduce_f32 /
 * avid_allgather_bytes / avid_broadcast; they were deliberately not built.  The reference itself speaks to
 * torch.distributed (utils/distributed_utils.py:12-19, utils/main_utils.py:105-117: DistributedDataParallel;
 * criterions/avid.py:99-111, criterions/nce.py:27-33), whose "nccl" backend IS RCCL on ROCm — a second communicator
 * behind this ABI would duplicate rendezvous, error handling and the process group the driver already owns.
 * What the step needs from the collectives' side lives in the binding: avid-cma_amd/avid_hip/parallel.py (bucketed
 * all-reduce over one flat gradient buffer issued on a placed stream, one fused bank all-gather, one flat buffer
 * broadcast), with avid_stream_wait / avid_probe_spin below as the only native pieces (stream ordering and placement).
 * ---------------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------------
 * Launch programs: the host side of a whole forward / backward pass as ONE call.
 *
 * The reference drives its step from Python, one ATen call per layer (main-avid.py:155-180 ->
 * models/av_wrapper.py:50-61 -> models/video.py:44-54, models/audio.py:33-44, models/network_blocks.py:23-27,
 * 52-60, and autograd's mirror image of that for loss.backward()).  A binding that does the same with the
 * entry points above issues ~330 launches per step through its interpreter.  A program is that launch
 * sequence compiled once per (model, input shape): an array of avid_instr records — opcode, stream index,
 * geometry, tensor references — that avid_program_run() walks, calling the SAME entry points above with the
 * SAME arguments (results are bit-identical to issuing the calls one by one), plus the cross-stream
 * dependencies (AVID_OP_WAIT) that keep the four-stream arrangement of the step (two towers, each with a
 * trailing weight-gradient stream).
 *
 * A tensor reference is (slot, byte offset): slots[] is an array of base pointers the caller fills for each
 * run (activation arenas, the parameter / gradient / BatchNorm-buffer tensors, the inputs), so the compiled
 * program never holds an address and the caller's allocator stays in charge.  slot < 0 = NULL.
 * The program owns nothing on the device; the only state of the executor is a pool of hipEvents for the waits.
 * ---------------------------------------------------------------------------------------------- */
typedef struct avid_ref {
  int32_t slot;
  int32_t reserved;
  int64_t off;
} avid_ref;

enum {
  AVID_OP_NOP = 0,
  AVID_OP_WAIT = 1,          /* stream i[0] waits for everything issued so far on stream i[1] (event record + wait) */
  AVID_OP_MEMSET0 = 2,       /* t0 <- zeros, n[0] bytes */
  AVID_OP_CONV_FWD = 3,      /* d; t: x w u addend bias y bn_partials [in_s4]; i0 relu, i1 input BatchNorm (0 none, 1 affine, 2 + ReLU:
                                x is then that BatchNorm's input and t7 its saved [4][C] vectors, i2 = C: avid_conv_fwd_in) */
  AVID_OP_CONV_DGRAD = 4,    /* d; t: dy w wt u addend dx bn.x bn.scale bn.shift bn.mean bn.invstd bn.partials;
                                i0..2 addend strides (0 = dense addend), i3 bn.relu, i4 bn present */
  AVID_OP_CONV_WGRAD = 5,    /* d; t: x dy dw [in_s4]; i0 input BatchNorm as AVID_OP_CONV_FWD's i1 (t3, i1 = C: avid_conv_wgrad_in) */
  AVID_OP_WGRAD_GROUP = 6,   /* i0 = n, followed by n AVID_OP_WGRAD_ITEM records (d; t: x dy dw) */
  AVID_OP_WGRAD_ITEM = 7,
  AVID_OP_BN_FWD = 8,        /* n0 M; i0 C, i1 relu, i2 nparts; f0 momentum, f1 eps;
                                t: x gamma beta running_mean running_var y save4 counter partials
                                (save4 = mean | invstd | scale | shift, C floats each) */
  AVID_OP_BN_BWD = 9,        /* n0 M; i0 C, i1 relu, i2 nparts, i3 frozen; t: x dy gamma save4 dx dgamma dbeta partials */
  AVID_OP_BN_POOL_FWD = 10,  /* i0..4 B T H W C, i5 nparts; f0 momentum, f1 eps;
                                t: x gamma beta running_mean running_var y argmax save4 counter partials */
  AVID_OP_BN_POOL_BWD = 11,  /* i0..4 B T H W C; t: x dy argmax gamma save4 dx dgamma dbeta */
  AVID_OP_GPOOL_FWD = 12,    /* i0 B, i1 S, i2 C; t: x y argmax */
  AVID_OP_GPOOL_BWD = 13,    /* i0 B, i1 S, i2 C; t: dy argmax dx */
  AVID_OP_RELU_BWD = 14,     /* n0 elements; t: y dy dx */
  AVID_OP_COLSUM = 15,       /* n0 M; i0 C; t: x out */
  AVID_OP_WT_BATCH = 16,     /* i0 descriptors, n0 max_elems; t: table (avid_wt_desc[] in device memory) */
  AVID_OP_ADAM = 17,         /* n0 elements; f0 lr, f1 beta1, f2 beta2, f3 eps, f4 weight_decay, f5 grad_scale; n1 step;
                                t: p g m v step_dev lr_dev (step_dev, if given, is advanced by one first) */
  AVID_OP_COUNT_
};

#define AVID_INSTR_REFS 12
typedef struct avid_instr {
  int32_t op;
  int32_t stream; /* index into the run's stream table (ignored by AVID_OP_WAIT) */
  int32_t mark;   /* free for the caller (segment labels) */
  int32_t reserved;
  avid_conv_desc d;
  int32_t i[6];
  int64_t n[2];
  float f[6];
  avid_ref t[AVID_INSTR_REFS];
} avid_instr;

/* Scratch of one stream of a run (what the *_workspace_bytes queries ask for, maximum over the stream's records). */
typedef struct avid_stream_ws {
  void* ptr;
  size_t bytes;
} avid_stream_ws;

/* One single-wave kernel that spins for about `us` microseconds on `stream`: the probe with which a binding finds out
 * which of its streams the hardware serialises (streams that share a hardware queue, or queues that share a dispatch
 * pipe: 4 pipes serve GPU_MAX_HW_QUEUES queues) before it places the step's four streams — avid_hip/streams.py. */
int avid_probe_spin(int us, avid_stream_t stream);
/* The shader clock while other work runs: one wave on `stream` spins for `us` microseconds of the constant 100 MHz clock
 * and writes out2[0] = shader-clock cycles elapsed (s_memtime), out2[1] = 100 MHz ticks elapsed (device int64 x 2):
 * GHz = out2[0] / out2[1] / 10.  bench.py runs it beside the timed region (roofline.shader_clock_ghz). */
int avid_clock_probe(int us, long long* out2, avid_stream_t stream);
/* Everything issued to `waiter` after this call runs behind everything issued to `waited` before it (one event of the
 * executor's pool: record + wait) — what AVID_OP_WAIT does inside a program, for a binding that orders its collectives'
 * stream behind the streams that produced a gradient bucket.
 * The event pool's contract: 64 events per device, drawn round-robin through an atomic index by avid_stream_wait and by
 * every AVID_OP_WAIT record (the forward's thread and autograd's backward thread may both draw).  An event is re-recorded
 * after 64 further draws whether or not the GPU has passed the earlier wait: that is safe because hipStreamWaitEvent
 * captures the record that is current WHEN IT IS CALLED (a later hipEventRecord on the same event does not move an
 * earlier wait), and the pair record + wait is issued back to back on the calling thread.  It would NOT be safe for a
 * caller to keep an event of its own across calls, and none is handed out.  Under a stream capture the same pair
 * becomes a graph dependency. */
int avid_stream_wait(avid_stream_t waiter, avid_stream_t waited);
/* sizeof(avid_instr) as the library was compiled: a binding checks its mirror of the record against it. */
size_t avid_program_instr_bytes(void);
/* Scratch bytes records [begin, end) need on each of n_streams streams (out_bytes[n_streams]). */
int avid_program_workspace_bytes(const avid_instr* prog, int begin, int end, int n_streams, size_t* out_bytes);
/* Issue records [begin, end) of prog.  streams[n_streams]: hipStream_t handles, ws[n_streams]: their scratch.
 * Asynchronous like every other entry point.  On error the message names the failing record. */
int avid_program_run(const avid_instr* prog, int begin, int end, void* const* slots, int n_slots,
                     const avid_stream_t* streams, const avid_stream_ws* ws, int n_streams);

#ifdef __cplusplus
}
#endif
#endif /* AVID_HIP_H */
