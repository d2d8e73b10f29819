/* egaze_hip.h -- C ABI of libegaze_hip.so (gfx950 / MI355X only).
 *
 * The drop-in boundary of the SP / AT / LF hot path of hyf015/egocentric-gaze-prediction.  The reference has no
 * FFI layer of its own: its hot path is the set of torch.nn ops that models/model_SP.py, models/LSTMnet.py,
 * models/late_fusion.py and floss.py execute.  Each entry point below replaces one of those
 * op instances (forward and backward); the reference-side binding is the ctypes table in
 * egocentric-gaze-prediction_amd/_lib.py (INTEGRATION.md shows how a maintainer wires it into the reference).
 *
 * Conventions
 *   - device pointers only (fp32 unless stated); no allocation and no synchronisation inside: the caller passes
 *     workspaces (sizes from the *_ws_bytes / *_rows / *_elems queries) and a hipStream_t; work is stream-ordered.
 *   - activations are NHWC ([B][H][W][C]); network inputs (first conv) are NCHW as the reference's DataLoader
 *     yields them; 1-channel maps are identical in both layouts; weights keep the reference layout (Cout,Cin,3,3).
 *   - every function returns 0 on success or a non-zero hipError_t-style code; egz_last_error() returns the message
 *     of the calling thread's last failure (the Python shim raises EgazeHipError / RuntimeError, matching the
 *     reference's exception-based error behaviour, e.g. utils.py:151-152, extractLSTMw.py:111).
 *   - reentrant per stream; one process per GPU (no cross-device state).
 */
#ifndef EGAZE_HIP_H
#define EGAZE_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

const char* egz_version(void);
const char* egz_last_error(void);

/* ---- nn.Conv2d(Cin, Cout, 3, padding=1): utils.py:70 (encoders), models/model_SP.py:10,13-29 (fusion Conv3d
 *      k=(1,3,3) == the same 3x3 conv on each stream; decoder), models/late_fusion.py:10-12.
 *      Implicit GEMM on v_mfma_f32_32x32x2_f32.  Weights are re-packed on the device first:
 *      fwd  : wp[tap][Cout_pad][Cin_pad]         dgrad: wp[8-tap][Cin_pad][Cout_pad]   (pad = round up to 32, zeros;
 *      one reduction-index-contiguous row per output channel of the GEMM -- private layout, may change) */
size_t egz_pack_w3x3_elems(int C, int K, int kind /* 0: fwd / dgrad, 1: ups_fwd / ups_dgrad */);
int egz_pack_w3x3_fwd(const float* w, float* wp, int C, int K, hipStream_t stream);
int egz_pack_w3x3_dgrad(const float* w, float* wp, int C, int K, hipStream_t stream);
/* nn.Upsample(scale_factor=2) followed by the conv (models/model_SP.py:16-17,20-21,24-25,27-28): per output phase the
 * 3x3 taps collapse to 2x2 pre-summed taps on the low-res input (4/9 of the MACs); ups_dgrad is its transpose. */
int egz_pack_w3x3_ups_fwd(const float* w, float* wp, int C, int K, hipStream_t stream);
int egz_pack_w3x3_ups_dgrad(const float* w, float* wp, int C, int K, hipStream_t stream);
/* rows of the BatchNorm statistic partials written by the stats epilogue: stat_partial is [rows][2][K] fp64 */
int egz_conv3x3_stat_rows(int B, int H, int W, int K, int flags);
/* y[B][H][W][K] = conv3x3(x) + bias.  H, W are OUTPUT dims.  flags: bit0 = x is [B][H/2][W/2][C] and is nearest-x2
 * upsampled on the fly (nn.Upsample(scale_factor=2), models/model_SP.py:16,20,24,27); bit1 (with bit0) = use the
 * phase-decomposed form (weights from egz_pack_w3x3_ups_fwd); bits 4-5 epilogue: 0 bias,
 * 1 bias+ReLU, 2 bias + per-channel (sum, sumsq) partials for train-mode BatchNorm; 0x100 / 0x200 / 0x400 force the
 * 128x64 / 128x128 / 64x64 tile (same flags must be given to egz_conv3x3_stat_rows).  Called with dgrad-packed weights (and K = Cin) it computes the data gradient. */
int egz_conv3x3_fwd(const float* x, const float* wp, const float* bias, float* y, double* stat_partial, int B, int H,
                    int W, int C, int K, int flags, hipStream_t stream);
/* data gradient of [upsample x2 -> conv] w.r.t. the low-res input: dy [B][H][W][K] -> dx [B][H/2][W/2][C] */
int egz_conv3x3_ups_dgrad(const float* dy, const float* wp, float* dx, int B, int H, int W, int C, int K, int flags,
                          hipStream_t stream);
/* Error-compensated split-half variant for the wide layers (Cout % 128 == 0, Cin % 32 == 0): operands as hi + lo in a
 * 16-bit type, a*b ~= hi*hi + hi*lo + lo*hi accumulated in fp32 on v_mfma_f32_32x32x16_{f16,bf16} (16x the exact-f32
 * MFMA rate, 3 instructions per product).  dtype 1 = f16 x3 (weights pre-scaled 2^10; err ~3e-7 = fp32 class),
 * 2 = bf16 x3 (fp32 exponent range, err ~5e-6; used for gradients).  kind: 0 fwd, 1 dgrad, 2 ups_fwd, 3 ups_dgrad.
 * egz_conv3x3_fwd_split: flags as egz_conv3x3_fwd (bits 0-1, 4-5) plus bit2 = the 16-tap data gradient of an
 * upsampled conv (then C / K are the GEMM's reduction / output channel counts). */
int egz_pack_w3x3_split(const float* w, void* wp, int C, int K, int kind, int dtype, hipStream_t stream);
/* Optional tile schedule of egz_conv3x3_fwd_split (flags bit 14 = 0x4000): tiles beyond the last full round of resident
 * blocks are run split-K through `workspace` (raw partial accumulators) and reduced in a fixed order;
 * egz_conv3x3_fwd_split_ws_bytes gives the workspace size (0 without the flag: workspace may then be NULL). */
size_t egz_conv3x3_fwd_split_ws_bytes(int B, int H, int W, int C, int K, int flags);
int egz_conv3x3_fwd_split(const float* x, const void* wp, const float* bias, float* y, double* stat_partial, int B,
                          int H, int W, int C, int K, int flags, int dtype, void* workspace, size_t ws_bytes,
                          const unsigned int* x_absmax, unsigned int* absmax_out, hipStream_t stream);
/* absmax_out (optional; bias + ReLU epilogue, forward forms): receives max |y| in the egz_absmax layout -- the f16 x3
 * scaling of the convolution that consumes y (per-block partials from the epilogue + one fold launch, no extra pass over y
 * for launches of at most 16384 blocks).
 * Streamed-weight form of the plain split-half conv (csrc/conv3x3_igemm_x3s.hip): the weights are packed in MFMA
 * fragment order (kind 4 = forward, 5 = data gradient of the same nn.Conv2d: utils.py:66, models/model_SP.py:13-29) and go
 * L2 -> registers, the activation halo goes through LDS.  egz_conv3x3_streamed_ok: 1 when a geometry is covered
 * (C = reduction channels % 32 == 0, K = GEMM columns % 64 == 0).  mode 0 = plain conv; mode 1 = data gradient of
 * [nn.Upsample(x2) -> nn.Conv2d] (models/model_SP.py:17-18,22-23,25-26,28-29) w.r.t. the low-res input, x = hi-res dy,
 * kind-6 packing; mode 2 = the FORWARD of [nn.Upsample(x2) -> nn.Conv2d] as four 2 x 2-tap phase convolutions over the
 * low-res input (x = [B][H/2][W/2][C], y = [B][H][W][K], H x W passed = the hi-res output, kind-7 packing =
 * [phase][channel block][pre-summed tap], f16 x3 only, epi 0 / 1, K % 64 == 0).  epi: 0 bias, 1 bias + ReLU, 2 bias + BN partials (egz_conv3x3_streamed_stat_rows rows), 3 = data gradient
 * masked by mask_src > 0 with per-channel sums and per-tile abs-max (see below). */
int egz_conv3x3_streamed_ok(int B, int H, int W, int C, int K, int mode);
/* rows of stat_partial ([rows][2][K] doubles) of an epi 2 / epi 5 launch of egz_conv3x3_fwd_streamed (mode 0) */
int egz_conv3x3_streamed_stat_rows(int B, int H, int W, int C, int K);
int egz_pack_w3x3_split_frag(const float* w, void* wq, int C, int K, int kind, int dtype, hipStream_t stream);
/* Every fragment-ordered packing an optimizer step made stale in ONE launch (no reference counterpart: the reference's
 * Conv2d reads its fp32 weights directly, utils.py:70; this is the repack after optim.Adam.step, SP.py:137).  table: n rows of
 * 8 x int64 in DEVICE memory [w, wq, C, K, kind 4..7, dtype 1 f16 / 2 bf16, first block, unused]; first block = running sum of
 * egz_pack_w3x3_frag_blocks(C, K) over the preceding rows, total_blocks = the sum over all rows.  Bit-identical to n calls of
 * egz_pack_w3x3_split_frag. */
int egz_pack_w3x3_frag_blocks(int C, int K);
int egz_pack_w3x3_frag_batch(const long long* table, int n, int total_blocks, hipStream_t stream);
int egz_conv3x3_fwd_streamed(const float* x, const void* wq, const float* bias, float* y, double* stat_partial, int B,
                             int H, int W, int C, int K, int epi, int dtype, int mode, const unsigned int* x_absmax,
                             const float* mask_src, unsigned int* absmax_out, const float* bn_coef, float* minmax_out,
                             hipStream_t stream);
/* dtype | 0x10 (with dtype 1, f16 split halves; the wide tiles K % 32 == 0 of modes 0 and 1): TWO MFMA products per MAC instead of
 * three -- x_hi w_hi + x_hi w_lo: the x operand (the halo image that goes through LDS; dy in a data gradient) enters with its f16 hi
 * half only, rounded to nearest from the fp32 value or from hi + lo of a pre-split pair, the packed weights keep their 22 bits.
 * Meant for the data gradients (autograd of utils.py:70, models/model_SP.py:13-29): the result moves by ~2e-4 relative L2
 * (tests/test_hip_ops.py::test_backward_two_products).  Honoured by the narrow (late-fusion) kernel for epi 0 and 5 too; ignored by
 * egz_conv3x3_fwd_streamed_splitk, which stays on three products. */
/* Deferred BatchNorm (narrow geometry, epi 0 / 1 / 2): with bn_coef != NULL, x is the PRE-BatchNorm conv output of the block
 * below and bn_coef that BatchNorm's 4 x C coefficient rows; relu(x * scale + shift) is applied while the halo is staged, so
 * the normalised tensor of late_fusion.py:11-12 is never materialised (x_absmax = its max, from egz_bn_finalize_deferred).
 * minmax_out (epi 2, narrow geometry): [egz_conv3x3_streamed_stat_rows][2][K] per-channel max / min of y. */
/* epi 5 (mode 0, no bias; the narrow geometry C, K <= 32 with H and W multiples of 16, or K % 64 == 0): data gradient w.r.t. the
 * output of a train-mode [BatchNorm2d -> ReLU] (late_fusion.py:10-12; the conv -> BN -> ReLU -> conv pairs of the VGG
 * encoders, utils.py:64-76) that ALSO accumulates that BatchNorm's backward sums: mask_src =
 * the layer's pre-BN conv output (layout of y), bn_coef = 4 rows of K floats (batch mean, 1/std, scale, shift);
 * stat_partial receives egz_conv3x3_streamed_stat_rows rows of (sum dz, sum dz * xhat) with dz = (mask_src * scale + shift > 0) ? y : 0,
 * the `sums` argument of egz_bn_relu_pool_bwd.
 * epi 1 (bias + ReLU) with absmax_out != NULL (K % 64 == 0, at most 16384 tiles): absmax_out receives max |y|, folded.
 * Split-K form of a plain egz_conv3x3_fwd_streamed launch for small pixel counts (batch-1 inference as in
 * run_spatialstream.py:85-138 / AT.py:216, the 14 x 14 layers at the reference's default --batch_size_sp 8): the channel
 * blocks of a tile are divided over nsplit blocks (raw partial sums in the workspace, nsplit x B x H x W x K floats) and a
 * fix-up launch sums them in split order and applies the epilogue (epi 0 / 1 / 2).  egz_conv3x3_streamed_splits recommends
 * the split count; 1 = use egz_conv3x3_fwd_streamed.  K % 128 == 0, C % 32 == 0. */
int egz_conv3x3_streamed_splits(int B, int H, int W, int C, int K);
int egz_conv3x3_fwd_streamed_splitk_stat_rows(int B, int H, int W);   /* rows of stat_partial for epi 2 (one per 32 pixels) */
size_t egz_conv3x3_fwd_streamed_splitk_ws_bytes(int B, int H, int W, int K, int nsplit);
int egz_conv3x3_fwd_streamed_splitk(const float* x, const void* wq, const float* bias, float* y, double* stat_partial, int B,
                                    int H, int W, int C, int K, int epi, int dtype, const unsigned int* x_absmax,
                                    void* workspace, size_t ws_bytes, int nsplit, unsigned int* absmax_out,
                                    hipStream_t stream);
/* helpers of the epi = 3 (ReLU mask + bias-gradient sums + abs-max) form of egz_conv3x3_fwd_streamed, which folds the
 * nn.ReLU backward of a decoder layer (models/model_SP.py:13-29) into the data gradient of the layer above it */
int egz_colsum_f64(const double* part, int rows, int cols, int ncols_out, float* out, void* workspace, size_t ws_bytes,
                   hipStream_t stream);
/* dw (K,C,3,3) = sum_pixels dy (x) x   (autograd of the same conv; loss.backward() at SP.py:136, LF.py:99) */
size_t egz_conv3x3_wgrad_ws_bytes(int B, int H, int W, int C, int K, int flags);
int egz_conv3x3_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int C, int K, int flags,
                      void* workspace, size_t ws_bytes, const unsigned int* dy_absmax, const unsigned int* x_absmax,
                      const float* x_bn, hipStream_t stream);
/* flags | 0x20000 (with 0x2000 and dy_absmax: f16 split halves on the wide kernels, C and K multiples of 32 / 64): TWO MFMA products
 * per MAC -- x_hi dy_hi + x_hi dy_lo: x enters with 11 significant bits, rounded to nearest (from the fp32 value, or from hi + lo
 * of a pre-split pair), dy keeps 22; dw moves by ~2e-4 relative L2.  Also on the narrow (late-fusion) kernels; ignored by the bf16 and exact-f32 launches. */
/* x_bn (optional): x is the PRE-BatchNorm conv output of the block below and x_bn that BatchNorm's 4 x C coefficient rows;
 * relu(x * scale + shift) is applied while x is staged (deferred BatchNorm, late_fusion.py:11-12), x_absmax = the max of the
 * normalised values.  Only where egz_conv3x3_wgrad_narrow_ok(B, H, W, C, K) (C, K <= 32, W % 16 == 0, flags 0x2000). */
int egz_conv3x3_wgrad_narrow_ok(int B, int H, int W, int C, int K);
/* Pre-split activations (round 5; the x operand of the encoder convolutions, utils.py:64-76).  The [BatchNorm -> ReLU (-> pool)]
 * pass that writes a block output can store it as f16 hi / lo PAIRS instead of fp32 -- per 4-channel quad the 16 bytes
 * [4 hi halves | 4 lo halves] of (value * 2^s), s from the tensor's exact abs-max, the very pair the split-half kernels
 * would form while staging the fp32 value (same footprint, bit-identical products).  Consumers: egz_conv3x3_fwd_streamed with
 * mode | 0x100 (mode 0, dtype 1, epi 2, K % 64 == 0, C % 32 == 0, x_absmax = the abs-max the pairs were scaled with) and
 * egz_conv3x3_wgrad with flags | 0x8000 where egz_conv3x3_wgrad_presplit_ok(B, H, W, C, K) (C, K multiples of 64, the
 * split-half 9-tap kernel's geometry; flags 0x2000, dy_absmax and x_absmax given, plain conv).  Producer chain:
 * egz_conv3x3_fwd_streamed(epi 2, K % 64 == 0, K <= 512, minmax_out = 1024 zero-filled uints: 1024 / (2 K) sets of order-preserving
 * images of the per-channel max of y and of -y) -> egz_bn_finalize_bound -> egz_bn_relu_pool_fwd_presplit. */
int egz_conv3x3_wgrad_presplit_ok(int B, int H, int W, int C, int K);

/* ---- first conv of a stack, small Cin, NCHW input: Conv2d(3,64) / Conv2d(20,64) (utils.py:70 at SP.py:53, inputs
 *      per data/STdatas.py:50-73) and Conv2d(2,32) (models/late_fusion.py:10).  K in {64, 32}. */
int egz_conv_first_stat_rows(int B, int H, int W);
int egz_conv_first_stat_rows_for(int B, int H, int W, int C, int K);   /* rows of stat_partial egz_conv_first_fwd writes for C -> K
                                                                        * (C <= 3: the direct kernel, one row per block) */
int egz_conv_first_fwd(const float* x_nchw, const float* w, const float* bias, float* y_nhwc, double* stat_partial,
                       int B, int H, int W, int C, int K, float* minmax_out, unsigned int* minmax_ordered, hipStream_t stream);
/* minmax_out (optional; C <= 3 with stat_partial): [rows][2][K] per-channel max / min of y, rows as
 * stat_partial -- input of egz_bn_finalize_deferred.  minmax_ordered (optional, same conditions): 1024 zero-filled uints (the first 2 K are written), the
 * atomic-max form egz_bn_finalize_bound reads (see egz_conv3x3_wgrad_presplit_ok). */
size_t egz_conv_first_wgrad_ws_bytes(int B, int H, int W, int C);
int egz_conv_first_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw, int B, int H, int W, int C, int K,
                         void* workspace, size_t ws_bytes, hipStream_t stream);

/* ---- nn.BatchNorm2d (train: batch statistics + running-stat update with unbiased variance; eval: affine) fused with
 *      the nn.ReLU and optional nn.MaxPool2d(2,2) that follow it (utils.py:68,72; models/model_SP.py:12,45-47;
 *      models/late_fusion.py:10-12).  mean/invstd/scale/shift are [K] outputs kept for the backward pass. */
size_t egz_bn_ws_bytes(int K);
int egz_bn_finalize(const double* stat_partial, int rows, int K, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* mean_out,
                    float* invstd_out, float* scale, float* shift, long long* num_batches_tracked, void* workspace,
                    size_t ws_bytes, hipStream_t stream);
/* The same for a BatchNorm whose output is never materialised -- the convolution above it applies relu(y * scale + shift) while
 * staging y (egz_conv3x3_fwd_streamed bn_coef; late_fusion.py:11-12).  minmax: [mm_rows][2][K] per-channel max / min rows of y
 * from the producing conv; absmax_out (egz_absmax layout) receives the exact max of the normalised + ReLU'd values, the f16
 * split scale of that convolution.  K in {16, 32, 64}; one launch, no workspace. */
int egz_bn_finalize_deferred(const double* stat_partial, int rows, int K, double count, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, float* mean_out,
                             float* invstd_out, float* scale, float* shift, long long* num_batches_tracked,
                             const float* minmax, int mm_rows, unsigned int* absmax_out, hipStream_t stream);
/* egz_bn_finalize that also bounds the block output: minmax as written by egz_conv3x3_fwd_streamed on the 64- / 128-column tiles
 * (1024 uints), absmax_out (egz_absmax layout, zero-filled by the caller) receives the EXACT max of relu(y * scale + shift)
 * (the map is monotonic in y per channel).  K % 64 == 0. */
int egz_bn_finalize_bound(const double* stat_partial, int rows, int K, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* mean_out,
                          float* invstd_out, float* scale, float* shift, long long* num_batches_tracked, void* workspace,
                          size_t ws_bytes, const unsigned int* minmax, unsigned int* absmax_out, hipStream_t stream);
int egz_bn_eval_coeffs(int K, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, hipStream_t stream);
/* egz_bn_relu_pool_fwd with the output stored pre-split (see egz_conv3x3_wgrad_presplit_ok): absmax is an INPUT, the exact max of
 * the output from egz_bn_finalize_bound. */
int egz_bn_relu_pool_fwd_presplit(const float* y, const float* scale, const float* shift, float* out, int B, int H, int W,
                                  int K, int pool, const unsigned int* absmax, hipStream_t stream);
int egz_bn_relu_pool_fwd(const float* y, const float* scale, const float* shift, float* out, int B, int H, int W,
                         int K, int pool, unsigned int* absmax, hipStream_t stream);   /* absmax: optional, max |out| */
size_t egz_bn_relu_pool_bwd_ws_bytes(int K);
int egz_bn_relu_pool_bwd(const float* y, const float* dout, const float* scale, const float* shift, const float* mean,
                         const float* invstd, float* dy, float* dgamma, float* dbeta, int B, int H, int W, int K,
                         int pool, void* workspace, size_t ws_bytes, unsigned int* absmax, const double* sums,
                         int sums_rows, hipStream_t stream);
/* The same with dy stored PRE-SPLIT (f16 hi / lo pairs; see egz_conv3x3_wgrad_presplit_ok): scaled by a BOUND of max |dy| that the
 * finalize step derives before the apply pass runs from y_minmax (the 1024 uints of egz_conv3x3_fwd_streamed's minmax_out for y),
 * dout_absmax (max |dout|, egz_absmax layout: egz_conv3x3_fwd_streamed's absmax_out of the data gradient that produced dout)
 * and the two per-channel sums; the bound is left in `absmax` (zero-filled by the caller) and is dy's abs-max for its consumers
 * (egz_conv3x3_fwd_streamed mode | 0x100 with epi 0 / 5, egz_conv3x3_wgrad flags | 0x10000).  K % 64 == 0, K <= 512. */
int egz_bn_relu_pool_bwd_presplit(const float* y, const float* dout, const float* scale, const float* shift, const float* mean,
                                  const float* invstd, float* dy, float* dgamma, float* dbeta, int B, int H, int W, int K,
                                  int pool, void* workspace, size_t ws_bytes, unsigned int* absmax, const double* sums,
                                  int sums_rows, const unsigned int* y_minmax, const unsigned int* dout_absmax,
                                  hipStream_t stream);
/* Backward of a FIRST block [Conv2d(C -> K, 3x3) -> BatchNorm2d(train) -> ReLU] with C <= 3 and K = 32 (late_fusion.py:10-12)
 * or 64 (the RGB encoder, utils.py:70 at SP.py:53; the network input needs no data gradient): dgamma / dbeta and dw (K, C, 3, 3) in one pass
 * over y (pre-BN conv output, NHWC) and dout, x = the block input [B][C][H][W]; the gradient w.r.t. the conv output is never
 * stored.  sums / sums_rows as for egz_bn_relu_pool_bwd. */
size_t egz_bn_bwd_first_wgrad_ws_bytes(int C, int K);
int egz_bn_bwd_first_wgrad(const float* y, const float* dout, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* x, float* dw, float* dgamma, float* dbeta, int B, int H, int W,
                           int C, int K, void* workspace, size_t ws_bytes, const double* sums, int sums_rows,
                           hipStream_t stream);
/* sums (optional, pool = 0): [sums_rows][2][K] partial rows of (sum dz, sum dz * xhat) produced together with dout by
 * egz_conv3x3_fwd_streamed epi 5 -- the reduce pass over y and dout is skipped.
 * `absmax` of the three gradient producers (egz_bn_relu_pool_bwd, egz_pairmax_bwd, egz_relu_bwd_bias; optional): a buffer of
 * egz_absmax_elems() uints that the CALLER ZERO-FILLS before the producing call (true for every `absmax` / `absmax_out`
 * argument of this header except egz_absmax's, which zero-fills its own).  Layout: 32 slots, one per 128-byte line, each the
 * bit pattern of a non-negative float; a producing block folds its maximum into one slot with at most one device-scope atomic
 * max (max is exact and order independent: deterministic), a consumer (`x_absmax`, `dy_absmax`, `a_absmax` arguments) takes
 * the maximum of the slots at the top of its kernel -- no fold launch between producer and consumer.  It is the scale source
 * of the f16 x3 split-half forward / data / weight gradients; egz_absmax computes it for a bare tensor (n % 4 == 0). */
int egz_absmax_elems(void);
int egz_absmax(const float* x, long n, unsigned int* absmax, hipStream_t stream);

/* ---- nn.MaxPool3d((2,1,1)) over the depth-2 stack == element-wise max of the two streams (model_SP.py:11,43);
 *      y2 = [2][n] (stream s, then t), the first stream wins ties like torch's pooling scan. */
int egz_pairmax_fwd(const float* y2, float* z, long n, hipStream_t stream);
int egz_pairmax_bwd(const float* y2, const float* dz, float* dy2, long n, unsigned int* absmax, hipStream_t stream);
/* per-channel (sum, sumsq) partials of an NHWC tensor, in the conv-epilogue format ([rows][2][K] fp64) */
int egz_channel_stats_rows(void);
int egz_channel_stats(const float* x, long rows, int K, double* stat_partial, hipStream_t stream);

/* ---- nn.ReLU backward (decoder, models/model_SP.py:13-29), optionally fused with the conv bias gradient;
 *      nn.Upsample(scale_factor=2) backward (2x2 sum); column sums; NCHW <-> NHWC transposes. */
int egz_relu_bwd(const float* out, const float* dout, float* dy, long n, hipStream_t stream);
size_t egz_relu_bwd_bias_ws_bytes(int K);
int egz_relu_bwd_bias(const float* out, const float* dout, float* dy, float* db, long rows, int K, void* workspace,
                      size_t ws_bytes, unsigned int* absmax, hipStream_t stream);
int egz_upsample2x_bwd(const float* dxu, float* dx, int B, int H, int W, int C, hipStream_t stream);
int egz_colsum(const float* x, long rows, int K, float* out, void* workspace, size_t ws_bytes, hipStream_t stream);
int egz_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, hipStream_t stream);
/* the same transpose with the channel dimension zero-padded to Cp (the 20-channel flow stack -> 32 channels so that the
 * temporal encoder's first conv, SP.py:53, runs on the split-half kernels) */
int egz_nchw_to_nhwc_pad(const float* in, float* out, int B, int C, int H, int W, int Cp, hipStream_t stream);
int egz_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, hipStream_t stream);
/* stream-ordered device copy of n floats / zero fill of `bytes` bytes: torch.cat of the two encoder maps
 * (models/model_SP.py:39) when they do not already share a buffer; optimizer.zero_grad() (SP.py:138) */
int egz_copy(const float* src, float* dst, long n, hipStream_t stream);
int egz_fill_zero(void* dst, size_t bytes, hipStream_t stream);

/* ---- nn.Conv2d(C, 1, 1) + nn.Sigmoid head (models/model_SP.py:30,32,49; models/late_fusion.py:13,15,22) */
int egz_conv1x1_sigmoid_fwd(const float* x, const float* w, const float* bias, float* out, float* logits, long M,
                            int C, hipStream_t stream);
size_t egz_conv1x1_sigmoid_bwd_ws_bytes(int C);
int egz_conv1x1_sigmoid_bwd(const float* x, const float* w, const float* out, const float* dout, float* dx, float* dw,
                            float* db, long M, int C, void* workspace, size_t ws_bytes, hipStream_t stream);
/* The same with the ReLU backward of the block below folded in, for `Conv2d 3x3 -> ReLU -> Conv2d 1x1 -> Sigmoid`
 * (models/model_SP.py:28-32): x is that block's post-ReLU output, dx = (x > 0) ? dlogit * w : 0.  mstat: fp64 partial rows
 * [egz_conv1x1_sigmoid_bwd_rows(M, C)][C] whose column sums (egz_colsum_f64) are the 3x3 conv's bias gradient; absmax: an
 * egz_absmax_elems() buffer, slot 0 = max |dx| on return.  Replaces the autograd ReLU-backward pass between the two convs. */
int egz_conv1x1_sigmoid_bwd_rows(long M, int C);
int egz_conv1x1_sigmoid_bwd_masked(const float* x, const float* w, const float* out, const float* dout, float* dx, float* dw,
                                   float* db, double* mstat, unsigned int* absmax, long M, int C, void* workspace,
                                   size_t ws_bytes, hipStream_t stream);

/* ---- floss.forward / build_weight_from_target (floss.py:9-41): W = width / (||p - centroid(argmax set)|| + 1),
 *      F.binary_cross_entropy(input, target, W) with torch's -100 log clamp, mean reduction.  weighted = 0 gives
 *      torch.nn.BCELoss() (SP.py:103-106).  Backward follows aten: w*(x-t)/max((1-x)x, 1e-12)/N.
 *      nn.MSELoss (AT.py:83,138) likewise.  loss_out / grad_out are device scalars. */
size_t egz_loss_ws_bytes(int B);
int egz_floss_fwd(const float* inp, const float* target, float* weights_out, float* loss_out, int B, int H, int W,
                  int weighted, void* workspace, size_t ws_bytes, hipStream_t stream);
int egz_floss_bwd(const float* inp, const float* target, const float* weights, const float* grad_out, float* dinp,
                  long n, hipStream_t stream);
/* nn.MSELoss (AT.py:83); tanh_b != 0: the target is tanh(b), i.e. `criterion(pred, tanh(target))` of AT.py:138 in one pass */
int egz_mse_fwd(const float* a, const float* b, float* loss_out, long n, void* workspace, size_t ws_bytes, int tanh_b,
                hipStream_t stream);
int egz_mse_bwd(const float* a, const float* b, const float* grad_out, float* da, long n, int tanh_b, hipStream_t stream);
/* nn.MSELoss + its gradient for a unit seed in ONE single-block launch, n <= 4096 (AT.py:138-141 on one (1, 1, 512) sample);
 * bit-identical to egz_mse_fwd + egz_mse_bwd(grad_out = 1).  ring (optional): the loss is also parked in ring[counter[0] % ring_n]. */
int egz_mse_fwd_grad(const float* a, const float* b, float* loss_out, float* da, long n, int tanh_b, float* ring, int ring_n,
                     const int* counter, hipStream_t stream);

/* ---- AT: nn.LSTM(512,512,2) + nn.Linear + tanh (models/LSTMnet.py:18-37) from a strided f32-MFMA GEMM and fused
 *      cell kernels.  egz_gemm: C[M][N] (row stride ldc) = op(A) op(B) (+C if flags&1) (+bias[n]) (ReLU if flags&2),
 *      op(A)(m,k) = A[m*sam + k*sak], op(B)(k,n) = B[k*sbk + n*sbn].  Gate order i,f,g,o. */
int egz_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sam, long sak,
             long sbk, long sbn, long ldc, int flags, hipStream_t stream);
/* count <= 8 products of ONE shape and stride set in one launch (blockIdx.z picks the operands): the weight-gradient products
 * dgates^T [x | h_prev] of nn.LSTM's backward (models/LSTMnet.py:18 under autograd).  A, B, C: HOST arrays of `count` device
 * pointers; flags bit 0 = accumulate; no bias, no ReLU.  Outside egz_gemm's fast-path geometry: `count` egz_gemm calls. */
int egz_gemm_batched(const float* const* A, const float* const* B, float* const* C, int count, int M, int N, int K, long sam,
                     long sak, long sbk, long sbn, long ldc, int flags, hipStream_t stream);
int egz_lstm_cell_fwd(const float* gates, const float* c_prev, float* h_out, float* c_out, float* act, int B, int Hd,
                      hipStream_t stream);
int egz_lstm_cell_bwd(const float* act, const float* c, const float* c_prev, const float* dh, const float* dc_in,
                      float* dgates, float* dc_prev, int B, int Hd, hipStream_t stream);
/* nn.LSTM(H, H, num_layers = L) as a wavefront over (layer, step) (models/LSTMnet.py:18,26-35: self.lstm(input, hidden)): launch
 * s runs step s - l of every layer l, T + L - 1 dependent launches instead of T x L, the upper layers' input projections reduced
 * in the same launch (K = 2H).  w_ih / w_hh / bsum: HOST arrays of L device pointers ([4H][H], [4H][H], [4H] = b_ih + b_hh;
 * w_ih[0] / bsum[0] unused: gx0 [T][B][4H] = x W_ih0^T + b_ih0 + b_hh0 for every step); h0, c0, hn, cn: [L][B][H];
 * hs: [L][T + 1][B][H] (slot 0 of a layer = a copy of its h0, slots 1 .. T = the outputs); cs: [L][T][B][H]; acts: [L][T][B][4H]
 * or null.  1 <= L <= 4, H % 256 == 0. */
int egz_lstm_wave_fwd(const float* gx0, const float* const* w_ih, const float* const* w_hh, const float* const* bsum,
                      const float* h0, const float* c0, float* hs, float* cs, float* acts, float* hn, float* cn, int L, int T,
                      int B, int H, hipStream_t stream);
/* The same forward recurrence for the AT network's own geometry (L = 2, H = 512, B <= 32) as ONE persistent, weight-stationary
 * launch: each of 128 x ceil(B / 16) blocks keeps its 16 x 1536 weights in registers for the whole sequence and the blocks hand
 * h_t to each other inside the launch (write-through stores + arrival counters, csrc/lstm_seq.hip).  Outputs as
 * egz_lstm_wave_fwd; inputs differ in two places: gx0 = x W_ih0^T WITHOUT bias, and b_ih / b_hh = HOST arrays of L device pointers
 * to the module's own bias vectors ([4H] each), summed inside the kernel.  `sync`: egz_lstm_persist_sync_words() uints of device
 * scratch (its counters are zeroed by the call); after the launch word [1024] is 0, or 1 + s when a block gave up waiting in
 * global step s (0.5 s of wall clock without progress; outputs then NaN from that step on).  The LAST 32 words of the scratch are
 * sticky: a call writes them only on such a time-out (word 0: forward launches, word 1: backward launches; atomic max of 1 + s) and
 * never clears them -- zero the scratch once when allocating it, keep it across calls, read the two words wherever the host
 * synchronises anyway.  Any other geometry (or T * B beyond the 32-bit offsets of this form): returns hipErrorNotSupported (801)
 * and launches nothing -- call egz_lstm_wave_fwd. */
int egz_lstm_persist_sync_words(void);
int egz_lstm_persist_fwd(const float* gx0, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                         const float* const* b_hh, const float* h0, const float* c0, float* hs, float* cs, float* acts, float* hn,
                         float* cn, unsigned int* sync, int L, int T, int B, int H, hipStream_t stream);
/* Its backward through time (autograd of the same call): T + 2 L - 1 launches of uniform K = 4H blocks; the gradient a lower layer
 * receives from the layer above (dgates_above,t W_ih_above) is formed by blocks of its own one launch ahead of the cell backward
 * that consumes it and parked in dhin.  dh_top: [T][B][H] or null; dhn, dcn: [L][B][H] or null; w_hh_t / w_ih_t: HOST arrays of L
 * device pointers to the TRANSPOSED weights [H][4H] (w_ih_t[0] unused); dgates: [L][T][B][4H] out; dh0, dc0: [L][B][H] out;
 * dhin: [L - 1][T][B][H] scratch (may be null at L = 1). */
int egz_lstm_wave_bwd(const float* dh_top, const float* dhn, const float* dcn, const float* acts, const float* cs,
                      const float* c0, const float* const* w_hh_t, const float* const* w_ih_t, float* dgates, float* dh0,
                      float* dc0, float* dhin, int L, int T, int B, int H, hipStream_t stream);
/* egz_lstm_wave_bwd for the AT network's own geometry (L = 2, H = 512, B <= 32) as ONE persistent launch: 64 x ceil(B / 8) blocks,
 * each keeping 8 units x 2048 of W_hh_l1^T, W_ih_l1^T, W_hh_l0^T in registers; the dgates of a step are handed over inside the
 * launch (csrc/lstm_seq.hip, lstm_persist_bwd_kernel).  Inputs and outputs as egz_lstm_wave_bwd without dhin, except: w_hh / w_ih
 * are the weights AS THE MODULE HOLDS THEM ([4H][H]; HOST arrays of L device pointers, w_ih[0] unused) -- no transposed copies --
 * and the launch also forms the bias gradients: db = HOST array of 2 L device pointers (b_ih_l0, b_hh_l0, b_ih_l1, b_hh_l1: [4H]
 * each = the sum of dgates_l over steps and batch rows; null entries skipped; db may be null).  `sync` as in
 * egz_lstm_persist_fwd.  Any other geometry: hipErrorNotSupported, nothing launched. */
int egz_lstm_persist_bwd(const float* dh_top, const float* dhn, const float* dcn, const float* acts, const float* cs,
                         const float* c0, const float* const* w_hh, const float* const* w_ih, float* dgates, float* dh0,
                         float* dc0, float* const* db, unsigned int* sync, int L, int T, int B, int H, hipStream_t stream);
/* The same network at T = 1, B = 1 -- the reference's own stepping (AT.py:127-145 training loop, AT.py:246 inference): the
 * whole step in ONE call (L + 1 launches forward, 2L + 1 backward; csrc/lstm_b1.hip).  params / grads: HOST arrays of
 * 4L + 2 device pointers in state-dict order (w_ih, w_hh, b_ih, b_hh per layer, lin.weight [N][H], lin.bias [N]); a null
 * grads entry skips that gradient.  inp [C] raw (tanh applied inside, saved to xt); h0, c0, hn, cn [L][H]; acts [L][4H]
 * (null in no-grad runs); out [N] after ReLU.  No gradient w.r.t. inp / h0 / c0 (the loop detaches them, AT.py:143). */
size_t egz_lstm_b1_ws_bytes(int L, int C, int H, int N);
int egz_lstm_b1_fwd(const void* const* params, int L, const float* inp, const float* h0, const float* c0, float* xt,
                    float* acts, float* hn, float* cn, float* out, int C, int H, int N, hipStream_t stream);
int egz_lstm_b1_bwd(const void* const* params, void* const* grads, int L, const float* dout, const float* dhn,
                    const float* dcn, const float* xt, const float* acts, const float* h0, const float* c0,
                    const float* hn, const float* cn, const float* out, int C, int H, int N, void* workspace,
                    size_t ws_bytes, hipStream_t stream);
int egz_tanh_fwd(const float* x, float* y, long n, hipStream_t stream);
int egz_tanh_bwd(const float* y, const float* dy, float* dx, long n, hipStream_t stream);
int egz_add(const float* a, const float* b, float* out, long n, hipStream_t stream);

/* ---- torch.optim.Adam step (defaults as configured at SP.py:110-113, AT.py:84, LF.py:77) over one flat buffer;
 *      grad_scale multiplies the gradient first (1/world_size after the RCCL sum all-reduce).
 *      nonfinite: device word (may be null; zeroed by its owner).  An element whose scaled gradient is NaN / inf is skipped (p, m, v
 *      keep their values) and bit 0 of *nonfinite is set -- torch.optim.Adam would write NaN into the weights there; every finite
 *      gradient gives torch's bits. */
int egz_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                  double eps, int step, double grad_scale, unsigned int* nonfinite, hipStream_t stream);
/* the same step with the counter of COMPLETED steps on the device (applies step step[0] + 1, then increments step[0]): for
 * optimizer steps inside a captured hipGraph, where a replay cannot receive a new host scalar.  step -> TWO ints {completed
 * steps, reserved (0)} */
int egz_adam_step_dev(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                      double eps, int* step, double grad_scale, unsigned int* nonfinite, hipStream_t stream);

/* utils.computeAAEAUC (utils.py:96-140) per sample on the device: res (B,6) doubles = (AAE deg, fp = #{z > z[gp]}, gaze row,
 * gaze col, centroid row, centroid col); gw = the 2R+1 weights of scipy's gaussian kernel (sigma 14, R = 56),
 * dist = 112 / tan(pi/6).  224 x 224 maps only, like the reference.  Called by LF.trainLate / SP.testSP every batch
 * (LF.py:92-94, SP.py:170-174) instead of a device-to-host copy of both maps + 148 ms of scipy per 32 frames. */
int egz_aae_auc(const float* out, const float* gt, int B, int H, int W, const double* gw, int R, double dist,
                double* res, hipStream_t stream);

/* torch.cat((f, g), dim=1) of two one-channel maps, the late-fusion stack's input (models/late_fusion.py:19):
 * f, g [B][1][H][W] -> out [B][2][H][W]; 16-byte copies when HW % 4 == 0 and the pointers are 16-byte aligned, 4-byte ones otherwise. */
int egz_cat2_planes(const float* f, const float* g, float* out, int B, long HW, hipStream_t stream);
/* Input pipeline on the device (data/STdatas.py:50-68, data/lateDataset.py:22-33): uint8 planes [...][C][plane] ->
 * (u8 / 255 - mean[c]) / std[c] in fp32, the reference's three correctly-rounded operations (bit-exact). */
int egz_u8_normalize(const unsigned char* src, float* dst, long n, long plane, int C, const float* mean,
                     const float* std, hipStream_t stream);
/* AT extraction glue (AT.py:25-39,58-66,229; extractLSTMw.py:81-90): chn_weight = mean of the size x size window of the
 * channels-last (B,H,W,C) feature map around gaze_point / cell; weighted map = min-max normalised sum_c feat * w. */
int egz_crop_mean(const float* feat, const int* gp, float* out, int B, int H, int W, int C, int size, int cell,
                  hipStream_t stream);
/* extractLSTMw.crop_feature_var (extractLSTMw.py:46-58) + mean: explicit window {y0, y1, x0, x1} per sample */
int egz_window_mean(const float* feat, const int* win, float* out, int B, int H, int W, int C, hipStream_t stream);
/* AT.crop_align_feature + mean (AT.py:41-56,229; extractLSTMw.py:32-44, `--align`): the mean of a window of the bilinearly
 * x16-upsampled map is a linear functional of the map -- out[b][c] = sum_p wmap[b][p] * feat[b][p][c] */
int egz_pixel_weighted_sum(const float* feat, const float* wmap, float* out, int B, int HW, int C, hipStream_t stream);
int egz_weighted_minmax(const float* feat, const float* w, float* out, int B, int HW, int C, hipStream_t stream);

/* Config-1 glue (run_spatialstream.py:99-104,130-136): centre of mass of the uint8-quantised gaze map exactly as
 * scipy.ndimage.center_of_mass((map * 255).astype(uint8)) computes it (com (B,2) doubles, gp = floor(com) (B,2) int32, q8
 * optional (B,H,W) uint8 image), and nn.functional.interpolate(mode='bilinear', scale_factor=scale) of (B,h,w) maps with either
 * align_corners convention (False: run_spatialstream.py:136; True: upsample_bilinear at AT.py:47, extractLSTMw.py:33); sample b
 * of the result starts at dst + b * dst_bstride floats. */
int egz_u8_center_of_mass(const float* map, int B, int H, int W, double* com, int* gp, unsigned char* q8, hipStream_t stream);
int egz_bilinear_up(const float* src, float* dst, int B, int h, int w, int scale, int align_corners, long dst_bstride,
                    hipStream_t stream);

/* ---- measurement aid (bench.py): sustained v_mfma_f32_32x32x16_f16 rate of this chip on the given operand bits; frag =
 *      16 x 64 x 16 bytes, out = blocks x 256 floats; executes blocks x 4 x iters x 8 MFMAs of 32768 flop. */
int egz_mfma_probe(const void* frag, float* out, int blocks, int iters, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EGAZE_HIP_H */
