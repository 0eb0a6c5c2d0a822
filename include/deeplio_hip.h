/*
 * deeplio_hip.h -- C-ABI of libdeeplio_hip.so (MI355X / gfx950 only).
 *
 * The reference (ArashJavan/DeepLIO) has no FFI: every op on its training hot
 * path is a stock torch.nn call.  This header declares the device entry points
 * that replace those calls, one group per reference site (file:line relative
 * to the reference checkout).  A maintainer binds them with ctypes exactly as
 * deeplio_amd/_lib.py does.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned storage (the Python
 *     host passes torch storage); the library never allocates or frees.
 *   - workspace is caller-owned; query its size with the *_ws_bytes call.
 *   - all tensors are fp32, NCHW contiguous.  A "channel slice" (ctot, coff, C)
 *     addresses channels [coff, coff+C) of a buffer that has ctot channels, so
 *     that concat (torch.cat(dim=1)) never materialises a copy.
 *   - `stream` is a hipStream_t passed as void* (0 = null stream); kernels are
 *     enqueued, never synchronised.
 *   - return 0 on success, negative DLIO_E* otherwise.
 */
#ifndef DEEPLIO_HIP_H
#define DEEPLIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLIO_OK 0
#define DLIO_EINVAL (-1)   /* bad shape / null pointer            -> ValueError   */
#define DLIO_EUNSUP (-2)   /* unsupported configuration           -> ValueError   */
#define DLIO_ELAUNCH (-3)  /* hipGetLastError() after launch      -> RuntimeError */
#define DLIO_EWS (-4)      /* workspace too small                 -> RuntimeError */

typedef void* dlio_stream_t;

/* ---- library info ----------------------------------------------------
 * DLIO_ABI_VERSION is bumped whenever a prototype changes; dlio_version() returns the value the
 * library was built with and dlio_abi_hash() the CRC-32 of the comment-stripped text of THIS header
 * at build time (deeplio_amd/build.py passes it in).  The ctypes binding (deeplio_amd/_lib.py)
 * compares both against the header next to it when it loads the library, so a stale .so fails at
 * import instead of being called with a changed signature. */
#define DLIO_ABI_VERSION 243
int dlio_version(void);
uint32_t dlio_abi_hash(void);
const char* dlio_arch(void);
/* != 0: the library was built with a TIMING PROBE macro (kernels that skip work and give wrong results: DLIO_SPLIT_Q0 bit 0,
 * BX3_ABLATE bit 1, W1_COAL_PROBE bit 2); the host side refuses such a library unless DLIO_ALLOW_PROBES=1 */
int dlio_build_probes(void);           /* "gfx950" */
const char* dlio_strerror(int code);
/* shared[0] = 1 when HIP streams a and b are served by the same hardware queue (their launches then never overlap), found
 * by timing: a 1 us launch on b issued behind a 150 us spin on a.  Synchronises both streams.  The host side uses it to
 * put the step's four heavy streams (the two siamese encoders of LidarPointSegFeat, lidar_feat_nets.py:62-118, and their
 * weight-gradient companions) on four different queues whatever the process created before (deeplio_amd.functional.assign_streams). */
int dlio_streams_share_queue(dlio_stream_t a, dlio_stream_t b, int* shared);
/* hipGetErrorString of the HIP error behind the last DLIO_ELAUNCH in this thread */
const char* dlio_last_hip_error_string(void);

/* ---- profiling hooks (bench.py roofline leg) ---------------------------
 * When enabled, the launchers bracket each launch with hipEvents on the launch stream.
 * dlio_prof_collect synchronises those events and returns the summed milliseconds / algorithmic
 * FLOPs / bytes / launch count per kernel kind:
 *   0  multi-tap conv forward / data gradient on the fp32 MFMA (stems, strided layers)   MFMA-bound
 *   1  conv weight gradients other than kinds 4 and 5                                      MFMA-bound
 *   2  1x1 conv forward / data gradient                                                    HBM-bound
 *   3  3x3 stride-1 forward / data gradient on the split-bf16 kernel (dlio_conv3x3_bx3_fwd) MFMA-bound
 *   4  3x3 stride-1 weight gradient                                                        MFMA-bound
 *   5  1x1 weight gradient                                                                 HBM-bound
 *   6  BatchNorm forward statistics      (bytes: 1 pass over the conv output)              HBM-bound
 *   7  BatchNorm forward apply           (2 passes, +1 with a residual)                    HBM-bound
 *   8  BatchNorm backward reductions     (2 passes: dy, x)                                 HBM-bound
 *   9  BatchNorm backward apply          (3 passes: dy, x, dx)                             HBM-bound
 *  10  max-pool / SE scale / global average pool, forward and backward                     HBM-bound
 *  11  native-bf16 convolutions of the mixed-precision path (forward / data / weight gradient)
 * For the HBM-bound kinds `bytes` are the bytes the launch moves by construction (every operand once).
 * dlio_prof_enable takes a bit mask of the kinds to time (0 = off): an event pair costs ~1.3 us of
 * stream time, so the timed region of bench.py times the dominant kind only. */
#define DLIO_PROF_KINDS 16
int dlio_prof_enable(int kinds_mask);
/* time only every stride-th launch of each enabled kind (1 = all): the per-kind sums then cover the
 * sampled launches only.  A stride coprime with a kind's launches per step walks through all layers
 * over consecutive steps. */
int dlio_prof_sample(int stride);
int dlio_prof_reset(void);
/* destroys the event pools (after dlio_prof_collect; the next enabled launch re-creates them) */
int dlio_prof_release(void);
int dlio_prof_collect(int kind, double* ms, double* flops, double* bytes, int64_t* launches);

/* ---- convolution ------------------------------------------------------
 * replaces nn.Conv2d forward/backward at: pointseg_net.py:18 (conv1a),
 * pointseg_modules.py:96-106 (Fire squeeze/expand), base_net.py:55-71
 * (FlowNet conv), resnet.py:36 + torchvision BasicBlock, lidar_feat_nets.py:
 * 279-304 (Simple-1). */
typedef struct DlioConvDesc {
  int32_t N, Cin, H, W;          /* logical input                         */
  int32_t in_ctot, in_coff;      /* channel slice of the input buffer     */
  int32_t Cout, OH, OW;          /* logical output                        */
  int32_t out_ctot, out_coff;    /* channel slice of the output buffer    */
  int32_t KH, KW, SH, SW, PH, PW;
  int32_t res_ctot, res_coff;    /* channel slice of the residual buffer  */
  int32_t in_relu;               /* with in_scale: x' = max(0, affine(x)) */
} DlioConvDesc;

/* w [Cout][Cin][KH][KW] -> wt [KH*KW][KP][Cout], KP = Cin  rounded up to 16 (mode 0, forward layout)
 * w [Cout][Cin][KH][KW] -> wt [KH*KW][KP][Cin],  KP = Cout rounded up to 16, taps reversed
 *                          (mode 1, data-gradient layout for stride-1 convs)
 * rows k >= K are zero.  dlio_conv2d_prep_weight_floats = number of floats of wt. */
size_t dlio_conv2d_prep_weight_floats(int Cout, int Cin, int KH, int KW, int mode);
/* The same transform for many weight tensors in ONE launch (all convolutions of a model, both
 * modes, once per optimizer step).  items_dev: DEVICE array; item i covers the floats
 * [start, start + dlio_conv2d_prep_weight_floats(...)) of the concatenated output index space,
 * starts ascending from 0; total_floats = end of the last item. */
typedef struct DlioPrepItem {
  const float* w;   /* [Cout][Cin][KH][KW] */
  float* wt;        /* prepped layout, dlio_conv2d_prep_weight_floats floats */
  int32_t Cout, Cin, taps, mode;
  int64_t start;
} DlioPrepItem;
int dlio_conv2d_prep_weights_batched(const DlioPrepItem* items_dev, int n_items,
                                     int64_t total_floats, dlio_stream_t stream);
int dlio_conv2d_prep_weight(const float* w, float* wt, int Cout, int Cin, int KH, int KW,
                            int mode, dlio_stream_t stream);

/* y[n,co,oh,ow] = bias[co] + residual[n,co,oh,ow]
 *               + sum_{ci,dy,dx} wt[dy*KW+dx][ci][co] * x'[n,ci,oh*SH-PH+dy,ow*SW-PW+dx]
 * x' = x, or max(0,(x-in_mean[ci])*in_scale[ci]+in_shift[ci]) when in_scale != NULL
 * (zero padding is applied AFTER the transform).  bias/residual/in_* may be NULL.
 * residual may alias y (accumulate).  Implicit GEMM on v_mfma_f32_32x32x2_f32. */
int dlio_conv2d_fwd(const float* x, const float* wt, const float* bias,
                    const float* in_mean, const float* in_scale, const float* in_shift,
                    const float* residual, float* y, const DlioConvDesc* d,
                    dlio_stream_t stream);

/* 3x3 stride-1 convolution with fp32 accuracy on the bf16 matrix cores (same nn.Conv2d call sites
 * as dlio_conv2d_fwd, 3x3 layers): operands are split into three bf16 pieces and each product is
 * formed from six bf16 MFMAs with fp32 accumulation (csrc/conv_bx3.hip).  Weights are pre-split
 * by dlio_conv3x3_bx3_prep (mode 0 forward, mode 1 data gradient: taps reversed, channels
 * transposed) into dlio_conv3x3_bx3_prep_floats floats of storage; desc as for dlio_conv2d_fwd
 * (KH = KW = 3, SH = SW = 1), bias / residual nullable. */
size_t dlio_conv_bx3_prep_floats(int Cout, int Cin, int taps, int mode);      /* taps = KH*KW: 9 or 1 */
int dlio_conv_bx3_prep(const float* w, void* wt, int Cout, int Cin, int taps, int mode, dlio_stream_t stream);
/* 1x1 stride-1 convolution on the same scheme (pixels % 4 == 0 and 16-byte aligned rows, else
 * DLIO_EUNSUP: use dlio_conv2d_fwd); weights from dlio_conv_bx3_prep(..., taps = 1, mode) */
int dlio_conv1x1_bx3_fwd(const float* x, const void* wt, const float* bias, const float* residual,
                         float* y, const DlioConvDesc* desc, dlio_stream_t stream);
/* the same with the apply-on-load transform of dlio_conv2d_fwd on the input (in_scale NULL = none): the stored
 * input is the producer's raw output, (x - in_mean[ci]) * in_scale[ci] + in_shift[ci] (+ ReLU when desc->in_relu)
 * is formed while the operand is split */
int dlio_conv1x1_bx3_fwd_aff(const float* x, const void* wt, const float* bias, const float* in_mean,
                             const float* in_scale, const float* in_shift, const float* residual,
                             float* y, const DlioConvDesc* desc, dlio_stream_t stream);
/* stride-1 convolution with 3x3 / 3x2 / 2x2 / 2x1 / 1x2 / 1x1 taps and an EXPLICIT output extent (desc->OH / OW up to
 * K-1 beyond the symmetric-padding formula: the extra outputs read zero padding) on the split-bf16 kernel: the input phases
 * of a strided layer's data gradient (FlowNet conv2-6, ResNet layer2-4: lidar_feat_nets.py:248-257, resnet.py:27-47, whose
 * backward is torch autograd's conv2d data gradient).  weights from dlio_conv_bx3_prep(taps = KH * KW, mode).
 * DLIO_EUNSUP for other tap windows or strides (use dlio_conv2d_fwd). */
int dlio_conv_bx3_fwd_taps(const float* x, const void* wt, const float* bias, const float* residual, float* y,
                           const DlioConvDesc* desc, dlio_stream_t stream);
/* dlio_conv3x5s2_bx3_fwd / dlio_conv_bx3_fwd_taps on the two-piece fp16 split: *amax_x = the largest |x| (one device float
 * left by the producer of x: amax_out of the BatchNorm launches), wt from dlio_conv_h2_prep(taps = KH * KW, mode); layers with
 * more than 32 output channels, else DLIO_EUNSUP.  FlowNet conv2-6 (lidar_feat_nets.py:248-257), ResNet's strided stage heads
 * (resnet.py:27-47) and the phases of their data gradients. */
int dlio_conv_h2_fwd_strided(const float* x, const float* amax_x, const void* wt, const float* bias,
                             const float* residual, float* y, const DlioConvDesc* d, dlio_stream_t stream);
int dlio_conv_h2_fwd_taps(const float* x, const float* amax_x, const void* wt, const float* bias,
                          const float* residual, float* y, const DlioConvDesc* d, dlio_stream_t stream);
/* the split-bf16 kernel on strided layers, forward: 3x5 taps with stride (1, 2) (the PointSeg stem, pointseg_net.py:18-20;
 * FlowNet conv2 / conv3) and 3x3 taps with stride (2, 2) (FlowNet conv4-6, lidar_feat_nets.py:252-257; ResNet layer2-4,
 * resnet.py:27-47); weights from dlio_conv_bx3_prep(taps = KH * KW, mode 0); DLIO_EUNSUP for anything else */
int dlio_conv3x5s2_bx3_fwd(const float* x, const void* wt, const float* bias, const float* residual, float* y,
                           const DlioConvDesc* desc, dlio_stream_t stream);
/* dlio_conv1x1_bx3_fwd_aff with scratch: narrowing layers on few pixels (>= 192 input channels, < 65536 pixels, too
 * few workgroups to fill the chip) split their channel loop over workgroups, write fp32 partial tiles to ws and sum
 * them (fixed order, + bias + residual) in a second launch.  dlio_conv1x1_bx3_ws_bytes(desc) = bytes that split
 * wants (0: the layer runs unsplit); with less (or ws NULL) the call runs unsplit. */
size_t dlio_conv1x1_bx3_ws_bytes(const DlioConvDesc* desc);
int dlio_conv1x1_bx3_fwd_ws(const float* x, const void* wt, const float* bias, const float* in_mean,
                            const float* in_scale, const float* in_shift, const float* residual,
                            float* y, void* ws, size_t ws_bytes, const DlioConvDesc* desc,
                            dlio_stream_t stream);
/* dlio_conv3x3_bx3_fwd with scratch: long channel loops on small feature maps (>= 6 chunks of 16 channels, a launch that
 * fills at most 3/4 of the chip's workgroup slots: the blk4 / blk5 data gradients) split their channel loop over
 * workgroups, write fp32 partial tiles to ws and sum them (fixed order, + bias + residual) in a second launch.
 * dlio_conv3x3_bx3_ws_bytes(desc) = bytes that split wants (0: unsplit); with less (or ws NULL) the call runs unsplit. */
size_t dlio_conv3x3_bx3_ws_bytes(const DlioConvDesc* desc);
int dlio_conv3x3_bx3_fwd_ws(const float* x, const void* wt, const float* bias, const float* residual, float* y,
                            void* ws, size_t ws_bytes, const DlioConvDesc* desc, dlio_stream_t stream);
/* y = conv3x3(x, wt) + conv1x1(x1, wt1) (+ residual) in one launch: the data gradient of a Fire block's expand pair,
 * dS = W3^T * dE3 + W1^T dE1 (autograd's conv2d backward of pointseg_modules.py:126-133).  desc as for
 * dlio_conv3x3_bx3_fwd_ws with PH = PW = 1 and OH x OW = H x W; x1 contiguous [N][C1][H][W]; wt / wt1 from
 * dlio_conv_bx3_prep(taps = 9 / 1, mode 1); ws as dlio_conv3x3_bx3_ws_bytes(desc). */
int dlio_fire_expand_dgrad(const float* x, const void* wt, const float* x1, const void* wt1, int C1, const float* residual,
                           float* y, void* ws, size_t ws_bytes, const DlioConvDesc* desc, dlio_stream_t stream);
size_t dlio_conv3x3_bx3_prep_floats(int Cout, int Cin, int mode);
int dlio_conv3x3_bx3_prep(const float* w, void* wt, int Cout, int Cin, int mode, dlio_stream_t stream);
/* every split-bf16 layout of a model in one launch (DlioPrepItem as for dlio_conv2d_prep_weights_batched;
 * taps = 9 or 1; start / total count taps * ceil(K/16) * N * 16 elements per item) */
int dlio_conv3x3_bx3_prep_batched(const DlioPrepItem* items_dev, int n_items, int64_t total,
                                  dlio_stream_t stream);
int dlio_conv3x3_bx3_fwd(const float* x, const void* wt, const float* bias, const float* residual,
                         float* y, const DlioConvDesc* desc, dlio_stream_t stream);
/* the same convolution on a TWO-piece fp16 split of x 2^k (three MFMAs per product instead of six): *amax_x = the
 * largest magnitude in x (on the device, from the kernel that produced x: dlio_bn_coop_bwd's amax_out), wt from
 * dlio_conv_h2_prep(taps 9).  Only the launch sizes of the producer / consumer kernel (the 3x3 data gradients of
 * fire_blk1-3: dlio_conv3x3_h2_ok), DLIO_EUNSUP otherwise. */
int dlio_conv3x3_h2_ok(const DlioConvDesc* desc);
/* the 1x1 stride-1 convolution of dlio_conv1x1_bx3_fwd_ws on the same two-piece split (the squeeze / expand1x1 data
 * gradients, pointseg_modules.py:96,101 in backward): wt from dlio_conv_h2_prep(taps 1); ws as dlio_conv1x1_bx3_ws_bytes. */
int dlio_conv1x1_h2_fwd(const float* x, const float* amax_x, const void* wt, const float* bias, const float* residual,
                        float* y, void* ws, size_t ws_bytes, const DlioConvDesc* desc, dlio_stream_t stream);
int dlio_conv3x3_h2_fwd(const float* x, const float* amax_x, const void* wt, const float* bias, const float* residual,
                        float* y, const DlioConvDesc* desc, dlio_stream_t stream);

/* ---- Fire expand pair as one launch (csrc/fire_expand.hip) ----------------------------------------------------------
 * Replaces  torch.cat([self.expand1x1(s), self.expand3x3(s)], 1)  of Fire.forward (pointseg_modules.py:126-133; the
 * two Conv2d of pointseg_modules.py:100-106) where both halves have E channels.
 * The squeeze activation travels as "planes": [N][ceil(S/16)][3][H + 2][W + 2][16] bf16 -- every fp32 value as its
 * three bf16 pieces (hi, mid, lo: exact), position-major, 16-channel chunks, with the zero border of the 3x3 padding
 * stored; dlio_fire_planes_bytes(N, S, H, W) bytes.
 * dlio_bn_split16: the squeeze BatchNorm2d (+ ReLU) (pointseg_modules.py:98-99,122-124) applied to the raw squeeze
 * output x (channel slice of x_ctot), writing the activated fp32 tensor y (nullable) AND the planes (border included).
 * mode 0: train-mode statistics + apply (two launches; mean / invstd / scale / running statistics as
 * dlio_bn_train_apply, ws = dlio_chan_stats_ws_bytes(N, C, H * W)); mode 1: the statistics partials only; mode 2:
 * apply from the partials in ws with count = N * H * W * count_scale (SyncBN: all-reduce between 1 and 2); mode 3:
 * eval, mean / scale are inputs (dlio_bn_eval_params).  W % 4 == 0, 16-byte aligned pointers, else DLIO_EUNSUP.
 * dlio_fire_expand_fwd: y[:, y_coff : y_coff + E] = expand1x1(s) + bias1, y[:, y_coff + E : y_coff + 2 E] =
 * expand3x3(s) + bias3 (3x3, pad 1); w3t / w1t from dlio_conv_bx3_prep(taps = 9 / 1, mode 0); biases nullable. */
size_t dlio_fire_planes_bytes(int N, int S, int H, int W);
int dlio_bn_split16(const float* x, int N, int x_ctot, int x_coff, int C, int H, int W, int post_relu,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean, float* invstd, float* scale, float* y, int y_ctot, int y_coff,
                    void* planes, void* ws, size_t ws_bytes, int mode, double count_scale, float* bound_out,
                    dlio_stream_t stream);
int dlio_fire_expand_fwd(const void* planes, const void* w3t, const void* w1t, const float* bias3, const float* bias1,
                         float* y, int N, int S, int H, int W, int E, int y_ctot, int y_coff, int planes_fmt,
                         dlio_stream_t stream);
/* planes_fmt 1 -- the two-piece format: planes written by dlio_bn_split16 with mode + 16 (train modes 0 / 2 only: two fp16
 * pieces of x 2^k per value, [N][ceil(S/16)][2][H + 2][W + 2][16], 2^-k as a float behind them; same allocation size) and
 * weights from dlio_conv_h2_prep (mode 0; two fp16 pieces of w 2^j, [tap][chunk][2][n][16], then { 2^-j, 2^j }): three
 * v_mfma_f32_32x32x16_f16 per product instead of six bf16 ones, error ~1e-7 of the result.  2^k comes from the
 * tensor's EXACT largest magnitude in mode 0 (the statistics pass leaves every channel's min / max behind its partial sums in
 * ws -- dlio_chan_stats_ws_bytes has room for them --, BatchNorm is monotone per channel: max |BN(x)| = max over channels of
 * max(|BN(min)|, |BN(max)|)); in mode 2 (partials computed elsewhere) from the bound |beta| + |gamma| sqrt(N H W), which holds
 * for batch statistics.  bound_out (nullable): that magnitude as a device float, for later two-piece consumers of y
 * (dlio_conv3x3_wgrad_h2); 2^j from the weight tensor's largest magnitude. */
/* (the last two of the dlio_conv_h2_prep_floats floats are scratch of the magnitude pass -- several workgroups per tensor
 * meet there -- and must be ZERO before dlio_conv_h2_prep_batched; dlio_conv_h2_prep zeroes them and every launch leaves
 * them zero, so a layout that went through dlio_conv_h2_prep once can be refreshed by the batched call from then on;
 * wt -- here and in the items of the two batched calls -- 16-byte aligned: the layouts are written in 16-byte pieces,
 * DLIO_EUNSUP from dlio_conv_h2_prep otherwise) */
size_t dlio_conv_h2_prep_floats(int Cout, int Cin, int taps, int mode);
int dlio_conv_h2_prep(const float* w, void* wt, int Cout, int Cin, int taps, int mode, dlio_stream_t stream);
int dlio_conv_h2_prep_batched(const DlioPrepItem* items_dev, int n_items, int64_t total, dlio_stream_t stream);
/* The same launch that also leaves per-tile channel sums, + one small launch that turns them into the train-mode BatchNorm
 * statistics of BOTH expand layers (mean / invstd / scale = gamma * invstd / shift = beta over the 2 E channels of the concat
 * buffer, running statistics updated with `momentum`): an apply-on-load block (functional.FireFn(defer=True)) needs no pass
 * over the concat buffer for its statistics.  ws: dlio_fire_expand_stats_ws_bytes() bytes. */
size_t dlio_fire_expand_stats_ws_bytes(int N, int H, int W, int E);
int dlio_fire_expand_fwd_stats(const void* planes, const void* w3t, const void* w1t, const float* bias3, const float* bias1,
                               float* y, int N, int S, int H, int W, int E, int y_ctot, int y_coff, const float* gamma1,
                               const float* beta1, float* running_mean1, float* running_var1, const float* gamma3,
                               const float* beta3, float* running_mean3, float* running_var3, float eps, float momentum,
                               float* mean, float* invstd, float* scale, float* shift, void* ws, size_t ws_bytes,
                               int planes_fmt, dlio_stream_t stream);
/* dst [planes][HU][WU] = src [planes][OH][OW] with SH-1 / SW-1 zeros inserted between rows /
 * columns (and zero tail rows/columns up to HU, WU): turns the data gradient of a strided
 * convolution into dlio_conv2d_fwd with stride 1 on the data-gradient weight layout
 * (HU = (OH-1)*SH+1 + (H+2*PH-KH)%SH, pad KH-1-PH). */
int dlio_zero_upsample2d(const float* src, float* dst, int64_t planes, int OH, int OW, int HU,
                         int WU, int SH, int SW, dlio_stream_t stream);
/* Phase decomposition of a strided data gradient (same nn.Conv2d backward lines): input rows
 * ih = SH*i + a only receive the taps kh = (a+PH) mod SH, +SH, ... -- so dX[a::SH, b::SW] is a
 * STRIDE-1 convolution of dY with that tap subset (dlio_conv2d_fwd on the subset's data-gradient
 * layout, no MFMA work on inserted zeros).  This call weaves the SH*SW phase results
 * ([N][C][ceil((H-a)/SH)][ceil((W-b)/SW)] each, index a*SW+b, NULL = phase without taps) back
 * into dX (channel slice) and adds the optional residual.  SH*SW <= 4. */
int dlio_phase_interleave2d(const float* const* phases, int SH, int SW, const float* residual,
                            int r_ctot, int r_coff, float* dx, int dx_ctot, int dx_coff, int N,
                            int C, int H, int W, dlio_stream_t stream);
/* strided data gradient (any stride), scalar reference implementation: dx[n,ci,ih,iw] = sum_{co,dy,dx} dy[...] * w[co][ci][dy][dx]
 * w in the STANDARD [Cout][Cin][KH][KW] layout.  d describes the forward conv. */
int dlio_conv2d_dgrad_strided(const float* dy, const float* w, float* dx,
                              const DlioConvDesc* d, dlio_stream_t stream);

/* dw[co][ci][dy][dx] = sum_{n,oh,ow} dy[n,co,oh,ow] * x'[n,ci,oh*SH-PH+dy,ow*SW-PW+dx]
 * standard weight layout; deterministic two-stage split-K through ws.  accumulate: dw += ...
 * (gradient written straight into the optimizer's flat gradient buffer). */
size_t dlio_conv2d_wgrad_ws_bytes(const DlioConvDesc* d);
int dlio_conv2d_wgrad(const float* x, const float* dy, float* dw,
                      const float* in_mean, const float* in_scale, const float* in_shift,
                      void* ws, size_t ws_bytes, int accumulate, const DlioConvDesc* d,
                      dlio_stream_t stream);
/* the 3x3 stride-1 pad-1 weight gradient (pointseg_modules.py:103 expand3x3 in backward) on the two-piece fp16 split: both
 * operands as two fp16 pieces of x 2^k, three v_mfma_f32_32x32x16_f16 per product instead of six bf16 ones.
 * amax_x / amax_dy: device floats with the largest magnitude of x / dy or a bound on it (dlio_bn_split16 bound_out,
 * dlio_bn_coop_bwd amax_out); ws as dlio_conv2d_wgrad_ws_bytes(d).  DLIO_EUNSUP where dlio_conv3x3_wgrad_h2_ok(d) is 0.
 * Also takes the 3x5 stride-(1, 2) pad-(1, 2) layers (FlowNet conv2 / conv3, lidar_feat_nets.py:248-251): their two column
 * phases are 3x3 stride-1 weight gradients (dlio_conv2d_wgrad does the same on three pieces). */
int dlio_conv3x3_wgrad_h2_ok(const DlioConvDesc* d);
int dlio_conv3x3_wgrad_h2(const float* x, const float* amax_x, const float* dy, const float* amax_dy, float* dw, void* ws,
                          size_t ws_bytes, int accumulate, const DlioConvDesc* d, dlio_stream_t stream);

/* ---- per-channel reductions / batch norm --------------------------------
 * replaces nn.BatchNorm2d (train + eval) at pointseg_net.py:19,
 * pointseg_modules.py:98-106, base_net.py:63, resnet.py:38, lidar_feat_nets.py:
 * 280-301 and the conv bias gradient.
 * Statistics are accumulated in fp64. */
size_t dlio_chan_stats_ws_bytes(int N, int C, int HW);
/* sum[c] = sum x, sumsq[c] = sum x^2 over (n, hw) of channel slice; x' = relu(x) if pre_relu */
int dlio_chan_stats(const float* x, int N, int ctot, int coff, int C, int HW, int pre_relu,
                    double* sum, double* sumsq, void* ws, size_t ws_bytes, dlio_stream_t stream);
/* mean/var(biased)/invstd, scale=gamma*invstd; running stats updated in place
 * with momentum and the unbiased variance (nn.BatchNorm2d semantics). */
int dlio_bn_finalize(const double* sum, const double* sumsq, int C, double count,
                     const float* gamma, float eps, float momentum,
                     float* running_mean, float* running_var,
                     float* mean, float* invstd, float* scale, dlio_stream_t stream);
/* dlio_chan_stats + dlio_bn_finalize in two launches instead of three (count = N*HW).
 * shift_out (nullable, [C]): receives beta -- with mean and scale the (mean, scale, shift) table the
 * in_mean / in_scale / in_shift operands of the consumers take when this BatchNorm is applied on load. */
int dlio_bn_train_stats(const float* x, int N, int ctot, int coff, int C, int HW, int pre_relu,
                        const float* gamma, float eps, float momentum, float* running_mean,
                        float* running_var, float* mean, float* invstd, float* scale, void* ws,
                        size_t ws_bytes, const float* beta, float* shift_out, int phase, double count_scale,
                        dlio_stream_t stream);       /* phase / count_scale: SyncBN split as for dlio_bn_train_apply */
/* eval mode: mean=running_mean, invstd=rsqrt(running_var+eps), scale=gamma*invstd */
int dlio_bn_eval_params(const float* running_mean, const float* running_var, const float* gamma,
                        float eps, int C, float* mean, float* invstd, float* scale,
                        dlio_stream_t stream);
/* y = post( (pre(x) - mean[c]) * scale[c] + beta[c] ) + residual
 * pre = relu if pre_relu, post = relu if post_relu.  y may alias x. */
int dlio_bn_apply(const float* x, int x_ctot, int x_coff, const float* mean, const float* scale,
                  const float* beta, const float* residual, int r_ctot, int r_coff,
                  float* y, int y_ctot, int y_coff, int N, int C, int HW,
                  int pre_relu, int post_relu, dlio_stream_t stream);
/* backward reductions: g = dy * [post_relu ? y>0 : 1];  xh = (pre(x)-mean)*invstd
 * sum_g[c] = sum g, sum_gx[c] = sum g*xh; optional fp32 copies dgamma = sum_gx, dbeta = sum_g */
int dlio_bn_bwd_reduce(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot,
                       int x_coff, const float* mean, const float* invstd, const float* scale,
                       const float* beta, int N, int C, int HW, int pre_relu, int post_relu,
                       double* sum_g, double* sum_gx, float* dgamma, float* dbeta, int accumulate,
                       void* ws, size_t ws_bytes, dlio_stream_t stream);
/* dgamma = sum_gx, dbeta = sum_g (fp32 out);  train: dx = scale*(g - sum_g/M - xh*sum_gx/M)
 * eval (use_batch_stats=0): dx = scale*g.  pre_relu masks dx by x>0. */
int dlio_bn_bwd_apply(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot,
                      int x_coff, const float* mean, const float* invstd, const float* scale,
                      const float* beta, const double* sum_g, const double* sum_gx,
                      float* dx, int dx_ctot, int dx_coff, float* dgamma, float* dbeta,
                      int N, int C, int HW, int pre_relu, int post_relu, int use_batch_stats,
                      dlio_stream_t stream);
/* out[c] (fp32) = sum over (n,hw) of channel slice -- conv bias gradient */
int dlio_chan_sum(const float* x, int N, int ctot, int coff, int C, int HW, float* out,
                  int accumulate, void* ws, size_t ws_bytes, dlio_stream_t stream);

/* ---- pooling ---------------------------------------------------------
 * nn.MaxPool2d(kernel 3) at pointseg_net.py:21-46, resnet.py:40,
 * lidar_feat_nets.py:281-299 (ceil_mode); adaptive_avg_pool2d(1,1) at
 * lidar_feat_nets.py:84-85,258, resnet.py:49, pointseg_modules.py:218.
 * idx holds the argmax window position (dy*KW+dx) as uint8, first max wins
 * (ATen tie rule).  x_scale (nullable, [N][C]) multiplies the input plane
 * (fused SELayer channel re-weighting: pointseg_modules.py:220). */
int dlio_maxpool2d_fwd(const float* x, const float* x_scale, float* y, uint8_t* idx,
                       int N, int C, int H, int W, int OH, int OW, int K, int SH, int SW,
                       int PH, int PW, dlio_stream_t stream);
/* the 3x3, pad 1, stride (1|2, 2) pool over max(0, (x - aff[0][c]) * aff[1][c] + aff[2][c]) (apply-on-load: x is the
 * raw output of the convolution in front of the pool, aff = [3][C] mean / scale / shift; PointSeg stem -> pool1,
 * pointseg_net.py:18-21); DLIO_EUNSUP for other pool shapes */
int dlio_maxpool2d_fwd_aff(const float* x, const float* aff, float* y, uint8_t* idx, int N, int C, int H, int W,
                           int OH, int OW, int K, int SH, int SW, int PH, int PW, dlio_stream_t stream);
/* dx = scatter(dy) * x_scale[n][c] + x_add[n][c]; x_scale / x_add nullable (x_add only on the
 * 3x3, pad 1, stride (1|2, 2) fast path: SELayer + pool backward in one pass) */
int dlio_maxpool2d_bwd(const float* dy, const uint8_t* idx, const float* x_scale,
                       const float* x_add, float* dx, int N, int C, int H, int W, int OH, int OW,
                       int K, int SH, int SW, int PH, int PW, dlio_stream_t stream);
/* out[plane] = (sum_hw a[plane][hw] * b[plane][hw]) / div[plane]  (div nullable = 1; div == 0 gives 0): the SELayer scale
 * gradient behind a max-pool from POOLED tensors -- with y = maxpool(x * s), s > 0: d loss / d s = sum dy * y / s
 * (pointseg_modules.py:203-221 + the MaxPool2d behind it; autograd's mul / max_pool2d backward) */
int dlio_plane_dot(const float* a, const float* b, const float* div, float* out, int planes, int HW, dlio_stream_t stream);
/* pair_fuse_fc -- the head of the siamese lidar feature nets (lidar_feat_nets.py:84-94, :131-141) in one launch:
 * feat [N][C] = mean_hw a[n][c] (+ mode 0 | - mode 1) mean_hw b[n][c]; y [N][F] = act(feat w^T + bias) (w [F][C] = nn.Linear.weight,
 * bias nullable, act as dlio_linear_fwd: 0 none, 1 relu, 2 leaky 0.01, 3 sigmoid, 4 tanh).  a, b contiguous [N][C][HW].
 * ws: dlio_pair_fuse_fc_ws_bytes() bytes of scratch; counters: N ints, ZERO before the first use, restored by the kernel
 * (the workgroup that arrives last at an image's counter adds the per-channel-block partial products in block order).
 * dlio_pair_fuse_bwd: da [N][C][HW] = df [N][C] / HW broadcast, db = (+|-) the same -- autograd's backward of the two
 * adaptive_avg_pool2d and the add / sub. */
size_t dlio_pair_fuse_fc_ws_bytes(int N, int C, int F);
int dlio_pair_fuse_fc_fwd(const float* a, const float* b, int N, int C, int HW, int mode, const float* w, const float* bias,
                          int F, int act, float* feat, float* y, void* ws, size_t ws_bytes, int* counters,
                          dlio_stream_t stream);
int dlio_pair_fuse_bwd(const float* df, float* da, float* db, int N, int C, int HW, int mode, dlio_stream_t stream);
/* The two bias-free fully connected layers of an SELayer (pointseg_modules.py:207-212, 217-219: Linear(C, R) -> ReLU ->
 * Linear(R, C) -> Sigmoid on the [N, C] plane averages g) in one launch: h [N][R] = relu(g w1^T), s [N][C] = sigmoid(h w2^T);
 * w1 [R][C], w2 [C][R] row-major (nn.Linear.weight).  C <= 1024, R <= 512, both multiples of 4 (dlio_se_fc_ok), else
 * DLIO_EUNSUP.  Backward in two launches: dz2 = ds * s (1 - s), dz1 = (dz2 w2) [h > 0], dg = dg_scale * dz1 w1 (dz2 [N][C],
 * dz1 [N][R] are scratch outputs); dw1 [R][C] (+)= dz1^T g, dw2 [C][R] (+)= dz2^T h (accumulate != 0: added to what is there). */
int dlio_se_fc_ok(int N, int C, int R);
int dlio_se_fc_fwd(const float* g, const float* w1, const float* w2, float* h, float* s, int N, int C, int R,
                   dlio_stream_t stream);
int dlio_se_fc_bwd(const float* ds, const float* s, const float* h, const float* g, const float* w1, const float* w2,
                   float* dz2, float* dz1, float* dg, float dg_scale, float* dw1, float* dw2, int accumulate, int N, int C,
                   int R, dlio_stream_t stream);
/* ds[n][c] = sum_hw scatter(dy) * x  without materialising scatter(dy) (fast-path shapes only,
 * DLIO_EUNSUP otherwise) */
int dlio_maxpool2d_bwd_dot(const float* dy, const uint8_t* idx, const float* x, float* ds, int N,
                           int C, int H, int W, int OH, int OW, int K, int SH, int SW, int PH,
                           int PW, dlio_stream_t stream);
/* out[n][c] = mean over HW of channel slice */
int dlio_gap_fwd(const float* x, int ctot, int coff, float* out, int N, int C, int HW,
                 dlio_stream_t stream);
/* dx[n][c][hw] (+)= dout[n][c] / HW */
int dlio_gap_bwd(const float* dout, float* dx, int N, int C, int HW, int accumulate,
                 dlio_stream_t stream);
/* SELayer scale (pointseg_modules.py:220): y = x * s[n][c] */
int dlio_chan_scale_fwd(const float* x, const float* s, float* y, int N, int C, int HW,
                        dlio_stream_t stream);
/* dx = dy * s ; ds[n][c] = sum_hw dy*x */
int dlio_chan_scale_bwd(const float* dy, const float* x, const float* s, float* dx, float* ds,
                        int N, int C, int HW, dlio_stream_t stream);

/* ---- dense layers -----------------------------------------------------
 * nn.Linear at lidar_feat_nets.py:66, pointseg_modules.py:209-214,
 * imu_feat_nets.py:27-30, fusion_nets.py:59, odom_feat_nets.py:18-20,
 * deeplio_nets.py:57-58 and the LSTM/GRU input/recurrent projections.
 * act: 0 none, 1 relu, 2 leaky_relu(0.01), 3 sigmoid, 4 tanh.
 * y[m][n] = act( sum_k x[m*ldx+k] * w[n*K+k] + b[n] + addend[m*ldadd+n] ) */
int dlio_linear_fwd(const float* x, int ldx, const float* w, const float* b,
                    const float* addend, int ldadd, float* y, int ldy,
                    int M, int N, int K, int act, dlio_stream_t stream);
/* dz = dy * act'(y) (in terms of the OUTPUT y); dz may alias dy */
int dlio_act_bwd(const float* dy, const float* y, float* dz, int64_t n, int act,
                 dlio_stream_t stream);
/* dx[m][k] (+)= sum_n dz[m*lddz+n] * w[n*K+k].  With a workspace of
 * dlio_linear_bwd_data_ws_bytes the large weight-streaming form is used (ws may be NULL). */
size_t dlio_linear_bwd_data_ws_bytes(int M, int N, int K);
int dlio_linear_bwd_data(const float* dz, int lddz, const float* w, float* dx, int lddx,
                         int M, int N, int K, int accumulate, void* ws, size_t ws_bytes,
                         dlio_stream_t stream);
/* dw[n][k] (+)= sum_m dz[m][n]*x[m][k];  db[n] (+)= sum_m dz[m][n]  (db may be NULL) */
int dlio_linear_bwd_weight(const float* dz, int lddz, const float* x, int ldx, float* dw,
                           float* db, int M, int N, int K, int accumulate,
                           dlio_stream_t stream);

/* ---- elementwise helpers ---------------------------------------------- */
/* op: 0 a+b, 1 a-b, 2 a*b, 3 relu(a+b) (BasicBlock tail) */
/* *amax_out (zero before the launch) = max |x|: the operand scale of a two-piece fp16 convolution for a tensor no kernel of
 * ours produced (the stem's input images) */
int dlio_abs_max(const float* x, int64_t n, float* amax_out, dlio_stream_t stream);
int dlio_ew_binary(const float* a, const float* b, float* y, int64_t n, int op, float* amax_out,
                   dlio_stream_t stream);
/* torch.sum(y, dim=0) per IMU window (imu_feat_nets.py:49): y[g][c] = sum_r x[g][r][c];
 * backward broadcasts dy over r */
int dlio_seg_sum_fwd(const float* x, float* y, int groups, int rows, int cols, dlio_stream_t stream);
int dlio_seg_sum_bwd(const float* dy, float* dx, int groups, int rows, int cols, dlio_stream_t stream);
/* y = alpha*a */
int dlio_ew_scale(const float* a, float alpha, float* y, int64_t n, dlio_stream_t stream);
/* strided 2-D copy: dst[r*ldd + c] = src[r*lds + c] (concat / split along features) */
int dlio_copy2d(const float* src, int lds, float* dst, int ldd, int rows, int cols,
                int accumulate, dlio_stream_t stream);
/* nn.Dropout: mask from Philox4x32-10(seed, offset); y = x*mask/(1-p); mask saved as u8 */
int dlio_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p,
                     uint64_t seed, uint64_t offset, dlio_stream_t stream);
int dlio_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p,
                     dlio_stream_t stream);
/* trainer.py:221-243 NaN/Inf guards: flag[0] |= 1 if any non-finite */
int dlio_nonfinite_flag(const float* x, int64_t n, int32_t* flag, dlio_stream_t stream);

/* ---- recurrent layers ---------------------------------------------------
 * nn.LSTM / nn.GRU at imu_feat_nets.py:63-70 (state carried over sub-
 * sequences, :79-83) and odom_feat_nets.py:61-68.  One call = one layer, one
 * direction, one sequence.  gx = x W_ih^T + b_ih is computed by dlio_linear_fwd
 * beforehand.  Gate order i,f,g,o (LSTM) / r,z,n (GRU), G = 4 / 3.
 * Sequence tensors are addressed by ROW r(t,b) = t*rst + b*rsb (batch-first:
 * rst=1, rsb=T; time-major: rst=B, rsb=1):
 *   gx, dgates : [rows][G*H]      gates (saved, post-activation) : [rows][4*H]
 *   cs, hp     : [rows][H]        (c_t, and the h that ENTERED step t)
 *   hs / dhs   : row stride ldhs / lddhs floats (one half of a bidirectional
 *                [rows][2H] buffer)
 * Persistent single-workgroup kernel (W_hh resident in registers, h/c/gates in
 * LDS) when H is 32, 64 or 128; per-step streamed kernels otherwise.
 * GRU saves r,z,n in gates[..][0..3H) and hn = W_hn h + b_hn in [3H..4H). */
size_t dlio_rnn_ws_bytes(int T, int B, int H);
int dlio_lstm_seq_fwd(const float* gx, const float* w_hh, const float* b_hh,
                      const float* h0, const float* c0, float* hs, int ldhs, float* cs,
                      float* hp, float* gates, float* hT, float* cT, int T, int B, int H,
                      int rst, int rsb, int reverse, void* ws, size_t ws_bytes,
                      dlio_stream_t stream);
/* dhs: gradient wrt every h_t as seen by the consumer; dhT/dcT: gradient wrt
 * the final state (nullable).  Outputs dgates (pre-activation gate grads, same
 * row addressing as gx), dh0, dc0. */
int dlio_lstm_seq_bwd(const float* dhs, int lddhs, const float* dhT, const float* dcT,
                      const float* gates, const float* cs, const float* c0,
                      const float* w_hh, float* dgates, float* dh0, float* dc0,
                      int T, int B, int H, int rst, int rsb, int reverse, void* ws,
                      size_t ws_bytes, dlio_stream_t stream);
int dlio_gru_seq_fwd(const float* gx, const float* w_hh, const float* b_hh, const float* h0,
                     float* hs, int ldhs, float* hp, float* gates, float* hT, int T, int B,
                     int H, int rst, int rsb, int reverse, void* ws, size_t ws_bytes,
                     dlio_stream_t stream);
/* outputs dgx[rows][3H] (grad wrt gx) and dgh[rows][3H] (grad wrt W_hh h + b_hh), dh0 */
int dlio_gru_seq_bwd(const float* dhs, int lddhs, const float* dhT, const float* gates,
                     const float* hp, const float* w_hh, float* dgx, float* dgh, float* dh0,
                     int T, int B, int H, int rst, int rsb, int reverse, void* ws,
                     size_t ws_bytes, dlio_stream_t stream);

/* ---- streamed LSTM LAYER, both directions per launch ---------------------
 * nn.LSTM(256 -> 1024, num_layers 2, bidirectional, batch_first) of OdomFeatRNN
 * (odom_feat_nets.py:61-68, forward :72-83) over the S axis: one call = one layer,
 * D = 1 | 2 directions, the whole sequence of T steps, ZERO initial state.  Rows are
 * r(t, b) = b T + t (batch-first).  x [rows][ldx >= I]; hs [rows][ldhs >= D H]: direction d
 * writes columns d H .. d H + H (torch's layout of a bidirectional output); saved for
 * backward: cs, hp [D][rows][H] (c_t and the h that ENTERED step t), gates [D][rows][4H]
 * (post-activation, order i, f, g, o).  Parameters of direction 1 may be NULL when D = 1.
 * Geometry: B <= 8, H and I multiples of 256 (dlio_lstm_layer_ok); ws from
 * dlio_lstm_layer_ws_bytes (the K-slice / N-slab partial sums; fixed summation order).
 * Backward: dhs [rows][lddhs] = gradient w.r.t. hs (both directions' columns); writes
 * dgates [D][rows][4H] (scratch the caller owns), the eight parameter gradients
 * (accumulate != 0: added to what is there) and, unless NULL, dx [rows][lddx >= I] summed
 * over both directions. */
int dlio_lstm_layer_ok(int T, int B, int I, int H, int D);
size_t dlio_lstm_layer_ws_bytes(int T, int B, int I, int H, int D);
int dlio_lstm_layer_fwd(const float* x, int ldx, const float* w_ih0, const float* w_hh0,
                        const float* b_ih0, const float* b_hh0, const float* w_ih1,
                        const float* w_hh1, const float* b_ih1, const float* b_hh1, float* hs,
                        int ldhs, float* cs, float* hp, float* gates, int T, int B, int I, int H,
                        int D, void* ws, size_t ws_bytes, dlio_stream_t stream);
/* the weight-gradient launch of dlio_lstm_layer_bwd on its own (dlio_lstm_layer_bwd with dw_ih0 == NULL skips it): nothing
 * downstream on the tape reads these gradients, the caller may issue it on a companion stream behind the data path */
int dlio_lstm_layer_wgrad(const float* dgates, const float* x, int ldx, const float* hp, float* dw_ih0,
                          float* dw_hh0, float* db_ih0, float* db_hh0, float* dw_ih1, float* dw_hh1,
                          float* db_ih1, float* db_hh1, int accumulate, int T, int B, int I, int H,
                          int D, dlio_stream_t stream);
int dlio_lstm_layer_bwd(const float* dhs, int lddhs, const float* x, int ldx, const float* hp,
                        const float* gates, const float* cs, const float* w_ih0,
                        const float* w_hh0, const float* w_ih1, const float* w_hh1, float* dgates,
                        float* dw_ih0, float* dw_hh0, float* db_ih0, float* db_hh0, float* dw_ih1,
                        float* dw_hh1, float* db_ih1, float* db_hh1, int accumulate, float* dx,
                        int lddx, int T, int B, int I, int H, int D, void* ws, size_t ws_bytes,
                        dlio_stream_t stream);

/* ---- the small layers of the step's serial middle, one launch each ------
 * DeepLIOFusionSoft.forward (fusion_nets.py:64-75): cat = [a | b] ([R][Fa], [R][Fb]);
 * s1 = sigmoid(cat W1^T + b1) ([Fa][Fa+Fb] weights), s2 = sigmoid(cat W2^T + b2);
 * out [R][Fa+Fb] = [a s1 | b s2]; gate [R][Fa+Fb] = [s1 | s2] (saved for backward, the
 * module's s1_feat / s2_feat); a / b are read with row strides lda / ldb (the IMU feature is
 * a strided slice of the RNN output: no copy).  Fa + Fb <= 512 (dlio_soft_fusion_ok).  Backward writes
 * da, db and the four parameter gradients (accumulate != 0: added). */
int dlio_soft_fusion_ok(int R, int Fa, int Fb);
int dlio_soft_fusion_fwd(const float* a, int lda, const float* b, int ldb, const float* w1,
                         const float* b1, const float* w2, const float* b2, float* out,
                         float* gate, int R, int Fa, int Fb, dlio_stream_t stream);
int dlio_soft_fusion_bwd(const float* dout, const float* a, int lda, const float* b, int ldb,
                         const float* gate, const float* w1, const float* w2, float* da,
                         float* db, float* dw1, float* dbias1, float* dw2, float* dbias2, int R,
                         int Fa, int Fb, int accumulate, dlio_stream_t stream);
/* DeepLIO.forward's last lines (deeplio_nets.py:84-90): y = dropout(x, p); pos = fc_pos(y),
 * ori = fc_ori(y) (Linear(K, 3) each).  x [R][ldx >= K] is read in place (the forward half
 * of the odometry LSTM's [.., 2H] output: no slice copy); mask [R][K] u8 (NULL: no dropout)
 * is drawn at the Philox position (seed, offset) dlio_dropout_fwd over a contiguous [R][K]
 * tensor would use.  Backward: dx [R][lddx >= K] (columns K.. are zeroed; NULL: skipped),
 * the four parameter gradients. */
int dlio_heads_ok(int R, int K, int ldx);
int dlio_heads_fwd(const float* x, int ldx, uint8_t* mask, const float* wp, const float* bp,
                   const float* wo, const float* bo, float* pos, float* ori, int R, int K,
                   float p, uint64_t seed, uint64_t offset, dlio_stream_t stream);
int dlio_heads_bwd(const float* dpos, const float* dori, const float* x, int ldx,
                   const uint8_t* mask, const float* wp, const float* wo, float* dx, int lddx,
                   float* dwp, float* dbp, float* dwo, float* dbo, int R, int K, float p,
                   int accumulate, dlio_stream_t stream);

/* ---- pose chain + loss --------------------------------------------------
 * Trainer.se3_to_SE3 (trainer.py:324-351): per batch element chain
 * R_s = R_{s-1} exp(w_s), t_s = R_{s-1} t + t_{s-1}; q_s = quat_wxyz(R_s)
 * (tester.py:223-251 variant: order=1 -> xyzw).  status[0] bit 0 is set when a
 * determinant check (|det-1| > 1e-5+1e-8) fails, mirroring the ValueError; bit 1 when a
 * product failed liegroups' validity test (tol 1e-6) and the quaternion was taken from its
 * re-orthonormalisation, as SO3.from_matrix(normalize=True) does (trainer.py:349).
 * R_all [B][S][9] is saved for backward. */
int dlio_se3_chain_fwd(const float* t, const float* w, float* p, float* q, float* R_all,
                       int32_t* status, int B, int S, int order, dlio_stream_t stream);
int dlio_se3_chain_bwd(const float* t, const float* w, const float* R_all, const float* dp,
                       const float* dq, float* dt, float* dw, int B, int S, int order,
                       dlio_stream_t stream);
/* SO3.normalize of liegroups (the SVD projection behind from_matrix(normalize=True),
 * trainer.py:349) as an SVD-free Newton polar iteration, for n 3x3 matrices: Q = R when R passes
 * the validity test (valid[i] = 1, nullable), else its orthogonal polar factor.  _bwd: dR from
 * G = dL/dQ (first order in the defect of R, which is <= 1e-5 for a chain of exponentials). */
int dlio_so3_project(const float* R, float* Q, int32_t* valid, int n, dlio_stream_t stream);
int dlio_so3_project_bwd(const float* R, const float* G, float* dR, int n, dlio_stream_t stream);
/* HWSLoss / LWSLoss (losses/losses.py:21-39, 68-86).  Eight [B][S*][D] inputs
 * are passed as 4 (pred, gt) pairs with element counts n[4] (0 = term off).
 * mode bit 0: 0 = HWS with sx,sq; 1 = LWS with beta.  mode bit 1 (BASELINE configs[4], no
 * reference counterpart): the two rotation terms are geodesic instead of MSE --
 * L_w = mean_rows theta(exp(w_pred), exp(w_gt))^2, L_q = mean_rows theta(q_pred, q_gt)^2 with
 * theta = 2 atan2(|a ^ b|, |<a,b>|) (= 2 acos|<a,b>| for unit quaternions = |log(Ra^T Rb)|).
 * out: loss[0], the four terms [1..4].  Order of pairs: f2f_t, f2f_w, f2g_p, f2g_q. */
int dlio_pose_loss_fwd(const float* const* pred, const float* const* gt, const int32_t* n,
                       const float* sx, const float* sq, float beta, int mode, float* out,
                       dlio_stream_t stream);
/* dpred[i] = gscale * d loss / d pred[i]; dsx, dsq (HWS only, nullable) */
int dlio_pose_loss_bwd(const float* const* pred, const float* const* gt, const int32_t* n,
                       const float* sx, const float* sq, float beta, int mode,
                       const float* out, const float* gscale, float* const* dpred,
                       float* dsx, float* dsq, dlio_stream_t stream);

/* The trainer's tail -- se3_to_SE3 (trainer.py:324-351), the criterion on (f2f_t, f2f_w, f2g_p[:, g0:g1], f2g_q[:, g0:g1])
 * against gt_f2f [B][S][6] = (t | w) and gt_f2g [B][S][7] = (p | q) (trainer.py:245-252; losses/losses.py:21-39,68-86) and the
 * torch.isnan / isinf check of the model output (trainer.py:240-243) -- in two launches each way; the slices are read in
 * place.  terms: bit 0 local (f2f), bit 1 global (f2g) (losses/__init__.py:10-18); mode, sx, sq, beta, out [5] as
 * dlio_pose_loss_fwd; order, status, p, q, R_all as dlio_se3_chain_fwd; nonfinite (nullable): |= 1 on a non-finite t / w.
 * _bwd: dt, dw = gscale * d loss / d (t, w) (local terms + the chain's backward of the global terms); dsx, dsq (HWS, nullable):
 * acc_hyper != 0 adds to what they hold (they are views of the flat gradient buffer then).  ws: dlio_pose_tail_ws_floats. */
int dlio_pose_tail_fwd(const float* t, const float* w, const float* gt_f2f, const float* gt_f2g, int B, int S,
                       int g0, int g1, int terms, const float* sx, const float* sq, float beta, int mode,
                       int order, float* p, float* q, float* R_all, int32_t* status, int32_t* nonfinite,
                       float* out, dlio_stream_t stream);
int dlio_pose_tail_bwd(const float* t, const float* w, const float* gt_f2f, const float* gt_f2g, int B, int S,
                       int g0, int g1, int terms, const float* sx, const float* sq, float beta, int mode,
                       int order, const float* p, const float* q, const float* R_all, const float* out,
                       const float* gscale, float* ws, float* dt, float* dw, float* dsx, float* dsq,
                       int acc_hyper, dlio_stream_t stream);
size_t dlio_pose_tail_ws_floats(int B, int S, int g0, int g1);

/* ---- batch preparation ----------------------------------------------------
 * DataCombiCreater.process (models/misc.py:24-125) on the device.
 * pair_stack: images [B][F][Ctot][H][W], combinations [S][2] (int32, device) ->
 *   xyz [B][S][2][c_split][H][W] and normals [B][S][2][Ctot-c_split][H][W]
 *   (misc.py:65-69: imgs[:, combinations] then channel split 0:3 | 3:).  H*W % 4 == 0.
 * gt_relative: gts [B][F][15] rows [x(3), R(9), v(3)] -> f2f [B][S][6] = [dx, log(R)] of
 *   T_i^-1 T_{i+1} and f2g [B][S][7] = [p, quat wxyz] of T_0^-1 T_{i+1} (misc.py:83-125);
 *   flag[0] |= 1 on a non-finite f2f entry (the reference raises ValueError). */
int dlio_pair_stack(const float* images, const int32_t* combinations, float* xyz, float* normals,
                    int B, int F, int Ctot, int c_split, int H, int W, int S, dlio_stream_t stream);
int dlio_gt_relative(const float* gts, const int32_t* combinations, float* f2f, float* f2g,
                     int32_t* flag, int B, int F, int S, dlio_stream_t stream);

/* Train-mode BatchNorm forward in two launches: split statistics of x (as dlio_bn_train_stats),
 * then ONE plane-structured kernel in which every workgroup sums the partials of its own channel,
 * finalises mean / invstd / scale (+ running statistics, published by the workgroup of plane
 * n=0) and applies y = post((pre(x)-mean)*scale+beta) + residual like dlio_bn_apply.
 * gap_out (optional, [N][gap_ctot] with channel offset gap_coff): mean over HW of every OUTPUT
 * plane, bit-identical to dlio_gap_fwd(y) -- the SELayer behind a Fire block needs it.
 * ws: dlio_chan_stats_ws_bytes(N, C, HW); its first C * dlio_chan_stats_splits(N,C,HW) * 2 doubles
 * are the (sum, sum of squares) partials.
 * Synchronised statistics across data-parallel replicas (SyncBN): call with phase = 1 (partials
 * only), all-reduce(sum) those doubles over the replicas, call again with phase = 2 (apply only)
 * and count_scale = number of replicas.  phase = 0, count_scale = 1: the whole thing.
 * r_mean / r_scale / r_shift (nullable, [r_ctot] each): the residual buffer holds the producer's RAW
 * convolution output and is activated on load, residual' = max(0, (r - r_mean) * r_scale + r_shift)
 * (apply-on-load: the producing Fire block's BatchNorm+ReLU output is never written). */
int dlio_chan_stats_splits(int N, int C, int HW);
/* amax_out (nullable; also on dlio_bn_small_fwd / dlio_bn_coop_fwd / dlio_bn_bwd / dlio_ew_binary): one device float, ZERO
 * before the launch, that receives the largest |y| written -- the operand scale of a two-piece fp16 consumer
 * (dlio_conv3x3_h2_fwd, dlio_conv3x3_wgrad_h2) without a pass of its own. */
int dlio_bn_train_apply(const float* x, int N, int x_ctot, int x_coff, int C, int HW, int pre_relu,
                        int post_relu, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, float* mean,
                        float* invstd, float* scale, const float* residual, int r_ctot, int r_coff,
                        float* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot, int gap_coff,
                        void* ws, size_t ws_bytes, int phase, double count_scale,
                        const float* r_mean, const float* r_scale, const float* r_shift,
                        float* amax_out, dlio_stream_t stream);
/* Train-mode BatchNorm2d (+ ReLU, + residual) of SMALL feature maps in ONE launch (csrc/bn_small.hip; the BatchNorm2d of
 * pointseg_modules.py:98-106 in fire_blk4 / fire_blk5): N <= 16 images, H * W in {256, 512, 1024, 2048} (dlio_bn_small_ok),
 * 16-byte aligned planes; DLIO_EUNSUP otherwise (use dlio_bn_train_apply / dlio_bn_bwd).  One workgroup holds a channel
 * in registers: each element is read once.  The launch covers the channels [0, C) of the slice; channels [0, C1) use
 * parameter set 1, [C1, C) set 2 (the expand1x1 / expand3x3 halves of a Fire block's concat buffer; C1 = C: one layer).
 * forward: mean / invstd / scale [C] are written (shift_out nullable: receives beta, the third row of an apply-on-load
 * table); y NULL = statistics only; residual / r_mean / r_scale / r_shift / gap_out as dlio_bn_train_apply.
 * backward: dx of channels [0, C1) -> dx1 [N][C1][HW], of [C1, C) -> dx2 [N][C - C1][HW]; dgamma / dbeta per set
 * (nullable; accumulated when accumulate != 0); beta1 / beta2 are needed for the ReLU mask (post_relu). */
int dlio_bn_small_ok(int N, int HW);
int dlio_bn_small_fwd(const float* x, int N, int x_ctot, int x_coff, int C, int C1, int HW, int post_relu,
                      const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                      const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                      float eps, float momentum, float* mean, float* invstd, float* scale, float* shift_out,
                      const float* residual, int r_ctot, int r_coff, const float* r_mean, const float* r_scale,
                      const float* r_shift, float* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot,
                      int gap_coff, float* amax_out, dlio_stream_t stream);
int dlio_bn_small_bwd(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot, int x_coff,
                      const float* mean, const float* invstd, const float* scale, const float* beta1,
                      const float* beta2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                      float* dbeta2, int accumulate, int N, int C, int C1, int HW, int post_relu,
                      float* amax_out, dlio_stream_t stream);
/* The same for LARGE feature maps (fire_blk1-3: 32-128 KB per (n, c) plane): the N * parts workgroups that hold a channel's
 * planes in registers exchange partial sums through `part` ([C][N * parts][2] 64-bit slots, every one holding
 * dlio_bn_coop_empty() before the first use: a slot is its own arrival flag) and `sync` ([C + 4] ints, zero before the first
 * use: [0] error flag, [1] ticket dispenser, [2] workgroups that have left, [4 + c] departures of channel c); the kernels
 * restore everything but the error flag; sync[0] != 0 afterwards = a spin limit was hit, results invalid (re-initialise both
 * buffers).  One launch, each element read once (dlio_bn_train_apply reads twice, dlio_bn_bwd's two launches five times
 * against three).  A plane may be cut into dlio_bn_coop_parts(N, H * W) workgroups (N * parts <= 256 slot pairs per channel;
 * with gap_out the parts of a plane exchange their plane sums through a third slot region of `part`; the bf16 kernels keep
 * the plane in one workgroup then: dlio_bn_coop_gap_ok).  2 <= N <= 64, H * W a multiple of 8192 up to 65536
 * (dlio_bn_coop_ok), else DLIO_EUNSUP.  Arguments as dlio_bn_small_fwd / _bwd (no statistics-only mode: y required).
 * Items are handed out in order by the ticket dispenser and a workgroup loads its next item under the exchange of the
 * current one; progress needs N * parts workgroups of a launch resident at a time (checked against the occupancy query for
 * two concurrent launches, else DLIO_EUNSUP), not the whole grid: any number of launches may be in flight. */
int dlio_bn_coop_ok(int N, int HW);
/* CUs a cooperative launch sizes its grid for (0 = the default, 5 / 8 of the chip; the grid is CUs x the occupancy query's
 * workgroups per CU).  A cap, not a correctness condition.  cus < 0 (tests): exactly -cus workgroups per launch, whatever the geometry --
 * fewer than half the cooperating workgroups of a channel cannot make progress and end in the error flag. */
int dlio_bn_coop_set_cus(int cus);
/* 2: persistent workgroups (grid as above), one item at a time, the next ticket drawn when the item is done;
 * 1: one item per workgroup, the grid covers the items -- the partners of a channel are then whichever workgroups the
 * dispatcher starts next, a workgroup that has drawn its ticket keeps its slot until its channel is complete, and the eight
 * XCDs dispatch their shares of a grid independently: the tickets still to be drawn may all belong to workgroups of ONE XCD,
 * which start only if that XCD has a free slot.  Two such launches of N * parts = 64 partners (the two encoders' fire_blk1
 * layers) can fill an XCD's 96 slots with waiting workgroups of both: measured in the training step, one step in ~150 stalls
 * for the 140 ms of the spin limit, both launches at once, and never when the launches are chained one after the other (the
 * host mirror does that in mode 1: deeplio_amd/ops.py _coop_enter / _coop_exit -- slower than mode 2);
 * 3 (default): per launch -- one item per workgroup when 3 (N * parts - 1) < occupancy x CUs / 8, i.e. when up to THREE
 * concurrent cooperative launches cannot fill an XCD with waiting workgroups, persistent otherwise (at N = 16: fire_blk2 /
 * blk3 one item per workgroup, fire_blk1 persistent; a caller that keeps more than three cooperative launches in flight at
 * once uses mode 2);
 * 0: persistent workgroups that load their next item under the exchange of the current one (twice the registers);
 * -1: back to the default (environment DLIO_BN_COOP_MODE). */
int dlio_bn_coop_set_mode(int oneshot);
int dlio_bn_coop_get_mode(void);                 /* the mode in force (3 / 2 / 1 / 0) */
/* 1: a cooperative launch of this geometry runs one item per workgroup under the mode in force -- it counts towards mode 3's
 * "at most three in flight" and, in mode 1, must not overlap another such launch; 0: persistent workgroups (or no cooperative
 * kernel for the geometry).  Needs the device (occupancy query). */
int dlio_bn_coop_one_item(int N, int HW);
int dlio_bn_coop_parts(int N, int HW);
int dlio_bn_coop_gap_ok(int N, int HW);        /* bf16 kernels with gap_out: the plane in one workgroup, H * W in {8192, 16384, 32768} */
size_t dlio_bn_coop_ws_bytes(int N, int C);      /* bytes of `part` */
unsigned long long dlio_bn_coop_empty(void);
int dlio_bn_coop_fwd(const float* x, int N, int x_ctot, int x_coff, int C, int C1, int HW, int post_relu,
                     const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                     const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                     float eps, float momentum, float* mean, float* invstd, float* scale,
                     const float* residual, int r_ctot, int r_coff, const float* r_mean, const float* r_scale,
                     const float* r_shift, float* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot,
                     int gap_coff, void* part, void* sync, float* amax_out, dlio_stream_t stream);
int dlio_bn_coop_bwd(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot, int x_coff,
                     const float* mean, const float* invstd, const float* scale, const float* beta1,
                     const float* beta2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                     float* dbeta2, int accumulate, int N, int C, int C1, int HW, int post_relu, void* part,
                     void* sync, float* amax_out, dlio_stream_t stream);
/* (amax_out, nullable: one float on the device, ZERO before the launch; afterwards max |dx1|, |dx2| -- what the two-piece
 *  split kernels take their power-of-two scale from, dlio_conv3x3_h2_fwd) */
/* The same launch behind an SELayer + MaxPool2d(3, stride (SH, 2), padding 1) (the end of a PSEncoder block,
 * pointseg_net.py:27-46; SELayer pointseg_modules.py:216-221): the gradient of the BatchNorm output is not stored but
 * formed while loading, x_scale[n, c] * route(dy_pooled through idx) + x_add[n, c] (+ dy, nullable: the part of the gradient
 * that is stored) -- what dlio_maxpool2d_bwd(x_scale, x_add) would have written, at a quarter of the traffic.  dy_pooled /
 * idx: [N][C][OH][OW] (C = all channels of this launch), x_scale / x_add [N][C] nullable.  dlio_bn_coop_pool_ok(N, H, W, SH):
 * the geometry rule (dlio_bn_coop_ok, W a multiple of 16 with W / 8 dividing 64 and the workgroup size). */
int dlio_bn_coop_pool_ok(int N, int H, int W, int SH);
int dlio_bn_coop_bwd_pool(const float* dy, int dy_ctot, int dy_coff, const float* dy_pooled, const unsigned char* idx,
                          const float* x_scale, const float* x_add, int H, int W, int SH, const float* x, int x_ctot,
                          int x_coff, const float* mean, const float* invstd, const float* scale, const float* beta1,
                          const float* beta2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                          float* dbeta2, int accumulate, int N, int C, int C1, int post_relu, void* part, void* sync,
                          float* amax_out, dlio_stream_t stream);
/* Train-mode BatchNorm2d + ReLU (+ bypass residual) as a streaming apply for layers whose statistics are known before the
 * pass (dlio_fire_expand_fwd_stats leaves mean / scale / shift rows over the concat buffer's channels; pointseg_modules.py:
 * 100-106,126-133): y = max(0, (x - mean[c]) * scale[c] + shift[c]) + r, r = residual as stored, or -- r_scale given: the
 * residual is a deferred block's raw output -- max(0, (r - r_mean) * r_scale + r_shift) with rows indexed by r_coff + c.
 * gap_out (nullable): plane averages of y, [N][gap_ctot] at gap_coff.  H * W a multiple of 4, 16-byte aligned tensors. */
int dlio_bn_aff_apply(const float* x, int N, int x_ctot, int x_coff, int C, int HW, const float* mean,
                      const float* scale, const float* shift, const float* residual, int r_ctot, int r_coff,
                      const float* r_mean, const float* r_scale, const float* r_shift, float* y, int y_ctot,
                      int y_coff, float* gap_out, int gap_ctot, int gap_coff, dlio_stream_t stream);
/* The same followed by MaxPool2d(3, stride (SH, 2), padding 1) WITHOUT writing y (a Fire block in front of SELayer + MaxPool,
 * pointseg_net.py:27-46; SELayer pointseg_modules.py:207-221: its scale s = sigmoid(..) > 0, so maxpool(s * y) = s *
 * maxpool(y) with the same arg-max): y_pooled [N][C][OH][OW] = the pooled maximum of y (to be scaled by the caller:
 * dlio_chan_scale_fwd), idx = the tap codes of dlio_maxpool2d_fwd (bit-identical to pooling the materialised y), gap_out
 * (nullable) = plane averages of y.  SH in {1, 2}, W a multiple of 4, H even for SH = 2 (dlio_bn_aff_pool_ok). */
int dlio_bn_aff_pool_ok(int H, int W, int SH);
int dlio_bn_aff_pool_fwd(const float* x, int N, int x_ctot, int x_coff, int C, int H, int W, int SH,
                         const float* mean, const float* scale, const float* shift, const float* residual,
                         int r_ctot, int r_coff, const float* r_mean, const float* r_scale, const float* r_shift,
                         float* y_pooled, unsigned char* idx, float* gap_out, int gap_ctot, int gap_coff,
                         dlio_stream_t stream);
/* the cooperative one-launch kernels over bf16 storage (train mode, one layer per launch; arithmetic and rounding as
 * dlio_bn_bf16_apply / dlio_bn_bf16_bwd, BASELINE configs[4]); geometry rule of dlio_bn_coop_ok in elements; part / sync as
 * for dlio_bn_coop_fwd */
int dlio_bn_bf16_coop_fwd(const void* x, int N, int x_ctot, int x_coff, int C, int HW, int post_relu, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          float* mean, float* invstd, float* scale, const void* residual, int r_ctot, int r_coff, void* y,
                          int y_ctot, int y_coff, float* gap_out, int gap_ctot, int gap_coff, void* part, void* sync,
                          dlio_stream_t stream);
int dlio_bn_bf16_coop_bwd(const void* dy, int dy_ctot, int dy_coff, const void* x, int x_ctot, int x_coff,
                          const float* mean, const float* invstd, const float* scale, const float* beta, void* dx,
                          int dx_ctot, int dx_coff, float* dgamma, float* dbeta, int accumulate, int N, int C, int HW,
                          int post_relu, void* part, void* sync, dlio_stream_t stream);
/* Train-mode BatchNorm (+ ReLU) backward behind a 3x3 / pad 1 / stride (SH, 2) max-pool (dlio_maxpool2d_fwd_aff): both
 * launches gather the gradient of the activated tensor from the pooled gradient dy_pool [N,C,OH,OW] and the arg-max map
 * while they stream x [N,C,H,W] -- the pool's own backward pass and its output are not needed.  dx contiguous;
 * ws as dlio_chan_stats_ws_bytes(N, C, H * W); DLIO_EUNSUP for other pool shapes / unaligned tensors.
 * Per-replica statistics only: there is no phase / count_scale pair here, so a caller that synchronises BatchNorm
 * statistics over ranks must take dlio_maxpool2d_bwd + dlio_bn_bwd (phases 1 / 2) instead. */
int dlio_bn_bwd_pool(const float* dy_pool, const uint8_t* idx, const float* x, const float* mean,
                     const float* invstd, const float* scale, const float* beta, float* dx, float* dgamma,
                     float* dbeta, int accumulate, int N, int C, int H, int W, int OH, int OW, int SH, void* ws,
                     size_t ws_bytes, dlio_stream_t stream);
/* BatchNorm backward in two launches: dlio_bn_bwd_reduce's reduction + a plane-structured
 * dlio_bn_bwd_apply that sums the partials itself; dgamma / dbeta (optional, += when accumulate)
 * are written by the workgroup of plane n=0.  phase / count_scale as above; local_ws (phase 2,
 * optional): a copy of this replica's partials taken BEFORE the all-reduce -- dgamma / dbeta
 * stay per-replica sums (the gradient all-reduce adds the others), dx uses the global sums. */
int dlio_bn_bwd(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot, int x_coff,
                const float* mean, const float* invstd, const float* scale, const float* beta,
                float* dx, int dx_ctot, int dx_coff, float* dgamma, float* dbeta, int accumulate,
                int N, int C, int HW, int pre_relu, int post_relu, int use_batch_stats, void* ws,
                size_t ws_bytes, int phase, double count_scale, const void* local_ws,
                float* amax_out, dlio_stream_t stream);

/* ---- lidar scan -> range image (the data step in front of the path) --------
 * LaserScan.do_range_projection (deeplio/common/laserscan.py:122-185): per point
 * depth = |p|, yaw = -atan2(y,x), pitch = asin(z/depth),
 * proj_x = floor(0.5*(yaw/pi+1)*W), proj_y = floor((1-(pitch+|fov_down|)/fov)*H), clamped;
 * float32 arithmetic in the reference's operation order, atan2/asin correctly rounded.
 * Pixels take the CLOSEST of their points (the reference scatters in decreasing-depth order);
 * equal depths: smallest point index.  Empty pixels are 0 in every output (laserscan.py:27-60).
 * points [N,3], remissions [N] or NULL; proj_x/proj_y/unproj_range [N];
 * proj_range/proj_remission/proj_idx/proj_mask [H,W], proj_xyz [H,W,3]; proj_mask may be NULL.
 * ws: dlio_scan_project_ws_bytes(H, W) bytes of scratch. */
size_t dlio_scan_project_ws_bytes(int H, int W);
int dlio_scan_project(const float* points, const float* remissions, int N, int H, int W,
                      double fov_up_deg, double fov_down_deg, int32_t* proj_x, int32_t* proj_y,
                      float* unproj_range, float* proj_range, float* proj_xyz,
                      float* proj_remission, int32_t* proj_idx, int32_t* proj_mask, void* ws,
                      size_t ws_bytes, dlio_stream_t stream);
/* LaserScan.do_normal_projection (laserscan.py:215-248): range-weighted cross products of the
 * four neighbour differences, normalised, zero border.  normals [H,W,3]. */
int dlio_scan_normals(const float* proj_xyz, const float* proj_range, float* normals, int H,
                      int W, dlio_stream_t stream);
/* Kitti.get_velo_image (deeplio/datasets/kitti.py:83-97) + transform_images (:345-364):
 * image = (xyz/max_depth, remission, normal, range) -> crop -> CHW -> minus mean -> channel select.
 * channels: n_channels HOST ints in 0..7; mean: 8 HOST floats indexed by original channel or
 * NULL.  out [n_channels][H-2*crop_top][W-2*crop_left]. */
int dlio_velo_image(const float* proj_xyz, const float* proj_remission, const float* normals,
                    const float* proj_range, float max_depth, const int32_t* channels,
                    const float* mean, int n_channels, int H, int W, int crop_top, int crop_left,
                    float* out, dlio_stream_t stream);

/* ---- mixed precision (BASELINE configs[4]) ---------------------------------
 * bf16 storage of the PointSeg encoders' activations and activation gradients (NCHW, channel slices
 * as above), fp32 master weights, fp32 accumulation, fp32/fp64 BatchNorm statistics, fp32 weight
 * gradients.  No reference counterpart (the reference is fp32 only); same call sites as the fp32
 * entry points they mirror.  void* tensors are bf16; H*W must be a multiple of 8 (16-byte accesses).
 *
 * Convolutions (csrc/conv_bf16.hip): one v_mfma_f32_32x32x16_bf16 per product.  Weights are re-laid-out
 * from the fp32 master copy into [tap][ceil(K/16)][N][16] bf16 (mode 0 forward: K = Cin, N = Cout;
 * mode 1 data gradient: K = Cout, N = Cin, taps reversed); the batched form takes DlioPrepItem as
 * dlio_conv3x3_bx3_prep_batched does.  bias fp32, residual / y bf16 (residual may alias y). */
size_t dlio_conv_bf16_prep_elems(int Cout, int Cin, int taps, int mode);
int dlio_conv_bf16_prep(const float* w, void* wt, int Cout, int Cin, int taps, int mode, dlio_stream_t stream);
int dlio_conv_bf16_prep_batched(const DlioPrepItem* items_dev, int n_items, int64_t total, dlio_stream_t stream);
int dlio_conv3x3_bf16_fwd(const void* x, const void* wt, const float* bias, const void* residual, void* y,
                          const DlioConvDesc* desc, dlio_stream_t stream);
int dlio_conv1x1_bf16_fwd(const void* x, const void* wt, const float* bias, const void* residual, void* y,
                          const DlioConvDesc* desc, dlio_stream_t stream);
/* dW (fp32, accumulated when accumulate != 0) from bf16 x and dy: 3x3 and 1x1 stride-1 layers; ws as
 * dlio_conv2d_wgrad_ws_bytes(desc) */
int dlio_conv2d_wgrad_bf16(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, int accumulate,
                           const DlioConvDesc* desc, dlio_stream_t stream);
/* train-mode BatchNorm (+ReLU, + residual, + plane averages of the stored output) over bf16:
 * statistics and apply in two launches; eval_mode != 0: apply only, mean / invstd / scale are inputs
 * (dlio_bn_eval_params).  ws: dlio_bf16_stats_ws_bytes(N, C, HW).  phase / count_scale as dlio_bn_train_apply:
 * 0 = both launches, 1 = the partial sums only ([C][splits][2] doubles at the start of ws: the caller all-reduces
 * them over the data-parallel replicas), 2 = apply from the partials in ws with count = N * HW * count_scale. */
int dlio_bf16_stats_splits(int N, int C, int HW);
size_t dlio_bf16_stats_ws_bytes(int N, int C, int HW);
int dlio_bn_bf16_apply(const void* x, int N, int x_ctot, int x_coff, int C, int HW, int post_relu,
                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, float* mean, float* invstd, float* scale, const void* residual,
                       int r_ctot, int r_coff, void* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot,
                       int gap_coff, int eval_mode, void* ws, size_t ws_bytes, int phase, double count_scale,
                       dlio_stream_t stream);
/* backward: reductions + dx (bf16), dgamma / dbeta (fp32, accumulated when accumulate != 0); phase / count_scale as
 * above, local_ws (phase 2, nullable) = this replica's partials before the all-reduce: dgamma / dbeta come from them */
int dlio_bn_bf16_bwd(const void* dy, int dy_ctot, int dy_coff, const void* x, int x_ctot, int x_coff,
                     const float* mean, const float* invstd, const float* scale, const float* beta, void* dx,
                     int dx_ctot, int dx_coff, float* dgamma, float* dbeta, int accumulate, int N, int C, int HW,
                     int post_relu, int use_batch_stats, void* ws, size_t ws_bytes, int phase, double count_scale,
                     const void* local_ws, dlio_stream_t stream);
/* 3x3 max-pool, padding 1, stride (1|2, 2), W % 16 == 0, with the fused SELayer scale (x_scale [N*C] fp32,
 * nullable); idx uint8 = kh*3 + kw of the first maximum.  bwd: dx = x_scale * scatter(dy) + x_add[plane];
 * bwd_dot: ds[plane] = sum dy * x[arg-max]. */
int dlio_maxpool_bf16_fwd(const void* x, const float* x_scale, void* y, uint8_t* idx, int N, int C, int H, int W,
                          int OH, int OW, int K, int SH, int SW, int PH, int PW, dlio_stream_t stream);
int dlio_maxpool_bf16_bwd(const void* dy, const uint8_t* idx, const float* x_scale, const float* x_add, void* dx,
                          int N, int C, int H, int W, int OH, int OW, int K, int SH, int SW, int PH, int PW,
                          dlio_stream_t stream);
int dlio_maxpool_bf16_bwd_dot(const void* dy, const uint8_t* idx, const void* x, float* ds, int N, int C, int H,
                              int W, int OH, int OW, int K, int SH, int SW, int PH, int PW, dlio_stream_t stream);
/* global average pool bf16 -> fp32 [N][C] and its backward fp32 -> bf16 */
int dlio_gap_bf16_fwd(const void* x, int ctot, int coff, float* out, int N, int C, int HW, dlio_stream_t stream);
int dlio_gap_bf16_bwd(const float* dout, void* dx, int N, int C, int HW, dlio_stream_t stream);
/* dir 0: fp32 -> bf16 (round to nearest even); dir 1: bf16 -> fp32; n % 8 == 0 */
int dlio_cast_bf16(const void* src, void* dst, int64_t n, int dir, dlio_stream_t stream);

/* ---- optimizer ----------------------------------------------------------
 * torch.optim.Adam / SGD(momentum) / RMSprop / Adadelta as built by create_optimizer
 * (optimizer.py:4-16) over ONE flat parameter buffer: weight decay is L2 added
 * to the gradient.  step is the 1-based step count. grad_scale multiplies g
 * first (data-parallel averaging). */
/* workgroups the sweeps below may use (0 = default, 8 per CU): a sweep issued under the backward pass (the tail bucket's early
 * update, FlatOptimizer.step_early) runs as background work on a small grid */
int dlio_optim_set_max_blocks(int blocks);
int dlio_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step,
                   float grad_scale, dlio_stream_t stream);
int dlio_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                  float weight_decay, int step, float grad_scale, dlio_stream_t stream);
/* torch.optim.RMSprop (optimizer.py:12-13): square_avg = alpha*square_avg + (1-alpha) g^2;
 * centered (grad_avg != NULL): avg = sqrt(square_avg - grad_avg^2) + eps; momentum != 0
 * (momentum_buf != NULL): buf = momentum*buf + g/avg, p -= lr*buf; else p -= lr*g/avg. */
int dlio_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buf,
                      float* grad_avg, int64_t n, float lr, float alpha, float eps,
                      float weight_decay, float momentum, float grad_scale, dlio_stream_t stream);
/* torch.optim.Adadelta (optimizer.py:14-15): square_avg = rho*square_avg + (1-rho) g^2;
 * delta = sqrt(acc_delta+eps)/sqrt(square_avg+eps) * g; acc_delta = rho*acc_delta + (1-rho) delta^2;
 * p -= lr*delta. */
int dlio_adadelta_step(float* p, const float* g, float* square_avg, float* acc_delta, int64_t n,
                       float lr, float rho, float eps, float weight_decay, float grad_scale,
                       dlio_stream_t stream);
/* out[0] = sum g^2 (fp64) -- calc_grad_norm, trainer.py:481-486 */
int dlio_sumsq(const float* g, int64_t n, double* out, dlio_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPLIO_HIP_H */
