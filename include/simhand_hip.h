/*
 * simhand_hip.h -- C ABI of libsimhand_hip.so (gfx950 / MI355X).
 *
 * The reference (ut-vision/SiMHand) has no C/FFI/plugin API: its operator
 * boundary is Python (torch ops dispatched to cuDNN/cuBLAS/ATen).  Each entry
 * point below names the reference call site (file:line under /root/reference)
 * whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a live torch
 *     tensor); no ownership transfer, no allocation inside the library;
 *   - workspaces are caller-allocated after a *_workspace_bytes() query;
 *   - every launch takes an explicit hipStream_t (as void*) and is asynchronous.
 *     The entry points of THIS header hold no mutable global state that a caller can observe: a call's result and kernel
 *     choice depend on its arguments only, concurrent callers on different streams
 *     do not interact.  The library's instruments -- the optional event profiler
 *     (simhand_prof_*), the route counters (simhand_route_*) and the simhand_test_*
 *     hooks (relaxed atomics that pick between kernels computing the same result) --
 *     are process-global by design, are NOT part of the drop-in surface and are
 *     declared in include/simhand_hip_test.h; a product caller never includes it.
 *     Two device-side scratch areas are written but never read: the SINK pages
 *     g_c64_sink (64 KB) and g_r128_sink (64 KB) of the 3x3 ring kernels
 *     (conv3x3_c64.hip, conv3x3_ring.hip) receive the stores of pad positions, so
 *     that every wave issues the same number of stores per step (the hand-counted
 *     s_waitcnt of the LDS-DMA rings depends on it).  Every block writes them,
 *     nobody loads from them: concurrent launches may interleave there freely;
 *     the zero pages (g_zero_page, g_c64_zero_page) are read-only after load.
 *   - entry points marked EXPERIMENTAL were built, tested and measured slower than
 *     the default path; they stay for the experiment record,
 *     the engine does not call them by default and they may go away;
 *   - return 0 on success, non-zero on error; simhand_last_error() returns a
 *     thread-local message;
 *   - activations are NHWC, conv weights KRSC ([Cout][R][S][Cin]); dtype enum
 *     selects fp32 (parity mode, f32 MFMA) or bf16 (bf16 MFMA, fp32 accumulate).
 */
#ifndef SIMHAND_HIP_H
#define SIMHAND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sh_stream_t; /* hipStream_t */

enum sh_dtype { SH_F32 = 0, SH_BF16 = 1, SH_FP8_E4M3 = 2 /* OCP e4m3fn; operands of the simhand_*_fp8 entry points only */ };

/* joint-distance definitions of get_weights_* (src/models/utils.py:218-388) */
enum sh_dist_mode {
  SH_DIST_MPJPE = 0,  /* mean_j ||dxy||                       :229-231,:251-253 */
  SH_DIST_W_ABS = 1,  /* pos: ||mean_j |d|||_2 ; neg: ||mean_xy |d|||_2 over joints  :224-227,:246-249 */
  SH_DIST_W_O_ABS = 2,/* pos: ||mean_j d||_2   ; neg: ||mean_xy d||_2 over joints    :219-222,:241-244 */
  SH_DIST_L2 = 3      /* plain L2 over F features (the *_with_pca variants :264-301) */
};
enum sh_weight_type { SH_W_NONE = 0, SH_W_LINEAR = 1, SH_W_NONLINEAR = 2,
                      SH_W_EXPLICIT = 3 /* D_loc / d_pos already hold the weights (functional surface) */ };

/* SH_ABI_VERSION is bumped whenever a signature, a struct layout or an enum value of this header changes (history: 1 = rounds 1-2;
 * 2 = round 3: `width` argument of simhand_proj_postprocess_fwd / _bwd / simhand_proj_stats, sh_dgrad_opts grew dy_src / dy_q / wt_q /
 * dy_state / w_state / sub_grad, tuning setters renamed simhand_test_*; 3 = round 4: this constant, the in-library environment switches
 * moved behind simhand_test_* hooks, the stem entry points of DESIGN 3; 4 = round 5: the two-pass / fused-backward stem entry points and
 * their route counters removed (sh_route renumbered from 36 on), SH_SW_COUNT unchanged, and the last argument of simhand_fp8_scale_update
 * became a mode 0 / 1 / 2 with state[1] = 1 / (scale of the existing codes) -- see the FP8 section; round 6 moved the instrument
 * declarations (simhand_test_* / simhand_prof_* / simhand_route_*) to include/simhand_hip_test.h without changing a signature;
 * 5 = round 6: simhand_conv2d_wgrad_colsum_sums, simhand_bn_apply_gram_sums, simhand_bn_bwd_finalize_raw_coefs, simhand_bn_finalize_ticket added (merged
 * parameter-sized launches), SH_SW_FOLD_LEGACY added to sh_test_switch).  A binding compares simhand_abi_version() with the
 * SH_ABI_VERSION it was written against before its first call (simhand_amd/_lib.py load() does) -- a caller built against an older header
 * would otherwise pass shifted arguments or a short options struct unnoticed. */
#define SH_ABI_VERSION 5
int simhand_abi_version(void);
/* The library ships in two builds of the same sources: libsimhand_hip.so, whose 16-bit storage type (SH_BF16 below) is bfloat16 -- and
 * libsimhand_hip_f16.so, where the same enum value means IEEE fp16 (11-bit significand: the storage type of the reference's
 * precision=16 / native AMP, src/experiments/main.py:158-159; the caller scales the loss, see simhand_amd/host/amp.py).  Every entry
 * point, layout and kernel is the same; only the unpack / round-to-nearest-even pack / MFMA operand type differ.  0 = bf16, 1 = fp16. */
int simhand_half_format(void);
const char* simhand_last_error(void);
/* 0 when a gfx950 device is usable from this process */
int simhand_device_check(void);

/* ===========================================================================
 * Loss: similarity-weighted NT-Xent over the gathered global batch
 *   replaces get_weights_linear / get_weights_nonlinear (+_with_pca)
 *   (src/models/utils.py:218-388) and vanila_{,weights_,pos_weights_,neg_weights_}
 *   contrastive_loss (:157-189,:391-501) incl. their autograd backward.
 *
 * Row order of Z_all / J_all is the reference's cat(view1, view2): row k and
 * row B+k form pair k (N = 2B).  This rank owns pairs [pair_off, pair_off+b_loc):
 * local row l < b_loc is global row pair_off+l, local row l >= b_loc is global
 * row B+pair_off+(l-b_loc).  rows_loc = 2*b_loc.
 * =========================================================================== */

/* stats layout (double[8]): [0]=max D, [1]=min D, [2]=sum D (row block),
 * [3]=max d+, [4]=min d+, [5]=sum d+ ; [6],[7] reserved.  The caller all-reduces
 * [0..2] (MAX, MIN, SUM) across ranks between dist and fwd. */
#define SH_NTXENT_NSTATS 8

/* d+_k for all B pairs + its stats; J_all fp32 [N][F] */
int simhand_pos_dist(const float* J_all, int B, int F, int dist_mode,
                     float* d_pos /*[B]*/, double* stats /*[8]*/, sh_stream_t stream);

/* D row block [rows_loc][N] fp32 + stats[0..2] of the block */
size_t simhand_neg_dist_workspace_bytes(int rows_loc, int N);
int simhand_neg_dist(const float* J_all, int B, int F, int dist_mode, int b_loc, int pair_off,
                     float* D_loc /*[rows_loc][N]*/, double* stats /*[8]*/,
                     void* workspace, size_t workspace_bytes, sh_stream_t stream);

/* explicit weights w = f(dist) as returned by get_weights_linear / get_weights_nonlinear
 * (src/models/utils.py:235,:259,:321,:344): positive != 0 uses stats[3..5], else stats[0..2];
 * mean_count = number of entries the mean runs over (B, resp. N*N). */
int simhand_weights_from_dist(const float* dist, int64_t count, int weight_type, const double* stats, int positive,
                              double mean_count, float lambda, float* weights, sh_stream_t stream);

typedef struct sh_ntxent_params {
  int B;            /* global pairs; N = 2B */
  int dim;          /* projection width, must be 128 (output_dim of *_config.json) */
  int b_loc;        /* local pairs */
  int pair_off;     /* first local pair */
  int weight_type;  /* sh_weight_type */
  int use_wpos;     /* pos_neg in {pos_neg,pos} */
  int use_wneg;     /* pos_neg in {pos_neg,neg} */
  float temperature;/* 0.5 */
  float lambda_pos; /* non_linear only */
  float lambda_neg;
} sh_ntxent_params;

size_t simhand_ntxent_workspace_bytes(const sh_ntxent_params* p);
/* forward: neg_loc[rows_loc] = sum_{j!=i} exp(w-_ij s_ij / t); loss_part[0] = (1/N) sum over local rows of
 * (log neg_i - w+_k s_{i,pair(i)} / t)  (sum of loss_part over ranks = the reference's loss). */
int simhand_ntxent_fwd(const sh_ntxent_params* p, const float* Z_all, const float* D_loc, const float* d_pos,
                       const double* stats, float* neg_loc, float* loss_part,
                       void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* backward for the local rows: dZ_loc[rows_loc][dim] = dloss * dL/dz_i (closed form).
 * neg_all[N] = gathered negative sums in global row order; dloss = device scalar or NULL (=1). */
int simhand_ntxent_bwd(const sh_ntxent_params* p, const float* Z_all, const float* D_loc, const float* d_pos,
                       const double* stats, const float* neg_all, const float* dloss, float* dZ_loc,
                       void* workspace, size_t workspace_bytes, sh_stream_t stream);

/* FUSED form (north star: "the B x B cosine-similarity matrix + adaptive MPJPE-weighted softmax-cross-entropy as one
 * LDS-tiled kernel"): the joint-distance tile is computed inside the loss tile kernel next to the similarity tile, from
 * J_all [N][F] (column tile staged in LDS, the lane's own row in registers) -- NO [rows_loc][N] distance block exists in HBM.
 * The statistics (max / min / sum of D) still come from simhand_neg_dist, called with D_loc = NULL (statistics only).
 * Distances are evaluated in the same operation order as simhand_neg_dist: results are bit-identical to the D-block form. */
int simhand_ntxent_fwd_fused(const sh_ntxent_params* p, const float* Z_all, const float* J_all, int F, int dist_mode, const float* d_pos,
                             const double* stats, float* neg_loc, float* loss_part, void* workspace, size_t workspace_bytes,
                             sh_stream_t stream);
int simhand_ntxent_bwd_fused(const sh_ntxent_params* p, const float* Z_all, const float* J_all, int F, int dist_mode, const float* d_pos,
                             const double* stats, const float* neg_all, const float* dloss, float* dZ_loc, void* workspace,
                             size_t workspace_bytes, sh_stream_t stream);

/* ===========================================================================
 * Projection post-process (per row of 128 = 64 2-D points)
 *   replaces F.normalize -> translate_encodings -> rotate_encoding -> F.normalize
 *   (simhand_w_model.py:56-93, src/models/utils.py:606-684) and its backward,
 *   and get_projection_stats (simhand_w_model.py:138-151).
 *   jitter_x/jitter_y: int64 [N] or NULL (flag "crop" off); angle: float64 [N]
 *   degrees or NULL (flag "rotate" off) -- dtypes as collated.
 * =========================================================================== */
enum sh_pp_flags {
  SH_PP_NORM_IN = 1,        /* F.normalize before the un-warp  (simhand_w_model.py:57-58) */
  SH_PP_NORM_OUT = 2,       /* F.normalize after the un-warp   (simhand_w_model.py:92-93) */
  SH_PP_ANGLE_AS_GIVEN = 4  /* angle[] is rotate_encoding's own argument (already negated by the caller) */
};
#define SH_PP_FUSED (SH_PP_NORM_IN | SH_PP_NORM_OUT)
/* the projection kernels view a row as 64 2-D points: `width` (floats per row) MUST be SH_PROJ_DIM (output_dim of every
 * *_config.json of the reference); any other value is refused with an error code, never read out of bounds */
#define SH_PROJ_DIM 128
/* translation: either raw collated jitters (int64; the kernel applies -(j / float(size))) or the ready
 * factors translate_x/translate_y that translate_encodings() receives (fp32) -- never both. */
int simhand_proj_postprocess_fwd(const float* P /*[N][width]*/, int N, int width, const int64_t* jitter_x, const int64_t* jitter_y,
                                 const float* translate_x, const float* translate_y, const double* angle,
                                 int img_h, int img_w, int flags, float* Z /*[N][width]*/, sh_stream_t stream);
int simhand_proj_postprocess_bwd(const float* P, int N, int width, const int64_t* jitter_x, const int64_t* jitter_y,
                                 const float* translate_x, const float* translate_y, const double* angle,
                                 int img_h, int img_w, int flags, const float* dZ, float* dP, sh_stream_t stream);
/* out[8] = batch means of per-row {x_mean,x_median,x_min,x_max,y_mean,y_median,y_min,y_max}; row_ws [N][8] */
int simhand_proj_stats(const float* P, int N, int width, float* row_ws, float* out, sh_stream_t stream);

/* ===========================================================================
 * Backbone operators (replace torchvision resnet -> torch.nn.Conv2d /
 * BatchNorm2d / ReLU / MaxPool2d / AdaptiveAvgPool2d / Linear dispatched to
 * cuDNN/cuBLAS/ATen; call sites src/models/resnet_model.py:13-58,
 * src/models/unsupervised/simclr_model.py:22-39)
 * =========================================================================== */
typedef struct sh_conv_desc {
  int n, h, w, cin;      /* input NHWC */
  int cout, r, s;        /* filter KRSC */
  int stride, pad;
  int ho, wo;            /* output spatial */
  int dtype;             /* sh_dtype of activations and weights */
} sh_conv_desc;

/* y = conv(x, w).  If bn_partial != NULL the epilogue also writes per-row-block
 * partial (sum, sum of squares) of the fp32 accumulators per output channel:
 * bn_partial [nblk_m][2][cout] fp32 with nblk_m = simhand_conv2d_fwd_stat_blocks(). */
int simhand_conv2d_fwd_stat_blocks(const sh_conv_desc* d);
int simhand_conv2d_fwd(const sh_conv_desc* d, const void* x, const void* w, void* y, float* bn_partial, sh_stream_t stream);
/* simhand_bn_apply (ReLU, no residual) + simhand_conv2d_fwd in ONE launch, for the 3x3 layers whose kernel keeps its activation rows in an
 * LDS ring (simhand_conv2d_fwd_bnin_ok(d): the 64 -> 64 ring kernel, e.g. 56 x 56, AND the 128 -> 128 ring kernel, e.g. 28 x 28, as long as
 * n*h*w*cin < 2^32 -- the by-product's element offsets are 32-bit in the kernels): y_in is the previous unit's RAW conv output [n][h][w][cin], the ring
 * rows are rewritten in place as a = relu(y_in * in_scale + in_shift) before any tap reads them (pad positions fetch NaNs, which the ReLU
 * turns into the exact zeros the padding needs), and a leaves as a by-product (a_out, same shape: the weight gradient's operand).  The
 * stand-alone pass -- one read and one write of the tensor -- disappears.  y, bn_partial as simhand_conv2d_fwd(d, a, ...); a_out, y and
 * bn_partial are bit-identical to the two-launch sequence.  Replaces (reference): bn1 + relu + conv2 of torchvision's Bottleneck.forward
 * (src/models/resnet_model.py:13-58). */
int simhand_conv2d_fwd_bnin_ok(const sh_conv_desc* d);
int simhand_conv2d_fwd_bnin(const sh_conv_desc* d, const void* y_in, const float* in_scale, const float* in_shift, const void* w, void* a_out,
                            void* y, float* bn_partial, sh_stream_t stream);
/* Direct 7x7 / stride 2 / pad 3 / 3 -> 64 stem (torchvision ResNet conv1, src/models/resnet_model.py:13-26) without an
 * im2col matrix.  simhand_stem_pad_input repacks the NCHW fp32 image batch to zero-padded NHWC4
 * xp [n][hp][wp][4] (dtype elements; geometry from simhand_stem_geometry: hp = h + 8, wp = roundup8(w + 8),
 * input pixel (ih, iw) at (ih + 3, iw + 3)); simhand_stem_pack_weights lays the OIHW fp32 [64][3][7][7] filter out as
 * [64][256] (column r*32 + tap*4 + c).  fwd: y [n*ho*wo][64] (+ BN partials [ceil(n*ho*wo/128)][2][64] if non-NULL);
 * wgrad: dw_oihw fp32 [64][3][7][7] from xp and dy [n*ho*wo][64], deterministic split-K through `workspace`. */
int simhand_stem_geometry(int h, int w, int* hp, int* wp, int* ho, int* wo);
int simhand_stem_pad_input(const float* x_nchw, void* xp, int n, int h, int w, int dtype, sh_stream_t stream);
int simhand_stem_pack_weights(const float* w_oihw, void* wp, int dtype, sh_stream_t stream);
/* rows of the bn_partial buffer simhand_stem_conv_fwd fills: [blocks][2][64] */
int simhand_stem_conv_fwd_stat_blocks(int n, int h, int w, int dtype);
int simhand_stem_conv_fwd(const void* xp, const void* wp, void* y, float* bn_partial, int n, int h, int w, int dtype, sh_stream_t stream);
size_t simhand_stem_conv_wgrad_workspace_bytes(int n, int h, int w, int dtype);
int simhand_stem_conv_wgrad(const void* xp, const void* dy, float* dw_oihw, void* workspace, size_t workspace_bytes, int n, int h, int w, int dtype, sh_stream_t stream);
/* dx = conv_transpose(dy, w).  wt = weights permuted to [Cin][R][S][Cout] (simhand_oihw_f32_to_crsk).
 * accumulate != 0: dx += result (residual branch merge). */
int simhand_conv2d_dgrad(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate, sh_stream_t stream);
/* identity-block merge without materialising the ReLU-masked residual gradient:
 * dx = conv_transpose(dy, w) + res_grad * [bit of res_mask]  (res_grad = gradient of the block output,
 * res_mask = the block output's ReLU bit mask from simhand_bn_apply; both laid out like dx). */
int simhand_conv2d_dgrad_masked_residual(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, const void* res_grad,
                                         const uint8_t* res_mask, sh_stream_t stream);
/* bf16 forward with the whole BatchNorm + residual + ReLU tail in the epilogue:
 *   out = act(conv(x) * scale[c] + shift[c] (+ residual)),  relu_mask (optional) = its 1-bit ReLU mask [pixel][cout/8]
 * for statistics that are known BEFORE the convolution runs: eval mode, or train mode for a 1x1 convolution whose
 * batch statistics follow from the Gram matrix of its input (mean_c = W_c . sum(x) / M, E[y^2]_c = W_c^T (x^T x) W_c / M).
 * The raw convolution output is never stored.  Replaces (reference): conv3 -> bn3 -> (+identity) -> relu and the
 * downsample conv -> bn of torchvision's Bottleneck (src/models/resnet_model.py:13-58). */
int simhand_conv2d_fwd_bnact(const sh_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                             const void* residual, int relu, void* out, uint8_t* relu_mask, sh_stream_t stream);

/* Data gradient with the BatchNorm-backward partial sums of the PREVIOUS conv+BN unit fused into the epilogue.
 * The dx this call stores is that unit's incoming gradient da (its raw conv output y has dx's layout), so instead
 * of a separate simhand_bn_bwd_partial pass (reads da and y) the epilogue reads y once and emits per tile
 *   partial[blk][0][c] = sum g,  partial[blk][1][c] = sum g * y,   g = da * relu'(.)
 * relu_mode 0: no ReLU; 2: mask recomputed as y*scale + shift > 0; 3: 1-bit mask written by simhand_bn_apply;
 * 4: the STORED dx is the masked gradient g = dx * bit(mask) (y unused; partial[blk][1] = 0) -- the form the folded
 *    BatchNorm backward of a 1x1 conv + BN unit consumes (simhand host: ResNetEngine._unit3_bwd_folded).
 * blk runs over simhand_conv2d_dgrad_stat_blocks(d, accumulate, relu_mode, c2) rows (the launch the same arguments
 * select; c2 = channels of sh_dgrad_opts.x2, 0 without); finish with simhand_bn_bwd_finalize_raw.
 * accumulate: 0 store, 1 dx += result, 2 dx = result + res_grad * bit(res_mask) (as the two entry points above).
 * Replaces (reference): autograd's native_batch_norm_backward reduction after each Conv2d input-gradient in
 * torchvision's Bottleneck / BasicBlock (src/models/resnet_model.py:13-58). */
typedef struct sh_bn_bwd_fuse {
  const void* y;
  const float* scale;
  const float* shift;
  const uint8_t* mask;
  int32_t relu_mode;
  float* partial;
} sh_bn_bwd_fuse;
int simhand_conv2d_dgrad_stat_blocks(const sh_conv_desc* d, int accumulate, int relu_mode, int c2);
/* 1 if the fused form is the faster choice for this layer (callers keep the standalone pass otherwise);
 * simhand_test_conv2d_dgrad_fuse_1x1(1) forces it for the short-K 1x1 layers too (tuning hook) */
int simhand_conv2d_dgrad_fuse_pays(const sh_conv_desc* d);
int simhand_conv2d_dgrad_fused(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate, const void* res_grad,
                               const uint8_t* res_mask, const sh_bn_bwd_fuse* fuse, sh_stream_t stream);
/* General form: accumulate modes as above, optional fusion, optional fp32 per-channel bias (length cin) added to the
 * result before accumulation / rounding, and an optional SECOND reduction segment for 1x1 / stride-1 layers:
 *   dx = dy * wt^T + x2 * wt2^T   (x2 [n][h][w][c2], wt2 [cin][c2], c2 a multiple of 64)
 * accumulated in fp32 in one pass -- the two terms of the folded BatchNorm backward's input gradient
 * (g (diag(A) W) - a (W^T diag(B) W)) without a second launch re-reading and re-writing dx.  Only where
 * simhand_conv2d_dgrad_concat_ok(d, c2) says so (bf16 1x1 / stride 1, c2 a multiple of 64); x2 = NULL: off. */
/* dy_src: the `dy` operand is DERIVED on load instead of read -- dy = coef_a * (da [y*scale + shift > 0]) - coef_b * y + coef_c (the
 * BatchNorm-backward apply of the unit this convolution belongs to, its sums already reduced: simhand_bn_bwd_coefs) -- and
 * written once to dy_out for the weight gradient that follows: the stand-alone simhand_bn_bwd_apply pass over the unit is gone
 * (2 reads + 1 write of a dy-sized tensor become 1 extra read + 1 write inside this launch).  The `dy` argument of
 * simhand_conv2d_dgrad_ex is then ignored.  Only where simhand_conv2d_dgrad_dysrc_ok(d) says so (the bf16 1x1 / stride-1 layers
 * the activation-stationary kernel takes, single reduction segment).  relu = 0: no gate.
 * Replaces (reference): native_batch_norm_backward in front of conv1's input gradient in torchvision's Bottleneck
 * (src/models/resnet_model.py:13-58). */
typedef struct sh_dy_src {
  const void* da;
  const void* y;
  const float* scale;
  const float* shift;
  const float* coef_a;
  const float* coef_b;
  const float* coef_c;
  int32_t relu;
  void* dy_out;
} sh_dy_src;
typedef struct sh_dgrad_opts {
  int32_t accumulate;
  const void* res_grad;
  const uint8_t* res_mask;
  const float* bias;
  const sh_bn_bwd_fuse* fuse;
  const void* x2;
  const void* wt2;
  int32_t c2;
  const sh_dy_src* dy_src;
  /* fp8 data gradient (BASELINE configs[4]; only where simhand_conv2d_dgrad_fp8_pays(d)): the reduction runs over e4m3 operands on the
   * 256 x 256 kernel's scaled-MFMA variant -- dy_q [pixels][cout] e4m3 codes of dy (scale state dy_state), wt_q CRSK e4m3 weights
   * (state w_state); result = acc / (scale_dy scale_w) before every epilogue option above (bias, accumulate, merges, fused sums).
   * `dy` / `wt` may still be passed (ignored).  No second reduction segment, no dy_src. */
  const void* dy_q;
  const void* wt_q;
  const float* dy_state;
  const float* w_state;
  /* sub_grad [n][h/2][w/2][cin] (h, w even; NULL = off): dx = gate(result + S), S = sub_grad at the EVEN pixels of dx and zero elsewhere --
   * the data gradient of a stride-2 1x1 shortcut, computed densely at its output resolution, merged into the main branch's data
   * gradient (the gate = the masked store of fuse->relu_mode 4 when given).  Merged inside the launch where the kernel the arguments
   * select has that epilogue (the masked-store forms of the activation-stationary 1x1 kernel and of the 256 x 256 kernel); otherwise
   * the call finishes with simhand_scatter2_add -- same result, one more pass over the even pixels.  Needs accumulate == 0.
   * Replaces (reference): autograd's accumulation of the two input-gradient branches of a stage-entry Bottleneck
   * (src/models/resnet_model.py:13-58). */
  const void* sub_grad;
} sh_dgrad_opts;
int simhand_conv2d_dgrad_fp8_pays(const sh_conv_desc* d);
/* e4m3 WEIGHT gradient (BASELINE configs[4]; round 5): dW [cout][cin][3][3] fp32 (the reference's nn.Conv2d.weight.grad layout) = sum over
 * pixels of dy^T x with BOTH operands as e4m3 codes -- x_q [n][h][w][cin] (the codes the BatchNorm-apply in front of the fp8 forward
 * emitted, state x_state) and dy_q [n][h][w][cout] (the codes the BatchNorm-backward apply emitted for the fp8 data gradient, state
 * dy_state) -- on v_mfma_scale_f32_16x16x128_f8f6f4 with the reduction running over 128 pixels per instruction; result = acc / (scale_x
 * scale_dy).  Only where simhand_conv2d_wgrad_fp8_pays(d): 3x3 / stride 1 / pad 1, >= 256 channels on both sides, w + 2 <= 128.
 * Replaces (reference): the weight gradient of conv2 of torchvision's Bottleneck (src/models/resnet_model.py:13-58) -- the reference
 * itself has no fp8 path (src/experiments/main.py:158-159 is fp16 AMP): parity n/a, checked against the fp32 gradient of the dequantised operands. */
int simhand_conv2d_wgrad_fp8_pays(const sh_conv_desc* d);
size_t simhand_conv2d_wgrad_fp8_workspace_bytes(const sh_conv_desc* d);
int simhand_conv2d_wgrad_fp8(const sh_conv_desc* d, const void* x_q, const void* dy_q, const float* x_state, const float* dy_state, float* dw_oihw,
                             void* workspace, size_t workspace_bytes, sh_stream_t stream);
int simhand_conv2d_dgrad_concat_ok(const sh_conv_desc* d, int c2);
int simhand_conv2d_dgrad_dysrc_ok(const sh_conv_desc* d);
int simhand_conv2d_dgrad_ex(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, const sh_dgrad_opts* opts, sh_stream_t stream);

/* simhand_conv2d_fwd_bnact with the NEXT 1x1 convolution chained on (a Bottleneck's conv3 -> bn3 -> + identity -> ReLU followed by
 * the next block's conv1): chain_y [pixels][cin] = out * chain_w^T (chain_w: KRSC [cin][cout] bf16 of that conv1, which maps
 * d->cout channels back to d->cin) is computed from the bf16 output chunks while they are still in registers -- the block output is
 * written (the residual path and the backward pass need it) but not read back -- together with its BatchNorm partial sums
 * chain_partial [simhand_conv2d_fwd_chain_stat_blocks(d)][2][cin].  Results are bit-identical to simhand_conv2d_fwd_bnact followed by
 * simhand_conv2d_fwd.  residual, relu_mask are required (ReLU is implied).  Only where simhand_conv2d_fwd_chain_ok(d) says so.
 * Replaces (reference): the bn3 / add / relu tail of one torchvision Bottleneck and conv1 of the next (src/models/resnet_model.py:13-58). */
int simhand_conv2d_fwd_chain_ok(const sh_conv_desc* d);
int simhand_conv2d_fwd_chain_stat_blocks(const sh_conv_desc* d);
int simhand_conv2d_fwd_bnact_chain(const sh_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                                   const void* residual, void* out, uint8_t* relu_mask, const void* chain_w, void* chain_y,
                                   float* chain_partial, sh_stream_t stream);

/* dw (fp32, KRSC) = sum over output pixels of dy (x) patches(x); deterministic two-stage split-K */
size_t simhand_conv2d_wgrad_workspace_bytes(const sh_conv_desc* d);
int simhand_conv2d_wgrad(const sh_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* bf16 1x1 / stride-1 only: the same gradient (KRSC fp32) plus, as a by-product of the dy tiles the kernel stages anyway,
 * the per-channel sums of dy: dy_colsum[split][0][cout] for split < simhand_conv2d_wgrad_splits(d) ([split][1][.] = 0, so the
 * buffer has the layout simhand_bn_bwd_finalize_raw folds).  Used by the folded BatchNorm backward of conv3 + bn3. */
int simhand_conv2d_wgrad_splits(const sh_conv_desc* d);
int simhand_conv2d_wgrad_colsum(const sh_conv_desc* d, const void* x, const void* dy, float* dw, float* dy_colsum, void* workspace,
                                size_t workspace_bytes, sh_stream_t stream);
/* Round 6: the same with the channel sums FOLDED -- dy_sum[cout] = sum over splits of dy_colsum[split][0][.], added in the order
 * simhand_bn_bwd_finalize_raw uses (bit-identical) -- by the launch that reduces the split-K partials: two launches instead of three
 * (dy_colsum stays the scratch buffer it was; needs simhand_conv2d_wgrad_splits(d) < 4096). */
int simhand_conv2d_wgrad_colsum_sums(const sh_conv_desc* d, const void* x, const void* dy, float* dw, float* dy_colsum, float* dy_sum,
                                     void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* BatchNorm passes fused into the operand loaders of the bf16 1x1 / stride-1 weight-gradient kernel (its tiles are staged
 * global -> registers -> LDS, so a per-channel transform between load and store costs VALU only):
 *  simhand_bn_apply_gram: y = raw conv output [m][c] of a conv + BN (+ReLU) unit whose activation feeds a FOLDED 1x1
 *    convolution (simhand_conv2d_fwd_bnact).  One launch writes a = act(y*scale + shift), s2 = a^T a [c][c] fp32 and
 *    colsum_partial [simhand_conv2d_wgrad_splits(d)][2][c] (row 0 = sum a) -- the stand-alone simhand_bn_apply pass over y
 *    and the separate Gram launch's read of a are gone.  d = the c -> c 1x1 descriptor over the m pixels.
 *  simhand_conv2d_wgrad_bnbwd: weight gradient of a conv whose output y feeds a BN (+ReLU): the dy operand is computed on
 *    the fly as dy = coef_a * (da * [y*scale + shift > 0]) - coef_b * y + coef_c  (BatchNorm backward with its sums already
 *    reduced: coef_a = gamma invstd, coef_b = coef_a invstd dgamma / M, coef_c = -coef_a dbeta / M + mean coef_b; see
 *    simhand_bn_bwd_coefs) and written to dy_out for the data gradient that follows -- the stand-alone
 *    simhand_bn_bwd_apply pass is gone.  dw_oihw as simhand_conv2d_wgrad_oihw (c_real = 0: KRSC fp32).
 * Replaces (reference): native_batch_norm / native_batch_norm_backward around conv2 / conv1 of torchvision's Bottleneck
 * (src/models/resnet_model.py:13-58). */
int simhand_bn_apply_gram(const sh_conv_desc* d, const void* y, const float* scale, const float* shift, int relu, void* a, float* s2,
                          float* colsum_partial, void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* Round 6: + colsum[c] = sum a, folded inside the split-K reduction launch (as simhand_conv2d_wgrad_colsum_sums) */
int simhand_bn_apply_gram_sums(const sh_conv_desc* d, const void* y, const float* scale, const float* shift, int relu, void* a, float* s2,
                               float* colsum_partial, float* colsum, void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* EXPERIMENTAL (off in the engine: measured -4.3 ms BatchNorm / +8.5 ms weight gradient per step) */
int simhand_conv2d_wgrad_bnbwd(const sh_conv_desc* d, const void* x, const void* da, const void* y, const float* scale, const float* shift,
                               const float* coef_a, const float* coef_b, const float* coef_c, int relu, void* dy_out, float* dw_oihw,
                               int c_real, void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* coefficient vectors of the line above from the finished BatchNorm-backward sums */
int simhand_bn_bwd_coefs(const float* mean, const float* invstd, const float* gamma, const float* dgamma, const float* dbeta, int64_t m,
                         int c, float* coef_a, float* coef_b, float* coef_c, sh_stream_t stream);
/* Same, but the split-K reduction writes the gradient directly in the reference's nn.Conv2d.weight.grad layout
 * OIHW fp32 [cout][c_real][r][s]; c_real <= cin drops zero-padded input channels (the im2col'd stem: cin = 192
 * columns, c_real = 147 = 3*7*7 -> exactly weight.grad.view(64, 147)). */
int simhand_conv2d_wgrad_oihw(const sh_conv_desc* d, const void* x, const void* dy, float* dw_oihw, int c_real, void* workspace, size_t workspace_bytes, sh_stream_t stream);

/* 224-row tiles of the 256 x 256 kernel (7 x 32 rows: 401 408 pixels = exactly 7 rounds of 256 CUs instead of 6.125): 0 off, 1 auto
 * (default: when they fill whole rounds and the 256-row plan does not), 2 forced whenever the pixel count is a multiple of 224 */

/* layout / dtype transforms.  k_pad = padded length of one flattened KRSC weight row
 * (>= r*s*c; rows are zero padded) -- r*s*c for ordinary convs, 192 for the im2col'd stem. */
int simhand_nchw_f32_to_nhwc(const float* src, void* dst, int n, int c, int h, int w, int c_pad, int dtype, sh_stream_t stream);
int simhand_oihw_f32_to_krsc(const float* src, void* dst, int k, int c, int r, int s, int k_pad, int dtype, sh_stream_t stream);
int simhand_oihw_f32_to_crsk(const float* src, void* dst, int k, int c, int r, int s, int dtype, sh_stream_t stream);
int simhand_krsc_f32_to_oihw(const float* src, float* dst, int k, int c, int r, int s, int k_pad, sh_stream_t stream);
int simhand_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t count, sh_stream_t stream);
/* stem lowering: NCHW fp32 images -> [n*ho*wo][k_pad] patch matrix, column (c*R + r)*S + s (the OIHW
 * flattening of the filter, so the weight matrix is W.view(cout, cin*R*S) row-padded to k_pad), zero padded.
 * The 7x7/2 stem then runs as a 1x1 conv with cin = k_pad through conv2d_fwd / conv2d_wgrad. */
int simhand_im2col_nchw_f32(const float* x, void* col, int n, int cin, int h, int w, int r, int s, int stride, int pad,
                            int k_pad, int dtype, sh_stream_t stream);

/* train-mode BatchNorm over [M][C] (M = N*H*W), eps 1e-5, momentum 0.1 (torch defaults).
 * Level-1 partial sums [nblk][2][C] fp32 come either from the conv epilogue (nblk =
 * simhand_conv2d_fwd_stat_blocks) or from simhand_bn_partial_stats (nblk = simhand_bn_stat_blocks). */
int simhand_bn_stat_blocks(int64_t m, int c);
int simhand_bn_partial_stats(const void* y, int64_t m, int c, int dtype, float* partial, sh_stream_t stream);
/* partials -> mean / invstd, scale = gamma*invstd, shift = beta - mean*scale; updates running_mean /
 * running_var (unbiased, momentum) and num_batches_tracked like torch.  pre_bias (or NULL): bias of a
 * Linear feeding this BN -- it only shifts the batch mean that goes into running_mean. */
size_t simhand_bn_finalize_workspace_bytes(int nblk, int c);
int simhand_bn_finalize(const float* partial, int nblk, int64_t m, int c, const float* gamma, const float* beta,
                        const float* pre_bias, float eps, float momentum, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                        void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* Round 6: both levels of the fold in ONE launch -- the block that draws the last ticket finalizes (bit-identical statistics).  ticket: a
 * caller-owned uint32 on the device, ZERO before the first call, not shared by launches that may run concurrently (one per stream); every
 * call leaves it zero (stream-ordered). */
int simhand_bn_finalize_ticket(const float* partial, int nblk, int64_t m, int c, const float* gamma, const float* beta,
                               const float* pre_bias, float eps, float momentum, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                               void* workspace, size_t workspace_bytes, uint32_t* ticket, sh_stream_t stream);
/* eval mode (model.eval(), validation_step): scale/shift from the running statistics */
int simhand_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                            int c, float* scale, float* shift, sh_stream_t stream);
/* a = act(y*scale + shift (+ residual)), act = relu if relu != 0.  relu_mask (or NULL): [M][C/VE] bytes, bit e of
 * byte (row, cv) = element (row, cv*VE + e) passed the ReLU (VE = 8 for bf16, 4 for fp32) -- lets the backward of
 * units with a residual add read 1 bit instead of the stored activation. */
int simhand_bn_apply(const void* y, const float* scale, const float* shift, const void* residual, int relu,
                     void* a, uint8_t* relu_mask, int64_t m, int c, int dtype, sh_stream_t stream);
/* backward: g = da * [relu input > 0]; partial [simhand_bn_stat_blocks][2][C] sums of g and g*xhat;
 * finalize -> dbeta, dgamma; apply: dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)), dres (optional) = g.
 * relu: 0 = none, 1 = mask from the stored activation a (units with a residual add),
 *       2 = mask recomputed from y*scale+shift (no residual; a is not read at all),
 *       3 = `a` points to the bit mask written by simhand_bn_apply (residual units). */
int simhand_bn_bwd_partial(const void* da, const void* a, const void* y, const float* mean, const float* invstd,
                           const float* scale, const float* shift, int relu, int64_t m, int c, int dtype, float* partial,
                           sh_stream_t stream);
int simhand_bn_bwd_finalize(const float* partial, int nblk, int c, float* dgamma, float* dbeta, sh_stream_t stream);
/* Folded BatchNorm of a 1x1 convolution y = a W^T: the parameter-sized algebra.  w = fp32 master weight
 * [cc][cw] (rounded to bf16 first when round_bf16), s2 = a^T a [cw][cw], t2 = sum a [cw] (simhand_conv2d_wgrad_colsum with
 * x = dy = a), m = pixels.
 * fwd: batch statistics of y without y: mean / invstd / scale / shift (+ running statistics), ws2 = W s2 [cc][cw] (kept for
 *      the backward).
 * bwd: gmat = g^T a [cc][cw] (simhand_conv2d_wgrad_colsum), s = sum g [cc]  ->  dgamma, dbeta, dw [cc][cw] (= OIHW),
 *      wa = diag(A) W as the CRSK operand [cw][cc] of the first data-gradient term, wm = -(W^T diag(B) W) [cw][cw] the
 *      operand of the second (accumulating) one, bias = C W [cw];  bw [cc][cw], ccoef [cc]: scratch.
 * Replaces (reference): native_batch_norm(_backward) around conv3 / downsample of torchvision's Bottleneck. */
size_t simhand_bn_fold_workspace_bytes(int cc, int cw); /* split-K scratch of the two small GEMMs (either call) */
int simhand_bn_fold_fwd(const float* w, int round_bf16, const float* s2, const float* t2, int cc, int cw, int64_t m, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift, float* ws2,
                        void* workspace, size_t workspace_bytes, sh_stream_t stream);
int simhand_bn_fold_bwd(const float* w, int round_bf16, const float* gmat, const float* s, const float* ws2, const float* t2,
                        const float* mean, const float* invstd, const float* gamma, int cc, int cw, int64_t m, float* dgamma,
                        float* dbeta, float* dw, void* wa, float* bw, float* ccoef, void* wm, float* bias, int dtype,
                        void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* out = x gated by the ReLU bit mask simhand_bn_apply / simhand_conv2d_fwd_bnact wrote ([m][c/VE] bytes) */
int simhand_apply_relu_bitmask(const void* x, const uint8_t* mask, void* out, int64_t m, int c, int dtype, sh_stream_t stream);
/* Stem block fused: out, idx = MaxPool(3, 2, 1)(ReLU(y*scale + shift)) without storing the activation in between, and
 * its backward: the BatchNorm-backward passes gather the pooled gradient dz through idx (partial: simhand_bn_stat_blocks
 * (n*h*w, c) blocks of [2][c] sums of g and g*xhat -> simhand_bn_bwd_finalize; apply: dy).  y is [n][h][w][c].
 * ywin (optional, laid out like out): the RAW conv output y of each window's winning tap.  Every pooled gradient reaches
 * exactly one input pixel -- its winner -- so the BatchNorm-backward sums over the h x w grid equal
 * simhand_bn_bwd_partial(dz, NULL, ywin, ..., relu_mode 2, m = n*ho*wo): 2 passes over pooled-size tensors instead of a
 * gather pass over the 4x larger y (the only difference: gradients of windows sharing a winner are summed unrounded).
 * Replaces (reference): bn1 -> relu -> maxpool of the torchvision ResNet stem (src/models/resnet_model.py:13-26). */
int simhand_bn_relu_maxpool_fwd(const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, void* ywin, int n, int h,
                                int w, int c, int dtype, sh_stream_t stream);
int simhand_maxpool_bn_bwd_partial(const void* dz, const uint8_t* idx, const void* y, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, int n, int h, int w, int c, int dtype, float* partial,
                                   sh_stream_t stream);
int simhand_maxpool_bn_bwd_apply(const void* dz, const uint8_t* idx, const void* y, const float* mean, const float* invstd,
                                 const float* gamma, const float* dgamma, const float* dbeta, const float* scale, const float* shift,
                                 void* dy, int n, int h, int w, int c, int dtype, sh_stream_t stream);
/* same for the RAW partial sums of simhand_conv2d_dgrad_fused (sum g, sum g*y):
 * dbeta = sum g, dgamma = invstd * (sum g*y - mean * sum g), folded in fp64 */
size_t simhand_bn_bwd_finalize_raw_workspace_bytes(int nblk, int c);
int simhand_bn_bwd_finalize_raw(const float* partial, int nblk, int c, const float* mean, const float* invstd, float* dgamma,
                                float* dbeta, void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* Round 6: the same + the coefficients of dy = A g - B y + C (simhand_bn_bwd_coefs: coefs [3][c] = A, B, C) out of the same launch */
int simhand_bn_bwd_finalize_raw_coefs(const float* partial, int nblk, int c, const float* mean, const float* invstd, const float* gamma,
                                      int64_t m, float* dgamma, float* dbeta, float* coefs, void* workspace, size_t workspace_bytes,
                                      sh_stream_t stream);
int simhand_bn_bwd_apply(const void* da, const void* a, const void* y, const float* mean, const float* invstd,
                         const float* gamma, const float* dgamma, const float* dbeta, const float* scale, const float* shift,
                         int relu, void* dy, void* dres, int64_t m, int c, int dtype, sh_stream_t stream);

/* pooling (NHWC): MaxPool2d(3, stride 2, pad 1) (idx = winning tap 0..8 per output element, uint8)
 * and AdaptiveAvgPool2d(1) */
int simhand_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, int dtype, sh_stream_t stream);
int simhand_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int n, int h, int w, int c, int dtype, sh_stream_t stream);
/* y[n][ceil(h/2)][ceil(w/2)][c] = x[n][2i][2j][c]: the pixels a stride-2 1x1 convolution reads (dense operand for the folded
 * BatchNorm backward of the strided shortcut) */
int simhand_subsample2(const void* x, void* y, int n, int h, int w, int c, int dtype, sh_stream_t stream);
/* its transpose with accumulation: dx[n][2i][2j][:] = gate(dx[n][2i][2j][:] + src[n][i][j][:]), dx is [n][h][w][c];
 * gate = optional ReLU bit mask over dx's pixels ([n*h*w][c/VE]) */
int simhand_scatter2_add(const void* src, void* dx, const uint8_t* mask, int n, int h, int w, int c, int dtype, sh_stream_t stream);
int simhand_avgpool_fwd(const void* x, void* y, int n, int hw, int c, int dtype, sh_stream_t stream);
int simhand_avgpool_bwd(const void* dy, void* dx, int n, int hw, int c, int dtype, sh_stream_t stream);
/* the same with the gradient gated by the ReLU bit mask [n * hw][c / VE] of the pooled tensor (NULL = no gate): avgpool_bwd followed by
 * simhand_apply_relu_bitmask in one pass, bit for bit */
int simhand_avgpool_bwd_masked(const void* dy, const uint8_t* mask, void* dx, int n, int hw, int c, int dtype, sh_stream_t stream);

/* column sums of [M][C] (Linear bias gradient); partial: (2*simhand_bn_stat_blocks(m,c) + 1) * c floats */
int simhand_colsum(const void* x, int64_t m, int c, int dtype, float* partial, float* out, sh_stream_t stream);

/* ===========================================================================
 * FP8 first slice (BASELINE configs[4], "next" row 8f-4 mixed-precision policy).  The reference has no fp8 path
 * (fp16 autocast + GradScaler, src/experiments/main.py:158-159): parity n/a, the gate is agreement with the bf16 path.
 *   per-tensor scaling: q = e4m3(clamp(v * scale, +-448)).  A scale lives in a device `state` vector of
 *   simhand_fp8_state_floats(history) floats: [0] scale, [1] 1/scale, [2] ring position, [3] updates, [4..] amax ring.
 *   simhand_fp8_amax folds max|x| into *amax_bits (uint bits of a non-negative float, atomicMax: exact and deterministic);
 *   simhand_fp8_scale_update(state, amax_bits, history, margin, mode) consumes *amax_bits.  state[0] is always the scale the NEXT
 *   quantisation uses; state[1] is 1 / (the scale the codes that exist NOW were made with) -- what a convolution launched behind the
 *   update de-scales with:
 *     mode 0 (current scaling; call it BEFORE the quantisation it serves): scale = 448 / (amax * margin_pow2), state[1] = 1 / scale;
 *     mode 1 (delayed scaling; call it BEHIND the quantisation whose amax it consumes): amax enters the ring, state[0] = 448 /
 *             (max(ring) * margin_pow2) for the next call, state[1] = 1 / (the scale that quantisation just used) -- NOT 1 / state[0];
 *     mode 2 (a delayed site's FIRST update, before its first quantisation): the ring's first entry, state[1] = 1 / state[0].
 *   (Until round 5 the argument was a 0 / 1 flag `delayed` and state[1] was always 1 / state[0]: a caller that ran the delayed update
 *   behind its quantisation de-scaled the following convolutions with the NEXT scale.  Values outside 0..2 are treated as 1.)
 *   A consumer that runs LATER than the next quantisation of the same site (a weight gradient in the backward) must keep its own copy
 *   of state[0:2] taken right behind the quantisation (the engine does: host/resnet_model.py _conv_fwd_fp8).
 *   simhand_fp8_quantize converts with state[0] and (optionally) records the tensor's own amax for that next update.
 *   simhand_conv2d_fwd_fp8: y (bf16) = conv(q_x, q_w) / (scale_x scale_w) on v_mfma_scale_f32_16x16x128_f8f6f4 (all block
 *   scales 2^0), + the fused BatchNorm partial sums [ceil(n*ho*wo/128)][2][cout] of simhand_conv2d_fwd.  Needs cin, cout
 *   multiples of 128 (simhand_conv2d_fwd_fp8_supported).  x_q [n][h][w][cin], w_q KRSC, both e4m3.
 * =========================================================================== */
int simhand_fp8_state_floats(int history);
int simhand_fp8_amax(const void* x, int64_t count, int dtype, uint32_t* amax_bits, sh_stream_t stream);
int simhand_fp8_scale_update(float* state, uint32_t* amax_new_bits, int history, float margin_pow2, int mode, sh_stream_t stream);
int simhand_fp8_quantize(const void* x, void* q, int64_t count, int dtype, const float* state, uint32_t* amax_bits, sh_stream_t stream);
int simhand_fp8_pack_krsc(const float* w_oihw, void* q, int k, int c, int r, int s, const float* state, sh_stream_t stream);
int simhand_conv2d_fwd_fp8_supported(const sh_conv_desc* d);
/* rows of the bn_partial buffer simhand_conv2d_fwd_fp8 fills ([rows][2][cout]): the layers with >= 256 output channels run on the e4m3
 * variant of the 256 x 256 LDS-DMA kernel (one scaled K = 128 MFMA per tile pair and k-step), the others on the 128-row kernel */
int simhand_conv2d_fwd_fp8_stat_blocks(const sh_conv_desc* d);
/* 1 where the engine routes a layer's forward through fp8 by default: where it measured faster than the bf16 kernel it replaces */
int simhand_conv2d_fwd_fp8_pays(const sh_conv_desc* d);
/* BatchNorm-apply (+ReLU) of the unit in front of an fp8 convolution, bf16: a = act(y*scale + shift) AND q = e4m3(clamp(a * q_state[0]))
 * in one pass, max|a| folded into amax_bits (for simhand_fp8_scale_update) -- replaces simhand_bn_apply + simhand_fp8_quantize. */
/* simhand_bn_bwd_apply (modes as there, bf16) that ALSO emits q = e4m3(clamp(dy * q_state[0])) and folds max|dy| into amax_bits: the
 * operand of the fp8 data gradient leaves the BatchNorm-backward pass that produces dy. */
int simhand_bn_bwd_apply_fp8(const void* da, const void* a, const void* y, const float* mean, const float* invstd, const float* gamma,
                             const float* dgamma, const float* dbeta, const float* scale, const float* shift, int relu, void* dy, void* q,
                             const float* q_state, uint32_t* amax_bits, int64_t m, int c, sh_stream_t stream);
int simhand_bn_apply_fp8(const void* y, const float* scale, const float* shift, int relu, void* a, void* q, const float* q_state,
                         uint32_t* amax_bits, int64_t m, int c, sh_stream_t stream);
int simhand_conv2d_fwd_fp8(const sh_conv_desc* d, const void* x_q, const void* w_q, const float* x_state, const float* w_state, void* y,
                           float* bn_partial, sh_stream_t stream);

/* ===========================================================================
 * GPU batch producer ("next" row 8f-2): the per-sample augmentation chain of the contrastive recipes for a whole batch.
 * Replaces (reference): SampleAugmenter.transform_sample with rotate / crop (+random_crop) / resize / color_jitter
 * (src/data_loader/sample_augmenter.py:50-136, :173-318, :424-474), ToTensor + Normalize (src/data_loader/utils.py:279-285)
 * and the angle / jitter bookkeeping of Data_Set.get_random_augment_param (src/data_loader/data_set.py:804-838).
 *   images [n][h][w][3] uint8; joints [n][21][3] fp32 (x, y, depth) in image pixels;
 *   draws (made by the caller): angle [n] degrees or NULL (rotate off), crop_margin [n], jitter [n][2] int32 (the
 *   override / drawn crop-box jitter), hsab [n][4] = hue, saturation, alpha, beta factors or NULL (color_jitter off);
 *   out_images [n][3][out_h][out_w] fp32 normalised; joints_aug [n][21][3]; rec [n][6] int32 = jitter_x, jitter_y (the
 *   batch entries of App. B), origin_x, origin_y, crop width, crop height.
 * OpenCV's own rounding is un-vendored: resampling / colour arithmetic follow the published definitions (oracle/augment.py
 * says what is pinned). */
size_t simhand_augment_workspace_bytes(int n);
int simhand_augment_batch(const uint8_t* images, const float* joints, const float* angle, const float* crop_margin, const int32_t* jitter,
                          const float* hsab, int n, int h, int w, int out_w, int out_h, float* out_images, float* joints_aug, int32_t* rec,
                          void* workspace, size_t workspace_bytes, sh_stream_t stream);
/* The same chain with the augmenter's COIN-FLIP operations (sample_augmenter.py:138-171, :254-272, :302-388): the flips and
 * their draws are inputs.  flags [n]: bit 0 sobel_filter (first, on the raw frame: gray -> Sobel dx + dy, 3x3), bit 1 cut_out
 * (rows [b0,b1) x columns [b2,b3) of the raw frame = cut_fill; box from get_random_cut_out_box -- computed by the caller),
 * bit 2 gaussian_blur (kernel blur_kx x blur_ky [both odd: odd(0.1 * rows), odd(0.1 * cols)], sigma per sample), bit 3
 * gaussian_noise (after the colour jitter: image += saturate_u8(rint(noise_std * z)), uint8 wrap-around; z = standard-normal
 * draws [n][out_h][out_w][3]), bit 4 color_drop (last: all channels = BGR2GRAY).  any_* = whether any sample has that bit set
 * (the pre-passes / scratch buffers of an operation nobody uses are skipped).  --flip exists on the reference's CLI but its
 * augmenter never implements it.  Workspace: simhand_augment_workspace_bytes_ex(n, h, w, any pre-pass op, any blur). */
typedef struct sh_augment_extra {
  const int32_t* flags;      /* [n] */
  const int32_t* cut_box;    /* [n][4] */
  const uint8_t* cut_fill;   /* [n] */
  const float* blur_sigma;   /* [n] */
  const float* noise;        /* [n][out_h][out_w][3] */
  float noise_std;
  int blur_kx, blur_ky;
  int any_sobel, any_cut_out, any_blur, any_noise;
} sh_augment_extra;
size_t simhand_augment_workspace_bytes_ex(int n, int h, int w, int pre_ops, int blur);
int simhand_augment_batch_ex(const uint8_t* images, const float* joints, const float* angle, const float* crop_margin, const int32_t* jitter,
                             const float* hsab, const sh_augment_extra* extra, int n, int h, int w, int out_w, int out_h, float* out_images,
                             float* joints_aug, int32_t* rec, void* workspace, size_t workspace_bytes, sh_stream_t stream);

/* ===========================================================================
 * Thin RCCL wrappers behind an opaque communicator handle -- the exchange steps of the path for a caller
 * that does not go through torch.distributed: all-gather of the projection / joint rows into the reference row order,
 * all-reduce of the distance statistics (MAX / MIN / SUM) and of the parameter gradients (SUM).
 * Replaces (reference): nn.DataParallel's scatter / gather / reduce-add under strategy="dp" (src/experiments/main.py:152-163).
 * librccl.so is resolved at run time (the copy the process already mapped, else dlopen); every entry point returns an
 * error if it cannot be loaded.  simhand_comm_init binds the communicator to the CURRENT HIP device; rank 0 creates the id
 * and hands its SH_COMM_ID_BYTES to the other ranks over any side channel.  `count` = elements PER RANK for all_gather.
 * =========================================================================== */
#define SH_COMM_ID_BYTES 128
enum sh_comm_dtype { SH_COMM_F32 = 0, SH_COMM_F64 = 1, SH_COMM_BF16 = 2, SH_COMM_I64 = 3 };
enum sh_comm_op { SH_COMM_SUM = 0, SH_COMM_MAX = 1, SH_COMM_MIN = 2 };
int simhand_comm_unique_id(uint8_t* id /*[SH_COMM_ID_BYTES]*/);
int simhand_comm_init(const uint8_t* id, int world, int rank, void** comm);
int simhand_comm_destroy(void* comm);
int simhand_comm_world(void* comm, int* world, int* rank);
int simhand_comm_all_gather(void* comm, const void* send, void* recv, int64_t count, int dtype, sh_stream_t stream);
int simhand_comm_all_reduce(void* comm, const void* send, void* recv, int64_t count, int dtype, int op, sh_stream_t stream);

/* ===========================================================================
 * Optimizer ("next" row 8f-1): LARSWrapper(Adam) step, pl_bolts 0.2.2 semantics
 * (call site src/models/base_model.py:59-106) -- PARITY UNPINNED, restated from
 * the published source.  One launch per parameter tensor group element.
 * =========================================================================== */
int simhand_sumsq_partial(const float* x, int64_t count, float* partial, int nblk, sh_stream_t stream);
int simhand_lars_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                           const float* p_sumsq, const float* g_sumsq, int nblk_norm,
                           float lr, float beta1, float beta2, float adam_eps, float weight_decay,
                           float lars_eta, float lars_eps, int lars_clip, int use_lars, int step, sh_stream_t stream);

/* Multi-tensor form: the whole parameter list in two launches.  `tensors` (n_tensors records) and `chunks`
 * (n_chunks pairs {tensor index, chunk index inside that tensor}, a chunk = simhand_opt_chunk_elems() consecutive
 * elements, tensors in table order so that a tensor's chunks are first_chunk .. first_chunk+n_chunks-1) are DEVICE
 * arrays; bc1 = 1-beta1^step and bc2_sqrt = sqrt(1-beta2^step) are computed by the caller per tensor.
 * norm_partials: 2*n_chunks floats of scratch. */
typedef struct sh_opt_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t count;
  int32_t first_chunk, n_chunks;
  float lr, weight_decay, bc1, bc2_sqrt;
  int32_t use_lars, reserved;
} sh_opt_tensor;
int simhand_opt_chunk_elems(void);

/* Every convolution weight of the net re-packed in TWO launches: item = fp32 OIHW master -> KRSC rows [k][r*s*c] (forward / weight
 * gradient order) and, when crsk is not NULL, CRSK [c][r*s*k] (data-gradient operand), both in `dtype`.  items, chunks and
 * crsk_tiles are DEVICE arrays; chunk j = (item index, chunk number within that item), simhand_pack_chunk_elems() KRSC elements
 * each; crsk_tiles j = (item index, tile number): the CRSK copies are 64 x 64 tile transposes of the KRSC copies, tile number =
 * (tap * (k / 64) + k_tile) * (c / 64) + c_tile (items with a CRSK copy need k % 64 == 0 and c % 64 == 0).
 * Replaces (reference): nothing -- torch keeps one weight layout; here the packed copies follow every optimizer step
 * (src/models/base_model.py:59-106 is where the reference's parameters change). */
typedef struct sh_pack_item {
  const float* src;
  void* krsc;
  void* crsk;
  int32_t k, c, r, s;
} sh_pack_item;
int simhand_pack_chunk_elems(void);
int simhand_pack_weights_multi(const sh_pack_item* items, const int32_t* chunks, int n_chunks, const int32_t* crsk_tiles, int n_tiles,
                               int dtype, sh_stream_t stream);
int simhand_lars_adam_multi(const sh_opt_tensor* tensors, int n_tensors, const int32_t* chunks, int n_chunks, float* norm_partials,
                            float beta1, float beta2, float adam_eps, float lars_eta, float lars_eps, int lars_clip,
                            int64_t total_elems, sh_stream_t stream);
/* The same update under dynamic loss scaling (the reference's precision = 16 policy, src/experiments/main.py:158-159): found_inf is
 * GradScaler's DEVICE flag (one float; non-zero = this step's unscaled gradients held an inf / nan) -- the update launch then returns
 * without touching parameters or moments, so the skipped step costs no host synchronisation.  found_inf = NULL: the plain update.
 * The caller's step counters (bc1 / bc2_sqrt) must not advance over a skipped step (host/optim.py reads the flag one step late). */
int simhand_lars_adam_multi_guarded(const sh_opt_tensor* tensors, int n_tensors, const int32_t* chunks, int n_chunks, float* norm_partials,
                                    float beta1, float beta2, float adam_eps, float lars_eta, float lars_eps, int lars_clip,
                                    int64_t total_elems, const float* found_inf, sh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMHAND_HIP_H */
