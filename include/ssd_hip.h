/*
 * ssd_hip.h -- C ABI of the MI355X (gfx950) SSD hot-path library  (libssd_hip.so)
 *
 * Drop-in boundary for the data-parallel hot path of AcherStyx/SSD-Object-Detection.
 * The reference is pure Python (TensorFlow + numpy) and has no FFI; each entry point below
 * replaces one Python-level seam, cited as <file>:<line> relative to the reference tree.
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add at each seam.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Unless a parameter says HOST, every pointer is a DEVICE pointer owned by the caller.
 *     The library allocates nothing, creates no streams, reads no environment variables and keeps no state that
 *     a result depends on: calls are re-entrant and thread-safe.  (Its only process-wide state: the idempotent
 *     once-per-device registration of kernels that need > 64 KB of LDS, and the development overrides of
 *     ssd_dev_knob at the end of this header, which the product never sets.)
 *     Scratch is passed as (ws, ws_bytes) sized by the matching *_workspace_bytes().
 *   - `stream` (last argument) is a hipStream_t passed as void*; all work is enqueued on it
 *     and nothing synchronises the host, so every call can be captured into a hipGraph.
 *   - Return value: SSD_OK (0) or a negative ssd_status.  Kernels never abort.  The Python
 *     wrapper maps SSD_ERR_ASSERT to AssertionError and SSD_ERR_VALUE to ValueError, the
 *     exception types the reference raises at the same seams.
 *   - Layouts: boxes are (cx, cy, w, h); activations NHWC; conv weights [Cout][kh][kw][Cin].
 *   - Alignment: every tensor pointer and every workspace 16-byte aligned (hipMalloc gives 256); the loc / conf outputs of
 *     ssd_conv2d_head_fwd 4-byte aligned.  Kernels use 16-byte vector accesses on them without checking.
 */
#ifndef SSD_HIP_H
#define SSD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SSD_OK = 0,
    SSD_ERR_ASSERT = -1,    /* a reference `assert` would have fired (utils/bbox.py:50-51,95; models/ssd_model.py:347-351,375) */
    SSD_ERR_VALUE = -2,     /* bad argument (reference: ValueError / TF InvalidArgument)                    */
    SSD_ERR_WORKSPACE = -3, /* ws_bytes smaller than *_workspace_bytes()                                    */
    SSD_ERR_LAUNCH = -4,    /* HIP launch failure (hipGetLastError != hipSuccess)                           */
    SSD_ERR_UNSUPPORTED = -5
} ssd_status;

typedef enum { SSD_F32 = 0, SSD_BF16 = 1 } ssd_dtype;

int ssd_hip_abi_version(void);
const char* ssd_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * Default boxes -- replaces SSDObjectDetectionModel._build_prior_box (models/ssd_model.py:173-194)
 *   grid_hw    HOST int[2*levels]   (h, w) per level            (reference: loc-head output shapes, :164)
 *   s_ref      HOST double[levels+1]                            (:176)
 *   ratios     HOST int[ratio_off[levels]], ratio_off HOST int[levels+1]   (:177, CSR form)
 *   in_size    input side in pixels (300)                       (:184)
 *   out        DEVICE double[A*4], A = ssd_priors_count(...)    (cx,cy,w,h), unclipped, reference order
 * Bit-exact against the reference's float64 array.
 * ---------------------------------------------------------------------------------------- */
int ssd_priors_count(const int* grid_hw, int levels, const int* ratio_off);
int ssd_priors(const int* grid_hw, int levels, const double* s_ref, const int* ratios,
               const int* ratio_off, double in_size, double* out, void* stream);

/* Geometry of a prior set made by ssd_priors (levels <= SSD_MAX_LEVELS). */
#define SSD_MAX_LEVELS 8
#define SSD_GRID_VERIFIED 0x5D5D0001
typedef struct {
    int levels;
    int grid_h[SSD_MAX_LEVELS];
    int grid_w[SSD_MAX_LEVELS];
    int per_cell[SSD_MAX_LEVELS]; /* priors per cell = 2 + 2*len(ratios[level]) */
    int verified;                 /* SSD_GRID_VERIFIED once ssd_prior_grid_verify has checked it against a prior array; else 0 */
} ssd_prior_grid;

/* Check on the device that `grid` describes `priors` exactly: every prior at the cell centre ssd_priors computes, bit for
 * bit, in the reference's order, with one (w, h) per level and anchor type.  Sets grid->verified (HOST) accordingly and
 * returns SSD_OK either way; synchronises `stream` once -- call it when the prior set is built, not per step.
 *   scratch DEVICE int32[1].
 * ssd_match_encode takes its single-launch path only with a verified grid; a grid that was not verified (or does not fit)
 * is a pruning hint only and can never change a result. */
int ssd_prior_grid_verify(const double* priors, int A, ssd_prior_grid* grid, int32_t* scratch, void* stream);

/* Encoding of an all-zero (unmatched) target row against every prior: the value
 * apply_anchor_box (utils/bbox.py:94-101) produces for rows match_bbox left at 0, cast to f32 as
 * the TensorSpec at models/ssd_model.py:222 does.  Image-independent; computed once.
 *   priors DEVICE double[A*4] -> enc_zero DEVICE float[A*4] */
int ssd_encode_zero(const double* priors, int A, float* enc_zero, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched target assignment -- replaces, for a whole batch, the per-image generator body
 *   match_bbox(cls, bbox, prior_box, thresh)   utils/bbox.py:44-91   (called models/ssd_model.py:212)
 *   apply_anchor_box(matched_loc, prior_box)   utils/bbox.py:94-101  (called models/ssd_model.py:213)
 *   + the float32 cast of the TensorSpec        models/ssd_model.py:222
 *
 *   gt_box   float[total_gt*4]  all images' boxes, concatenated (cx,cy,w,h in [0,1], w,h >= 0, finite)
 *   gt_cls   float[total_gt]    class ids stored as float (as the loaders emit them)
 *   gt_off   int32[B+1]         image b owns rows gt_off[b] .. gt_off[b+1]-1   (DEVICE)
 *   total_gt, max_nt            HOST-known sum / max of per-image counts (max_nt <= A else SSD_ERR_ASSERT, :50)
 *   priors   double[A*4]; enc_zero float[A*4] from ssd_encode_zero
 *   grid     HOST, may be NULL.  Optional: the geometry `priors` was generated with (ssd_priors).  Verified
 *            (ssd_prior_grid_verify) it selects the single-launch path, which enumerates the columns that can matter to
 *            a gt row from the geometry; unverified it only seeds a pruning bound.  Results are identical with,
 *            without or with a wrong hint.
 *   thresh   > 0 else SSD_ERR_ASSERT (:51)
 *   out_cls  int32[B*A]   (0 where unmatched -- not the background id, as in the reference)
 *   out_loc  float[B*A*4] encoded offsets (finite "zero-row" values where unmatched, as in the reference)
 *   out_mask uint8[B*A]   1 = positive
 *   out_owner int32[B*A] or NULL: the matched gt row of each anchor within its image (-1 = unmatched),
 *            i.e. index_list of utils/bbox.py:60-79 as a dense map.  The training path passes NULL.
 *   ws       scratch of >= ssd_match_encode_workspace_bytes(B, A, total_gt) bytes (used by the three-launch path only:
 *            no verified grid, or an image with more than 64 boxes); no state is kept in it.
 * Anchor indices / classes / masks are bit-exact with the reference; out_loc is bit-exact in
 * the division terms and <= 1 float32 ulp in the log terms (device log vs numpy log).
 * ---------------------------------------------------------------------------------------- */
size_t ssd_match_encode_workspace_bytes(int B, int A, int total_gt);
int ssd_match_encode(const float* gt_box, const float* gt_cls, const int32_t* gt_off, int B,
                     int total_gt, int max_nt, const double* priors, const float* enc_zero, int A,
                     const ssd_prior_grid* grid, double thresh, int32_t* out_cls, float* out_loc,
                     uint8_t* out_mask, int32_t* out_owner, void* ws, size_t ws_bytes, void* stream);

/* apply_anchor_box (utils/bbox.py:94-101) on n paired rows: float32 boxes against float64 priors,
 * float64 result [n*4] exactly as numpy returns it (division terms bit-exact, log terms <= 1 ulp).
 * The same shapes are required of both inputs by the reference's assert (:95); the wrapper checks. */
int ssd_apply_anchor_box(const float* box, const double* priors, int n, double* out, void* stream);

/* Pairwise IoU, the reference's iou_n (utils/bbox.py:28-41) for float32 box_1 rows against
 * float64 box_2 rows (the dtype mix match_bbox feeds it): out[i] = iou(b1[i], b2[i]), float64,
 * bit-exact.  Exposed for parity tests and for callers of iou_n. */
int ssd_iou_n(const float* b1, const double* b2, int n, double* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training loss, forward + backward -- replaces SSDObjectDetectionModel._ssd_loss
 * (models/ssd_model.py:341-396) and the part of tape.gradient (:248) that flows into the two
 * network outputs.
 *   conf      [B*A*C] class logits, background = class C-1 (:47,364-365); dtype SSD_F32 or SSD_BF16
 *   loc       [B*A*4] predicted offsets, same dtype
 *   gt_cls    int32[B*A], gt_loc float[B*A*4], gt_mask uint8[B*A]: the outputs of ssd_match_encode
 *   grad_scale multiplies both gradients (1.0 = d total_loss)
 *   out8      float[8]: loc loss, cls loss pos, cls loss neg (the reference's three scalars, :392-394),
 *             their sum, P, N (= number of mined negatives, ties included :372), tau, status
 *             (0 ok; 1 = P==0 or 3P > B*A, where TF would raise; 2 = tau==0, where the assert at :375 fires;
 *             3 = a logit row or a positive's offsets were NaN / Inf: a diverged run, reported first)
 *   dconf     [B*A*C], dloc [B*A*4]: gradients, same dtype as conf/loc
 * Hard-negative mining is global over the B images handed in (one micro-batch), as in the reference.
 * fp32 losses agree with a float64 evaluation to <= 1e-4 relative.
 * ---------------------------------------------------------------------------------------- */
size_t ssd_loss_workspace_bytes(int B, int A, int C);
int ssd_loss_fwd_bwd(const void* conf, const void* loc, int dtype, const int32_t* gt_cls,
                     const float* gt_loc, const uint8_t* gt_mask, int B, int A, int C, float grad_scale,
                     float* out8, void* dconf, void* dloc, void* ws, size_t ws_bytes, void* stream);

/* The same loss with the gradient handed over in the form the head convolutions' backward pass consumes, instead of the
 * dense dconf / dloc.  Only the anchors the loss selects carry a gradient -- the positives and the mined negatives, 4P of
 * B*A rows (:355-380; every other row of tape.gradient's result is exactly zero) -- so per feature level l the gradient is
 * a COMPACT list of pixel rows: row r of level l is the head output gradient of pixel pixel_of_row[l][r] (flat index
 * b*hw + y*W + x, ascending in r), channels in the head GEMM's order [per_cell*4 loc | per_cell*C conf | zero pad to npad];
 * row_of_pixel[l][b*hw + pix] is the inverse map (-1: the pixel carries no gradient); count[l] rows are valid.  Scattered
 * back, the rows are bit-identical to ssd_loss_fwd_bwd's dconf / dloc (tests/test_sparse_heads_gpu.py).
 *   hg        HOST struct, device pointers inside; rows / maps sized for B*hw rows (every pixel selected)
 *   dtype     SSD_BF16 only
 *   ws        >= ssd_loss_heads_workspace_bytes(B, A, C) bytes
 *   ws_clean  0: the call clears its histogram words first (a memset node).  1: the caller states they are zero already -- they
 *             are after a COMPLETED call of this function on the same ws with the same B and A (the last launch re-zeroes
 *             them), and no other use of ws in between; a wrong 1 gives a wrong mining threshold, not a fault
 * Consumers: ssd_heads_bwd_data_sparse, ssd_heads_bwd_weight_sparse. */
typedef struct {
    int levels;                               /* <= SSD_MAX_LEVELS; sum of hw*per_cell over the levels == A          */
    int hw[SSD_MAX_LEVELS];                   /* pixels of a level's feature map (H*W)                              */
    int per_cell[SSD_MAX_LEVELS];             /* default boxes per pixel (:153)                                     */
    int npad[SSD_MAX_LEVELS];                 /* row length, >= per_cell*(4+C), multiple of 8                       */
    void* rows[SSD_MAX_LEVELS];               /* DEVICE bf16 [B*hw][npad]                                           */
    int32_t* row_of_pixel[SSD_MAX_LEVELS];    /* DEVICE int32 [B*hw]                                                */
    int32_t* pixel_of_row[SSD_MAX_LEVELS];    /* DEVICE int32 [B*hw]                                                */
    int32_t* count;                           /* DEVICE int32 [SSD_MAX_LEVELS]                                      */
} ssd_head_grads;
size_t ssd_loss_heads_workspace_bytes(int B, int A, int C);
int ssd_loss_fwd_bwd_heads(const void* conf, const void* loc, int dtype, const int32_t* gt_cls, const float* gt_loc,
                           const uint8_t* gt_mask, int B, int A, int C, float grad_scale, float* out8,
                           const ssd_head_grads* hg, void* ws, size_t ws_bytes, int ws_clean, void* stream);

/* ------------------------------------------------------------------------------------------
 * Inference scoring + decode -- replaces the scoring half of SSDObjectDetectionModel.visualize
 * (models/ssd_model.py:479-488, mask=None branch) and the box decode of visualize_dataset (:466-467).
 *   conf [B*A*C], loc [B*A*4] (SSD_F32 / SSD_BF16), priors double[A*4]
 *   score  float[B*A]  best non-background softmax probability
 *   cls    int32[B*A]  its class (== argmax over all classes wherever cand is set)
 *   box    float[B*A*4] decoded (cx,cy,w,h) in pixels (in_size = 300) for candidates, 0 elsewhere
 *   cand   uint8[B*A]  score > score_thresh && !(p_background > score_thresh)
 * ---------------------------------------------------------------------------------------- */
int ssd_score_decode(const void* conf, const void* loc, int dtype, const double* priors, int B, int A, int C,
                     float score_thresh, double in_size, float* score, int32_t* cls, float* box,
                     uint8_t* cand, void* stream);

/* Per-image, per-class greedy hard NMS.  The reference has NO suppression step; this
 * entry point is build-defined and its oracle is oracle/ssd_oracle.py:nms.  Candidates (cand != 0) are
 * ordered by (score desc, anchor asc); the first max_cand (<= ssd_nms_max_candidates()) take part; a
 * candidate is kept iff its IoU -- the reference's scalar iou (utils/bbox.py:6-25) evaluated in float32 --
 * with every already kept candidate of its class is <= iou_thresh.
 *   keep uint8[B*A] (fully written), keep_count int32[B] or NULL.  Bit-exact against the oracle. */
int ssd_nms_max_candidates(void);
int ssd_nms(const float* score, const int32_t* cls, const float* box, const uint8_t* cand, int B, int A,
            float iou_thresh, int max_cand, uint8_t* keep, int32_t* keep_count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution stack (NHWC bf16 activations, bf16 weights [Cout][k][k][Cin], fp32 accumulate on MFMA).
 * Replaces the TensorFlow kernels behind SSDObjectDetectionModel._build (models/ssd_model.py:74-171)
 * and behind tape.gradient (:248) for those layers.  TF "SAME" padding is passed explicitly:
 * pad_total = max((Ho-1)*stride + k - H, 0), pad_t = pad_total/2 (the remainder goes after).
 * Channel counts of activation tensors must be multiples of 8 (the 3-channel image is expanded to 8
 * zero-padded channels by ssd_image_prep).
 * (ws, ws_bytes) of the forward / data-gradient calls is OPTIONAL scratch (may be NULL/0): when given, skinny
 * problems (few output tiles, long reduction) are split over k with a fixed-order fp32 reduction; a few MB suffice
 * (ksplit * B*Ho*Wo * Cout * 4 bytes, only used when the layer has < 160 output tiles).
 * ---------------------------------------------------------------------------------------- */
/* y[B,Ho,Wo,Cout] = relu?(conv(x[B,H,W,Cin], w) + bias)          Conv2D, :86-151 and the VGG blocks :77-82 */
int ssd_conv2d_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin,
                   int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int relu, void* ws,
                   size_t ws_bytes, void* stream);
/* Conv2D + bias + ReLU followed by MaxPool2D 2x2 / stride 2 (models/ssd_model.py:77-84: block1_conv2 -> block1_pool etc.):
 * y as ssd_conv2d_fwd, y_pool [B,Hp,Wp,Cout] and pool_code as ssd_maxpool2x2_fwd_argmax of y.  Layers served by a
 * 16-wide block kernel pool the tile they already hold on chip; otherwise two launches.  Cout % 8 == 0.
 * y may be NULL when nothing but the pooling reads the full-resolution map (true for every pooled layer of the SSD300
 * trunk: the backward pass needs the pooled map and the winner codes only): the fused kernels then skip that store
 * (half of block1_conv2's traffic); a layer shape they do not serve returns SSD_ERR_VALUE before anything is launched. */
int ssd_conv2d_fwd_pool(const void* x, const void* w, const float* bias, void* y, void* y_pool, void* pool_code, int B, int H,
                        int W, int Cin, int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int relu, int Hp,
                        int Wp, void* ws, size_t ws_bytes, void* stream);
/* One pyramid level's loc+conf heads as ONE 3x3 SAME GEMM (N = per_cell*(4+classes); weight rows: loc filters
 * then conf filters), written straight into the concatenated outputs loc[B,A,4] / conf[B,A,classes]
 * (:155-167: the Reshape + Concatenate are the store addressing). */
int ssd_conv2d_head_fwd(const void* x, const void* w, const float* bias, void* loc, void* conf, int B, int H, int W,
                        int Cin, int per_cell, int classes, int anchors_total, int level_off, void* ws, size_t ws_bytes,
                        void* stream);
/* dx[B,H,W,Cin] (+)= conv_transpose(dy[B,Ho,Wo,Cout_pad], w_t), then zeroed where relu_src <= 0 (relu_src =
 * the forward activation stored in dx's place, or NULL).  w_t from ssd_weight_transpose. */
int ssd_conv2d_bwd_data(const void* dy, const void* w_t, const void* relu_src, void* dx, int B, int H, int W, int Cin,
                        int Cout_pad, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int accumulate,
                        void* ws, size_t ws_bytes, void* stream);

/* ReLU sign bits.  ssd_conv2d_fwd_relubits = ssd_conv2d_fwd (relu = 1) that also writes one byte per pixel and 8 output
 * channels (bit k: channel 8c + k > 0), relu_bits [B*Ho*Wo][Cout/8]; ssd_conv2d_bwd_data_bits = ssd_conv2d_bwd_data with
 * its ReLU mask read from such bytes instead of the bf16 activation (relu_src): identical results, 16x fewer mask bytes --
 * re-reading whole activations for their sign cost the data gradients 13-130 us each.  Written / read in the staged store
 * of the kernels (and by the image-layer kernel): SSD_ERR_UNSUPPORTED, nothing launched, where a call resolves to a kernel
 * without it (split-K + finalize, scattered epilogues) -- use the plain functions there. */
int ssd_conv2d_fwd_relubits(const void* x, const void* w, const float* bias, void* y, void* relu_bits, int B, int H, int W, int Cin,
                            int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, void* ws, size_t ws_bytes,
                            void* stream);
int ssd_conv2d_bwd_data_bits(const void* dy, const void* w_t, const void* relu_bits, void* dx, int B, int H, int W, int Cin,
                             int Cout_pad, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, int accumulate, void* ws,
                             size_t ws_bytes, void* stream);
/* Second layer's data gradient and first layer's weight gradient in ONE launch (block1_conv2 / block1_conv1 of the VGG-16
 * trunk, models/ssd_model.py:62-66): the gradient w.r.t. the first layer's output has a single consumer, the 3-channel first
 * layer's weight gradient, so it is multiplied with the image patch where it is produced and never stored -- equal to
 *   ssd_conv2d_bwd_data_bits(dy, w_t, relu_bits, dx, B,H,W, 64, 64, 3,1,1,1, H,W, 0, ...)  followed by
 *   ssd_conv2d_bwd_weight(image, dx, dw0, dbias0, B,H,W, 8, 64, 64, 3,1,1,1, H,W, ...)
 * up to fp32 summation order (dx is rounded to bf16 in both).  dy [B,H,W,64] bf16; w_t [64][3][3][64] (ssd_weight_transpose
 * of the second layer); relu_bits [B*H*W][8] (ssd_conv2d_fwd_relubits of the first layer); image [B,H,W,8] bf16
 * (ssd_image_prep); dw0 f32 [64][3][3][8] (columns of the five pad channels: zeros); dbias0 f32 [64] or NULL.  Fixed summation
 * order (one slab per workgroup in ws, reduced in order): bitwise reproducible.  H, W >= 16. */
size_t ssd_conv2d_bwd_data_wgrad_first_workspace_bytes(int B, int H, int W);
int ssd_conv2d_bwd_data_wgrad_first(const void* dy, const void* w_t, const void* relu_bits, const void* image, float* dw0,
                                    float* dbias0, int B, int H, int W, void* ws, size_t ws_bytes, void* stream);
/* The 3x3 / stride 1 / pad 1 data gradient w.r.t. a POOLED map [B,H,W,Cin], carried on through the 2x2 / stride-2 max
 * pooling that produced the map: dx_full[B,Hf,Wf,Cin] = ssd_maxpool2x2_bwd_argmax(pool_code, ssd_conv2d_bwd_data(...)) in
 * one launch, bit-identical (the un-pooling runs in the convolution's store stage: no pooled gradient in HBM, no second
 * kernel).  Served by the LDS-patch kernels only: SSD_ERR_UNSUPPORTED (nothing launched) for any other layer shape --
 * call the two functions then.
 * dx_pooled (may be NULL): also receives the gradient w.r.t. the pooled map itself [B,H,W,Cin] -- the compressed form
 * ssd_conv2d_bwd_weight_unpooled reads. */
int ssd_conv2d_bwd_data_unpool(const void* dy, const void* w_t, const void* relu_src, const void* pool_code, void* dx_pooled,
                               void* dx_full, int B, int H, int W, int Cin, int Cout_pad, int Hf, int Wf, void* ws, size_t ws_bytes,
                               void* stream);
/* Weight gradient of a 3x3 / stride 1 / pad 1 convolution whose output is 2x2 / stride-2 max-pooled (block1_conv2, block2_conv2,
 * block3_conv3: models/ssd_model.py:77-84), from the gradient of the POOLED map dpool [B,Hp,Wp,Cout] and the pooling's winner codes
 * pool_code [B,Hp,Wp,Cout/8] (ssd_maxpool2x2_fwd_argmax / ssd_conv2d_fwd_pool):
 *   == ssd_conv2d_bwd_weight(x, ssd_maxpool2x2_bwd_argmax(pool_code, dpool), ...)   up to fp32 summation order,
 * without multiplying the three zeros of every pooling window: the window's four pixels are four consecutive k of a
 * structured-sparse MFMA (v_smfmac_f32_16x16x64_bf16), the pooled gradient is the compressed operand, the code its index.
 * Cin, Cout multiples of 64, H, W >= 16, Hp = H/2 or (H+1)/2 (VALID / SAME pooling); SSD_ERR_UNSUPPORTED otherwise (nothing
 * launched).  Deterministic (fixed-order split reduction). */
size_t ssd_conv2d_bwd_weight_unpooled_workspace_bytes(int B, int H, int W, int Cin, int Cout, int Hp, int Wp);
int ssd_conv2d_bwd_weight_unpooled(const void* x, const void* dpool, const void* pool_code, float* dw, float* dbias, int B, int H,
                                   int W, int Cin, int Cout, int Hp, int Wp, void* ws, size_t ws_bytes, void* stream);
/* dw f32 [Cout][k][k][Cin], dbias f32 [Cout] (or NULL) from x[B,H,W,Cin] and dy[B,Ho,Wo,ldy] (first Cout
 * channels).  Deterministic (fixed-order split reduction). */
size_t ssd_conv2d_bwd_weight_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ldy, int ksize);
int ssd_conv2d_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin,
                          int Cout, int ldy, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo, void* ws,
                          size_t ws_bytes, void* stream);
/* Several small layers' weight gradients in two launches instead of two per layer (the extras on the 10x10 ... 1x1 maps,
 * models/ssd_model.py:124-150): arguments per layer as ssd_conv2d_bwd_weight.  Served: layers that call would run on its
 * generic kernel with fewer than 32 pixel splits, at most 8 per call; SSD_ERR_UNSUPPORTED otherwise, nothing launched.
 * Bit-identical to the separate calls (same blocks, same slabs, same fixed-order sums). */
typedef struct {
    const void* x;
    const void* dy;
    float* dw;
    float* dbias;
    int B, H, W, Cin, Cout, ldy, ksize, stride, pad_t, pad_l, Ho, Wo;
} ssd_wgrad_item;
size_t ssd_conv2d_bwd_weight_batched_workspace_bytes(const ssd_wgrad_item* items, int count);
int ssd_conv2d_bwd_weight_batched(const ssd_wgrad_item* items, int count, void* ws, size_t ws_bytes, void* stream);
/* The weight-gradient entry points finish with a fixed-order sum of per-split fp32 slabs (a small HBM-bound launch).  With a
 * non-null stream set here (per calling thread; null = off, the default) that launch goes to `stream`, ordered behind the
 * slab kernel by an event: the caller's stream is free for the next layer's kernel while the sum runs.  The caller then owns
 * two obligations: dw / dbias are complete on `stream`, not on the call's own stream; and the workspace of a call must not be
 * reused before that call's sum has run (alternate two workspaces, wait for the sum two calls back). */
int ssd_set_wgrad_reduce_stream(void* stream);
/* w bf16 [Cout][k][k][Cin] -> w_t bf16 [Cin][k][k][Cout_pad], spatially flipped (data-gradient operand) */
int ssd_weight_transpose(const void* w, void* w_t, int Cout, int ksize, int Cin, int Cout_pad, void* stream);
/* the same for `ntensors` weight tensors in one launch: desc = device array of int64 rows {w, w_t, Cout, ksize, Cin,
 * Cout_pad} (device addresses and sizes), max_tiles = max over tensors of ceil(Cout_pad/32)*ceil(Cin/32)*ksize^2 */
int ssd_weight_transpose_batched(const long long* desc, int ntensors, int max_tiles, void* stream);
int ssd_cast_bf16(const float* src, void* dst, long long n, void* stream);
/* image f32 [B,H,W,3] -> bf16 [B,H,W,8]; normalize != 0 applies (x-0.5)*2 (models/ssd_model.py:214) */
int ssd_image_prep(const float* img, void* out, int B, int H, int W, int normalize, void* stream);
/* Input pipeline on the device for a ragged batch of decoded uint8 RGB images: '/255'
 * (data_loaders/coco/make_dataset.py:117), cv2.resize(image, (S, S)) with its default INTER_LINEAR
 * (data_loaders/ssd/make_dataset.py:40) and, if normalize != 0, (x-0.5)*2 (models/ssd_model.py:214) -> bf16 [B,S,S,8].
 * src: the images back to back (H x W x 3 bytes each), src_off int64 [B] byte offset of each image, src_hw int32 [B][2]. */
int ssd_image_resize_prep(const void* src, const int64_t* src_off, const int32_t* src_hw, void* out, int B, int S,
                          int normalize, void* stream);
/* Ground-truth boxes of the same batch: COCO [x, y, w, h] in pixels (top-left) -> centre form (coco/make_dataset.py:132)
 * divided by the image size (ssd/make_dataset.py:43-44).  gt_off int32 [B+1] as in ssd_match_encode. */
int ssd_box_prep(const float* box_tlwh, const int32_t* gt_off, const int32_t* src_hw, float* box_out, int B, int total_gt,
                 void* stream);
/* MaxPool2D 2x2 stride 2 (VGG block pools: VALID; models/ssd_model.py:84: SAME -> Ho = ceil(H/2)) */
int ssd_maxpool2x2_fwd(const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, void* stream);
/* pooling backward fused with the ReLU backward of the layer that produced x */
int ssd_maxpool2x2_bwd(const void* x, const void* y, const void* dy, void* dx, int B, int H, int W, int C, int Ho,
                       int Wo, void* stream);
/* The same pooling with a recorded winner (training): `code` u32 [B*Ho*Wo*C/8] receives one 4-bit code per pooled
 * element (window position 2*dy+dx of the first maximum, 4 = maximum <= 0: no gradient through the ReLU in front);
 * the backward pass then reads only dy and the codes.  Results are identical to ssd_maxpool2x2_fwd / _bwd. */
int ssd_maxpool2x2_fwd_argmax(const void* x, void* y, void* code, int B, int H, int W, int C, int Ho, int Wo, void* stream);
int ssd_maxpool2x2_bwd_argmax(const void* code, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo,
                              void* stream);
/* Block-scaled fp8 forward convolution (BASELINE configs[4]: "fp8 MFMA convs"; no reference counterpart -- the reference's
 * convolutions are fp32 TensorFlow ops, models/ssd_model.py:86-93 for the >= 256-channel 3x3 layers this serves).  MX format: OCP
 * e4m3 elements, one E8M0 scale byte (2^(s-127)) per 32 consecutive channels, multiplied on v_mfma_scale_f32_16x16x128_f8f6f4.
 *   ssd_quantize_mx_fp8     x bf16 [n] (n % 32 == 0) -> q u8 [n], scale u8 [n/32]: scale = smallest power of two with |x|/scale <= 448
 *   ssd_conv3x3_fwd_mxfp8   y bf16 [B,H,W,Cout] = relu?(conv3x3 SAME stride 1 of x with w + bias); x8 [B,H,W,Cin] / xscale [B,H,W,Cin/32],
 *                           w8 [Cout][3][3][Cin] / wscale [Cout][3][3][Cin/32] as ssd_quantize_mx_fp8 makes them; Cin % 128 == 0,
 *                           Cout % 8 == 0 (SSD_ERR_UNSUPPORTED otherwise); fp32 accumulation */
int ssd_quantize_mx_fp8(const void* x_bf16, void* q, void* scale, long long n, void* stream);
int ssd_conv3x3_fwd_mxfp8(const void* x8, const void* xscale, const void* w8, const void* wscale, const float* bias, void* y, int B,
                          int H, int W, int Cin, int Cout, int relu, void* stream);
/* A chain of small convolutions in ONE launch, one workgroup per image, activations in LDS (chain.hip) -- the reference's
 * "extras" behind the 19x19 map (models/ssd_model.py:124-150: six layers on 10x10 ... 1x1 maps), forward or data gradient.
 * Layer l reads the output of layer l-1 (layer 0: in0 [B][Hi*Wi][Kc] bf16) and writes out [B][Ho*Wo][N] bf16:
 *   out[o][n] = epilogue( sum over taps t, channels k of  in[(o * mul + t - pad) / div][k] * w[n][t][k] )
 * with the source pixel taken only when divisible by div and inside the map -- the implicit-GEMM geometry of
 * ssd_conv2d_fwd (mul = stride, div = 1, w = filters [Cout][k][k][Cin]) and of ssd_conv2d_bwd_data (mul = 1, div = stride,
 * pad = ksize - 1 - pad, w = ssd_weight_transpose's [Cin][k][k][Cout_pad]).  Epilogue: + bias (if not null), ReLU (relu != 0),
 * sign bits written to relu_bits (if not null); accumulate != 0: out += result BEFORE the mask (two gradients meet at a
 * feature map); mask_bits / mask_src (at most one): zero where the bit is clear / the value is <= 0.
 * Needs Kc % 128 == 0, N % 16 == 0, Ho*Wo <= 112, div in {1, 2}, every layer's input + output image within 160 KB of LDS
 * (rows padded by 16 bytes): SSD_ERR_UNSUPPORTED otherwise, nothing launched.  fp32 accumulation in one pass over k. */
#define SSD_CHAIN_MAX_LAYERS 8
#define SSD_CHAIN_PACK_MAX 16
/* w of a chain layer is NOT the filter tensor itself but its fragment-packed copy (ssd_chain_pack_weights): for filters
 * [N][K] bf16 with k = (tap, channel) contiguous -- the forward filters [Cout][k*k*Cin] or ssd_weight_transpose's
 * [Cin][k*k*Cout_pad] -- packed[N/16][K/32][64][8] with element (lane, e) of fragment (nt, s) = w[16 nt + (lane & 15)]
 * [32 s + 8 (lane >> 4) + e]: a wave's MFMA operand is one contiguous KB.  N % 16 == 0, K % 32 == 0; up to
 * SSD_CHAIN_PACK_MAX tensors per launch.  Re-pack after every optimizer step. */
typedef struct {
    const void* src;
    void* dst;
    int N, K;
} ssd_chain_pack;
int ssd_chain_pack_weights(const ssd_chain_pack* items, int count, void* stream);
/* Optional: read the packed copies (items[i].dst, N, K; src unused) into every XCD's L2 from a second stream shortly before
 * the ssd_conv_chain launch that uses them -- the chain's workgroups stream the filters at one miss latency per 128 KB in
 * flight when they are cold.  Reads only; no effect on results. */
int ssd_chain_prefetch(const ssd_chain_pack* items, int count, void* stream);
typedef struct {
    const void* w;
    const float* bias;
    void* out;
    const void* mask_bits;
    const void* mask_src;
    void* relu_bits;
    int Hi, Wi, Kc, Ho, Wo, N, ksize, mul, div, pad_t, pad_l, relu, accumulate;
} ssd_chain_layer;
int ssd_conv_chain(const void* in0, const ssd_chain_layer* layers, int nlayers, int B, void* stream);
/* Pieces of a ResNet-50 trunk (BASELINE configs[4]; the reference hard-codes its VGG trunk, models/ssd_model.py:46,75-97, so these
 * have no reference counterpart: semantics are Keras / TensorFlow's Add + ReLU, MaxPooling2D(3, strides=2, padding="same") and
 * their tape.gradient).  bf16 NHWC, n = element count (a multiple of 8).
 *   ssd_add_relu_fwd      out = relu(a + b)                               (residual add of a bottleneck block)
 *   ssd_relu_mask_bwd     out (+)= g where act > 0                        (its gradient on the identity-skip branch)
 *   ssd_maxpool3x3s2_fwd  3x3 / stride-2 max pooling with explicit top / left padding; code u32 [B*Ho*Wo*C/8]: one nibble per
 *                         element = 3 dy + dx of the first maximum, 15 = maximum <= 0 (no gradient through the ReLU in front)
 *   ssd_maxpool3x3s2_bwd  dx = gather of dy over the (overlapping) windows whose winner is the pixel: deterministic */
int ssd_add_relu_fwd(const void* a, const void* b, void* out, long long n, void* stream);
int ssd_relu_mask_bwd(const void* g, const void* act, void* out, int accumulate, long long n, void* stream);
int ssd_maxpool3x3s2_fwd(const void* x, void* y, void* code, int B, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l,
                         void* stream);
int ssd_maxpool3x3s2_bwd(const void* code, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int pad_t, int pad_l,
                         void* stream);
/* ------------------------------------------------------------------------------------------
 * Backward pass of ALL head convolutions (models/ssd_model.py:153-162; 3x3, stride 1, SAME, no activation) from the compact
 * gradient rows of ssd_loss_fwd_bwd_heads -- the part of tape.gradient (:248) behind the loc / conf outputs.  Work is
 * proportional to the rows that carry a gradient (4P of B*A anchors), not to the feature maps; results equal the dense
 * ssd_conv2d_bwd_data / ssd_conv2d_bwd_weight on the scattered rows up to fp32 summation order (dx rounds once to bf16).
 * One launch sequence serves every level; row counts are read on the device (no host synchronisation); deterministic.
 *   hl  HOST struct, per level: the feature map x bf16 [B,H,W,Cin] (Cin % 128 == 0 else SSD_ERR_UNSUPPORTED), cout =
 *       per_cell*(4+C) filters, w_tap = the head filters transposed tap-major bf16 [3][3][Cin][npad] (w_tap[kh][kw][ci][co]
 *       = w[co][kh][kw][ci], zero for co >= cout; ssd_weight_transpose_batched with bit 8 of the kernel-size field set),
 *       the ReLU mask of x as sign bits (relu_bits, uint8 [B,H,W,Cin/8]) or as the activation itself (relu_src) or neither,
 *       outputs dx bf16 [B,H,W,Cin] (overwritten: every pixel, zeros where no row reaches), dw f32 [cout][3][3][Cin], dbias
 *       f32 [cout] or NULL.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int levels;
    int H[SSD_MAX_LEVELS], W[SSD_MAX_LEVELS], Cin[SSD_MAX_LEVELS], cout[SSD_MAX_LEVELS];
    const void* x[SSD_MAX_LEVELS];
    const void* w_tap[SSD_MAX_LEVELS];
    const void* relu_bits[SSD_MAX_LEVELS];
    const void* relu_src[SSD_MAX_LEVELS];
    void* dx[SSD_MAX_LEVELS];
    float* dw[SSD_MAX_LEVELS];
    float* dbias[SSD_MAX_LEVELS];
} ssd_head_layers;
size_t ssd_heads_bwd_data_sparse_workspace_bytes(int B, const ssd_head_layers* hl);
int ssd_heads_bwd_data_sparse(const ssd_head_grads* hg, const ssd_head_layers* hl, int B, void* ws, size_t ws_bytes,
                              void* stream);
/* The same for a subset of the levels (bit l of level_mask): the small maps' gradients head the extras' data-gradient
 * chain, the 38x38 / 19x19 maps' are not read until that chain reaches them, so a caller may issue the two groups on two
 * streams.  Every level keeps its own slice of `ws`: calls with disjoint masks may share one workspace concurrently.
 * prezeroed != 0: the caller has cleared dx of the selected levels (anywhere earlier in the step, off the critical path) and
 * pixels that no gradient row reaches -- ~95 % of a large map, 140 MB of zero stores at batch 64 -- are not written again.
 * SSD_ERR_VALUE for an empty mask. */
int ssd_heads_bwd_data_sparse_levels(const ssd_head_grads* hg, const ssd_head_layers* hl, int B, unsigned level_mask, int prezeroed,
                                     void* ws, size_t ws_bytes, void* stream);
size_t ssd_heads_bwd_weight_sparse_workspace_bytes(int B, const ssd_head_grads* hg, const ssd_head_layers* hl);
int ssd_heads_bwd_weight_sparse(const ssd_head_grads* hg, const ssd_head_layers* hl, int B, void* ws, size_t ws_bytes,
                                void* stream);

/* dloc[B,A,4], dconf[B,A,classes] (bf16) -> one level's padded NHWC head gradient [B, hw, npad] */
int ssd_head_grad_pack(const void* dloc, const void* dconf, void* out, int B, int hw, int per_cell, int classes,
                       int npad, int anchors_total, int level_off, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer step over one flat fp32 parameter buffer -- replaces the per-tensor Python loop of
 * _train_step: tf.clip_by_norm(g, 0.01) (models/ssd_model.py:249), the micro-batch mean (:251-256) and
 * optimizer.apply_gradients (:258-260; Adam/SGD and their hyper-parameters: tools/train.py:42-53).
 * Every tensor starts on a multiple of ssd_opt_block_elems() elements (pad with zeros);
 *   tensor_block_off int32[ntensors+1]: first block of each tensor;  block_tensor int32[n/block]: owner.
 * ---------------------------------------------------------------------------------------- */
int ssd_opt_block_elems(void);
/* scale[t] = clip / max(||grad_t||_2, clip) (clip <= 0: 1), norms[t] optional; partial: double[n/block] scratch */
int ssd_grad_clip_scales(const float* grad, long long n, const int32_t* tensor_block_off, int ntensors, float clip,
                         double* partial, float* scale, float* norms, void* stream);
/* grad *= scale[tensor]  in place (clipped gradients are what data-parallel ranks all-reduce) */
int ssd_grad_apply_scale(float* grad, long long n, const int32_t* block_tensor, const float* scale, void* stream);
/* acc = (first ? 0 : acc) + grad * scale[tensor]: the micro-batch accumulation of clipped gradients (:251-255) */
int ssd_grad_accumulate(float* acc, const float* grad, long long n, const int32_t* block_tensor, const float* scale,
                        int first, void* stream);
/* Keras Adam with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) precomputed by the caller; the gradient is multiplied by
 * grad_scale (1/micro-batches or 1/world) and, if scale != NULL, by scale[tensor].  param_bf16 (may be NULL)
 * receives the bf16 copy the convolutions read. */
int ssd_adam_step(float* param, const float* grad, float* m, float* v, void* param_bf16, long long n,
                  const int32_t* block_tensor, const float* scale, float grad_scale, float lr_t, float beta1,
                  float beta2, float eps, void* stream);
int ssd_sgd_step(float* param, const float* grad, void* param_bf16, long long n, const int32_t* block_tensor,
                 const float* scale, float grad_scale, float lr, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dispatch queries (no reference counterpart; testing / reports): which kernel a convolution call with this shape
 * resolves to.  They run the library's own dispatch code with launching switched off, so the answer cannot drift from
 * what the call does.  Return: a plan word >= 0 (kernel id in SSD_PLAN_KERNEL_MASK, path flags above it) or a negative
 * ssd_status for a shape the call itself would reject.  pool: 0 = ssd_conv2d_fwd, 1 = ssd_conv2d_fwd_pool, 2 = the same
 * with y == NULL.  ws_bytes: the workspace the call would be given (0: none, no split-K).
 * ---------------------------------------------------------------------------------------- */
enum ssd_conv_plan {
    SSD_PLAN_KERNEL_MASK = 0xff,
    SSD_PLAN_C64B = 1, SSD_PLAN_P32_64, SSD_PLAN_P32_128, SSD_PLAN_8PH,
    SSD_PLAN_DMA_256_256, SSD_PLAN_DMA_256_128, SSD_PLAN_DMA_256_64, SSD_PLAN_DMA_128_64, SSD_PLAN_DMA_128_128,
    SSD_PLAN_CONV0_FWD, SSD_PLAN_P512, SSD_PLAN_PW,
    SSD_PLAN_WG_FIRST = 32, SSD_PLAN_WG_PATCH_16x16, SSD_PLAN_WG_PATCH_6x40, SSD_PLAN_WG_PATCH_10x24, SSD_PLAN_WG_TILE,
    SSD_PLAN_WG_GENERIC,
    SSD_PLAN_F_FLAT = 0x100,         /* strip blocks over a narrow map */
    SSD_PLAN_F_ROWFLAT = 0x200,      /* one strip of rows over all images */
    SSD_PLAN_F_SPLITK = 0x400,       /* split-K partial sums + k_igemm_finalize */
    SSD_PLAN_F_POOL_FUSED = 0x800,   /* 2x2 pooling computed in the convolution's epilogue */
    SSD_PLAN_F_S2 = 0x1000,          /* stride-2 data gradient by parity classes */
    SSD_PLAN_F_REDUCE_WIDE = 0x2000  /* >= 32 weight-gradient splits: k_wgrad_reduce_wide (else k_wgrad_reduce2) */
};
int ssd_conv2d_fwd_plan(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo,
                        int pool, size_t ws_bytes);
int ssd_conv2d_head_fwd_plan(int B, int H, int W, int Cin, int per_cell, int classes, size_t ws_bytes);
int ssd_conv2d_bwd_data_plan(int B, int H, int W, int Cin, int Cout_pad, int ksize, int stride, int pad_t, int pad_l, int Ho,
                             int Wo, int accumulate, size_t ws_bytes);
int ssd_conv2d_bwd_weight_plan(int B, int H, int W, int Cin, int Cout, int ldy, int ksize, int stride, int pad_t, int pad_l,
                               int Ho, int Wo);
const char* ssd_conv_plan_name(int plan);

/* Development only (no reference counterpart): override one of the library's kernel-selection defaults (names in DESIGN.md
 * section 4, e.g. "SSD_CONV_TILE") for A/B timing of kernel variants or to force a dispatch path in a test, inside one
 * process.  This is the ONE piece of mutable state the library has besides the once-per-device registration of large-LDS
 * kernels: relaxed atomics, never set by the product, never read from the environment, and results never depend on it (every
 * variant computes the same convolution).  SSD_ERR_VALUE for an unknown name. */
int ssd_dev_knob(const char* name, int value);

/* Measurement only (no reference counterpart, not on any product path): the inner loop of the convolution kernels without
 * their memory traffic -- 256 workgroups x 8 waves, each wave `iters` x 32 bf16 16x16x32 MFMAs fed by ds_read_b128 fragment
 * reads of random operands in LDS.  bench.py times it beside the train step: the rate this device sustains (devices of one
 * model differ by ~12 % on such a loop) and the in-kernel clock.
 *   stamps  DEVICE uint64[2 * ssd_dev_mfma_calibration_workgroups()]: per workgroup (delta s_memtime, delta s_memrealtime)
 *           around the loop -> clock = delta s_memtime / delta s_memrealtime x 100 MHz
 *   sink    DEVICE float[512 * workgroups], written and never read
 *   flop count of one call: ssd_dev_mfma_calibration_flops(iters) */
int ssd_dev_mfma_calibration_workgroups(void);
double ssd_dev_mfma_calibration_flops(int iters);
int ssd_dev_mfma_calibration(int iters, void* stamps, void* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSD_HIP_H */
