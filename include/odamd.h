/*
 * odamd.h -- C ABI of libodamd.so, the MI355X (gfx950) kernels for the YOLOv7 hot path.
 *
 * The reference (TWK2022/ObjectDetection) is pure Python and has no FFI of its own, so each
 * entry point below names the reference function whose arithmetic it replaces (file:line,
 * relative to the reference tree).  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the library allocates nothing: callers pass outputs and a workspace (sizes from the
 *     *_workspace_bytes queries); it keeps no mutable global state and is re-entrant;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work, they never
 *     synchronise;
 *   - return value: 0 on success, a negative ODAMD_E_* code for a rejected argument,
 *     a positive hipError_t if a launch failed.  Nothing throws across the boundary.
 */
#ifndef ODAMD_H
#define ODAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODAMD_OK 0
#define ODAMD_E_ARG (-1)       /* null pointer / non-positive size / unsupported value */
#define ODAMD_E_ALIGN (-2)     /* pointer or channel count violates the stated alignment */
#define ODAMD_E_WORKSPACE (-3) /* workspace too small */
#define ODAMD_E_SHAPE (-4)     /* shapes inconsistent with each other */
#define ODAMD_E_LIMIT (-5)     /* size beyond the fixed capacity of an allocation-free entry point */

#define ODAMD_DTYPE_F32 0
#define ODAMD_DTYPE_BF16 1

/* ABI version of this header (3); bumped on any change of a signature or of a structure the caller fills. */
int odamd_abi_version(void);
/* Human-readable text for a return code (static storage). */
const char* odamd_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * Box decode.  Replaces decode.forward (model/layer.py:273-293) and, with deploy != 0, the
 * sigmoid of deploy.forward (model/layer.py:302-305).
 *
 * head[i]   : raw output of head i, float32, the reference's (B, A, H_i, W_i, K) tensor, which
 *             is the NCHW conv output reinterpreted without a permute (model/layer.py:235).
 * out       : (B, R, K) float32, R = A * sum_i H_i*W_i, rows ordered scale0[a,y,x], scale1, scale2.
 * anchor_px : (3, A, 2) float32 HOST array, anchor * input_size already multiplied in float32.
 * stride    : (3) float32 HOST array, input_size / H_i.
 * deploy    : 0 -> channels 4: copied (training graph); 1 -> sigmoid applied to channels 4:.
 */
typedef struct {
    const float* head[3];
    int h[3];
    int w[3];
    float stride[3];
    float anchor_px[3][4][2]; /* up to 4 anchors per scale */
    int n_anchor;             /* A, 1..4 */
    int batch;                /* B */
    int kdim;                 /* K = 5 + classes */
    int deploy;
} odamd_decode_desc;

int odamd_decode(const odamd_decode_desc* d, float* out, void* stream);

/* deploy.forward (model/layer.py:304) on an already decoded (rows, K) float32 tensor:
 * sigmoid of channels 4: in place. */
int odamd_deploy_sigmoid(float* pred, size_t rows, int kdim, void* stream);

/* Backward of odamd_decode with deploy == 0: grad wrt the raw head tensors given grad wrt the
 * decoded tensor and the raw heads (autograd of model/layer.py:279-286).
 * grad_head[i] has the shape of head[i]. */
int odamd_decode_backward(const odamd_decode_desc* d, const float* grad_out, float* const grad_head[3],
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-processing.  Replaces predict_class._decode / _nms / iou (predict.py:67-92, :33-41)
 * for a whole batch: (1) x_min = cx - w/2, y_min = cy - h/2; (2) keep conf > conf_thr and
 * x_min > 0 and y_min > 0; (3) score = conf * max(cls), order by descending score (ties:
 * descending row index); (4) greedy class-agnostic NMS, candidate survives a kept box iff
 * IoU < iou_thr evaluated in float32 with the reference's operation order; (5) kept rows
 * converted back to cx,cy and scaled by (xs, ys) per image.
 *
 * pred      : (B, R, K) float32 deployed predictions (not modified).
 * scale_xy  : (B, 2) float32 device array {image_w/input_size, image_h/input_size}, or NULL for 1.
 * n_pad     : power of two >= the largest candidate count of any image (R rounded up is always
 *             safe and needs no host sync); 0 selects that safe bound.
 * out_rows  : (B, max_keep, K) float32, kept rows in keep order (may be NULL to skip step 5).
 * out_idx   : (B, max_keep) int32, row index into pred of every kept box, in keep order.
 * out_count : (B) int32 number of kept boxes (may exceed max_keep; only max_keep are stored).
 * cand_count: (B) int32 number of candidates that passed step (2) (may be NULL).
 */
size_t odamd_postprocess_workspace_bytes(int batch, int rows, int kdim, int n_pad);

int odamd_postprocess(const float* pred, int batch, int rows, int kdim, float conf_thr, float iou_thr,
                      const float* scale_xy, int n_pad, float* out_rows, int* out_idx, int* out_count,
                      int* cand_count, int max_keep, void* workspace, size_t workspace_bytes, void* stream);

/* Same chain fed from the three RAW head tensors: decode + deploy-sigmoid are evaluated on the
 * fly, so the (B,R,K) tensor is never written (model/layer.py:273-305 + predict.py:67-92).
 * out_rows holds fully decoded + deployed rows of the kept boxes. */
int odamd_postprocess_heads(const odamd_decode_desc* d, float conf_thr, float iou_thr, const float* scale_xy,
                            int n_pad, float* out_rows, int* out_idx, int* out_count, int* cand_count,
                            int max_keep, void* workspace, size_t workspace_bytes, void* stream);

/* Stand-alone predict_class._nms (predict.py:82-92): rows are (x_min,y_min,w,h,conf,cls...) and all
 * of them are candidates; kept rows are returned unchanged, in keep order. */
int odamd_nms_rows(const float* rows_xywh, int batch, int rows, int kdim, float iou_thr, int n_pad, float* out_rows,
                   int* out_idx, int* out_count, int max_keep, void* workspace, size_t workspace_bytes, void* stream);

/* odamd_nms_rows without a workspace argument and without any allocation.  It has no counterpart in
 * the reference: it exists for callers that cannot allocate device memory around the call (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The workspace is a static
 * device array of ODAMD_STATIC_SCRATCH_MIB MiB that every GPU the library runs on holds until the
 * process ends; n_pad is the power of two >= rows (at least 64).  Sizes whose workspace
 * (odamd_postprocess_workspace_bytes(batch, rows, kdim, 0)) exceeds the array are rejected with
 * ODAMD_E_LIMIT -- e.g. 1 image of 8192 rows, 6 of 4096, 24 of 2048 fit.  Results equal odamd_nms_rows
 * bit for bit.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU
 * (different streams or threads) race; serialise them.  The first call on a device looks the array's
 * address up (hipGetSymbolAddress); make that call outside stream capture. */
#define ODAMD_STATIC_SCRATCH_MIB 16
int odamd_nms_rows_static(const float* rows_xywh, int batch, int rows, int kdim, float iou_thr, float* out_rows,
                          int* out_idx, int* out_count, int max_keep, void* stream);

/* Candidate selection + NMS of the validation metric (train_class.py:340-346 with :76-83): pred is
 * the NON-deployed (B,R,K) tensor (cx,cy,w,h,raw conf,raw cls..).  Rows with conf > conf_thr become
 * (x_min,y_min,w,h) boxes, score = conf * max(cls); greedy NMS in stable descending score order
 * suppresses a candidate iff IoU > nms_thr (the caller passes 1 - iou_threshold, train_class.py:80;
 * the rule of torchvision.ops.nms on corner boxes); the first max_keep kept rows (the reference
 * uses 100, train_class.py:346) are written as (x_min,y_min,w,h,conf,cls..), out_count holds the
 * uncapped number kept.  Other arguments as in odamd_postprocess. */
int odamd_metric_nms(const float* pred, int batch, int rows, int kdim, float conf_thr, float nms_thr, int n_pad,
                     float* out_rows, int* out_idx, int* out_count, int* cand_count, int max_keep, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Stand-alone train_class.nms (train_class.py:77-83): rows are (x_min,y_min,w,h,conf,cls..), all of
 * them candidates; same ordering and suppression rule as odamd_metric_nms; kept rows are written
 * in keep order with w, h as the corner round trip (x_min + w) - x_min leaves them. */
int odamd_metric_nms_rows(const float* rows_xywh, int batch, int rows, int kdim, float nms_thr, int n_pad,
                          float* out_rows, int* out_idx, int* out_count, int max_keep, void* workspace,
                          size_t workspace_bytes, void* stream);

/* True / false positive counting of train_class.metric (train_class.py:347-365) for a batch:
 * kept_rows (B,max_keep,K) and keep_count (B) as written by odamd_metric_nms (max_keep <= 1024),
 * labels (B,lmax,K) rows (cx,cy,w,h,conf,cls..) with n_label (B) valid rows per image.  A label is
 * found when a kept row of the same argmax class has IoU > iou_thr with it (train_class.py:67-74).
 * counts: (B,3) int32 = (tp, fn, fp) per image. */
int odamd_metric_match(const float* kept_rows, const int* keep_count, int max_keep, const float* labels,
                       const int* n_label, int lmax, int batch, int kdim, float iou_thr, int* counts, void* stream);

/* Building blocks of the chain, exported for tests and for callers that bring their own
 * candidate set.  boxes: (n) float4 {x_min,y_min,w,h}; bit j of mask word (i, j/64) is set
 * iff j > i and not (IoU(i,j) < iou_thr).  predict.py:33-41. */
/* predict_class.iou (predict.py:33-41): out[i] = IoU(a[i], b[i]) for (x_min,y_min,w,h) float4 boxes;
 * na or nb may be 1 (that box is broadcast). */
int odamd_iou_pairs(const float* a_xywh, int na, const float* b_xywh, int nb, float* out, int n, void* stream);
int odamd_iou_mask(const float* boxes_xywh, int n, float iou_thr, uint64_t* mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution block.  Replaces cbs.forward in eval mode (model/layer.py:12-16): Conv2d without
 * bias, BatchNorm folded by the caller into per-channel (scale, shift), SiLU.  Also serves
 * head.forward (model/layer.py:234-236) with act = 0 and NCHW float32 output.
 *
 * Activations are NHWC ("channels last"): element (b, y, x, c) of a tensor lives at
 * base[((b*H + y)*W + x)*ld + c], ld >= C.  A concat (model/layer.py:25-27) is therefore a
 * set of producers writing at channel offsets of one buffer: pass y = buffer + offset and
 * y_ld = total channels.
 */
typedef struct {
    const void* x;     /* input activations, dtype `dtype`, NHWC with pixel stride x_ld */
    const void* w;     /* packed weights [cout][ky][kx][cin], dtype `dtype` */
    const float* scale;/* [cout] float32 (BN gamma/sqrt(var+eps)), NULL = 1 */
    const float* shift;/* [cout] float32 (BN beta - mean*scale, or conv bias), NULL = 0 */
    void* y;           /* output */
    int batch, h, w_in;/* input spatial size */
    int cin, cout;     /* cin % 16 == 0 unless first-layer kernel; any cout */
    int x_ld, y_ld;    /* pixel strides in elements (NHWC), y_ld ignored for NCHW output */
    int ksize;         /* 1 or 3, padding (ksize-1)/2 */
    int stride;        /* 1 or 2 */
    int act;           /* 0 none, 1 SiLU */
    int dtype;         /* ODAMD_DTYPE_* of x, w (and of y when out_nchw_f32 == 0) */
    int in_pool2;      /* 1: input is max-pooled 2x2/2 on the fly (mp.maxpool0, model/layer.py:140) */
    int out_nchw_f32;  /* 1: y is float32 NCHW (B, cout, Ho, Wo) -- the head layout */
    int in_upzero2;    /* 1: input is zero-upsampled x2 on the fly (value at even (y,x), zero elsewhere): with
                          flipped, transposed weights this is the data gradient of a stride-2 3x3 convolution */
    int accumulate;    /* 1: NHWC output is added to the destination (gradient accumulation) */
    int w_packed;      /* 1: w is in the fragment-ordered layout of odamd_conv_weight_layout (only where that
                          function returns 1 for this descriptor) */
} odamd_conv_desc;

int odamd_conv_forward(const odamd_conv_desc* d, void* stream);

/* Several bf16 layer classes are served by kernels that want the weights in the order their matrix fragments
 * are read: deep stride-1 layers on small maps (operands global -> LDS without registers), the 32-channel 3x3
 * layers on the largest maps (persistent, weights resident in LDS) and 1x1 layers with 256 / 512 / 1024 input
 * channels (persistent, weights resident in registers: one chunk = the whole reduction).  Returns 1 and sets
 * (cout_groups G, chunk_bytes CB) when such a form exists for the layer `d` describes (pointers in `d` are
 * ignored), else 0.  Packed layout, T = ksize^2 taps, cout padded with zero rows to a multiple of 32*G,
 * cin*2 a multiple of CB:
 *   [cout / (32 G)][cin * 2 / CB][G][T][CB / 32][2][32][8] bf16,
 *   element (cb, chunk, g, tap, s, hh, r, e) = w[cb*32*G + g*32 + r][tap][chunk*CB/2 + s*16 + hh*8 + e]
 * with w the plain [cout][ky][kx][cin] array.  Same arithmetic as the plain form (cbs.forward,
 * model/layer.py:12-16). */
int odamd_conv_weight_layout(const odamd_conv_desc* d, int* cout_groups, int* chunk_bytes);

/* The two 1x1 cbs blocks at the head of an mp block in ONE launch (mp.forward, model/layer.py:139-145: cbs1 on
 * maxpool0(x), cbs2 on x): both read the same tensor, the largest of the block, which this entry point fetches
 * once.  `pooled` describes cbs1 (in_pool2 = 1, output at half resolution), `plain` cbs2; both must name the same
 * input (x, batch, h, w_in, cin, x_ld), bf16, plain weights, SiLU.  The fused form exists for whole 8 x 16 patches
 * (h % 8 == 0, w_in % 16 == 0), 32 / 64 / 128 input channels and 32 / 64 / 128 output channels per branch, and for
 * whole 8 x 8 patches (h % 8 == 0, w_in % 8 == 0) with 256 input channels and 128 + 128 output channels (input
 * channels walked in chunks, the weights streamed with them):
 * odamd_conv_mp_supported returns 1 then (pointers are ignored by the query), and odamd_conv_mp_forward gives the
 * same results as the two odamd_conv_forward calls.  Otherwise the forward call returns ODAMD_E_SHAPE. */
int odamd_conv_mp_supported(const odamd_conv_desc* pooled, const odamd_conv_desc* plain);
int odamd_conv_mp_forward(const odamd_conv_desc* pooled, const odamd_conv_desc* plain, void* stream);

/* Two chained 3x3 stride-1 cbs blocks in ONE launch: the 3x3 chains of elan (sequential2 -> sequential3,
 * model/layer.py:74-81) and elan_h (cbs2 -> cbs3, cbs4 -> cbs5, model/layer.py:109-118), whose every link is also a
 * slice of the block's concat.  `second` must read exactly what `first` writes (second->x == first->y, same x_ld /
 * y_ld, batch, h, w_in); bf16, w_packed = 1 with the whole tensor as one block of fragments (the layout of
 * odamd_conv_weight_layout with G = 1, CB = cin * 2), folded BatchNorm constants on both; 32 or 64 input channels, 32 channels
 * between and after the two.  Both outputs are written (the first one is a concat slice too); the second convolution
 * reads the first one's tile, halo included, from LDS.  odamd_conv_chain_pair_supported returns 1 where the fused
 * form exists (the descriptors' pointers take part: alignment and the x / y identity), and the forward call gives
 * the same results as the two odamd_conv_forward calls; otherwise it returns ODAMD_E_SHAPE. */
int odamd_conv_chain_pair_supported(const odamd_conv_desc* first, const odamd_conv_desc* second);
int odamd_conv_chain_pair_forward(const odamd_conv_desc* first, const odamd_conv_desc* second, void* stream);

/* First layer: float32 NCHW image (B,3,H,W) -> NHWC activations; replaces cbs0 of
 * yolov7.forward (model/yolov7.py:114).  w: [cout][ky][kx][3] float32; scale/shift may both be
 * NULL (raw convolution for train-mode BatchNorm); act: 0 none, 1 SiLU. */
int odamd_conv_first_forward(const float* image, const float* w, const float* scale, const float* shift,
                             void* y, int batch, int h, int w_in, int cout, int y_ld, int act, int dtype,
                             void* stream);

/* First two layers in one launch, bf16 compute: cbs0 (3 -> 16, 3x3 stride 1) and cbs1 (16 -> 32, 3x3
 * stride 2) of yolov7.forward (model/yolov7.py:114-115, preset 's'), eval-mode BatchNorm folded to
 * scale/shift, SiLU after both.  The 16-channel activation between them stays on chip.
 * w0: [16][ky][kx][3] float32; w1: [32][ky][kx][16] bf16; y: NHWC bf16 (B, Ho, Wo, y_ld), 32 channels. */
int odamd_conv_first2_forward(const float* image, const float* w0, const float* scale0, const float* shift0,
                              const void* w1, const float* scale1, const float* shift1, void* y, int batch, int h,
                              int w_in, int y_ld, void* stream);

/* MaxPool2d(2,2) (model/layer.py:125), three stride-1 max pools k=5,9,13 of one tensor
 * (model/layer.py:215-217) and nearest Upsample x2 (model/yolov7.py:32), all NHWC, each
 * able to write into a channel slice of a wider buffer. */
int odamd_maxpool2_forward(const void* x, void* y, int batch, int h, int w, int c, int x_ld, int y_ld,
                           int dtype, void* stream);
int odamd_spp_pool3_forward(const void* x, void* y5, void* y9, void* y13, int batch, int h, int w, int c,
                            int x_ld, int y_ld, int dtype, void* stream);
int odamd_upsample2_forward(const void* x, void* y, int batch, int h, int w, int c, int x_ld, int y_ld,
                            int dtype, void* stream);
/* Layout converters between NCHW float32 (the module API) and NHWC activations.  `c` channels
 * are converted; x_c / y_c is the channel count of the whole NCHW tensor (>= c), so a channel
 * slice of it can be addressed by offsetting the pointer. */
int odamd_nchw_to_nhwc(const float* x, void* y, int batch, int c, int h, int w, int x_c, int y_ld, int dtype,
                       void* stream);
int odamd_nhwc_to_nchw(const void* x, float* y, int batch, int c, int h, int w, int x_ld, int y_c, int dtype,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (train_class.py:193-218): train-mode BatchNorm + SiLU, their backward, pooling /
 * upsample backward, weight gradients, loss, optimizer.  Data gradients of convolutions reuse
 * odamd_conv_forward with flipped, transposed weights (odamd_pack_weights builds them).
 */

/* Workspace for the per-channel reductions below (BN backward needs 2*channels floats more). */
size_t odamd_bn_workspace_bytes(int channels);

/* cbs.bn + cbs.silu in train mode (model/layer.py:14-15): batch statistics of the raw convolution
 * output y over npix = B*H*W pixels, running buffers updated (momentum, unbiased variance),
 * out = silu((y - mean)*scale + shift) written with pixel stride out_ld, scale = gamma*invstd, shift =
 * beta.  mean/invstd/scale/shift (channels floats each) are kept for the backward pass. */
int odamd_bn_train_forward(const void* y, int y_ld, void* out, int out_ld, size_t npix, int channels,
                           const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                           float* running_var, float* mean, float* invstd, float* scale, float* shift, int dtype,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Autograd of the above: given dact = dL/d(out), overwrites y with dL/dy in place and ADDS the
 * parameter gradients to dgamma / dbeta. */
int odamd_bn_silu_backward(void* y_dy, int y_ld, const void* dact, int dact_ld, size_t npix, int channels,
                           const float* scale, const float* shift, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, int dtype, void* workspace, size_t workspace_bytes,
                           void* stream);

/* dbias += sum over pixels of dy (head.output.bias, model/layer.py:232). */
int odamd_bias_grad(const void* dy, int dy_ld, size_t npix, int channels, float* dbias, int dtype, void* workspace,
                    size_t workspace_bytes, void* stream);

/* Backward of MaxPool2d(2,2), of a stride-1 MaxPool2d(k, pad k/2) (bf16: planes of <= 1024 pixels) and of nearest
 * Upsample x2.  All ADD into dx.  Ties go to the first maximum in row-major window order. */
int odamd_maxpool2_backward(const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld, int batch, int h,
                            int w, int c, int dtype, void* stream);
int odamd_maxpool_s1_backward(const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld, int batch,
                              int h, int w, int c, int ksize, int dtype, void* stream);
/* Backward of the three stride-1 max pools of an SPP block at once (sppcspc.MaxPool2d4 / 5 / 6 on one input,
 * model/layer.py:186-188, :215-217, as torch.autograd differentiates them): dx += the three window gradients, each sent to the
 * first maximum of its window in row-major order.  dy5 / dy9 / dy13: the gradients of the three pooled tensors (same
 * leading dimension).  bf16 planes of up to 1024 pixels run as one launch, everything else as three
 * odamd_maxpool_s1_backward launches. */
int odamd_spp_backward(const void* x, int x_ld, const void* dy5, const void* dy9, const void* dy13, int dy_ld, void* dx,
                       int dx_ld, int batch, int h, int w, int c, int dtype, void* stream);
int odamd_upsample2_backward(const void* dy, int dy_ld, void* dx, int dx_ld, int batch, int h, int w, int c,
                             int dtype, void* stream);

/* Weight gradient of a convolution (autograd of cbs.conv / head.output): dw (cout, cin, k, k) float32
 * in the reference's parameter layout, ADDED to.  x: (B,H,W) NHWC input, dy: NHWC gradient of the
 * raw output.  Workgroups store partial sums to `workspace` (odamd_conv_wgrad_workspace_bytes of the
 * same shape) and a second launch folds them into dw in a fixed order: no atomics, the result is
 * reproducible run to run.  dtype: element type of x and dy.  Both run on v_mfma_f32_32x32x16_bf16 with
 * float32 accumulation and float32 dw; float32 operands are split into hi + lo bf16 parts (three MFMAs
 * per product, ~2^-17 relative), the 3-channel first layer uses an exact float32 im2col tile. */
size_t odamd_conv_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout, int ksize, int stride, int dtype);
int odamd_conv_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int batch, int h, int w, int cin,
                     int cout, int ksize, int stride, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* loss_class.__call__ (train_class.py:410-427) with focal objectness (:463-474), CIoU (:449-460)
 * and BCE / soft-target cross-entropy class loss (:405-408): out4 = {loss, frame, confidence,
 * class}; dpred = d loss / d pred.  screen[i], screen_img[i]: row and image of positive i; label:
 * (n_pos, K).  n_pos == 0 gives frame = class = 1 without gradient, like the reference. */
size_t odamd_loss_workspace_bytes(void);
int odamd_detection_loss(const float* pred, float* dpred, const int* screen, const int* screen_img,
                         const float* label, int batch, int rows, int kdim, int n_pos, float label_smooth,
                         float* out4, void* workspace, size_t workspace_bytes, void* stream);

/* Same with the number of positives read on the device: n_pos_dev[0] entries (clamped to `capacity`,
 * the allocated length of screen / screen_img / label) -- the output of odamd_assign_labels; no host
 * synchronisation between label assignment and loss. */
int odamd_detection_loss_dev(const float* pred, float* dpred, const int* screen, const int* screen_img,
                             const float* label, int batch, int rows, int kdim, const int* n_pos_dev, int capacity,
                             float label_smooth, float* out4, void* workspace, size_t workspace_bytes, void* stream);

/* predict_class._image_process (predict.py:60-65): (h, w, 3) uint8 image on the device ->
 * cv2.resize(image, (size, size)) [8-bit bilinear, OpenCV's fixed-point scheme] -> / 255 -> (1, 3, size, size)
 * float32.  The scheme is restated from OpenCV's documentation and source, which is not available in
 * the build environment: parity with cv2 is unpinned (the test pins it against a numpy restatement and
 * against exact bilinear interpolation within one grey level). */
int odamd_preprocess_u8(const uint8_t* image_hwc, int h, int w, float* out_chw, int size, void* stream);

/* The same for `batch` images of one size stored back to back ((batch, h, w, 3) uint8 -> (batch, 3, size, size)
 * float32) in ONE launch: what a batched predict (predict.py:11 parses --batch, :60-65 processes one image) feeds
 * the network with. */
int odamd_preprocess_u8_batch(const uint8_t* images_hwc, int batch, int h, int w, float* out_chw, int size, void* stream);

/* torch_dataset label -> grid-row assignment (train_class.py:537-543 ranges = decode(-4) .. decode(+4) of
 * every grid row, :578-607 strict containment of the label's cx, cy, w, h) for a whole batch on the
 * device.  geometry: grid sizes, strides and anchors as for odamd_decode (head pointers ignored, kdim =
 * K); sig_lo / sig_hi: sigmoid(-4), sigmoid(+4) as float32; labels: (batch, lmax, K) float32 rows
 * (cx, cy, w, h, conf, classes...), n_label[b] of them valid.  A row matched by several labels is
 * emitted once per label, every copy carrying the last matching label (what the reference's loop does:
 * its conflict branch never runs).  Output in the reference's order (image, then ascending row):
 * screen[i], screen_img[i], label_out (i, K); n_pos[0] = entries written (<= capacity), n_pos[1] =
 * entries found; totals: (batch) ints of scratch. */
int odamd_assign_labels(const odamd_decode_desc* geometry, float sig_lo, float sig_hi, const float* labels,
                        const int* n_label, int lmax, int* screen, int* screen_img, float* label_out, int capacity,
                        int* n_pos, int* totals, void* stream);


/* torch.optim.Adam with coupled L2 (train_class.py:161-162) over flat float32 buffers, fused with
 * model_ema.update (train_class.py:485-493) when ema != NULL.  step counts from 1; grad_scale
 * multiplies the gradient first (1/world_size after a summing all-reduce).  Hyper-parameters are
 * doubles: the reference forms 1-beta and 1-decay in Python doubles before they meet float32 data. */
int odamd_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, size_t n,
                        double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                        double ema_decay, double grad_scale, void* stream);
int odamd_ema_update(float* ema, const float* src, size_t n, double decay, void* stream);

/* Packs master float32 weights into the kernels' layouts for every convolution at once.
 * table_dev: device array of n_entries descriptors of odamd_pack_desc_bytes() bytes each:
 *   { const float* w; void* fwd; void* bwd; int cout, cin, k, cin_phys, cout_phys, n_seg;
 *     int seg_off[8], seg_len[8]; int fwd_f32, fwd_frag, bwd_frag, reserved; }
 * fwd: [cout][ky][kx][cin_phys]; bwd: [cin_phys][2-ky][2-kx][cout_phys] (data-gradient weights), both in
 * `dtype`; fwd_f32 = 1 keeps the forward copy float32 (odamd_conv_first_forward takes float32 weights).
 * fwd_frag / bwd_frag != 0 (bf16 only) write that copy in the fragment order of odamd_conv_weight_layout
 * instead: value = cout_groups | chunk_bytes << 8 as that call returned them for the launch which will read the
 * copy (rows = its output channels, columns = its input channels); the destination must be zero-filled and
 * hold the row count rounded up to 32 * cout_groups. */
int odamd_pack_weights(const void* table_dev, int n_entries, int dtype, void* stream);
size_t odamd_pack_desc_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* ODAMD_H */
