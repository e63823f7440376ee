/*
 * lmi_hip.h -- C ABI of liblmi_hip.so: the MI355X (gfx950) kernels for the YOLOv8
 * pre/post-processing path that LMI's `Yolov8` wrapper runs around the network.
 *
 * The reference exposes a Python class API only (object_detectors/od_base.py:6-37,
 * object_detectors/yolov8_lmi/model.py:42-324); it has no FFI.  Each entry point
 * below replaces the Python/torch/cv2/ultralytics call sequence cited in its comment
 * (paths relative to the reference repository root).  INTEGRATION.md shows the
 * ctypes stubs a maintainer would add to the reference wrapper.
 *
 * Conventions
 *  - Every pointer marked [device] is a device pointer (e.g. torch `tensor.data_ptr()`),
 *    every buffer is caller-allocated; the library allocates nothing and keeps no
 *    per-call global state, so calls on different streams / processes are independent.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *    enqueue work; nothing synchronises the host.
 *  - Return value: 0 = LMI_OK, negative = error (see lmi_status_string); a message for
 *    the last error on the calling thread is available from lmi_last_error().
 *  - Arithmetic is compiled with -ffp-contract=off and IEEE-correct fp32 division so
 *    results are bit-reproducible against the CPU oracle (`oracle/`).
 */
#ifndef LMI_HIP_H
#define LMI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMI_VERSION 102 /* 0.1.2: letterbox plan / execute flags (0.1.1: lmi_finalize_args grew the keypoint fields) */

enum lmi_status {
  LMI_OK = 0,
  LMI_ERR_BAD_ARG = -1,     /* null pointer, non-positive size, unsupported combination */
  LMI_ERR_ALIGNMENT = -2,   /* pointer / stride alignment requirement not met */
  LMI_ERR_WORKSPACE = -3,   /* workspace too small (see *_workspace_bytes) */
  LMI_ERR_LIMIT = -4,       /* size beyond a compiled-in limit (e.g. max_det > LMI_MAX_DET_CAP) */
  LMI_ERR_LAUNCH = -5,      /* HIP launch / runtime error (message in lmi_last_error) */
  LMI_ERR_NO_DEVICE = -6
};

enum lmi_dtype { LMI_F32 = 0, LMI_F16 = 1, LMI_U8 = 2 };

/* resampler of the letterbox entry: the fixed-point cv2.INTER_LINEAR (default) */
enum lmi_resize_mode { LMI_RESIZE_CV2_LINEAR = 0, LMI_RESIZE_TORCH_BILINEAR = 1, LMI_RESIZE_TORCH_BILINEAR_1T = 2 };
/* OR-ed into `mode`: the source frames are single grey planes (uint8 HW, one byte per pixel) */
#define LMI_SRC_GRAY 0x100
/* OR-ed into `mode` (plan / execute, r03): the coefficient tables in `workspace` depend on the sizes alone
 * (source size, content size, pads, destination size, resampler) -- not on the pixels, the channel order, the pad
 * value or the output dtype.  LMI_LB_TABLES_ONLY builds them and returns (src / dst may be NULL; the frame
 * descriptors of lmi_letterbox_u8 are still read); LMI_LB_TABLES_READY says `workspace` still holds the tables of a
 * call with the same sizes and resampler and leaves the table pass out: one launch per call instead of two. */
#define LMI_LB_TABLES_ONLY 0x200
#define LMI_LB_TABLES_READY 0x400

#define LMI_NMS_SWEEP_ONLY 0x100 /* OR-ed into the `rotated` argument of lmi_nms_v8_ex / lmi_nms_suppress_ex */

#define LMI_MAX_DET_CAP 1024   /* largest max_det lmi_nms_v8 accepts */
#define LMI_MAX_MASK_DIM 32    /* number of prototype channels (nm) the mask kernels are built for */
#define LMI_MAX_POINT_OPS 16   /* longest operator list lmi_revert_points_f32 accepts */
#define LMI_NMS_STATIC_BYTES (4 << 20) /* built-in workspace of lmi_nms_v8_static, per GPU */

int lmi_version(void);
const char *lmi_status_string(int status);
const char *lmi_last_error(void);
/* number of visible HIP devices, or a negative lmi_status */
int lmi_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Preprocess -- replaces `Yolov8.preprocess` (object_detectors/yolov8_lmi/model.py:112-134):
 * one uint8 image (H,W,C) with C in {1,3} (a 2-D image is C=1), HWC -> (1,3,H,W), cast,
 * `/255` by IEEE division (fp16: the fp32 quotient rounded to fp16, as torch-CPU computes it).
 * No resize, pad or channel swap.
 *   src  [device] uint8, row_stride bytes between rows (>= W*C)
 *   dst  [device] (3,H,W) of dst_dtype (LMI_F32 | LMI_F16), contiguous
 */
int lmi_preprocess_u8(const void *src, int h, int w, int c, int64_t row_stride,
                      void *dst, int dst_dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * Letterbox geometry (host only, no GPU): aspect-preserving target size and centred pad split
 * used by lmi_letterbox_u8*.  r = min(dst_h/h, dst_w/w); new = (round(w r), round(h r));
 * pads follow ultralytics LetterBox `round(d/2 -+ 0.1)`, which equals the reference helpers'
 * `pad_L = (W-w)//2` (lmi_utils/gadget_utils/pipeline_utils.py:110-131, :152-172) and is the
 * convention `ops.scale_boxes` (model.py:222) inverts.  stretch != 0 gives the callers'
 * `cv2.resize(im,(w,h))` stretch (object_detectors/yolov8_lmi/run_model.py:83-87): no pad.
 *   out6 = {new_w, new_h, pad_left, pad_right, pad_top, pad_bottom}
 */
int lmi_letterbox_geometry(int h, int w, int dst_h, int dst_w, int stretch, int32_t out6[6]);

/* one source frame of a ragged batch; the array lives in device memory */
typedef struct lmi_frame {
  uint64_t src;        /* [device] uint8 HWC, 3 channels (one with LMI_SRC_GRAY); any byte alignment */
  int32_t h, w;        /* source size */
  int32_t row_stride;  /* bytes between source rows, >= 3*w (>= w with LMI_SRC_GRAY) */
  int32_t new_w, new_h;/* resized content size (lmi_letterbox_geometry) */
  int32_t pad_l, pad_t;/* where the content starts in the destination */
  int32_t reserved;
} lmi_frame;

/* Fused letterbox -- replaces the callers' chain around preprocess:
 *   cv2 BGR->RGB            object_detectors/yolov8_lmi/run_model.py:79, model.py:151-152
 *   cv2.resize INTER_LINEAR object_detectors/yolov8_lmi/run_model.py:83-87,
 *                           tests/object_detectors/yolov8_lmi/test_yolov8_model.py:84
 *   centred constant pad    lmi_utils/gadget_utils/pipeline_utils.py:145-173 (fit_array_to_size)
 *   HWC->NCHW, cast, /255   object_detectors/yolov8_lmi/model.py:128-133
 * for a batch of uint8 HWC 3-channel frames -> dst (B,3,dst_h,dst_w) fp32|fp16.  The resampler
 * reproduces OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic bit for bit (11-bit
 * coefficients, see oracle/cv2_resize.py).
 *   frames_host [host]   lmi_frame[batch]: validated (alignment, strides, geometry) and used to
 *                        size the launch; never dereferenced on the device
 *   frames_dev  [device] the same array in device memory, read by the kernels
 *   swap_rb != 0 reverses the channel order (BGR<->RGB); pad_value in 0..255.
 *   workspace   [device] >= lmi_letterbox_workspace_bytes(batch, dst_h, dst_w) bytes, 256-byte
 *               aligned: the per-frame coefficient tables, rebuilt by every call (unless mode carries
 *               LMI_LB_TABLES_READY, above)
 * mode = LMI_RESIZE_TORCH_BILINEAR swaps the resampler for the reference's own helper flavour
 *   `resize_image` + `fit_im_to_size`   lmi_utils/gadget_utils/pipeline_utils.py:23-142, as chained by
 *                                        object_detectors/ultralytics_lmi/yolo/run_model.py:87-100
 * i.e. F.interpolate(bilinear, align_corners=False) on the fp32 cast of the frame, truncating cast
 * back to uint8, then the centred pad -- or centred CROP: in this mode pad_l / pad_t may be negative
 * and the content may overhang the destination.  Bit-exact against aten's CPU kernels of torch 2.10
 * on >= 2 threads (aten computes differently on exactly one thread; tests/golden/letterbox_torch.npz,
 * oracle/geometry.py; aten switches between two kernels at new_h + new_w = 128, so does this one, per
 * frame).  LMI_RESIZE_TORCH_BILINEAR_1T is what the same helpers compute when torch runs on exactly ONE thread
 * (torch.set_num_threads(1), OMP_NUM_THREADS=1): aten then takes its 4-weight kernel for every 3-channel frame.
 * workspace: as for the cv2 mode -- the tables then hold aten's indices and fp32 weights and the same LDS-staged kernel
 * runs (as fast as the cv2 mode); NULL or too small a workspace selects a table-free kernel that reads its taps straight
 * from global memory (2-3 x slower, same result).
 * mode | LMI_SRC_GRAY (either resampler, both entries): every frame is ONE grey plane.  The callers turn such an image
 * into three equal channels before they resize it -- cv2.cvtColor(im0, COLOR_GRAY2BGR),
 * object_detectors/yolov8_lmi/run_model.py:77-78 and ultralytics_lmi/yolo/run_model.py:78-79; preprocess itself
 * expands a 1-channel input, model.py:123-126 -- and both resamplers treat channels independently, so the plane is
 * resampled once and the value written to the three output planes: the same tensor, a third of the source bytes.
 */
int64_t lmi_letterbox_workspace_bytes(int table_sets, int dst_h, int dst_w);
int lmi_letterbox_u8(const lmi_frame *frames_host, const lmi_frame *frames_dev, int batch,
                     void *dst, int dst_h, int dst_w, int dst_dtype, int pad_value, int swap_rb,
                     int mode, void *workspace, int64_t workspace_bytes, void *stream);

/* Same, for a batch of equally sized frames stored with a constant image stride (any number of bytes that
 * covers a frame: a packed batch of 375x500 frames, 562500 bytes apart, is read in place): no
 * descriptor array, geometry is derived from (h,w,dst_h,dst_w,stretch) on the host; one table
 * set: workspace >= lmi_letterbox_workspace_bytes(1, dst_h, dst_w).  mode must be LMI_RESIZE_CV2_LINEAR: the
 * geometry derived here is cv2's / ultralytics' (round(), centred pad), not the reference helpers' (int()
 * truncation) that the torch flavours go with -- those take lmi_letterbox_u8 (LMI_ERR_BAD_ARG otherwise). */
int lmi_letterbox_u8_uniform(const void *src, int batch, int h, int w, int64_t row_stride,
                             int64_t image_stride, void *dst, int dst_h, int dst_w,
                             int dst_dtype, int pad_value, int swap_rb, int stretch, int mode,
                             void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * NMS -- replaces `ops.non_max_suppression(preds, conf2, iou, agnostic=, max_det=, nc=)`
 * (object_detectors/yolov8_lmi/model.py:212; in-tree YOLOv5 analogue
 * object_detectors/yolov5_lmi/trt/yolov5_trt.py:172-282) including the torchvision greedy NMS
 * it calls (:267).  Semantics: SURVEY.md 8(a) row a4 / oracle/nms.py.
 *   pred        [device] fp32 (batch, 4+nc+nm, anchors), contiguous (the head output)
 *   conf_thres  compared in fp32, strictly (`>`)
 *   iou_thres   compared as torchvision does: fp32 IoU promoted to double, strictly
 *   max_wh      class offset (7680 upstream); ignored when agnostic != 0
 *   max_nms     candidates kept for NMS by score (30000 upstream)
 *   out_det     [device] fp32 (batch, max_det, 6+nm): x1,y1,x2,y2,conf,cls,coef..., score order;
 *               rows >= out_count[b] are left untouched
 *   out_count   [device] int32 (batch)
 *   out_anchor  [device] int32 (batch, max_det): anchor index of every kept row (parity checks)
 *   workspace   [device] >= lmi_nms_workspace_bytes(batch, anchors) bytes, 256-byte aligned
 */
int64_t lmi_nms_workspace_bytes(int batch, int anchors);
int lmi_nms_v8(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres,
               double iou_thres, int agnostic, int max_det, int max_nms, float max_wh,
               float *out_det, int32_t *out_count, int32_t *out_anchor, void *workspace,
               int64_t workspace_bytes, void *stream);
/* The two halves of lmi_nms_v8 / lmi_nms_v8_rotated (same reference lines), exported so that a caller can
 * schedule between them: lmi_nms_decode is the candidate filter (HBM-bound, reads the head once into
 * `workspace`); lmi_nms_suppress is sort + suppression from `workspace` (one workgroup per image, latency-bound,
 * next to no HBM traffic -- it shares the chip well with an HBM-bound kernel on another stream, e.g. the next
 * batch's letterbox started on an event recorded between the two calls; bench.py does that).  Same arguments
 * and workspace as lmi_nms_v8; rotated != 0 selects the oriented-box suppression.
 */
int lmi_nms_decode(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres,
                   void *workspace, int64_t workspace_bytes, void *stream);
int lmi_nms_suppress(const float *pred, int batch, int nc, int nm, int anchors, double iou_thres,
                     int agnostic, int max_det, int max_nms, float max_wh, int rotated,
                     float *out_det, int32_t *out_count, int32_t *out_anchor, void *workspace,
                     int64_t workspace_bytes, void *stream);

/* The general form of the three calls above ("_ex"): same reference lines, same kernels, plus
 *   pred_dtype  LMI_F32 or LMI_F16: the head of an `fp16=True` model (object_detectors/yolov8_lmi/model.py:68,132)
 *               is read as it is; every value is promoted to fp32 (exactly) and all comparisons and box arithmetic
 *               run in fp32 -- the result of `pred.float()` + the fp32 call, without that extra pass over the head.
 *               (The reference would run ultralytics' NMS on the fp16 tensor itself, where `cls * 7680` overflows to
 *               inf from class 9 on and the score threshold is rounded to fp16: deliberately not reproduced.)
 *   rotated     low byte != 0: oriented boxes (lmi_nms_v8_rotated); OR LMI_NMS_SWEEP_ONLY to keep the one-workgroup
 *               sorted sweep also where the suppressor-list kernel would run first (agnostic NMS or nc <= 8, and a
 *               batch that leaves every image >= 16 CUs: same result, for measurement and tests)
 *   fin         NULL, or the arguments of lmi_finalize_detections (below): `ops.scale_boxes` + clip, the per-class
 *               threshold and the survivors' dense ranks (object_detectors/yolov8_lmi/model.py:222-232) then run in
 *               the suppression kernel's epilogue -- rows leave in original-image space and keep / keep_count are
 *               written by the same launch; when mask_slot / slot_base are asked for, one small launch turns the
 *               per-image ranks into dense global slots (without slots, e.g. detection models: no extra launch).
 *               Results are identical to lmi_nms_v8 followed by lmi_finalize_detections.  Not with rotated != 0.
 */
typedef struct lmi_finalize_args {
  int32_t model_h, model_w;        /* network input size (img.shape[2:]) */
  int32_t uniform_h0, uniform_w0;  /* original size of every image when orig_hw is NULL */
  const int32_t *orig_hw;          /* [device] int32 (batch,2) = (H0,W0) per image, or NULL */
  int32_t pad_round;               /* 1 = ultralytics 8.3.x round(pad - 0.1); 0 = un-rounded pad */
  int32_t reserved;
  const double *thres;             /* [device] double (nc) per-class thresholds, or NULL -> uniform_thres */
  double uniform_thres;
  uint8_t *keep;                   /* [device] uint8 (batch, max_det) */
  int32_t *keep_count;             /* [device] int32 (batch) */
  int32_t *mask_slot;              /* [device] int32 (batch, max_det), may be NULL */
  int32_t *slot_map;               /* [device] int32 (batch*max_det), may be NULL */
  int32_t *slot_base;              /* [device] int32 (batch+1), may be NULL */
  /* Keypoint models (`Yolov8Pose.postprocess`, object_detectors/yolov8_lmi/model.py:568-569): kpt_n > 0 also runs
   * `ops.scale_coords` -- lmi_scale_coords, same arithmetic -- on the row columns kpt_col0 .. kpt_col0 + kpt_n * kpt_ndim
   * (kpt_ndim 2 or 3: x, y [, visibility]) while the rows are written.  0: none. */
  int32_t kpt_col0, kpt_n, kpt_ndim;
  int32_t reserved2;
} lmi_finalize_args;
int lmi_nms_v8_ex(const void *pred, int pred_dtype, int batch, int nc, int nm, int anchors, float conf_thres,
                  double iou_thres, int agnostic, int max_det, int max_nms, float max_wh, int rotated,
                  const lmi_finalize_args *fin, float *out_det, int32_t *out_count, int32_t *out_anchor,
                  void *workspace, int64_t workspace_bytes, void *stream);
int lmi_nms_decode_ex(const void *pred, int pred_dtype, int batch, int nc, int nm, int anchors, float conf_thres,
                      void *workspace, int64_t workspace_bytes, void *stream);
int lmi_nms_suppress_ex(const void *pred, int pred_dtype, int batch, int nc, int nm, int anchors, double iou_thres,
                        int agnostic, int max_det, int max_nms, float max_wh, int rotated,
                        const lmi_finalize_args *fin, float *out_det, int32_t *out_count, int32_t *out_anchor,
                        void *workspace, int64_t workspace_bytes, void *stream);

/* lmi_nms_v8 without a workspace argument, for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the reference: same
 * arguments, kernels and results as lmi_nms_v8, but the workspace is a 4 MiB (LMI_NMS_STATIC_BYTES) device array
 * built into the library -- every GPU the library runs on holds its own copy until the process ends -- and the
 * call makes no hipMalloc / hipFree.  Sizes with lmi_nms_workspace_bytes(batch, anchors) > LMI_NMS_STATIC_BYTES
 * (e.g. more than 11 images of 8400 anchors, or more than 2 of 33600) are rejected with LMI_ERR_LIMIT.  The
 * workspace of the CURRENT device is used: call with the device current that owns `stream` and the buffers.  All
 * calls on one GPU share that one array: calls that overlap on a GPU (two streams, two threads) race -- serialise
 * them, or use lmi_nms_v8 with separate workspaces.
 */
int lmi_nms_v8_static(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres,
                      double iou_thres, int agnostic, int max_det, int max_nms, float max_wh,
                      float *out_det, int32_t *out_count, int32_t *out_anchor, void *stream);

/* ------------------------------------------------------------------------------------------
 * Finalize -- replaces, per image, `ops.scale_boxes(img.shape[2:], pred[:, :4], orig_img.shape)`
 * + clip (object_detectors/yolov8_lmi/model.py:222; in-tree form yolov5_trt.py:316-341) IN
 * PLACE on det[..., :4], and the per-class confidence filter `M = confs > thres`
 * (model.py:223-232, comparison in float64, strict).  It also ranks the survivors so the mask
 * kernels can write them densely.
 *   det          [device] fp32 (batch, max_det, det_stride) from lmi_nms_v8 (det_stride = 6+nm)
 *   count        [device] int32 (batch)
 *   model_h/w    network input size (img.shape[2:])
 *   orig_hw      [device] int32 (batch,2) = (H0,W0) per image, or NULL -> every image is
 *                (uniform_h0, uniform_w0)
 *   pad_round    1 = ultralytics 8.3.x `round(pad - 0.1)`; 0 = un-rounded pad (YOLOv5 in-tree form)
 *   thres        [device] double (nc) per-class thresholds, or NULL -> uniform_thres for all
 *   keep         [device] uint8 (batch, max_det): 1 where M is true
 *   keep_count   [device] int32 (batch): m_b = number of survivors
 *   mask_slot    [device] int32 (batch, max_det): dense global rank of each survivor
 *                (image-major, score order), -1 for dropped rows; may be NULL
 *   slot_map     [device] int32 (batch*max_det): slot -> b*max_det + i ; may be NULL
 *   slot_base    [device] int32 (batch+1): first slot of every image, [batch] = number of
 *                survivors in the whole batch; may be NULL
 */
int lmi_finalize_detections(float *det, const int32_t *count, int batch, int max_det,
                            int det_stride, int nc, int model_h, int model_w,
                            const int32_t *orig_hw, int uniform_h0, int uniform_w0,
                            int pad_round, const double *thres, double uniform_thres,
                            uint8_t *keep, int32_t *keep_count, int32_t *mask_slot,
                            int32_t *slot_map, int32_t *slot_base, void *stream);

/* ------------------------------------------------------------------------------------------
 * Oriented boxes -- replaces, in `Yolov8Obb.postprocess`
 * (object_detectors/yolov8_lmi/model.py:398-465):
 *   lmi_nms_v8_rotated  `ops.non_max_suppression(preds, conf2, iou, agnostic=, max_det=, nc=, rotated=True)` :423.
 *     Same arguments as lmi_nms_v8; pred is (batch, 4+nc+nm, anchors) with xywh boxes and the angle as
 *     the LAST of the nm >= 1 extra channels.  Rows come out as [x, y, w, h, conf, cls, extras...]
 *     (no xyxy conversion).  Suppression is ultralytics' `nms_rotated`: a candidate is dropped when its
 *     probiou with ANY higher-scored candidate (kept or not) is >= iou_thres -- not a greedy pass.
 *     Limit: min(anchors, max_nms) <= 65536 candidates per image (LMI_ERR_LIMIT beyond).
 *   lmi_finalize_obb  `regularize_rboxes` :436, `scale_boxes(..., xywh=True)` :438, `xywhr2xyxyxyxy` :446
 *     and `M = confs > thres` :448-455.  det is NOT modified; corners [device] fp32 (batch, max_det, 4, 2)
 *     receive the four corner points of rows i < count[b]; keep / keep_count / thres as in
 *     lmi_finalize_detections.  The angle is read from column det_stride-1.
 */
int lmi_nms_v8_rotated(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres,
                       double iou_thres, int agnostic, int max_det, int max_nms, float max_wh,
                       float *out_det, int32_t *out_count, int32_t *out_anchor, void *workspace,
                       int64_t workspace_bytes, void *stream);
int lmi_finalize_obb(const float *det, const int32_t *count, int batch, int max_det, int det_stride,
                     int nc, int model_h, int model_w, const int32_t *orig_hw, int uniform_h0,
                     int uniform_w0, const double *thres, double uniform_thres, float *corners,
                     uint8_t *keep, int32_t *keep_count, void *stream);

/* ------------------------------------------------------------------------------------------
 * Keypoints -- replaces `ops.scale_coords(img.shape[2:], pred_kpts,
 * orig_img.shape)` of `Yolov8Pose.postprocess` (object_detectors/yolov8_lmi/model.py:568-569; in-tree
 * analogue scale_segments yolov5_trt.py:354-370) on the keypoint columns of the NMS rows, in place:
 * x -= pad_x, y -= pad_y with gain/pad as Python floats and the pads NOT rounded (unlike scale_boxes),
 * then /= gain and clip to [0,w0] x [0,h0].  Columns col0 + k*ndim + {0,1} of row i hold keypoint k's
 * (x, y); a third value (visibility, ndim == 3) is left alone.  Call after lmi_nms_v8 (nm = nkpt*ndim,
 * col0 = 6); boxes are handled by lmi_finalize_detections as for detection.
 */
int lmi_scale_coords(float *det, const int32_t *count, int batch, int max_det, int det_stride,
                     int col0, int nkpt, int ndim, int model_h, int model_w, const int32_t *orig_hw,
                     int uniform_h0, int uniform_w0, void *stream);

/* ------------------------------------------------------------------------------------------
 * Masks -- replaces `ops.process_mask_native(proto[i], pred[:, 6:], pred[:, :4],
 * orig_img.shape[:2])` + `masks[M]` (object_detectors/yolov8_lmi/model.py:235-236; in-tree
 * YOLOv5 analogue yolov5_trt.py:373-415): logits = coef @ proto as a k-ordered fp32 fma chain
 * (only the rows under each survivor's box; dense fp32 MFMA tiles for odd prototype widths), then fused
 * letterbox-pad crop -> bilinear upsample (align_corners=False) to the ORIGINAL frame -> crop to
 * the rescaled box -> `> 0`, written once as uint8 (or fp32 0./1.).
 *   proto      [device] fp32 (batch, 32, mh, mw)
 *   det, slot_map, slot_base: as produced by lmi_finalize_detections (boxes already rescaled)
 *   out_masks  [device] survivors' masks written back to back in slot order; with uniform
 *              originals this is an array (capacity_masks, H0, W0) of out_dtype (LMI_U8|LMI_F32)
 *   capacity_masks  number of mask planes out_masks can hold; slots beyond it are skipped and
 *              their number is written to overflow[0] ([device] int32, may be NULL)
 *   workspace  [device] >= lmi_mask_workspace_bytes(capacity_masks, mh, mw), 256-byte aligned
 * These three entries serve uniform originals (every image H0 x W0); the _ex forms above take a table of sizes.
 * lmi_process_mask = lmi_mask_logits (the coefficient x prototype product, and one box record per slot, into
 * `workspace`) followed by lmi_mask_upsample (crop + bilinear + box crop + threshold from the `workspace`
 * lmi_mask_logits left, same arguments); the two halves are exported so a caller can time or schedule
 * them separately.
 */
/* General forms ("_ex") of the three calls below: same reference lines, plus
 *   images_host / images_dev  NULL, or one lmi_mask_image per image, the same array in host memory (validated, used to
 *               size the launch) and in device memory (read by the kernels): a batch of DIFFERENTLY sized originals
 *               -- `Yolov8.postprocess` takes a list of them, object_detectors/yolov8_lmi/model.py:215-236 -- in one
 *               launch.  h0 / w0 are then ignored; the masks of image b are written back to back from element
 *               out_offset (h0*w0 elements each, score order), wherever the caller laid them out in out_masks
 *               (out_elems = elements out_masks holds).  The caller knows the survivor counts at that point (it
 *               sized the buffer from them): n_masks must equal slot_base[b+1] - slot_base[b].
 *   flags       LMI_MASK_LOGITS_MFMA: the coefficient x prototype product as dense fp32 MFMA tiles for every shape;
 *               LMI_MASK_UPSAMPLE_SINGLE: the one-pass upsample kernel for every shape (both: measurement);
 *               LMI_MASK_PROTO_F16: `proto` holds fp16 values, the prototypes of an `fp16=True` model
 *               (object_detectors/yolov8_lmi/model.py:68,132).  process_mask_native multiplies `protos.float()`: every
 *               value is promoted to fp32 as it is loaded (exact), so the masks equal those of the converted tensor
 *               without the conversion pass.  Even prototype widths only (the row kernel); LMI_ERR_BAD_ARG otherwise.
 */
typedef struct lmi_mask_image {
  int32_t h0, w0;      /* original frame size of this image */
  int32_t n_masks;     /* survivors of this image */
  int32_t reserved;
  int64_t out_offset;  /* element offset of the image's first mask in out_masks */
} lmi_mask_image;
enum lmi_mask_flags { LMI_MASK_LOGITS_MFMA = 1, LMI_MASK_UPSAMPLE_SINGLE = 2, LMI_MASK_PROTO_F16 = 4 };
int lmi_mask_logits_ex(const void *proto, int batch, int nm, int mh, int mw, const float *det, int max_det,
                       int det_stride, const int32_t *slot_map, const int32_t *slot_base, int h0, int w0,
                       const lmi_mask_image *images_host, const lmi_mask_image *images_dev,
                       int64_t capacity_masks, void *workspace, int64_t workspace_bytes, int flags, void *stream);
int lmi_mask_upsample_ex(void *workspace, int batch, int nm, int mh, int mw, const float *det, int max_det,
                         int det_stride, const int32_t *slot_map, const int32_t *slot_base, int h0, int w0,
                         const lmi_mask_image *images_host, const lmi_mask_image *images_dev, void *out_masks,
                         int out_dtype, int64_t capacity_masks, int64_t out_elems, int32_t *overflow, int flags,
                         void *stream);
int lmi_process_mask_ex(const void *proto, int batch, int nm, int mh, int mw, const float *det, int max_det,
                        int det_stride, const int32_t *slot_map, const int32_t *slot_base, int h0, int w0,
                        const lmi_mask_image *images_host, const lmi_mask_image *images_dev, void *out_masks,
                        int out_dtype, int64_t capacity_masks, int64_t out_elems, int32_t *overflow,
                        void *workspace, int64_t workspace_bytes, int flags, void *stream);
int64_t lmi_mask_workspace_bytes(int64_t capacity_masks, int mh, int mw);
int lmi_mask_logits(const float *proto, int batch, int nm, int mh, int mw, const float *det,
                    int max_det, int det_stride, const int32_t *slot_map, const int32_t *slot_base,
                    int h0, int w0, int64_t capacity_masks, void *workspace,
                    int64_t workspace_bytes, void *stream);
int lmi_mask_upsample(void *workspace, int batch, int nm, int mh, int mw, const float *det,
                      int max_det, int det_stride, const int32_t *slot_map,
                      const int32_t *slot_base, int h0, int w0, void *out_masks, int out_dtype,
                      int64_t capacity_masks, int32_t *overflow, void *stream);
int lmi_process_mask(const float *proto, int batch, int nm, int mh, int mw, const float *det,
                     int max_det, int det_stride, const int32_t *slot_map,
                     const int32_t *slot_base, int h0, int w0, void *out_masks, int out_dtype,
                     int64_t capacity_masks, int32_t *overflow, void *workspace,
                     int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Operator-list revert on the device, batched -- replaces the per-mask and
 * per-segment Python loops around `pipeline_utils.revert_to_origin`
 * (lmi_utils/gadget_utils/pipeline_utils.py:422-476) and `pipeline_utils.revert_masks_to_origin`
 * (:374-418 -> resize_image :23-68, fit_im_to_size :72-142) in `Yolov8.predict`
 * (object_detectors/yolov8_lmi/model.py:308-318).
 *
 * lmi_revert_points_f32: pts/out [device] fp32 (n_pairs, 2) (x, y) pairs -- an (N,4) box array is
 *   2N pairs.  `ops` [host] lists the steps in RECORDED order; they are undone last to first in
 *   fp32 (IEEE division), then rounded half-to-even and clamped at 0, exactly `revert_to_origin`:
 *     {'resize':[tw,th,ow,oh]} -> {LMI_POINT_DIV, (float)(tw/ow), (float)(th/oh)}   (ratio in double first)
 *     {'stretch':[sx,sy]}      -> {LMI_POINT_DIV, sx, sy}
 *     {'pad':[L,R,T,B]}        -> {LMI_POINT_SUB, L, T}
 *     {'flip':[lr,ud,w,h]}     -> {LMI_POINT_FLIP_X | _Y | _XY, w, h}
 *   pts == out is allowed.
 * lmi_revert_masks_stage: one fused step of `revert_mask_to_origin` on a stack of m planes
 *   (m, src_h, src_w) of dtype LMI_F32 or LMI_U8:
 *     1. fit_h/fit_w > 0: centred crop / zero pad to (fit_h, fit_w)  (fit_im_to_size; 0 = keep)
 *     2. resize_h/resize_w > 0: bilinear resize, align_corners=False (resize_image; integer masks are
 *        interpolated in fp32 and truncated; 0 = none)
 *     3. flip_lr / flip_ud
 *   dst [device] (m, out_h, out_w), out = resize size if given, else the fit size; dst != src.
 *   A reverted 'pad' operator [L,R,T,B] is step 1 with (fit_w, fit_h) = (w-L-R, h-T-B); a reverted
 *   'resize' [tw,th,ow,oh] is step 2 with (ow, oh).  Longer lists are chains of stages (see
 *   object_detectors_amd/hip_ops.py:revert_masks).
 */
enum lmi_point_kind { LMI_POINT_DIV = 1, LMI_POINT_SUB = 2, LMI_POINT_FLIP_X = 3, LMI_POINT_FLIP_Y = 4, LMI_POINT_FLIP_XY = 5 };
typedef struct lmi_point_op {
  int32_t kind; /* enum lmi_point_kind */
  float x, y;
} lmi_point_op;
int lmi_revert_points_f32(const float *pts, int64_t n_pairs, const lmi_point_op *ops, int n_ops,
                          float *out, void *stream);
int lmi_revert_masks_stage(const void *src, int dtype, int64_t m, int src_h, int src_w, int fit_h,
                           int fit_w, int resize_h, int resize_w, int flip_lr, int flip_ud,
                           void *dst, void *stream);

/* ------------------------------------------------------------------------------------------
 * Mask paste -- replaces `rescale_masks(masks, boxes, image_shape, threshold)`
 * (lmi_utils/postprocess_utils/mask_utils.py:66-140, used by detectron2_lmi/model.py:257,489): N soft
 * masks (n, mask_h, mask_w) fp32 are sampled into their boxes (x0,y0,x1,y1 = first four floats of each
 * `box_stride`-float row) over the whole (img_h, img_w) image with aten's bilinear grid_sample
 * (align_corners=False, zero padding) and thresholded with `>=`; out [device] uint8 (n, img_h, img_w) of
 * 0/1 (the reference returns bool).  threshold < 0 selects the reference's visualisation mode (:131-133): the
 * sampled value times 255, truncated like `.to(torch.uint8)`, 0 outside the masks.  No (n, H, W) fp32
 * intermediate: the constant outside the boxes is written at the memset rate, then the chunks under each box.
 */
int lmi_paste_masks(const float *masks, int64_t n, int mask_h, int mask_w, const float *boxes,
                    int box_stride, int img_h, int img_w, float threshold, uint8_t *out,
                    void *stream);

/* ------------------------------------------------------------------------------------------
 * Segments -- replaces `ops.masks2segments(masks)`
 * (object_detectors/yolov8_lmi/model.py:238-243; yolov5_trt.py:418-431: cv2.findContours with
 * RETR_EXTERNAL, CHAIN_APPROX_SIMPLE on masks copied to the host).
 * lmi_pack_mask_boxes [device]: masks (m, img_h, img_w) of LMI_U8 or LMI_F32; boxes int32 (m,4)
 *   x0,y0,x1,y1 half-open pixel ranges inside the image (a mask must be zero outside its range);
 *   writes, for mask i, (y1-y0) rows of ceil((x1-x0)/64) 64-bit words (bit c of word w = pixel
 *   x0 + 64w + c != 0) starting at out_words[word_offset[i]].  All pointers are device pointers.
 * lmi_trace_contours_host [host, no GPU work]: border following (Suzuki-Abe outer borders of the
 *   hole-filled 8-connected components, collinear runs collapsed) on the packed rows copied to the
 *   host, `threads` worker threads over the masks.  strategy_all != 0: every external contour of a
 *   mask back to back; 0: the longest one.  Points (x, y) in image coordinates go to out_xy
 *   (capacity_points pairs), mask after mask, points_per_mask[i] of them for mask i.  Returns the
 *   total number of points; if that exceeds capacity_points nothing is written (call again with a
 *   larger buffer); negative = error status.
 */
int lmi_pack_mask_boxes(const void *masks, int dtype, int64_t m, int img_h, int img_w,
                        const int32_t *boxes, const int64_t *word_offset, uint64_t *out_words,
                        void *stream);
int64_t lmi_trace_contours_host(const uint64_t *words, int64_t m, const int32_t *boxes,
                                const int64_t *word_offset, int strategy_all, int threads,
                                int32_t *out_xy, int64_t capacity_points, int64_t *points_per_mask);

#ifdef __cplusplus
}
#endif
#endif /* LMI_HIP_H */
