"""Oracle: `ops.process_mask_native(proto[i], pred[:, 6:], pred[:, :4], orig_img.shape[:2])`
as called at object_detectors/yolov8_lmi/model.py:235.  TEST INFRASTRUCTURE ONLY.

ultralytics is not installed (unpinned, tests/dockerfile:12): PARITY UNPINNED
against its binary.  Restated from the 8.3.x semantics in SURVEY.md 8(a) row a7,
with the shared pieces following the in-tree YOLOv5 copy
object_detectors/yolov5_lmi/trt/yolov5_trt.py:373-388 (crop_mask inequality
directions `>= x1, < x2, >= y1, < y2`) and :402-404/:414 (mask GEMM,
bilinear `align_corners=False`).  v8 "native" deltas: logits (no sigmoid);
`scale_masks` crops the letterbox pad in proto space
(`[int(pad_y) : int(mh - pad_y), int(pad_x) : int(mw - pad_x)]`, gain =
min(mh/H0, mw/W0)) and upsamples straight to the ORIGINAL frame (H0, W0); crop to
the already-rescaled boxes; threshold `> 0`.

`process_mask_native` uses torch-CPU itself (matmul + F.interpolate), exactly as
the reference's CPU path would; `process_mask_native_np` is an independent numpy
restatement of the same interpolation arithmetic (aten upsample_bilinear2d,
align_corners=False: scale = in/out, src = scale*(d+0.5)-0.5 clamped at 0) used to
cross-check it.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

F32 = np.float32


def mask_crop_window(mh: int, mw: int, h0: int, w0: int):
    """scale_masks' proto-space window: (top, bottom, left, right)."""
    gain = min(mh / h0, mw / w0)
    pad_x = (mw - w0 * gain) / 2
    pad_y = (mh - h0 * gain) / 2
    top, left = int(pad_y), int(pad_x)
    bottom, right = int(mh - pad_y), int(mw - pad_x)
    return top, bottom, left, right


def crop_indicator(boxes: np.ndarray, h: int, w: int) -> np.ndarray:
    """(n,h,w) bool: the factor `crop_mask` multiplies masks with (yolov5_trt.py:383-388): column index >= x1 and
    < x2, row index >= y1 and < y2, indices and boxes compared in fp32.  Pinned by tests/golden/yolov5_helpers.npz."""
    b = boxes.astype(F32)
    xs = np.arange(w, dtype=F32)[None, None, :]
    ys = np.arange(h, dtype=F32)[None, :, None]
    return (xs >= b[:, 0, None, None]) & (xs < b[:, 2, None, None]) & (ys >= b[:, 1, None, None]) & (ys < b[:, 3, None, None])


def process_mask_native(proto: np.ndarray, coef: np.ndarray, boxes: np.ndarray, shape) -> np.ndarray:
    """proto (32,mh,mw), coef (n,32), boxes (n,4) in original-image pixels, shape (H0,W0)
    -> (n,H0,W0) fp32 of 0./1."""
    h0, w0 = int(shape[0]), int(shape[1])
    n = coef.shape[0]
    if n == 0:
        return np.zeros((0, h0, w0), F32)
    with torch.inference_mode():
        p = torch.from_numpy(np.ascontiguousarray(proto)).float()
        c, mh, mw = p.shape
        logits = (torch.from_numpy(np.ascontiguousarray(coef)).float() @ p.view(c, -1)).view(-1, mh, mw)
        top, bottom, left, right = mask_crop_window(mh, mw, h0, w0)
        logits = logits[:, top:bottom, left:right]
        up = F.interpolate(logits[None], (h0, w0), mode="bilinear", align_corners=False)[0]
        b = torch.from_numpy(np.ascontiguousarray(boxes)).float()
        x1, y1, x2, y2 = torch.chunk(b[:, :, None], 4, 1)
        r = torch.arange(w0, dtype=x1.dtype)[None, None, :]
        cc = torch.arange(h0, dtype=x1.dtype)[None, :, None]
        up = up * ((r >= x1) * (r < x2) * (cc >= y1) * (cc < y2))
        return up.gt_(0.0).numpy()


def bilinear_axis(in_n: int, out_n: int):
    """aten `compute_indices_weights` for linear, align_corners=False, fp32."""
    scale = F32(in_n) / F32(out_n)
    d = np.arange(out_n, dtype=F32)
    src = scale * (d + F32(0.5)) - F32(0.5)
    src = np.maximum(src, F32(0))
    i0 = src.astype(np.int64)
    i1 = i0 + (i0 < in_n - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    l0 = (F32(1) - l1).astype(F32)
    return i0, i1, l0, l1


def mask_logits_upsampled_np(proto, coef, shape) -> np.ndarray:
    """Independent numpy restatement: (n,H0,W0) fp32 upsampled logits before crop/threshold."""
    h0, w0 = int(shape[0]), int(shape[1])
    c, mh, mw = proto.shape
    logits = (coef.astype(F32) @ proto.reshape(c, -1).astype(F32)).reshape(-1, mh, mw)
    top, bottom, left, right = mask_crop_window(mh, mw, h0, w0)
    logits = logits[:, top:bottom, left:right]
    y0, y1, wy0, wy1 = bilinear_axis(logits.shape[1], h0)
    x0, x1, wx0, wx1 = bilinear_axis(logits.shape[2], w0)
    r0 = logits[:, y0][:, :, x0] * wx0 + logits[:, y0][:, :, x1] * wx1
    r1 = logits[:, y1][:, :, x0] * wx0 + logits[:, y1][:, :, x1] * wx1
    return r0 * wy0[None, :, None] + r1 * wy1[None, :, None]


def process_mask_native_np(proto, coef, boxes, shape) -> np.ndarray:
    h0, w0 = int(shape[0]), int(shape[1])
    if coef.shape[0] == 0:
        return np.zeros((0, h0, w0), F32)
    up = mask_logits_upsampled_np(proto, coef, shape)
    b = boxes.astype(F32)
    xs = np.arange(w0, dtype=F32)[None, None, :]
    ys = np.arange(h0, dtype=F32)[None, :, None]
    inside = (xs >= b[:, 0, None, None]) & (xs < b[:, 2, None, None]) & (ys >= b[:, 1, None, None]) & (ys < b[:, 3, None, None])
    return ((up * inside) > 0).astype(F32)


def mask_iou(a: np.ndarray, b: np.ndarray) -> float:
    """IoU of two boolean mask stacks pooled over all pixels (1.0 when both are empty)."""
    a = a.astype(bool)
    b = b.astype(bool)
    union = np.logical_or(a, b).sum()
    if union == 0:
        return 1.0
    return float(np.logical_and(a, b).sum() / union)
