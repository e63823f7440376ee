"""Oracle: OpenCV `cv2.resize(..., interpolation=INTER_LINEAR)` for uint8 images,
plus the letterbox composition built on it.  TEST INFRASTRUCTURE ONLY.

Reference call sites (the callers' resize; the wrapper itself has none):
  object_detectors/yolov8_lmi/run_model.py:79-87          (BGR->RGB, cv2.resize stretch)
  tests/object_detectors/yolov8_lmi/test_yolov8_model.py:84
  lmi_utils/gadget_utils/pipeline_utils.py:145-173        (centred constant pad)
  object_detectors/yolov8_lmi/model.py:112-134            (HWC->NCHW, cast, /255)

Third-party algorithm restated (opencv-python is NOT installed here; the
reference pins 4.8.0.74 only in tests/dockerfile.v0:30, otherwise unpinned):
OpenCV 4.x `modules/imgproc/src/resize.cpp`, functions `cv::hal::resize`
(coefficient tables), `HResizeLinear<uchar,int,short,2048,...>` and
`VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>,...>`:

  inv_scale = (double)dst/src ; scale = 1./inv_scale
  f  = (float)((d + 0.5) * scale - 0.5) ; s = floor(f) ; f -= s        (float32)
  x only: s < 0      -> f = 0, s = 0
          s >= src-1 -> f = 0, s = src-1
  coef = { cvRound((1.f - f) * 2048), cvRound(f * 2048) }              (int16, half-to-even)
  rows: the two source rows floor(fy), floor(fy)+1 are each clamped to
        [0, src_h-1]; their weights are NOT reset.
  H[r][dx] = S[r][sx] * a0 + S[r][sx+1] * a1                           (int32, scale 2^11)
  D[dy][dx] = ( ((b0 * (H[r0][dx] >> 4)) >> 16)
              + ((b1 * (H[r1][dx] >> 4)) >> 16) + 2 ) >> 2             (uint8)

The IPP path is disabled for 8-bit linear unless "IPP not exact" is enabled, and
the 2x2 `INTER_LINEAR -> INTER_AREA` shortcut gives the same integers
((a+b+c+d+2)>>2), so one formula covers every size.  PARITY UNPINNED against an
OpenCV binary; cross-checked in tests/test_oracle_cv2_resize.py by an
independent per-pixel scalar implementation and hand-derived cases.
"""

from __future__ import annotations

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def linear_coeffs(src_n: int, dst_n: int, clamp_taps: bool):
    """Source index / int16 weight tables for one axis.

    clamp_taps=True reproduces the x-axis border rule (weight reset at both
    borders); False reproduces the y-axis rule (indices clamped later, weights
    kept).  Returns (s0, s1, w0, w1) with s* int32 and w* int32 (int16 range).
    """
    inv_scale = float(dst_n) / float(src_n)
    scale = 1.0 / inv_scale
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_taps:
        lo = s < 0
        f[lo] = 0.0
        s[lo] = 0
        hi = s >= src_n - 1
        f[hi] = 0.0
        s[hi] = src_n - 1
    one = np.float32(1.0)
    k = np.float32(INTER_RESIZE_COEF_SCALE)
    w0 = np.rint((one - f) * k).astype(np.int32)
    w1 = np.rint(f * k).astype(np.int32)
    s0 = np.clip(s, 0, src_n - 1).astype(np.int32)
    s1 = np.clip(s + 1, 0, src_n - 1).astype(np.int32)
    return s0, s1, w0, w1


def resize_linear_u8(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(src, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for uint8 HWC / HW."""
    if src.dtype != np.uint8:
        raise TypeError("uint8 only")
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    h, w, _ = src.shape
    if (dst_w, dst_h) == (w, h):
        out = src.copy()
        return out[:, :, 0] if squeeze else out
    x0, x1, a0, a1 = linear_coeffs(w, dst_w, clamp_taps=True)
    y0, y1, b0, b1 = linear_coeffs(h, dst_h, clamp_taps=False)
    s = src.astype(np.int32)
    a0 = a0[None, :, None]
    a1 = a1[None, :, None]

    def hpass(rows):
        r = s[rows]  # (dst_h, w, c)
        return r[:, x0, :] * a0 + r[:, x1, :] * a1  # int32, scale 2^11

    h0 = hpass(y0)
    h1 = hpass(y1)
    b0 = b0[:, None, None]
    b1 = b1[:, None, None]
    v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
    out = v.astype(np.uint8)  # values are provably within 0..255
    return out[:, :, 0] if squeeze else out


def letterbox_geometry(h: int, w: int, dst_h: int, dst_w: int):
    """Aspect-preserving resize size + centred pad split.

    r = min(S_h/h, S_w/w); new = (round(w r), round(h r)); pad halves follow
    ultralytics LetterBox (`round(d/2 -+ 0.1)`), which for every size equals the
    reference helpers' `pad_L = (W-w)//2` (pipeline_utils.py:110-131): smaller
    half first.  This is the convention `ops.scale_boxes` inverts
    (in-tree form: yolov5_lmi/trt/yolov5_trt.py:328-341).
    Returns (new_w, new_h, pad_l, pad_r, pad_t, pad_b).
    """
    r = min(dst_h / h, dst_w / w)
    new_w = max(1, min(dst_w, int(round(w * r))))
    new_h = max(1, min(dst_h, int(round(h * r))))
    dw = (dst_w - new_w) / 2
    dh = (dst_h - new_h) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_w, new_h, left, right, top, bottom


def normalize_u8(x: np.ndarray, fp16: bool) -> np.ndarray:
    """`img.float(); img /= 255` / `img.half(); img /= 255` (model.py:132-133).

    torch-CPU divides fp32 by the fp32 scalar 255 with IEEE division; for fp16 it
    computes in fp32 and rounds the quotient to fp16.
    """
    q = x.astype(np.float32) / np.float32(255.0)
    return q.astype(np.float16) if fp16 else q


def letterbox_u8(
    frame: np.ndarray,
    dst_h: int,
    dst_w: int,
    pad_value: int = 114,
    swap_rb: bool = False,
    fp16: bool = False,
    stretch: bool = False,
):
    """(resize | stretch) -> centred constant pad -> optional BGR<->RGB -> /255 -> CHW.

    Returns (chw float tensor as numpy (3, dst_h, dst_w), operators list) where
    operators is what `revert_to_origin` consumes:
    [{'resize': [new_w, new_h, w, h]}, {'pad': [L, R, T, B]}].
    """
    h, w = frame.shape[:2]
    if stretch:
        new_w, new_h, left, right, top, bottom = dst_w, dst_h, 0, 0, 0, 0
    else:
        new_w, new_h, left, right, top, bottom = letterbox_geometry(h, w, dst_h, dst_w)
    small = resize_linear_u8(frame, new_w, new_h)
    canvas = np.full((dst_h, dst_w, 3), pad_value, dtype=np.uint8)
    canvas[top : top + new_h, left : left + new_w] = small
    if swap_rb:
        canvas = canvas[:, :, ::-1]
    chw = np.ascontiguousarray(canvas.transpose(2, 0, 1))
    ops = [{"resize": [new_w, new_h, w, h]}, {"pad": [left, right, top, bottom]}]
    return normalize_u8(chw, fp16), ops
