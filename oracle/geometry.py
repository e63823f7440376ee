"""Oracle: the reference's own L1 geometry helpers, restated in numpy scalars.
TEST INFRASTRUCTURE ONLY.

Follows lmi_utils/gadget_utils/pipeline_utils.py:
  revert_to_origin   :422-476   (reverse operator list; round half-to-even; clamp >= 0)
  apply_operations   :479-509   (invert each operator, then revert in reverse)
  fit_im_to_size     :72-142    (centred pad / crop; pad_L = (W-w)//2)
  resize_image       :23-68     (target size rule `int(w*H/h)`)
  revert_mask_to_origin :374-405 (shape bookkeeping only here)
PINNED: tests/golden/l1_geometry.json holds outputs of those reference functions
run in the build container (tests/golden/make_golden_l1.py), plus the
known-answer cases of tests/lmi_utils/gadget_utils/test_pipeline_utils.py:117-186,
:327-390.

dtype rules reproduced (they change the rounding): the per-operator ratio tensors
are built with `torch.tensor([...])` and are therefore fp32 even when the points
are float64; Python lists of floats become fp32 points, integer points become fp32
at the first division.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


def _as_points(pts):
    if isinstance(pts, np.ndarray):
        return pts.copy()
    a = np.asarray(pts)
    if a.dtype == np.float64:  # torch.as_tensor(list of python floats) -> fp32
        a = a.astype(F32)
    return a


def _div(p, r32):
    if np.issubdtype(p.dtype, np.integer):
        p = p.astype(F32)
    return p / r32.astype(p.dtype)


def revert_to_origin(pts, operations):
    p = _as_points(pts)
    if p.ndim != 2 or p.shape[1] not in (2, 4):
        raise Exception(f"pts should be Nx2 or Nx4, got shape: {p.shape}")
    c = p.shape[1]
    rep = c // 2
    for op in reversed(operations):
        if "resize" in op:
            tw, th, ow, oh = op["resize"]
            p = _div(p, np.tile(np.array([tw / ow, th / oh], dtype=F32), rep))
        elif "pad" in op:
            left, _, top, _ = op["pad"]
            p = p - np.tile(np.array([left, top]), rep).astype(p.dtype if not np.issubdtype(p.dtype, np.integer) else np.int64)
        elif "stretch" in op:
            p = _div(p, np.tile(np.array(op["stretch"], dtype=F32), rep))
        elif "flip" in op:
            lr, ud, im_w, im_h = op["flip"]
            if lr:
                p[:, 0::2] = im_w - p[:, 0::2]
            if ud:
                p[:, 1::2] = im_h - p[:, 1::2]
        else:
            raise Exception(f"unsupported operation: {op}")
    p = np.maximum(np.rint(p), 0)
    return p


def apply_operations(pts, operations):
    inv = []
    for op in operations:
        if "resize" in op:
            tw, th, ow, oh = op["resize"]
            inv.append({"resize": [ow, oh, tw, th]})
        elif "pad" in op:
            left, right, top, bottom = op["pad"]
            inv.append({"pad": [-left, -right, -top, -bottom]})
        elif "stretch" in op:
            sx, sy = op["stretch"]
            inv.append({"stretch": [1 / sx, 1 / sy]})
        elif "flip" in op:
            inv.append({"flip": list(op["flip"])})
        else:
            raise Exception(f"unsupported operation: {op}")
    return revert_to_origin(pts, inv[::-1])


def fit_pads(h: int, w: int, W=None, H=None):
    """(pad_L, pad_R, pad_T, pad_B) of fit_im_to_size; negative = centred crop."""
    if W is None and H is None:
        return 0, 0, 0, 0
    W = w if W is None else W
    H = h if H is None else H
    if W >= w:
        pl = (W - w) // 2
        pr = W - w - pl
    else:
        pl = (w - W) // 2
        pr = w - W - pl
        pl, pr = -pl, -pr
    if H >= h:
        pt = (H - h) // 2
        pb = H - h - pt
    else:
        pt = (h - H) // 2
        pb = h - H - pt
        pt, pb = -pt, -pb
    return pl, pr, pt, pb


def fit_im_to_size(im: np.ndarray, W=None, H=None, value=0):
    h, w = im.shape[:2]
    pl, pr, pt, pb = fit_pads(h, w, W, H)
    out = im
    if pl < 0 or pr < 0:   # crop (a crop by one pixel is pads (0, -1))
        out = out[:, -pl : out.shape[1] + pr]
    elif pl or pr:
        widths = [(0, 0), (pl, pr)] + [(0, 0)] * (im.ndim - 2)
        out = np.pad(out, widths, constant_values=value)
    if pt < 0 or pb < 0:
        out = out[-pt : out.shape[0] + pb]
    elif pt or pb:
        widths = [(pt, pb), (0, 0)] + [(0, 0)] * (im.ndim - 2)
        out = np.pad(out, widths, constant_values=value)
    return out, pl, pr, pt, pb


def resize_target(h: int, w: int, W=None, H=None):
    """resize_image's size rule: the missing side is `int(w*H/h)` (truncation)."""
    if W is None and H is None:
        return w, h
    if W is None:
        W = int(w * H / h)
    elif H is None:
        H = int(h * W / w)
    return W, H


# ---------------------------------------------------------------------------------------------
# Mask revert (lmi_utils/gadget_utils/pipeline_utils.py:374-418 -> resize_image :23-68,
# fit_im_to_size :72-142, torch.flip).  PINNED: tests/golden/revert_masks.npz holds outputs of the
# reference's revert_masks_to_origin run in the build container (tests/golden/make_golden_revert_masks.py).
def _fma(a, b, c):
    """fp32 fused multiply-add: the fp64 product of two fp32 values is exact, so one rounding at the end."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F32)


def _bilinear_axis(in_n: int, out_n: int):
    """aten linear index/weight tables, align_corners=False, fp32.  The x86 build of torch 2.10 contracts
    `scale * (dst + 0.5) - 0.5` into one fused multiply-add (probed here: only the fused form reproduces
    F.interpolate bit for bit), unlike oracle/masks.py:bilinear_axis which keeps the two roundings."""
    scale = F32(in_n) / F32(out_n)
    d = np.arange(out_n, dtype=F32)
    src = np.maximum(_fma(scale, d + F32(0.5), F32(-0.5)), F32(0))
    i0 = np.minimum(src.astype(np.int64), in_n - 1)
    i1 = i0 + (i0 < in_n - 1)
    l1 = np.clip((src - i0.astype(F32)).astype(F32), F32(0), F32(1))
    return i0, i1, (F32(1) - l1).astype(F32), l1


def aten_linear_variant(channels: int, H: int, W: int, one_thread: bool = False) -> str:
    """Which of aten's two CPU bilinear kernels F.interpolate runs for a channels-last fp32 input (what
    `im.permute(2,0,1).unsqueeze(0)` of an HWC image is), torch 2.10 x86: the per-pixel 4-weight kernel when
    `out_H + out_W <= 128` (or C > 3, or C == 3 on exactly one thread), else the separable generic kernel.
    Probed in the build container: the switch sits at out_H + out_W == 129 for every width tried, 1 thread
    differs from 2/3/8 threads only for C == 3 outputs above that size.  The two round differently."""
    if channels > 3 or H + W <= 128 or (one_thread and channels == 3):
        return "weights4"
    return "separable"


def resize_bilinear(im: np.ndarray, W: int, H: int, one_thread: bool = False) -> np.ndarray:
    """resize_image(im, W, H) for an (H,W) plane or (H,W,C) image: fp32 bilinear with aten's operation order as
    built for x86 (bit-exact against F.interpolate, tests/test_oracle_geometry.py), then the truncating cast back
    for integer images.  separable kernel: horizontal taps first, then vertical, each pair as fma(a, wa, b*wb);
    4-weight kernel: w = wy*wx products in fp32, then fma(w11,i11, fma(w10,i10, fma(w00,i00, w01*i01)))."""
    src = im.astype(F32)
    y0, y1, wy0, wy1 = _bilinear_axis(src.shape[0], H)
    x0, x1, wx0, wx1 = _bilinear_axis(src.shape[1], W)
    tail = (1,) * (src.ndim - 2)   # (H,W,C) images: every channel gets the pixel's weights
    wx0, wx1 = wx0.reshape((1, W) + tail), wx1.reshape((1, W) + tail)
    wy0, wy1 = wy0.reshape((H, 1) + tail), wy1.reshape((H, 1) + tail)
    i00, i01, i10, i11 = src[y0][:, x0], src[y0][:, x1], src[y1][:, x0], src[y1][:, x1]
    if aten_linear_variant(src.shape[2] if src.ndim == 3 else 1, H, W, one_thread) == "weights4":
        w00, w01 = (wy0 * wx0).astype(F32), (wy0 * wx1).astype(F32)
        w10, w11 = (wy1 * wx0).astype(F32), (wy1 * wx1).astype(F32)
        out = _fma(w11, i11, _fma(w10, i10, _fma(w00, i00, (w01 * i01).astype(F32))))
    else:
        r0 = _fma(i00, wx0, (i01 * wx1).astype(F32))
        r1 = _fma(i10, wx0, (i11 * wx1).astype(F32))
        out = _fma(r0, wy0, (r1 * wy1).astype(F32))
    return out if np.issubdtype(im.dtype, np.floating) else out.astype(im.dtype)


def revert_mask_to_origin(mask: np.ndarray, operations) -> np.ndarray:
    for op in reversed(list(operations)):
        if 'resize' in op:
            _, _, nw, nh = op['resize']
            mask = resize_bilinear(mask, nw, nh)
        if 'pad' in op:
            h, w = mask.shape[:2]
            pl, pr, pt, pb = op['pad']
            mask, *_ = fit_im_to_size(mask, w - pl - pr, h - pt - pb)
        if 'flip' in op:
            lr, ud = op['flip'][0], op['flip'][1]
            if lr:
                mask = mask[:, ::-1]
            if ud:
                mask = mask[::-1]
    return np.ascontiguousarray(mask)


def revert_masks_to_origin(masks: np.ndarray, operations) -> np.ndarray:
    return np.stack([revert_mask_to_origin(m, operations) for m in masks]) if len(masks) else masks


# ---------------------------------------------------------------------------------------------
# LMI-flavoured caller-side resize + pad: `resize_image` then `fit_im_to_size`
# (pipeline_utils.py:23-142) as chained by the v1 runner (ultralytics_lmi/yolo/run_model.py:87-100), followed by
# `Yolov8.preprocess` (model.py:112-134).  PINNED: tests/golden/letterbox_torch.npz holds outputs of the
# reference's own two helpers run in the build container (tests/golden/make_golden_letterbox_torch.py), for torch
# on >= 2 threads (aten picks another kernel, with other rounding, on exactly one thread).
def letterbox_torch_u8(frame: np.ndarray, dst_h: int, dst_w: int, resize_w=None, resize_h=None, pad_value: int = 0,
                       swap_rb: bool = False, fp16: bool = False, one_thread: bool = False):
    """uint8 (h,w,3) -> resize_image(W=resize_w, H=resize_h) (fp32 bilinear, truncating cast; a missing side keeps
    the aspect ratio with `int()`; both missing: the aspect-preserving fit into (dst_h, dst_w), the side that does
    not touch the frame truncated the same way) -> fit_im_to_size(H=dst_h, W=dst_w, value=pad_value) (centred pad,
    or centred crop where the content is larger) -> optional channel reverse -> /255 -> CHW.
    Returns (chw, operators) with operators = [{'resize': [new_w,new_h,w,h]}, {'pad': [L,R,T,B]}]."""
    h, w = frame.shape[:2]
    if resize_w is None and resize_h is None:
        if dst_h * w <= dst_w * h:
            resize_h = dst_h
        else:
            resize_w = dst_w
    new_w, new_h = resize_target(h, w, resize_w, resize_h)
    small = resize_bilinear(frame, new_w, new_h, one_thread=one_thread)
    canvas, pl, pr, pt, pb = fit_im_to_size(small, W=dst_w, H=dst_h, value=pad_value)
    if swap_rb:
        canvas = canvas[:, :, ::-1]
    chw = np.ascontiguousarray(canvas.transpose(2, 0, 1))
    q = chw.astype(F32) / F32(255.0)
    ops = [{"resize": [new_w, new_h, w, h]}, {"pad": [pl, pr, pt, pb]}]
    return (q.astype(np.float16) if fp16 else q), ops
