"""Algorithmic byte counts per kernel launch (the numerators of `roofline.achieved`), as defined
in SURVEY.md 8(d) and DESIGN.md.  Pure arithmetic, no GPU."""
from __future__ import annotations


def _geometry(h, w, dst_h, dst_w):
    r = min(dst_h / h, dst_w / w)
    return max(1, min(dst_w, int(round(w * r)))), max(1, min(dst_h, int(round(h * r))))


def letterbox_bytes(h: int, w: int, dst_h: int, dst_w: int, out_elem: int = 4) -> int:
    """Unique source taps + output tensor, per image.  INTER_LINEAR reads 2 rows x 2 columns per
    output pixel, so a k-fold down-scale touches min(h, 2*new_h) rows and min(w, 2*new_w) columns:
    1080p/4K -> 640: 720*1280*3 + 3*640*640*4 = 7,680,000 B."""
    new_w, new_h = _geometry(h, w, dst_h, dst_w)
    return min(h, 2 * new_h) * min(w, 2 * new_w) * 3 + 3 * dst_h * dst_w * out_elem


def letterbox_row_granular_bytes(h: int, w: int, dst_h: int, dst_w: int, out_elem: int = 4) -> int:
    """Whole source rows that contain a tap (the attainable DRAM lower bound) + output."""
    new_w, new_h = _geometry(h, w, dst_h, dst_w)
    return min(h, 2 * new_h) * w * 3 + 3 * dst_h * dst_w * out_elem


def nms_decode_bytes(nc: int, anchors: int) -> int:
    """Class rows of the head, read once, per image."""
    return nc * anchors * 4


def mask_crop_rows(mh: int, mw: int, h0: int, w0: int):
    gain = min(mh / h0, mw / w0)
    pad_x, pad_y = (mw - w0 * gain) / 2, (mh - h0 * gain) / 2
    return int(mh - pad_y) - int(pad_y), int(mw - pad_x) - int(pad_x)


def mask_upsample_bytes(n_masks: int, h0: int, w0: int, mh: int, mw: int, out_elem: int = 1) -> int:
    """Per launch: every survivor's full-resolution mask written once + its cropped logits read once."""
    wh, ww = mask_crop_rows(mh, mw, h0, w0)
    return n_masks * (h0 * w0 * out_elem + wh * ww * 4)


def _aten_src_index(scale: float, d: int, in_n: int):
    """(i0, i1) of aten's bilinear source index for output index d (align_corners=False), fp32 like the kernels."""
    import numpy as np
    s = max(np.float32(scale) * (np.float32(d) + np.float32(0.5)) - np.float32(0.5), np.float32(0))
    i0 = int(s)
    return i0, i0 + (1 if i0 < in_n - 1 else 0)


def mask_upsample_bytes_exact(boxes, h0: int, w0: int, mh: int, mw: int, out_elem: int = 1) -> int:
    """Per launch, from the survivors' boxes (n,4) in original-image pixels: every full-resolution mask written once
    + the logits rows x columns that the pixels under each box read (what the box pass fetches; the logits outside a
    box are neither computed nor read).  Round 1 charged the whole crop window per survivor, which over-counts."""
    import math
    import numpy as np
    wh, ww = mask_crop_rows(mh, mw, h0, w0)
    sh, sw = np.float32(wh) / np.float32(h0), np.float32(ww) / np.float32(w0)
    total = 0
    for x1, y1, x2, y2 in np.asarray(boxes, dtype=np.float64):
        total += h0 * w0 * out_elem
        if not (x1 < x2 and y1 < y2):
            continue
        xa, xb = max(0, math.ceil(x1)), min(w0, math.ceil(x2))
        ya, yb = max(0, math.ceil(y1)), min(h0, math.ceil(y2))
        if xa >= xb or ya >= yb:
            continue
        r0, r1 = _aten_src_index(sh, ya, wh)[0], _aten_src_index(sh, yb - 1, wh)[1]
        c0, c1 = _aten_src_index(sw, xa, ww)[0], _aten_src_index(sw, xb - 1, ww)[1]
        total += (r1 - r0 + 1) * (c1 - c0 + 1) * 4
    return int(total)


def mask_logits_bytes(batch: int, n_masks: int, h0: int, w0: int, mh: int, mw: int, nm: int = 32) -> int:
    """Per launch: the crop-window rows of every prototype read once + logits written once + coefficients."""
    wh, _ = mask_crop_rows(mh, mw, h0, w0)
    return batch * nm * wh * mw * 4 + n_masks * (wh * mw * 4 + nm * 4)
