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


def mask_logits_bytes(batch: int, n_masks: int, h0: int, w0: int, mh: int, mw: int, nm: int = 32) -> int:
    """Per launch: the crop-window rows of every prototype read once + logits written once + coefficients."""
    wh, _ = mask_crop_rows(mh, mw, h0, w0)
    return batch * nm * wh * mw * 4 + n_masks * (wh * mw * 4 + nm * 4)
