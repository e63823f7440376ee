"""Oracle: `rescale_masks` (lmi_utils/postprocess_utils/mask_utils.py:12-140), the detectron2-style paste of
fixed-resolution soft masks (e.g. 28x28) into their boxes of a full image.  TEST INFRASTRUCTURE ONLY.

The reference module cannot be imported here (it needs numba and cv2 at import time), and the reference's tests
hold no fixture for it: PARITY UNPINNED against the reference's own output.  What is restated is its op sequence,
executed by torch-CPU's own `F.grid_sample` exactly as the reference's GPU branch does (whole image per mask,
skip_empty=False, :47-59): img = arange + 0.5; g = (img - lo) / (hi - lo) * 2 - 1; grid_sample(masks, grid,
align_corners=False) (bilinear, zero padding); `>= threshold` -> bool (:130-131).  `paste_np` is an independent
numpy restatement of aten's bilinear grid sampling used to cross-check it.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

F32 = np.float32


def rescale_masks(masks: np.ndarray, boxes: np.ndarray, image_shape, threshold: float = 0.5) -> np.ndarray:
    """masks (N,Hm,Wm) float in [0,1], boxes (N,4) xyxy -> (N,H,W) bool (uint8 0..255 for threshold < 0)."""
    img_h, img_w = int(image_shape[0]), int(image_shape[1])
    n = len(masks)
    if n == 0:
        return np.zeros((0, img_h, img_w), bool)
    m = torch.from_numpy(np.ascontiguousarray(masks, dtype=F32))[:, None]
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=F32))
    x0, y0, x1, y1 = torch.split(b, 1, dim=1)
    img_y = torch.arange(0, img_h, dtype=torch.float32) + 0.5
    img_x = torch.arange(0, img_w, dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(n, img_y.size(1), img_x.size(1))
    gy = img_y[:, :, None].expand(n, img_y.size(1), img_x.size(1))
    grid = torch.stack([gx, gy], dim=3)
    out = F.grid_sample(m, grid, align_corners=False)[:, 0]
    if threshold < 0:   # the reference's visualisation mode (:131-133)
        return (out * 255).to(dtype=torch.uint8).numpy()
    return (out >= threshold).numpy()


def paste_np(masks: np.ndarray, boxes: np.ndarray, image_shape) -> np.ndarray:
    """(N,H,W) fp32 sampled values before thresholding; aten's scalar bilinear formula (zeros padding)."""
    img_h, img_w = int(image_shape[0]), int(image_shape[1])
    n, hm, wm = masks.shape
    m = masks.astype(F32)
    out = np.zeros((n, img_h, img_w), F32)
    ys = np.arange(img_h, dtype=F32) + F32(0.5)
    xs = np.arange(img_w, dtype=F32) + F32(0.5)
    for i in range(n):
        x0, y0, x1, y1 = (F32(v) for v in boxes[i])
        gy = ((ys - y0) / (y1 - y0) * F32(2) - F32(1)).astype(F32)
        gx = ((xs - x0) / (x1 - x0) * F32(2) - F32(1)).astype(F32)
        iy = (((gy + F32(1)) * F32(hm) - F32(1)) / F32(2)).astype(F32)
        ix = (((gx + F32(1)) * F32(wm) - F32(1)) / F32(2)).astype(F32)
        iy0, ix0 = np.floor(iy), np.floor(ix)
        wy1, wx1 = (iy - iy0).astype(F32), (ix - ix0).astype(F32)
        wy0, wx0 = (F32(1) - wy1).astype(F32), (F32(1) - wx1).astype(F32)
        iy0, ix0 = iy0.astype(np.int64), ix0.astype(np.int64)

        def tap(yy, xx):
            ok = ((yy >= 0) & (yy < hm))[:, None] & ((xx >= 0) & (xx < wm))[None, :]
            v = m[i][np.clip(yy, 0, hm - 1)][:, np.clip(xx, 0, wm - 1)]
            return np.where(ok, v, F32(0))
        out[i] = (tap(iy0, ix0) * (wy0[:, None] * wx0[None, :]) + tap(iy0, ix0 + 1) * (wy0[:, None] * wx1[None, :])
                  + tap(iy0 + 1, ix0) * (wy1[:, None] * wx0[None, :]) + tap(iy0 + 1, ix0 + 1) * (wy1[:, None] * wx1[None, :]))
    return out
