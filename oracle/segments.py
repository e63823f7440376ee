"""Oracle: `ops.masks2segments(masks)` for the `segments` key of the result dict
(object_detectors/yolov8_lmi/model.py:238-243; in-tree YOLOv5 form
object_detectors/yolov5_lmi/trt/yolov5_trt.py:418-431).  TEST INFRASTRUCTURE ONLY.

The reference traces contours on the CPU with `cv2.findContours(x, RETR_EXTERNAL,
CHAIN_APPROX_SIMPLE)`; cv2 is not installed, so the border following (Suzuki & Abe 1985, outer
borders only, 8-connected foreground, holes filled first so that components nested in holes are
not external) is restated here on numpy / scipy arrays, one pixel at a time.  PARITY UNPINNED
against an OpenCV binary: point sets are the border pixels with collinear runs collapsed, start
pixel = first border pixel in raster order, traversal as in the paper.  The product's tracer
(object_detectors_amd/csrc/contours.hip) is compared with this file point for point.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy import ndimage

# 8-neighbourhood in clockwise order starting from west, as (dy, dx) in image coordinates (y down)
_CW = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]
_IDX = {d: k for k, d in enumerate(_CW)}


def _trace_outer(f: np.ndarray, i: int, j: int):
    """Border following from the raster-first pixel (i,j) of a component in the zero-padded 0/1
    image f.  Returns the (x,y) border points with collinear runs collapsed."""
    # 3.1: clockwise from the west neighbour, first nonzero neighbour
    first = None
    for k in range(8):
        dy, dx = _CW[k]
        if f[i + dy, j + dx]:
            first = (i + dy, j + dx)
            break
    if first is None:
        return [(j, i)]
    pts, dirs = [], []
    i2, j2 = first
    i3, j3 = i, j
    while True:
        # 3.3: counter-clockwise around (i3,j3), starting after (i2,j2)
        k0 = _IDX[(i2 - i3, j2 - j3)]
        for s in range(1, 9):
            dy, dx = _CW[(k0 - s) % 8]
            if f[i3 + dy, j3 + dx]:
                i4, j4 = i3 + dy, j3 + dx
                break
        pts.append((j3, i3))
        dirs.append((i4 - i3, j4 - j3))
        if (i4, j4) == (i, j) and (i3, j3) == first:
            break
        i2, j2, i3, j3 = i3, j3, i4, j4
    # CHAIN_APPROX_SIMPLE: keep a point when the direction leaving it differs from the one entering it
    n = len(pts)
    if n <= 2:
        return pts
    keep = [pts[k] for k in range(n) if dirs[k] != dirs[k - 1]]
    return keep or [pts[0]]


def find_external_contours(mask: np.ndarray):
    """List of (k,2) int32 arrays of (x,y) points, one per outermost 8-connected component."""
    m = np.asarray(mask) != 0
    if not m.any():
        return []
    ys, xs = np.nonzero(m)
    y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
    sub = m[y0:y1, x0:x1]
    filled = ndimage.binary_fill_holes(sub)   # components nested in holes are not external
    lab, n = ndimage.label(filled, structure=np.ones((3, 3), bool))
    f = np.pad(filled, 1).astype(np.uint8)
    starts = ndimage.find_objects(lab)
    out = []
    for c in range(1, n + 1):
        sl = starts[c - 1]
        r = sl[0].start
        cidx = np.nonzero(lab[r, sl[1]] == c)[0][0] + sl[1].start
        pts = _trace_outer(f, r + 1, cidx + 1)
        a = np.asarray(pts, dtype=np.int32).reshape(-1, 2)
        a[:, 0] += x0 - 1
        a[:, 1] += y0 - 1
        out.append(a)
    return out


def masks2segments(masks, strategy: str = "all"):
    """masks (n,H,W) tensor/array -> list of n float32 (k,2) arrays.  strategy 'all' concatenates
    every external contour (ultralytics 8.3.x default), 'largest' keeps the longest one (the
    in-tree YOLOv5 behaviour, yolov5_trt.py:426-427)."""
    if isinstance(masks, torch.Tensor):
        masks = masks.cpu().numpy()
    segments = []
    for x in masks:
        c = find_external_contours(x)
        if c:
            if strategy == "largest":
                c = c[int(np.argmax([len(v) for v in c]))]
            else:
                c = np.concatenate(c)
        else:
            c = np.zeros((0, 2))
        segments.append(c.astype("float32"))
    return segments
