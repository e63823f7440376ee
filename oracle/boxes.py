"""Oracle: `ops.scale_boxes` + clip (called at object_detectors/yolov8_lmi/model.py:222)
and the per-class confidence filter (model.py:223-232).  TEST INFRASTRUCTURE ONLY.

scale_boxes follows the in-tree form object_detectors/yolov5_lmi/trt/yolov5_trt.py:328-341
(+ clip_boxes :316-325) with the ultralytics 8.3.x delta that the pad is rounded:
`round((w1 - w0*gain)/2 - 0.1)`.  `pad_mode='exact'` keeps the un-rounded v5 pad.
gain/pad are Python doubles; the tensor ops are fp32 with the scalar cast to fp32
(torch-CPU: true division by fp32(gain), checked in tests).  PARITY UNPINNED
against an ultralytics binary (not installed, version unpinned).

The per-class filter is pinned by the reference text itself: thresholds are a
float64 numpy array turned into a float64 tensor, so `confs > thres` compares
double(conf) with the double threshold, strictly (model.py:227-232).
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


def scale_boxes(img1_shape, boxes: np.ndarray, img0_shape, pad_mode: str = "round") -> np.ndarray:
    """Boxes (n,4) xyxy in model space (img1 = (h1,w1)) -> original space (img0 = (h0,w0,...)),
    clipped to [0,w0] x [0,h0].  Returns a new fp32 array."""
    h1, w1 = img1_shape[:2]
    h0, w0 = img0_shape[:2]
    gain = min(h1 / h0, w1 / w0)
    px = (w1 - w0 * gain) / 2
    py = (h1 - h0 * gain) / 2
    if pad_mode == "round":
        px, py = round(px - 0.1), round(py - 0.1)
    b = boxes.astype(F32, copy=True)
    b[:, 0] -= F32(px)
    b[:, 1] -= F32(py)
    b[:, 2] -= F32(px)
    b[:, 3] -= F32(py)
    b[:, :4] /= F32(gain)
    b[:, 0] = np.clip(b[:, 0], F32(0), F32(w0))
    b[:, 1] = np.clip(b[:, 1], F32(0), F32(h0))
    b[:, 2] = np.clip(b[:, 2], F32(0), F32(w0))
    b[:, 3] = np.clip(b[:, 3], F32(0), F32(h0))
    return b


def min_conf(conf, names: dict) -> float:
    """`Yolov8.get_min_conf` (model.py:156-174): NMS threshold = min over the dict's known classes."""
    if isinstance(conf, float):
        return conf
    if isinstance(conf, dict):
        c2 = 1
        known = set(names.values())
        for k, v in conf.items():
            if k in known:
                c2 = min(c2, v)
        return c2
    raise TypeError(f"Confidence type {type(conf)} not supported")


def class_thresholds(conf, names: dict, nc: int) -> np.ndarray:
    """Per-class-id float64 thresholds equivalent to model.py:227-231
    (`conf.get(name, 1)`: classes missing from the dict can never pass)."""
    if isinstance(conf, float):
        return np.full((nc,), conf, dtype=np.float64)
    return np.array([conf.get(names[i], 1) for i in range(nc)], dtype=np.float64)


def class_filter(confs: np.ndarray, cls_ids: np.ndarray, thres_per_class: np.ndarray) -> np.ndarray:
    """M = confs > thres[cls] with the comparison done in float64 (model.py:232)."""
    t = thres_per_class[cls_ids.astype(np.int64)]
    return confs.astype(np.float64) > t


def scale_coords(img1_shape, coords: np.ndarray, img0_shape) -> np.ndarray:
    """ultralytics 8.3.x `ops.scale_coords(img1_shape, coords, img0_shape)` (normalize=False, padding=True) as
    called by Yolov8Pose.postprocess (object_detectors/yolov8_lmi/model.py:569; in-tree analogue
    scale_segments yolov5_trt.py:354-370): gain = min(h1/h0, w1/w0) and pad = ((w1 - w0*gain)/2, (h1 - h0*gain)/2)
    are Python floats -- the pads are NOT rounded here, unlike scale_boxes -- applied to the fp32 tensor in place
    (x -= pad_x, y -= pad_y, x /= gain, y /= gain), then clipped to [0,w0] x [0,h0].  PARITY UNPINNED against
    ultralytics (not installed); the op sequence is checked against torch-CPU executing it (tests/test_oracle_nms.py).
    coords (..., 2|3): a third component (visibility) passes through."""
    h1, w1 = img1_shape[:2]
    h0, w0 = img0_shape[:2]
    gain = min(h1 / h0, w1 / w0)
    pad = (w1 - w0 * gain) / 2, (h1 - h0 * gain) / 2
    c = np.array(coords, dtype=F32, copy=True)
    c[..., 0] = (c[..., 0] - F32(pad[0])) / F32(gain)
    c[..., 1] = (c[..., 1] - F32(pad[1])) / F32(gain)
    c[..., 0] = np.clip(c[..., 0], F32(0), F32(w0))
    c[..., 1] = np.clip(c[..., 1], F32(0), F32(h0))
    return c
