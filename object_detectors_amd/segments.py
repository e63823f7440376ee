"""`ops.masks2segments(masks)` for the `segments` key of the result dict
(object_detectors/yolov8_lmi/model.py:238-243; in-tree YOLOv5 form yolov5_lmi/trt/yolov5_trt.py:418-431).

The reference copies every full-resolution mask to the host and runs cv2.findContours per mask.  Here masks that
live on the GPU are first reduced to one bit per pixel of their box region (`lmi_pack_mask_boxes`, wave ballots:
about 1/50 of the bytes), copied once, and the border following runs in C++ on the host cores for all masks in
parallel (`lmi_trace_contours_host`).  Border following is sequential by nature; SURVEY.md 8(a) row a9 keeps it
off the device.  Semantics and their pinning status: oracle/segments.py.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch


def _lib():
    from . import _lib as L   # raises if liblmi_hip.so is missing
    return L, L.load()


def _int_boxes(boxes, n, h, w):
    """xyxy boxes -> integer half-open pixel ranges that contain every pixel crop_mask leaves on
    (x >= x1 && x < x2  <=>  ceil(x1) <= x < ceil(x2)); None -> the whole image."""
    if boxes is None:
        return np.tile(np.array([0, 0, w, h], np.int32), (n, 1))
    b = np.asarray(boxes.detach().cpu().numpy() if isinstance(boxes, torch.Tensor) else boxes, dtype=np.float64).reshape(n, -1)[:, :4]
    out = np.empty((n, 4), np.int32)
    out[:, 0] = np.clip(np.ceil(b[:, 0]), 0, w)
    out[:, 1] = np.clip(np.ceil(b[:, 1]), 0, h)
    out[:, 2] = np.clip(np.ceil(b[:, 2]), 0, w)
    out[:, 3] = np.clip(np.ceil(b[:, 3]), 0, h)
    out[:, 2] = np.maximum(out[:, 2], out[:, 0])
    out[:, 3] = np.maximum(out[:, 3], out[:, 1])
    return out


def masks2segments(masks, strategy: str = "all", boxes=None, threads: Optional[int] = None):
    """masks (n,H,W) tensor/array -> list of n float32 (k,2) arrays of (x,y) contour points.  strategy 'all'
    concatenates every external contour (ultralytics 8.3.x default), 'largest' keeps the longest one (the in-tree
    YOLOv5 behaviour, yolov5_trt.py:426-427).  boxes (n,4) xyxy, optional: regions outside of which the masks are
    known to be zero (the detector's own boxes) -- only those are packed and copied."""
    L, lib = _lib()
    n = len(masks)
    if n == 0:
        return []
    h, w = int(masks.shape[1]), int(masks.shape[2])
    ib = _int_boxes(boxes, n, h, w)
    rows = (ib[:, 3] - ib[:, 1]).astype(np.int64)
    row_words = ((ib[:, 2] - ib[:, 0]).astype(np.int64) + 63) // 64
    words = rows * row_words
    offset = np.zeros(n, np.int64)
    np.cumsum(words[:-1], out=offset[1:])
    total_words = int(words.sum())
    if isinstance(masks, torch.Tensor) and masks.is_cuda:
        ldt = {torch.uint8: L.LMI_U8, torch.float32: L.LMI_F32, torch.bool: L.LMI_U8}.get(masks.dtype)
        if ldt is None:
            raise TypeError("masks2segments handles uint8, bool and fp32 masks")
        m = masks.contiguous()
        packed = torch.empty((max(total_words, 1),), dtype=torch.int64, device=m.device)
        d_boxes = torch.from_numpy(ib).to(m.device)
        d_off = torch.from_numpy(offset).to(m.device)
        L.check(lib.lmi_pack_mask_boxes(m.data_ptr(), ldt, n, h, w, d_boxes.data_ptr(), d_off.data_ptr(), packed.data_ptr(),
                                        torch.cuda.current_stream(m.device).cuda_stream))
        bits = packed.cpu().numpy().view(np.uint64)   # the one device->host copy: one bit per box pixel
    else:
        a = masks.numpy() if isinstance(masks, torch.Tensor) else np.asarray(masks)
        bits = np.zeros(max(total_words, 1), np.uint64)
        for i in range(n):   # host arrays: pack with numpy (bit c of word w = column 64w + c)
            x0, y0, x1, y1 = ib[i]
            if x1 <= x0 or y1 <= y0:
                continue
            crop = a[i, y0:y1, x0:x1] != 0
            pad = int(row_words[i]) * 64 - crop.shape[1]
            pk = np.packbits(np.pad(crop, ((0, 0), (0, pad))), axis=1, bitorder="little")
            bits[offset[i]:offset[i] + words[i]] = np.ascontiguousarray(pk).view(np.uint64).reshape(-1)
    nthreads = int(threads or min(16, os.cpu_count() or 1))
    counts = np.zeros(n, np.int64)
    ib_c = np.ascontiguousarray(ib)
    capacity = max(4096, 64 * n)
    while True:
        out = np.empty((capacity, 2), np.int32)
        total = int(lib.lmi_trace_contours_host(bits.ctypes.data, n, ib_c.ctypes.data, offset.ctypes.data, 1 if strategy != "largest" else 0,
                                                nthreads, out.ctypes.data, capacity, counts.ctypes.data))
        if total < 0:
            L.check(total)
        if total <= capacity:
            break
        capacity = total
    ends = np.cumsum(counts)
    return [out[e - c:e].astype(np.float32) if c else np.zeros((0, 2), np.float32) for c, e in zip(counts, ends)]
