"""Oracle: the oriented-box path of `Yolov8Obb.postprocess`
(object_detectors/yolov8_lmi/model.py:398-465): `ops.non_max_suppression(..., rotated=True)` :423,
`ops.regularize_rboxes` :436, `ops.scale_boxes(..., xywh=True)` :438, `ops.xywhr2xyxyxyxy` :446.
TEST INFRASTRUCTURE ONLY.

All four live in ultralytics (unpinned, tests/dockerfile:12; not installed, no in-tree copy of the rotated
code): PARITY UNPINNED.  The restatement follows the 8.3.x sources as published:

  non_max_suppression(rotated=True): no xywh->xyxy conversion; candidates `cls.amax(1) > conf` (strict, fp32),
    best class by first argmax, rows [x, y, w, h, conf, cls, extras...] with the angle as the LAST extra;
    class offset added to the CENTRE only: boxes = cat(xy + cls*max_wh, wh, angle); i = nms_rotated(boxes,
    scores, iou)[:max_det].
  nms_rotated: NOT greedy.  Sort by score descending, ious = batch_probiou(b, b).triu_(1), keep column j iff
    max_i<j ious[i, j] < threshold (a box is dropped by ANY higher-scored box, kept or not).
  batch_probiou (Gaussian Bhattacharyya / Hellinger distance between the boxes' covariance ellipses, eps=1e-7):
    a, b, c = w^2/12 cos^2 + h^2/12 sin^2, w^2/12 sin^2 + h^2/12 cos^2, (w^2/12 - h^2/12) cos sin
    t1 = ((a1+a2)(y1-y2)^2 + (b1+b2)(x1-x2)^2) / ((a1+a2)(b1+b2) - (c1+c2)^2 + eps) * 0.25
    t2 = ((c1+c2)(x2-x1)(y1-y2)) / ((a1+a2)(b1+b2) - (c1+c2)^2 + eps) * 0.5
    t3 = log(((a1+a2)(b1+b2) - (c1+c2)^2) / (4 sqrt(max(a1b1-c1^2,0) max(a2b2-c2^2,0)) + eps) + eps) * 0.5
    bd = clamp(t1+t2+t3, eps, 100); hd = sqrt(1 - exp(-bd) + eps); iou = 1 - hd
  regularize_rboxes: (w,h,t) -> (max, min, t if w > h else t + pi/2) with t % pi.
  scale_boxes(xywh=True): x -= pad_x, y -= pad_y (pads rounded `round(p - 0.1)` as for xyxy), [:4] /= gain,
    then clip_boxes clamps columns 0,2 to [0,w0] and 1,3 to [0,h0] -- i.e. also the WIDTH and HEIGHT columns.
  xywhr2xyxyxyxy: corners ctr + v1 + v2, ctr + v1 - v2, ctr - v1 - v2, ctr - v1 + v2 with
    v1 = (w/2 cos, w/2 sin), v2 = (-h/2 sin, h/2 cos).

Everything is fp32 with torch's operation order; cos/sin/log/exp/sqrt come from numpy's fp32 routines, which
may differ from torch's (Sleef) and the GPU's by an ulp: kept-index parity is therefore only asserted where no
candidate's decisive IoU (its maximum over the higher-scored boxes) lies within 2e-5 of the threshold
(`nms_rotated` returns that margin).
"""

from __future__ import annotations

import math

import numpy as np

F32 = np.float32
EPS = F32(1e-7)


def covariance(wh_t: np.ndarray):
    """(n,3) [w,h,angle] -> a, b, c (n,) fp32 (ultralytics _get_covariance_matrix)."""
    w2 = (wh_t[:, 0] * wh_t[:, 0] / F32(12)).astype(F32)
    h2 = (wh_t[:, 1] * wh_t[:, 1] / F32(12)).astype(F32)
    cos, sin = np.cos(wh_t[:, 2]).astype(F32), np.sin(wh_t[:, 2]).astype(F32)
    cos2, sin2 = (cos * cos).astype(F32), (sin * sin).astype(F32)
    a = (w2 * cos2 + h2 * sin2).astype(F32)
    b = (w2 * sin2 + h2 * cos2).astype(F32)
    c = ((w2 - h2) * cos * sin).astype(F32)
    return a, b, c


def batch_probiou(obb1: np.ndarray, obb2: np.ndarray) -> np.ndarray:
    """(N,5) x (M,5) xywhr -> (N,M) fp32."""
    obb1, obb2 = obb1.astype(F32), obb2.astype(F32)
    x1, y1 = obb1[:, 0:1], obb1[:, 1:2]
    x2, y2 = obb2[None, :, 0], obb2[None, :, 1]
    a1, b1, c1 = (v[:, None] for v in covariance(obb1[:, 2:5]))
    a2, b2, c2 = (v[None, :] for v in covariance(obb2[:, 2:5]))
    sa, sb, sc = (a1 + a2).astype(F32), (b1 + b2).astype(F32), (c1 + c2).astype(F32)
    det = (sa * sb - sc * sc).astype(F32)
    den = (det + EPS).astype(F32)
    dy, dx = (y1 - y2).astype(F32), (x1 - x2).astype(F32)
    t1 = ((sa * (dy * dy) + sb * (dx * dx)) / den * F32(0.25)).astype(F32)
    t2 = ((sc * (x2 - x1) * (y1 - y2)) / den * F32(0.5)).astype(F32)
    d1 = np.maximum(a1 * b1 - c1 * c1, F32(0)).astype(F32)
    d2 = np.maximum(a2 * b2 - c2 * c2, F32(0)).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        t3 = (np.log((det / (F32(4) * np.sqrt(d1 * d2).astype(F32) + EPS) + EPS).astype(F32)).astype(F32) * F32(0.5)).astype(F32)
    bd = np.clip((t1 + t2 + t3).astype(F32), EPS, F32(100.0))
    hd = np.sqrt((F32(1.0) - np.exp(-bd).astype(F32) + EPS).astype(F32)).astype(F32)
    return (F32(1) - hd).astype(F32)


def nms_rotated(boxes: np.ndarray, scores: np.ndarray, threshold: float, max_det=None):
    """-> (kept indices in score order, cut to max_det; margin = min_j |max_i<j iou(i,j) - threshold| over the
    candidates up to the last kept one: how far the closest decision that matters is from flipping)."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64), np.inf
    order = np.argsort(-scores.astype(F32), kind="stable")
    b = boxes[order]
    ious = np.triu(batch_probiou(b, b), k=1)
    worst = ious.max(axis=0)
    thr = F32(threshold)
    pick = np.nonzero(worst < thr)[0]
    if max_det is not None:
        pick = pick[:max_det]
    last = int(pick[-1]) if len(pick) else 0
    if max_det is None or len(pick) < max_det:
        last = n - 1          # every candidate was decided
    # a NaN maximum (the formula's log of a negative determinant ratio, degenerate boxes only) is a drop whatever the
    # threshold (`NaN < thr` is False in torch too): it has no margin
    gaps = np.abs(worst[1:last + 1].astype(np.float64) - float(thr)) if last >= 1 else np.zeros(0)
    gaps = gaps[~np.isnan(gaps)]
    margin = float(gaps.min()) if len(gaps) else np.inf
    return order[pick], margin


def non_max_suppression_rotated(prediction, conf_thres=0.25, iou_thres=0.45, agnostic=False, max_det=300, nc=0,
                                max_nms=30000, max_wh=7680):
    """prediction (B, 4+nc+1, A) -> list of (rows (n, 7) [x,y,w,h,conf,cls,angle], kept anchors, margin)."""
    prediction = np.asarray(prediction, dtype=F32)
    bs, ch, _ = prediction.shape
    nc = nc or (ch - 4)
    nm = ch - nc - 4
    mi = 4 + nc
    thr = F32(conf_thres)
    out = []
    for xi in range(bs):
        p = prediction[xi]
        cand = np.nonzero(p[4:mi].max(axis=0) > thr)[0]
        x = np.ascontiguousarray(p[:, cand].T)
        cls = x[:, 4:mi]
        j = cls.argmax(axis=1) if len(x) else np.zeros((0,), np.int64)
        conf = cls[np.arange(cls.shape[0]), j] if len(x) else np.zeros((0,), F32)
        rows = np.concatenate([x[:, :4], conf[:, None], j[:, None].astype(F32), x[:, mi:]], axis=1).astype(F32)
        sel = conf > thr
        rows, anchors = rows[sel], cand[sel]
        if rows.shape[0] == 0:
            out.append((np.zeros((0, 6 + nm), F32), np.zeros((0,), np.int64), np.inf))
            continue
        if rows.shape[0] > max_nms:
            top = np.argsort(-rows[:, 4], kind="stable")[:max_nms]
            rows, anchors = rows[top], anchors[top]
        c = rows[:, 5:6] * F32(0 if agnostic else max_wh)
        boxes = np.concatenate([rows[:, :2] + c, rows[:, 2:4], rows[:, -1:]], axis=1).astype(F32)
        keep, margin = nms_rotated(boxes, rows[:, 4], iou_thres, max_det)
        out.append((rows[keep], anchors[keep], margin))
    return out


def regularize_rboxes(rboxes: np.ndarray) -> np.ndarray:
    x, y, w, h, t = (rboxes[:, i].astype(F32) for i in range(5))
    swap = ~(w > h)
    w_ = np.where(swap, h, w)
    h_ = np.where(swap, w, h)
    t = np.where(swap, (t + F32(math.pi / 2)).astype(F32), t)
    t = np.remainder(t, F32(math.pi)).astype(F32)   # floored remainder, like torch's %
    return np.stack([x, y, w_, h_, t], axis=-1).astype(F32)


def scale_boxes_xywh(img1_shape, boxes: np.ndarray, img0_shape) -> np.ndarray:
    h1, w1 = img1_shape[:2]
    h0, w0 = img0_shape[:2]
    gain = min(h1 / h0, w1 / w0)
    pad = round((w1 - w0 * gain) / 2 - 0.1), round((h1 - h0 * gain) / 2 - 0.1)
    b = np.array(boxes[:, :4], dtype=F32, copy=True)
    b[:, 0] = b[:, 0] - F32(pad[0])
    b[:, 1] = b[:, 1] - F32(pad[1])
    b = (b / F32(gain)).astype(F32)
    b[:, 0] = np.clip(b[:, 0], F32(0), F32(w0))
    b[:, 1] = np.clip(b[:, 1], F32(0), F32(h0))
    b[:, 2] = np.clip(b[:, 2], F32(0), F32(w0))
    b[:, 3] = np.clip(b[:, 3], F32(0), F32(h0))
    return b


def xywhr2xyxyxyxy(x: np.ndarray) -> np.ndarray:
    """(n,5) -> (n,4,2)."""
    ctr = x[:, :2].astype(F32)
    w, h, ang = x[:, 2:3].astype(F32), x[:, 3:4].astype(F32), x[:, 4:5].astype(F32)
    cos, sin = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    hw, hh = (w / F32(2)).astype(F32), (h / F32(2)).astype(F32)
    v1 = np.concatenate([hw * cos, hw * sin], axis=-1).astype(F32)
    v2 = np.concatenate([-hh * sin, hh * cos], axis=-1).astype(F32)
    return np.stack([ctr + v1 + v2, ctr + v1 - v2, ctr - v1 - v2, ctr - v1 + v2], axis=-2).astype(F32)


def postprocess_obb(rows: np.ndarray, model_hw, orig_shape) -> np.ndarray:
    """NMS rows (n,7) -> corner boxes (n,4,2) as Yolov8Obb.postprocess builds them (model.py:436-446)."""
    r = regularize_rboxes(np.concatenate([rows[:, :4], rows[:, -1:]], axis=1))
    r[:, :4] = scale_boxes_xywh(model_hw, r[:, :4], orig_shape)
    return xywhr2xyxyxyxy(r)
