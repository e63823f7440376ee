"""Oracle: `ultralytics.utils.ops.non_max_suppression` as called at
object_detectors/yolov8_lmi/model.py:212, and the `torchvision.ops.nms` it
reaches.  TEST INFRASTRUCTURE ONLY.

Neither package is installed or vendored (ultralytics is unpinned,
tests/dockerfile:12; torchvision = whatever nvcr.io/nvidia/pytorch:24.04-py3
ships): PARITY UNPINNED against those binaries.  The restatement follows the
YOLOv8 (8.3.x) semantics listed in SURVEY.md 8(a) row a4 and, for everything
that is shared with YOLOv5, the in-tree copy
object_detectors/yolov5_lmi/trt/yolov5_trt.py:172-282 (skeleton, class-offset
trick `boxes + cls*max_wh` :265-266, `max_det` cap :268-269), :306-313
(xywh2xyxy).  v5 -> v8 deltas applied: channel-major input (B, 4+nc+nm, A); no
objectness column (conf = best class score); candidate test
`cls.amax(1) > conf_thres` (strict, compared in fp32); no pre-sort unless
n > max_nms.

Greedy NMS = torchvision CPU kernel (`nms_kernel_impl`): order = stable
descending sort of the scores; area = (x2-x1)*(y2-y1); for each not-yet-
suppressed i in order keep i and suppress every later j with
  inter / (area_i + area_j - inter) > iou_threshold      (strict, no eps; the
  fp32 quotient is compared against the threshold as a double).
Tie rule (torch.sort stable=True): equal scores keep their input (= anchor)
order.  The wall-clock `time_limit` escape of the upstream function is
non-deterministic and deliberately not reproduced.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32


def xywh2xyxy(x: np.ndarray) -> np.ndarray:
    """(.., 4) cx,cy,w,h -> x1,y1,x2,y2 in fp32: xy -+ wh/2 (yolov5_trt.py:306-313)."""
    x = x.astype(F32, copy=False)
    half = x[..., 2:4] / F32(2)
    out = np.empty_like(x)
    out[..., 0:2] = x[..., 0:2] - half
    out[..., 2:4] = x[..., 0:2] + half
    return out


def box_iou(box1: np.ndarray, box2: np.ndarray, eps: float = 1e-7) -> np.ndarray:
    """(N,4) x (M,4) xyxy -> (N,M) fp32 IoU, the in-tree `box_iou` (yolov5_trt.py:285-303): inter =
    clamp(min(a2,b2) - max(a1,b1), 0).prod; inter / (area1 + area2 - inter + eps), every step in fp32.  With
    eps = 0 this is the quotient `nms_greedy` thresholds (torchvision's kernel has no eps)."""
    a = box1.astype(F32, copy=False)[:, None, :]
    b = box2.astype(F32, copy=False)[None, :, :]
    wh = np.maximum(np.minimum(a[..., 2:4], b[..., 2:4]) - np.maximum(a[..., 0:2], b[..., 0:2]), F32(0))
    inter = wh[..., 0] * wh[..., 1]
    area1 = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area2 = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area1 + area2 - inter + F32(eps))


def nms_greedy(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """torchvision.ops.nms (CPU kernel) -> kept indices in score order."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    boxes = boxes.astype(F32, copy=False)
    order = np.argsort(-scores.astype(F32), kind="stable")
    b = boxes[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = float(iou_thres)
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(n):
            if suppressed[i]:
                continue
            keep.append(order[i])
            if i + 1 == n:
                break
            xx1 = np.maximum(x1[i], x1[i + 1 :])
            yy1 = np.maximum(y1[i], y1[i + 1 :])
            xx2 = np.minimum(x2[i], x2[i + 1 :])
            yy2 = np.minimum(y2[i], y2[i + 1 :])
            w = np.maximum(F32(0), xx2 - xx1)
            h = np.maximum(F32(0), yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[i + 1 :] - inter)
            suppressed[i + 1 :] |= ovr.astype(np.float64) > thr
    return np.asarray(keep, dtype=np.int64)


def decode_candidates(p: np.ndarray, conf_thres: float, nc: int, agnostic: bool = False, max_nms: int = 30000,
                      max_wh: int = 7680):
    """One image's head p (4+nc+nm, A) -> what the reference hands to `torchvision.ops.nms` and indexes afterwards
    (yolov5_trt.py:243-266): (rows (n, 6+nm) sorted by descending score, anchors (n,), class-offset boxes (n,4)).
    Pinned by tests/golden/yolov5_helpers.npz (the arguments the reference's own decode passes to the NMS call)."""
    mi = 4 + nc
    thr = F32(conf_thres)
    cand = np.nonzero(p[4:mi].max(axis=0) > thr)[0]
    x = np.ascontiguousarray(p[:, cand].T)
    cls = x[:, 4:mi]
    j = cls.argmax(axis=1) if len(x) else np.zeros((0,), np.int64)
    conf = cls[np.arange(cls.shape[0]), j]
    rows = np.concatenate([xywh2xyxy(x[:, :4]), conf[:, None], j[:, None].astype(F32), x[:, mi:]], axis=1)
    sel = conf > thr
    rows, anchors = rows[sel], cand[sel]
    order = np.argsort(-rows[:, 4], kind="stable")[:max_nms]
    rows, anchors = rows[order], anchors[order]
    c = rows[:, 5:6] * F32(0 if agnostic else max_wh)
    return rows, anchors, rows[:, :4] + c


def non_max_suppression(
    prediction: np.ndarray,
    conf_thres: float = 0.25,
    iou_thres: float = 0.45,
    agnostic: bool = False,
    max_det: int = 300,
    nc: int = 0,
    max_nms: int = 30000,
    max_wh: int = 7680,
):
    """prediction (B, 4+nc+nm, A) fp32 -> list over images of
    (det (n, 6+nm) fp32 [x1,y1,x2,y2,conf,cls,coef...], kept_anchor (n,) int64)."""
    prediction = np.asarray(prediction, dtype=F32)
    bs, ch, _ = prediction.shape
    nc = nc or (ch - 4)
    nm = ch - nc - 4
    mi = 4 + nc
    thr = F32(conf_thres)  # tensor(fp32) > python float compares in fp32
    out = []
    for xi in range(bs):
        p = prediction[xi]  # (C, A)
        cand = np.nonzero(p[4:mi].max(axis=0) > thr)[0]
        if cand.size == 0:
            out.append((np.zeros((0, 6 + nm), F32), np.zeros((0,), np.int64)))
            continue
        x = np.ascontiguousarray(p[:, cand].T)  # (n, C) in anchor order
        box = xywh2xyxy(x[:, :4])
        cls = x[:, 4:mi]
        j = cls.argmax(axis=1)  # first maximal index, like torch.max(dim)
        conf = cls[np.arange(cls.shape[0]), j]
        rows = np.concatenate([box, conf[:, None], j[:, None].astype(F32), x[:, mi:]], axis=1)
        sel = conf > thr
        rows, anchors = rows[sel], cand[sel]
        n = rows.shape[0]
        if n == 0:
            out.append((np.zeros((0, 6 + nm), F32), np.zeros((0,), np.int64)))
            continue
        if n > max_nms:
            top = np.argsort(-rows[:, 4], kind="stable")[:max_nms]
            rows, anchors = rows[top], anchors[top]
        c = rows[:, 5:6] * F32(0 if agnostic else max_wh)
        boxes = rows[:, :4] + c  # fp32 add: class k quantises coordinates near k*7680
        keep = nms_greedy(boxes, rows[:, 4], iou_thres)[:max_det]
        out.append((rows[keep], anchors[keep]))
    return out
