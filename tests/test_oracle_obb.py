"""CPU checks of oracle/obb.py (PARITY UNPINNED against ultralytics: not installed).  The numpy restatement is
compared with torch-CPU executing the same published op sequences, plus analytic properties of probiou."""
import math

import numpy as np
import pytest
import torch

from oracle import obb as O

F32 = np.float32


def _cov_t(b):
    g = torch.cat((b[:, 2:4].pow(2) / 12, b[:, 4:]), dim=-1)
    a, bb, c = g.split(1, dim=-1)
    cos, sin = c.cos(), c.sin()
    cos2, sin2 = cos.pow(2), sin.pow(2)
    return a * cos2 + bb * sin2, a * sin2 + bb * cos2, (a - bb) * cos * sin


def _probiou_t(o1, o2, eps=1e-7):
    x1, y1 = o1[..., :2].split(1, dim=-1)
    x2, y2 = (x.squeeze(-1)[None] for x in o2[..., :2].split(1, dim=-1))
    a1, b1, c1 = _cov_t(o1)
    a2, b2, c2 = (x.squeeze(-1)[None] for x in _cov_t(o2))
    t1 = (((a1 + a2) * (y1 - y2).pow(2) + (b1 + b2) * (x1 - x2).pow(2)) / ((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2) + eps)) * 0.25
    t2 = (((c1 + c2) * (x2 - x1) * (y1 - y2)) / ((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2) + eps)) * 0.5
    t3 = (((a1 + a2) * (b1 + b2) - (c1 + c2).pow(2)) / (4 * ((a1 * b1 - c1.pow(2)).clamp_(0) * (a2 * b2 - c2.pow(2)).clamp_(0)).sqrt() + eps) + eps).log() * 0.5
    bd = (t1 + t2 + t3).clamp(eps, 100.0)
    hd = (1.0 - (-bd).exp() + eps).sqrt()
    return 1 - hd


def rboxes(rng, n, size=640):
    b = np.empty((n, 5), F32)
    b[:, :2] = rng.random((n, 2)) * size
    b[:, 2:4] = np.exp(rng.uniform(np.log(8), np.log(size / 3), (n, 2)))
    b[:, 4] = rng.uniform(-math.pi / 4, 3 * math.pi / 4, n)
    return b


def test_probiou_matches_torch_op_sequence():
    rng = np.random.default_rng(0)
    b = rboxes(rng, 200, 200)
    got = O.batch_probiou(b, b)
    ref = _probiou_t(torch.from_numpy(b), torch.from_numpy(b)).numpy()
    lo = ref < 0.98   # near IoU 1 the sqrt(1 - exp(-bd) + eps) amplifies an ulp of bd ~ eps to ~1e-4
    assert np.abs(got - ref)[lo].max() <= 4e-6      # cos/sin/log/exp of numpy vs torch differ by an ulp at most
    assert np.abs(got - ref)[~lo].max() <= 1e-3
    assert np.abs(np.diag(got) - 1.0).max() < 1e-3   # identical boxes: IoU ~ 1 (up to sqrt(eps))
    far = O.batch_probiou(np.array([[0, 0, 10, 10, 0]], F32), np.array([[500, 500, 10, 10, 0.3]], F32))
    assert far[0, 0] < 1e-3
    # invariances: a square ignores its angle; (w, h, t) and (h, w, t + pi/2) are the same ellipse
    sq = O.batch_probiou(np.array([[50, 50, 20, 20, 0.0]], F32), np.array([[50, 50, 20, 20, 1.0]], F32))
    assert abs(sq[0, 0] - 1) < 1e-3
    r = O.batch_probiou(np.array([[50, 50, 40, 10, 0.4]], F32), np.array([[50, 50, 10, 40, 0.4 + math.pi / 2]], F32))
    assert abs(r[0, 0] - 1) < 1e-3


def test_nms_rotated_is_not_greedy():
    """A (0.9) overlaps B (0.8), B overlaps C (0.7), A does not overlap C.  Greedy NMS keeps A and C; ultralytics'
    nms_rotated drops C as well, because ANY higher-scored box counts, kept or not."""
    boxes = np.array([[100, 100, 80, 20, 0.0], [130, 100, 80, 20, 0.0], [160, 100, 80, 20, 0.0]], F32)
    scores = np.array([0.9, 0.8, 0.7], F32)
    iou = O.batch_probiou(boxes, boxes)
    thr = float((iou[0, 2] + iou[0, 1]) / 2)
    assert iou[0, 1] > thr and iou[1, 2] > thr and iou[0, 2] < thr
    keep, margin = O.nms_rotated(boxes, scores, thr)
    assert keep.tolist() == [0] and margin > 1e-3
    keep, _ = O.nms_rotated(boxes[::-1].copy(), scores[::-1].copy(), thr)
    assert keep.tolist() == [2]                      # indices refer to the input order


def test_regularize_scale_and_corners_match_torch_ops():
    rng = np.random.default_rng(1)
    b = rboxes(rng, 300)
    t = torch.from_numpy(b.copy())
    x, y, w, h, a = t.unbind(dim=-1)
    w_, h_ = torch.where(w > h, w, h), torch.where(w > h, h, w)
    a = torch.where(w > h, a, a + math.pi / 2) % math.pi
    ref = torch.stack([x, y, w_, h_, a], dim=-1).numpy()
    got = O.regularize_rboxes(b)
    assert np.array_equal(got, ref)
    assert (got[:, 2] >= got[:, 3]).all() and (got[:, 4] >= 0).all() and (got[:, 4] < math.pi + 1e-6).all()
    # scale_boxes(xywh=True) on the regularised boxes
    S, h0, w0 = 640, 1080, 1920
    gain = min(S / h0, S / w0)
    pad = round((S - w0 * gain) / 2 - 0.1), round((S - h0 * gain) / 2 - 0.1)
    tb = torch.from_numpy(got[:, :4].copy())
    tb[..., 0] -= pad[0]
    tb[..., 1] -= pad[1]
    tb[..., :4] /= gain
    tb[..., 0] = tb[..., 0].clamp(0, w0)
    tb[..., 1] = tb[..., 1].clamp(0, h0)
    tb[..., 2] = tb[..., 2].clamp(0, w0)
    tb[..., 3] = tb[..., 3].clamp(0, h0)
    assert np.array_equal(O.scale_boxes_xywh((S, S), got[:, :4], (h0, w0, 3)), tb.numpy())
    # corners
    tx = torch.from_numpy(got.copy())
    ctr = tx[..., :2]
    ww, hh, ang = (tx[..., i:i + 1] for i in range(2, 5))
    cv, sv = torch.cos(ang), torch.sin(ang)
    v1 = torch.cat([ww / 2 * cv, ww / 2 * sv], -1)
    v2 = torch.cat([-hh / 2 * sv, hh / 2 * cv], -1)
    refc = torch.stack([ctr + v1 + v2, ctr + v1 - v2, ctr - v1 - v2, ctr - v1 + v2], -2).numpy()
    assert np.abs(O.xywhr2xyxyxyxy(got) - refc).max() <= 2.5e-4   # coordinates up to 640: two ulp


def test_nms_rotated_pipeline_shapes_and_class_offset():
    rng = np.random.default_rng(2)
    nc, A = 3, 400
    p = np.zeros((1, 4 + nc + 1, A), F32)
    p[0, :2] = rng.random((2, A)) * 300
    p[0, 2:4] = rng.uniform(10, 60, (2, A))
    p[0, 4:4 + nc] = rng.random((nc, A)) * 0.2
    p[0, -1] = rng.uniform(-0.7, 2.3, A)
    # two identical boxes of different classes survive per-class NMS, not agnostic NMS
    p[0, :4, 0] = p[0, :4, 1] = [150, 150, 50, 20]
    p[0, -1, 0] = p[0, -1, 1] = 0.3
    p[0, 4:4 + nc, 0] = [0.9, 0, 0]
    p[0, 4:4 + nc, 1] = [0, 0.8, 0]
    (rows, anchors, margin), = O.non_max_suppression_rotated(p, 0.5, 0.45, nc=nc)
    assert anchors.tolist() == [0, 1] and rows.shape == (2, 7) and rows[0, 5] == 0 and rows[1, 5] == 1
    (rows, anchors, _), = O.non_max_suppression_rotated(p, 0.5, 0.45, nc=nc, agnostic=True)
    assert anchors.tolist() == [0]
    (rows, anchors, _), = O.non_max_suppression_rotated(p, 0.95, 0.45, nc=nc)
    assert rows.shape == (0, 7)
