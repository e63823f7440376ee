"""Oracle `nms` / `boxes` cross-checks.  ultralytics / torchvision are not installed: PARITY
UNPINNED against the binaries.  Checked here: hand-built cases with exact answers (IoU exactly
at the threshold is kept, class offset separates identical boxes, agnostic merges them, max_det
truncation, class >= 69 coordinate quantisation, score ties keep anchor order), an independent
O(n^2) scalar greedy NMS, and the defining properties of a greedy NMS result."""
import numpy as np
import pytest
import torch

from oracle import boxes as B
from oracle import nms as N

F32 = np.float32


def _head(boxes_xywh, cls_scores, nc, nm=0, A=None, coef=None):
    """Build a (1, 4+nc+nm, A) head with the given rows on the first anchors, background elsewhere."""
    n = len(boxes_xywh)
    A = A or n
    p = np.zeros((1, 4 + nc + nm, A), F32)
    p[0, :4, :n] = np.asarray(boxes_xywh, F32).T
    p[0, 2:4, n:] = 1.0
    for i, sc in enumerate(cls_scores):
        for k, v in sc.items():
            p[0, 4 + k, i] = v
    if nm:
        p[0, 4 + nc:, :] = coef if coef is not None else np.arange(nm * A, dtype=F32).reshape(nm, A)
    return p


def test_iou_exactly_at_threshold_is_kept():
    # boxes [0,0,10,10] and [0,0,10,5]: inter 50, union 100, IoU = 0.5 exactly in fp32
    p = _head([[5, 5, 10, 10], [5, 2.5, 10, 5]], [{0: 0.9}, {0: 0.8}], nc=2)
    (det, idx), = N.non_max_suppression(p, 0.25, 0.5, nc=2)
    assert idx.tolist() == [0, 1]
    (det, idx), = N.non_max_suppression(p, 0.25, 0.49, nc=2)
    assert idx.tolist() == [0]
    assert det[0].tolist() == [0, 0, 10, 10, F32(0.9), 0]


def test_class_offset_and_agnostic():
    p = _head([[50, 50, 20, 20], [50, 50, 20, 20], [50, 50, 20, 20]], [{0: 0.9}, {1: 0.8}, {0: 0.7}], nc=3)
    (det, idx), = N.non_max_suppression(p, 0.25, 0.45, agnostic=False, nc=3)
    assert idx.tolist() == [0, 1] and det[:, 5].tolist() == [0, 1]
    (det, idx), = N.non_max_suppression(p, 0.25, 0.45, agnostic=True, nc=3)
    assert idx.tolist() == [0]


def test_strict_conf_and_first_max_class():
    thr = 0.25
    p = _head([[10, 10, 4, 4], [30, 30, 4, 4], [60, 60, 4, 4]], [{0: thr}, {1: 0.5, 2: 0.5}, {2: np.nextafter(F32(thr), F32(1))}], nc=3)
    (det, idx), = N.non_max_suppression(p, thr, 0.45, nc=3)
    assert idx.tolist() == [1, 2]          # conf == thr is dropped (strict >, compared in fp32)
    assert det[0, 5] == 1                  # ties between classes -> first maximal index


def test_max_det_and_score_order_and_ties():
    n = 12
    boxes = [[20 * i + 5, 5, 8, 8] for i in range(n)]
    scores = [{0: 0.5}] * 4 + [{0: 0.9 - 0.01 * i} for i in range(8)]
    p = _head(boxes, scores, nc=1)
    (det, idx), = N.non_max_suppression(p, 0.25, 0.45, nc=1, max_det=6)
    assert idx.tolist() == [4, 5, 6, 7, 8, 9]
    (det, idx), = N.non_max_suppression(p, 0.25, 0.45, nc=1, max_det=300)
    assert idx.tolist() == [4, 5, 6, 7, 8, 9, 10, 11, 0, 1, 2, 3]   # equal scores keep anchor order


def test_class_offset_quantisation_changes_iou():
    # class 79 offset = 606720: fp32 ulp there is 1/16 px, so sub-1/16 px detail is lost and two
    # boxes that differ only below that resolution behave as identical (IoU 1) for class 79 while
    # class 0 sees their true coordinates.
    b = [[100.02, 100.02, 0.05, 0.05], [100.05, 100.05, 0.05, 0.05]]
    for k, expect in ((0, [0, 1]), (79, None)):
        p = _head(b, [{k: 0.9}, {k: 0.8}], nc=80)
        (det, idx), = N.non_max_suppression(p, 0.25, 0.45, nc=80)
        xy = N.xywh2xyxy(np.asarray(b, F32)) + F32(k * 7680)
        ref = N.nms_greedy(xy, np.array([0.9, 0.8], F32), 0.45)
        assert idx.tolist() == ref.tolist()
        if expect:
            assert idx.tolist() == expect


def _greedy_scalar(boxes, scores, thr):
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    keep, dead = [], set()
    for a, i in enumerate(order):
        if i in dead:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if j in dead:
                continue
            xx1, yy1 = max(boxes[i][0], boxes[j][0]), max(boxes[i][1], boxes[j][1])
            xx2, yy2 = min(boxes[i][2], boxes[j][2]), min(boxes[i][3], boxes[j][3])
            w, h = max(F32(0), F32(xx2 - xx1)), max(F32(0), F32(yy2 - yy1))
            inter = F32(w * h)
            ai = F32(F32(boxes[i][2] - boxes[i][0]) * F32(boxes[i][3] - boxes[i][1]))
            aj = F32(F32(boxes[j][2] - boxes[j][0]) * F32(boxes[j][3] - boxes[j][1]))
            den = F32(F32(ai + aj) - inter)
            if den != 0 and float(F32(inter / den)) > thr:
                dead.add(j)
    return keep


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_vectorised_equals_scalar_greedy(seed):
    rng = np.random.default_rng(seed)
    n = 150
    c = rng.random((n, 2)).astype(F32) * 100
    wh = (rng.random((n, 2)).astype(F32) * 30 + 2)
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(F32)
    scores = rng.random(n).astype(F32)
    assert N.nms_greedy(boxes, scores, 0.45).tolist() == _greedy_scalar(boxes, scores, 0.45)


def synth_head(rng, B_, nc, nm, A, S, objects, conf_lo=0.3):
    """Clusters of jittered anchors around planted objects on a low background."""
    p = np.zeros((B_, 4 + nc + nm, A), F32)
    p[:, 0:2] = rng.random((B_, 2, A), dtype=F32) * S
    p[:, 2:4] = np.exp(rng.uniform(np.log(8), np.log(S / 2), (B_, 2, A))).astype(F32)
    p[:, 4:4 + nc] = (1 / (1 + np.exp(-rng.normal(-6, 1, (B_, nc, A))))).astype(F32)
    if nm:
        p[:, 4 + nc:] = rng.normal(0, 0.5, (B_, nm, A)).astype(F32)
    for b in range(B_):
        for _ in range(objects):
            k = int(rng.integers(5, 16))
            a = rng.choice(A, k, replace=False)
            cx, cy = rng.random(2) * S
            w, h = np.exp(rng.uniform(np.log(16), np.log(S / 2), 2))
            cls = int(rng.integers(0, nc))
            p[b, 0, a] = cx + rng.normal(0, 2, k)
            p[b, 1, a] = cy + rng.normal(0, 2, k)
            p[b, 2, a] = w * np.exp(rng.normal(0, 0.05, k))
            p[b, 3, a] = h * np.exp(rng.normal(0, 0.05, k))
            p[b, 4 + cls, a] = rng.uniform(conf_lo, 0.95) * rng.uniform(0.7, 1.0, k)
    return p


@pytest.mark.parametrize("agnostic", [False, True])
def test_greedy_properties_on_synthetic_head(agnostic):
    rng = np.random.default_rng(11)
    nc, A = 80, 2100
    p = synth_head(rng, 2, nc, 0, A, 320, objects=25)
    for b, (det, idx) in enumerate(N.non_max_suppression(p, 0.25, 0.45, agnostic=agnostic, nc=nc)):
        assert len(idx) > 5
        assert np.all(np.diff(det[:, 4]) <= 0)                       # score order
        # rows carry the un-offset boxes and the raw scores of their anchors
        xy = N.xywh2xyxy(p[b, :4, idx].reshape(-1, 4)) if False else N.xywh2xyxy(p[b, :4, :].T[idx])
        assert np.array_equal(det[:, :4], xy)
        assert np.array_equal(det[:, 4], p[b, 4:, :].max(0)[idx])
        # kept set is independent under IoU > thr (within a class unless agnostic)
        off = det[:, 5:6] * F32(0 if agnostic else 7680)
        bb = det[:, :4] + off
        for i in range(len(bb)):
            later = N.nms_greedy(np.vstack([bb[i], bb[i + 1:]]), np.arange(len(bb) - i, 0, -1).astype(F32), 0.45)
            assert len(later) == len(bb) - i
        # every candidate that was not kept overlaps a kept box with a higher (or equal, earlier) score
        cand = np.nonzero(p[b, 4:].max(0) > F32(0.25))[0]
        rest = np.setdiff1d(cand, idx)
        if len(idx) < 300:
            cb = N.xywh2xyxy(p[b, :4, :].T[rest]) + (p[b, 4:, rest].argmax(1)[:, None].astype(F32) * F32(0 if agnostic else 7680))
            for r, bx in zip(rest, cb):
                sc = p[b, 4:, r].max()
                hit = False
                for kb, ks, ka in zip(bb, det[:, 4], idx):
                    if ks > sc or (ks == sc and ka < r):
                        if len(N.nms_greedy(np.vstack([kb, bx]), np.array([2, 1], F32), 0.45)) == 1:
                            hit = True
                            break
                assert hit


def test_empty_and_zero_candidates():
    p = _head([[5, 5, 2, 2]], [{0: 0.1}], nc=2, nm=3)
    (det, idx), = N.non_max_suppression(p, 0.25, 0.45, nc=2)
    assert det.shape == (0, 9) and idx.shape == (0,)


def test_mask_coefficients_travel_with_rows():
    p = _head([[10, 10, 4, 4], [50, 50, 4, 4]], [{1: 0.6}, {0: 0.7}], nc=2, nm=4)
    (det, idx), = N.non_max_suppression(p, 0.25, 0.45, nc=2)
    assert idx.tolist() == [1, 0]
    assert np.array_equal(det[:, 6:], p[0, 6:, :].T[[1, 0]])


# ---- boxes ---------------------------------------------------------------------------------

def _scale_boxes_torch(img1_shape, boxes, img0_shape):
    """The 8.3.x op sequence executed by torch-CPU (fp32 tensor, python scalars)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    b = torch.from_numpy(boxes.copy())
    b[..., 0] -= pad[0]
    b[..., 1] -= pad[1]
    b[..., 2] -= pad[0]
    b[..., 3] -= pad[1]
    b[..., :4] /= gain
    b[..., 0] = b[..., 0].clamp(0, img0_shape[1])
    b[..., 1] = b[..., 1].clamp(0, img0_shape[0])
    b[..., 2] = b[..., 2].clamp(0, img0_shape[1])
    b[..., 3] = b[..., 3].clamp(0, img0_shape[0])
    return b.numpy()


@pytest.mark.parametrize("h0,w0,S", [(1080, 1920, 640), (2160, 3840, 640), (427, 640, 640), (375, 500, 640),
                                     (1080, 1920, 1280), (640, 640, 640), (889, 1333, 640)])
def test_scale_boxes_matches_torch_ops(h0, w0, S):
    rng = np.random.default_rng(h0)
    b = (rng.random((500, 4)).astype(F32) * (S + 40) - 20).astype(F32)
    got = B.scale_boxes((S, S), b, (h0, w0, 3))
    assert np.array_equal(got, _scale_boxes_torch((S, S), b, (h0, w0, 3)))
    assert got[:, [0, 2]].min() >= 0 and got[:, [0, 2]].max() <= w0
    assert got[:, [1, 3]].min() >= 0 and got[:, [1, 3]].max() <= h0


def test_class_filter_float64_compare_and_dict():
    names = {0: "a", 1: "b", 2: "c"}
    conf = {"a": 0.3, "b": 0.6}
    assert B.min_conf(conf, names) == 0.3
    assert B.min_conf(0.5, names) == 0.5
    with pytest.raises(TypeError):
        B.min_conf(1, names)
    thr = B.class_thresholds(conf, names, 3)
    assert thr.tolist() == [0.3, 0.6, 1.0]
    confs = np.array([0.3, 0.30000004, 0.7, 0.99], F32)    # fp32(0.3) = 0.30000001192 > 0.3 (double)
    cls = np.array([0, 0, 1, 2], F32)
    m = B.class_filter(confs, cls, thr)
    ref = torch.from_numpy(confs) > torch.tensor(thr[cls.astype(int)])
    assert m.tolist() == ref.tolist() == [True, True, True, False]


def _scale_coords_torch(img1_shape, coords, img0_shape):
    """ops.scale_coords' 8.3.x op sequence executed by torch-CPU (fp32 tensor, python scalars, pads not rounded)."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    c = torch.from_numpy(coords.copy())
    c[..., 0] -= pad[0]
    c[..., 1] -= pad[1]
    c[..., 0] /= gain
    c[..., 1] /= gain
    c[..., 0] = c[..., 0].clamp(0, img0_shape[1])
    c[..., 1] = c[..., 1].clamp(0, img0_shape[0])
    return c.numpy()


@pytest.mark.parametrize("h0,w0,S,ndim", [(1080, 1920, 640, 3), (427, 640, 640, 2), (375, 500, 640, 3), (889, 1333, 640, 3), (640, 640, 640, 2)])
def test_scale_coords_matches_torch_ops(h0, w0, S, ndim):
    rng = np.random.default_rng(w0)
    k = (rng.random((40, 17, ndim)).astype(F32) * (S + 40) - 20).astype(F32)
    got = B.scale_coords((S, S), k, (h0, w0, 3))
    assert np.array_equal(got, _scale_coords_torch((S, S), k, (h0, w0, 3)))
    if ndim == 3:
        assert np.array_equal(got[..., 2], k[..., 2])
