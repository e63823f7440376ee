"""GPU parity for oriented-box models (reference object_detectors/yolov8_lmi/model.py:391-531,
tests tests/object_detectors/yolov8_lmi/test_yolov8_model.py:213-256): lmi_nms_v8_rotated / lmi_finalize_obb against
oracle/obb.py (PARITY UNPINNED against ultralytics itself), and the `Yolov8Obb` drop-in end to end.
Kept indices are required to be identical whenever no decisive probiou lies within 2e-5 of the threshold (cos / sin /
log / exp differ by an ulp between numpy and the GPU); the seeds below satisfy that and the tests assert it."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "object_detectors_amd", "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)

from oracle import boxes as OB, geometry as OG, obb as OO  # noqa: E402
from object_detectors_amd import synth  # noqa: E402
from object_detectors_amd.nets.yolov8 import save_checkpoint  # noqa: E402

F32 = np.float32
NAMES = {0: "ship", 1: "plane", 2: "truck"}


def obb_head(rng, B, nc, size, objects):
    """synth.head with one extra channel, turned into an angle in [-pi/4, 3pi/4); clustered anchors share it."""
    p = synth.head(rng, B, nc, 1, size, objects=objects)
    p[:, -1] = (np.tanh(p[:, -1]) * 0.5 + 0.25) * math.pi
    return p


@pytest.mark.parametrize("agnostic", [False, True])
@pytest.mark.parametrize("seed,nc,size,objects,max_det", [(1, 3, 320, 25, 300), (2, 80, 640, 60, 300), (3, 15, 640, 120, 40)])
def test_rotated_nms_and_finalize(seed, nc, size, objects, max_det, agnostic):
    from object_detectors_amd import hip_ops as ops
    rng = np.random.default_rng(seed)
    B = 4
    p = obb_head(rng, B, nc, size, objects)
    det, count, anchor = ops.nms_v8(torch.from_numpy(p).cuda(), 0.25, 0.45, agnostic=agnostic, max_det=max_det, nc=nc, rotated=True)
    h0, w0 = 1080, 1920
    thres = torch.tensor(rng.uniform(0.2, 0.6, nc), dtype=torch.float64).cuda()
    fin = ops.finalize_obb(det, count, nc, (size, size), (h0, w0), thres)
    torch.cuda.synchronize()
    det_h, count_h, anchor_h = det.cpu().numpy(), count.cpu().numpy(), anchor.cpu().numpy()
    corners, keep = fin["corners"].cpu().numpy(), fin["keep"].cpu().numpy().astype(bool)
    ref = OO.non_max_suppression_rotated(p, 0.25, 0.45, agnostic=agnostic, max_det=max_det, nc=nc)
    for b in range(B):
        rows, ra, margin = ref[b]
        assert margin > 2e-5, "seed puts a decision on the threshold: pick another"
        n = int(count_h[b])
        assert anchor_h[b, :n].tolist() == ra.tolist()
        assert np.array_equal(det_h[b, :n], rows)                        # rows are copies of head values: bit-exact
        rc = OO.postprocess_obb(rows, (size, size), (h0, w0, 3))
        assert np.abs(corners[b, :n] - rc).max() <= 1e-3                  # boxes within 1e-3 px
        M = OB.class_filter(rows[:, 4], rows[:, 5], thres.cpu().numpy())
        assert keep[b, :n].tolist() == M.tolist() and int(fin["keep_count"][b]) == int(M.sum())
    assert count_h.max() <= max_det and (count_h.max() == max_det or max_det == 300)


def test_rotated_nms_not_greedy_and_errors():
    from object_detectors_amd import hip_ops as ops
    # A-B-C chain: greedy would keep A and C, nms_rotated keeps A only
    boxes = np.array([[100, 100, 80, 20, 0.0], [130, 100, 80, 20, 0.0], [160, 100, 80, 20, 0.0]], F32)
    iou = OO.batch_probiou(boxes, boxes)
    thr = float((iou[0, 2] + iou[0, 1]) / 2)
    p = np.zeros((1, 4 + 1 + 1, 64), F32)
    p[0, :4, :3] = boxes[:, :4].T
    p[0, 4, :3] = [0.9, 0.8, 0.7]
    det, count, anchor = ops.nms_v8(torch.from_numpy(p).cuda(), 0.5, thr, nc=1, rotated=True)
    assert count.item() == 1 and anchor[0, 0].item() == 0
    det, count, anchor = ops.nms_v8(torch.from_numpy(p).cuda(), 0.5, 0.999999, nc=1, rotated=True)
    assert count.item() == 3 and anchor[0, :3].tolist() == [0, 1, 2]
    # threshold 0: `ious.triu_(1).max(0) < 0` holds for nobody, not even the top candidate (the oracle agrees)
    det, count, anchor = ops.nms_v8(torch.from_numpy(p).cuda(), 0.5, 0.0, nc=1, rotated=True)
    assert count.item() == 0 and len(OO.nms_rotated(boxes, np.array([0.9, 0.8, 0.7], F32), 0.0)[0]) == 0
    with pytest.raises(ValueError):
        ops.nms_v8(torch.zeros((1, 5, 64), device="cuda"), 0.5, 0.45, nc=1, rotated=True)   # no angle channel


@pytest.fixture(scope="module")
def obb(tmp_path_factory):
    from object_detector import ObjectDetector
    import od_models  # noqa: F401
    from yolov8_lmi.model import Yolov8Obb
    path = str(tmp_path_factory.mktemp("o") / "obb.pt")
    save_checkpoint(path, "n", len(NAMES), names=NAMES, seed=5, task="obb")
    m = ObjectDetector(metadata=dict(version='v0', model_name='yolov8', task='obb', framework='ultralytics'), weights=path)
    assert isinstance(m, Yolov8Obb) and m.device.type == "cuda"
    for conv in m.model.head.cv3:
        conv[-1].bias.data.normal_(-2.0, 0.5)
        conv[-1].weight.data.mul_(8.0)
    return m


def test_obb_predict_against_oracle(obb):
    rng = np.random.default_rng(9)
    frame = synth.frames_u8(rng, 1, 256, 256)[0]
    operators = [{"resize": [256, 144, 640, 360]}, {"pad": [0, 0, 56, 56]}]
    im = obb.preprocess(frame)
    pred = obb.forward(im)[0].cpu().numpy()
    s = pred[0, 4:4 + len(NAMES)].max(0)
    conf = float((s.min() + 3 * s.max()) / 4)
    (rows, ra, margin), = OO.non_max_suppression_rotated(pred, conf, 0.4, nc=len(NAMES), max_det=50)
    assert len(rows) > 0 and margin > 2e-5
    rc = OO.postprocess_obb(rows, (256, 256), (256, 256, 3))
    ref = OG.revert_to_origin(rc.reshape(-1, 2), operators).reshape(-1, 4, 2)
    for as_tensor in (False, True):
        image = torch.from_numpy(frame).cuda() if as_tensor else frame
        out, tinfo = obb.predict(image, conf, operators=operators, iou=0.4, max_det=50)
        bx = out['boxes'].cpu().numpy() if as_tensor else out['boxes']
        assert bx.shape == ref.shape and np.abs(bx - ref).max() <= 1      # rounded pixels: an ulp before rounding may flip one
        assert (bx != ref).mean() < 0.01
        assert out['classes'] == [NAMES[int(c)] for c in rows[:, 5]]
        assert np.array_equal(out['scores'].cpu().numpy() if as_tensor else out['scores'], rows[:, 4])
    none, _ = obb.predict(frame, 0.9999999, operators=operators)
    assert len(none['boxes']) == 0


def test_rotated_nms_more_candidates_than_the_lds_sort_buffer():
    """4116 anchors, every one a candidate (> 4096): the sort and the covariance arrays fall back to global memory;
    max_det cuts the kept list at 300."""
    from object_detectors_amd import hip_ops as ops
    rng = np.random.default_rng(11)
    nc = 5
    p = synth.head(rng, 2, nc, 1, 448, objects=30)
    p[:, -1] = (np.tanh(p[:, -1]) * 0.5 + 0.25) * math.pi
    assert p.shape[2] == 4116
    det, count, anchor = ops.nms_v8(torch.from_numpy(p).cuda(), 0.0005, 0.3, nc=nc, max_det=300, rotated=True)
    torch.cuda.synchronize()
    ref = OO.non_max_suppression_rotated(p, 0.0005, 0.3, nc=nc, max_det=300)
    for b in range(2):
        rows, ra, margin = ref[b]
        assert margin > 2e-5, "seed puts a decision on the threshold: pick another"
        assert len(ra) == 300
        assert int(count[b]) == 300 and anchor[b].cpu().numpy().tolist() == ra.tolist()
        assert np.array_equal(det[b].cpu().numpy(), rows)


@pytest.mark.parametrize("agnostic", [False, True])
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_rotated_nms_far_pair_shortcut_is_exact(seed, agnostic):
    """The rotated pass skips the probiou formula for pairs it can prove to be below any threshold (nms.hip: distance
    against the covariance traces, well-conditioned boxes only).  Boxes that stress the proof: slivers and needle-thin
    boxes of aspect 1000:1 (ill-conditioned covariance: they must always take the formula), specks, huge boxes, and
    ordinary ones spread over several classes -- kept anchors identical to the oracle's dense evaluation of every
    pair.  (Exactly zero-width boxes and boxes of 1e-3 px are left out: for them the formula's log argument changes
    sign with the last bit of cos / sin, and numpy, torch-CPU and torch-GPU disagree among themselves.)"""
    from object_detectors_amd import hip_ops as ops
    rng = np.random.default_rng(seed)
    nc, A, size = 6, 768, 640
    p = np.zeros((1, 4 + nc + 1, A), F32)
    p[0, 0:2] = rng.uniform(0, size, (2, A))
    kind = rng.integers(0, 6, A)
    w = rng.uniform(8, 120, A)
    h = rng.uniform(8, 120, A)
    w[kind == 1], h[kind == 1] = 0.05, rng.uniform(5, 50, (kind == 1).sum())         # slivers
    w[kind == 2], h[kind == 2] = rng.uniform(100, 400, (kind == 2).sum()), rng.uniform(0.1, 0.4, (kind == 2).sum())   # needles
    w[kind == 3], h[kind == 3] = rng.uniform(1, 3, (kind == 3).sum()), rng.uniform(1, 3, (kind == 3).sum())   # specks
    w[kind == 4], h[kind == 4] = rng.uniform(400, 900, (kind == 4).sum()), rng.uniform(300, 800, (kind == 4).sum())    # huge
    p[0, 2], p[0, 3] = w, h
    p[0, 4:4 + nc] = 0.01
    score = rng.permutation(A).astype(F32) / A * 0.6 + 0.3                               # distinct, all above conf
    p[0, 4 + rng.integers(0, nc, A), np.arange(A)] = score
    p[0, -1] = rng.uniform(-math.pi / 4, 3 * math.pi / 4, A)
    det, count, anchor = ops.nms_v8(torch.from_numpy(p).cuda(), 0.25, 0.45, agnostic=agnostic, max_det=1000, nc=nc, rotated=True)
    (rows, ra, margin), = OO.non_max_suppression_rotated(p, 0.25, 0.45, agnostic=agnostic, max_det=1000, nc=nc)
    assert margin > 2e-5, "seed puts a decision on the threshold: pick another"
    n = int(count[0])
    assert 10 < n < A and anchor[0, :n].cpu().tolist() == ra.tolist()


def test_rotated_nms_any_number_of_workgroups_per_image():
    """The pair tests of an image are shared by as many workgroups as the batch leaves CUs for (16 for one image, one
    from 129 images on): every batch size gives each image the rows it gets when it is suppressed alone, also when the
    images of a batch differ in their candidate counts (one with none, one above the LDS buffers)."""
    from object_detectors_amd import hip_ops as ops
    rng = np.random.default_rng(21)
    nc = 4
    p = obb_head(rng, 12, nc, 320, 30)
    p[3, 4:4 + nc] = 0.0                      # an image without candidates
    big = torch.from_numpy(np.concatenate([p] * 11)[:130]).cuda()      # 130 images: one workgroup per image
    ref_rows = {}
    for b in range(12):                        # alone: 16 workgroups for the image
        d, c, a = ops.nms_v8(big[b:b + 1].contiguous(), 0.25, 0.3, nc=nc, rotated=True)
        ref_rows[b] = (int(c[0]), a[0, :int(c[0])].cpu().numpy().tolist(), d[0, :int(c[0])].cpu().numpy())
    assert ref_rows[3][0] == 0 and max(v[0] for v in ref_rows.values()) > 5
    for B in (130, 64, 24, 7, 2):
        d, c, a = ops.nms_v8(big[:B].contiguous(), 0.25, 0.3, nc=nc, rotated=True)
        torch.cuda.synchronize()
        for b in range(B):
            n, anchors, rows = ref_rows[b % 12]
            assert int(c[b]) == n and a[b, :n].cpu().numpy().tolist() == anchors, (B, b)
            assert np.array_equal(d[b, :n].cpu().numpy(), rows), (B, b)


def test_rotated_nms_global_path_image_among_split_images():
    """A batch small enough for several workgroups per image (G > 1) that mixes images whose candidates fit the LDS
    buffers (split over the workgroups) with one that has more than 4096 candidates (group 0 alone, covariances in
    global memory, where the per-image marker for nms_rotated_emit shares its address with the first covariance
    entry: the marker may only land after every wave's pair tests).  Every image equals the oracle and the same image
    of a 130-image call (one workgroup per image)."""
    from object_detectors_amd import hip_ops as ops
    rng = np.random.default_rng(11)
    nc = 5
    p = synth.head(rng, 4, nc, 1, 448, objects=30)
    p[:, -1] = (np.tanh(p[:, -1]) * 0.5 + 0.25) * math.pi
    assert p.shape[2] == 4116
    cls = p[1:, 4:4 + nc]
    cls[cls < 0.05] = 0.0                      # images 1..3: only the planted objects are candidates
    conf, iou = 0.0005, 0.3
    ref = OO.non_max_suppression_rotated(p, conf, iou, nc=nc, max_det=300)
    assert (p[0, 4:4 + nc].max(0) > conf).all() and all(0 < len(ref[b][1]) < 300 for b in (1, 2, 3))
    big = torch.from_numpy(np.concatenate([p] * 33)[:130]).cuda()
    for B in (130, 4, 2, 1):
        d, c, a = ops.nms_v8(big[:B].contiguous(), conf, iou, nc=nc, max_det=300, rotated=True)
        torch.cuda.synchronize()
        for b in range(B):
            rows, ra, margin = ref[b % 4]
            assert margin > 2e-5, "seed puts a decision on the threshold: pick another"
            n = int(c[b])
            assert n == len(ra) and a[b, :n].cpu().numpy().tolist() == ra.tolist(), (B, b)
            assert np.array_equal(d[b, :n].cpu().numpy(), rows), (B, b)


def test_rotated_nms_rejects_more_candidates_than_the_pair_queue_indexes():
    from object_detectors_amd import hip_ops as ops
    with pytest.raises((ValueError, RuntimeError)):
        ops.nms_v8(torch.zeros((1, 6, 70000), device="cuda"), 0.5, 0.45, nc=1, max_nms=70000, rotated=True)
