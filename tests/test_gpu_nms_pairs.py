"""GPU parity for the suppressor-list form of the axis-aligned NMS (round 3; csrc/nms.hip: nms_pair_lists +
nms_resolve_lists, nms_sort_suppress for flagged images): agnostic NMS and models with a handful of classes, in batches that leave every image several CUs.
Replaces the same reference call as the other NMS tests -- `ops.non_max_suppression` at
object_detectors/yolov8_lmi/model.py:212 (torchvision nms: yolov5_trt.py:267) -- and must give the kept anchors of
oracle/nms.py, and exactly what the one-workgroup sorted sweep (LMI_NMS_SWEEP_ONLY) gives, row for row."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nms as ON
from object_detectors_amd import synth

F32 = np.float32


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def same(a, b):
    (da, ca, aa), (db, cb, ab) = a, b
    assert torch.equal(ca, cb)
    for i, n in enumerate(ca.tolist()):
        assert torch.equal(aa[i, :n], ab[i, :n]) and torch.equal(da[i, :n], db[i, :n]), i


@pytest.mark.parametrize("agnostic,nc", [(True, 80), (False, 1), (False, 3), (True, 5), (False, 8)])
@pytest.mark.parametrize("objects,cluster", [(40, (5, 16)), (110, (8, 16))])
def test_lists_equal_the_sweep_and_the_oracle_over_batch_sizes(ops, agnostic, nc, objects, cluster):
    """One to 64 images per call: 64 / 16 / 4 workgroups per image share the pair tests; from 65 images on the sweep runs
    alone (fewer than 16 CUs per image; 17 to 64 images likewise on this part).  The typical and the dense head of the bench, 640 model size."""
    rng = np.random.default_rng(31 + nc + objects)
    p = synth.head(rng, 6, nc, 0, 640, objects, cluster=cluster)
    ref = ON.non_max_suppression(p, 0.25, 0.45, agnostic=agnostic, nc=nc)
    big = dev(np.concatenate([p] * 11))                       # 66 images
    for B in (1, 3, 16, 64, 66):
        x = big[:B].contiguous()
        got = ops.nms_v8(x, 0.25, 0.45, agnostic=agnostic, nc=nc)
        same(got, ops.nms_v8(x, 0.25, 0.45, agnostic=agnostic, nc=nc, sweep_only=True))
        for b in range(min(B, 6)):
            n = int(got[1][b])
            assert got[2][b, :n].cpu().numpy().tolist() == ref[b][1].tolist(), (B, b)
            assert np.array_equal(got[0][b, :n].cpu().numpy(), ref[b][0].astype(F32)), (B, b)


def test_a_list_longer_than_its_capacity_falls_back_to_the_sweep(ops):
    """70 nearly identical boxes with descending scores: the last has 69 suppressors, more than a list holds (48) -- the
    image takes the sorted sweep inside the kernel; its neighbour in the batch, an ordinary scene, keeps its lists."""
    rng = np.random.default_rng(5)
    A = 2100
    p = synth.head(rng, 2, 2, 0, 320, objects=12)
    p[0, 4:6] = 0.0
    k = 70
    p[0, 0, :k] = 100 + rng.uniform(-1, 1, k); p[0, 1, :k] = 120 + rng.uniform(-1, 1, k)
    p[0, 2, :k] = 60; p[0, 3, :k] = 40
    p[0, 4, :k] = np.linspace(0.9, 0.3, k, dtype=F32)
    p[0, 0, k:k + 5] = [200, 230, 260, 290, 300]; p[0, 1, k:k + 5] = 50; p[0, 2:4, k:k + 5] = 20
    p[0, 5, k:k + 5] = [0.8, 0.7, 0.6, 0.5, 0.4]
    for agnostic in (True, False):
        got = ops.nms_v8(dev(p), 0.25, 0.45, agnostic=agnostic, nc=2)
        same(got, ops.nms_v8(dev(p), 0.25, 0.45, agnostic=agnostic, nc=2, sweep_only=True))
        ref = ON.non_max_suppression(p, 0.25, 0.45, agnostic=agnostic, nc=2)
        for b in range(2):
            n = int(got[1][b])
            assert got[2][b, :n].cpu().numpy().tolist() == ref[b][1].tolist(), (agnostic, b)
        assert int(got[1][0]) >= 2 and 0 in got[2][0, :int(got[1][0])].tolist()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lists_random_settings_ties_caps_and_dtypes(ops, seed):
    """Thresholds from 0 to 1, max_det cuts, max_nms below and above the candidate count (below: the lists are not used),
    quantised scores (ties: anchor order decides), heads small enough that the list capacity (anchors / 4) is exceeded,
    fp16 heads, the fused finalize epilogue."""
    rng = np.random.default_rng(seed)
    for trial in range(20):
        nc = int(rng.choice([1, 2, 4, 8, 30]))
        A = int(rng.choice([256, 525, 2100, 8400]))
        size = {256: 128, 525: 160, 2100: 320, 8400: 640}[A]
        B = int(rng.integers(1, 5))
        objects = int(rng.integers(0, 60))
        p = synth.head(rng, B, nc, 0, size, objects=objects, anchors=A)
        if rng.random() < 0.5:
            p[:, 4:] = np.round(p[:, 4:] * 64) / 64
        conf = float(rng.choice([0.25, 0.1, 1 / 64, 0.5]))
        iou = float(rng.choice([0.45, 0.3, 0.7, 0.0, 1.0]))
        agnostic = bool(rng.integers(0, 2)) or nc == 30
        max_det = int(rng.choice([300, 7, 1, 64]))
        max_nms = int(rng.choice([30000, 30000, 50, 400]))
        half = bool(rng.random() < 0.3)
        x = dev(p).half() if half else dev(p)
        pr = x.float().cpu().numpy()
        kw = dict(agnostic=agnostic, max_det=max_det, nc=nc, max_nms=max_nms)
        tag = f"trial {trial}: nc={nc} A={A} B={B} conf={conf} iou={iou} {kw} half={half}"
        got = ops.nms_v8(x, conf, iou, **kw)
        same(got, ops.nms_v8(x, conf, iou, sweep_only=True, **kw))
        ref = ON.non_max_suppression(pr, conf, iou, **kw)
        for b in range(B):
            n = int(got[1][b])
            assert got[2][b, :n].cpu().numpy().tolist() == ref[b][1].tolist(), tag
        fin = dict(model_hw=(size, size), orig_hw=(270, 480), thres=0.3)
        f1 = ops.nms_v8(x, conf, iou, finalize=dict(fin), **kw)
        f2 = ops.nms_v8(x, conf, iou, finalize=dict(fin), sweep_only=True, **kw)
        same(f1[:3], f2[:3])
        assert torch.equal(f1[3]["keep_count"], f2[3]["keep_count"]) and torch.equal(f1[3]["slot_base"], f2[3]["slot_base"]), tag


def test_lists_full_size_dense_agnostic_batch(ops):
    """BASELINE config 5's agnostic leg at the size the lists serve best: 16 dense images (>= 1000 candidates each, 80
    classes incl. classes >= 69), kept anchors against the oracle on every image."""
    rng = np.random.default_rng(55)
    p = synth.head(rng, 16, 80, 32, 640, 110, cluster=(8, 16))
    assert ((p[:, 4:84].max(1) > 0.25).sum(1) >= 1000).all()
    got = ops.nms_v8(dev(p), 0.25, 0.45, agnostic=True, nc=80)
    ref = ON.non_max_suppression(p, 0.25, 0.45, agnostic=True, nc=80)
    for b in range(16):
        n = int(got[1][b])
        assert got[2][b, :n].cpu().numpy().tolist() == ref[b][1].tolist(), b
        assert np.array_equal(got[0][b, :n].cpu().numpy(), ref[b][0].astype(F32)), b
