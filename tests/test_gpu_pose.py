"""GPU parity for keypoint models (reference object_detectors/yolov8_lmi/model.py:535-657,
tests tests/object_detectors/yolov8_lmi/test_yolov8_model.py:258-313): lmi_scale_coords against the oracle, and the
`Yolov8Pose` drop-in end to end against the oracle chained the way the reference chains the ultralytics calls."""
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

from oracle import boxes as OB, geometry as OG, nms as ON  # noqa: E402
from object_detectors_amd import synth  # noqa: E402
from object_detectors_amd.nets.yolov8 import save_checkpoint  # noqa: E402

F32 = np.float32
NAMES = {0: "person", 1: "robot"}
KPT = (5, 3)


@pytest.mark.parametrize("h0,w0,S,ndim,ragged", [(1080, 1920, 640, 3, False), (427, 640, 640, 2, False), (375, 500, 320, 3, True)])
def test_scale_coords_kernel(h0, w0, S, ndim, ragged):
    from object_detectors_amd import hip_ops as ops
    rng = np.random.default_rng(h0)
    B, max_det, nk = 3, 20, 17
    det = (rng.random((B, max_det, 6 + nk * ndim)).astype(F32) * (S + 60) - 30).astype(F32)
    count = np.array([20, 0, 7], np.int32)
    shapes = [(h0, w0), (h0 // 2, w0), (h0, w0 // 3)] if ragged else [(h0, w0)] * B
    ohw = torch.tensor(shapes, dtype=torch.int32).cuda() if ragged else (h0, w0)
    d = torch.from_numpy(det).cuda()
    ops.scale_coords(d, torch.from_numpy(count).cuda(), nk, ndim, (S, S), ohw)
    got = d.cpu().numpy()
    for b in range(B):
        n = count[b]
        ref = OB.scale_coords((S, S), det[b, :n, 6:].reshape(n, nk, ndim), shapes[b] + (3,))
        assert np.array_equal(got[b, :n, 6:].reshape(n, nk, ndim), ref)
        assert np.array_equal(got[b, n:], det[b, n:]) and np.array_equal(got[b, :, :6], det[b, :, :6])   # nothing else touched


@pytest.fixture(scope="module")
def pose(tmp_path_factory):
    from object_detector import ObjectDetector
    import od_models  # noqa: F401
    from yolov8_lmi.model import Yolov8Pose
    path = str(tmp_path_factory.mktemp("p") / "pose.pt")
    save_checkpoint(path, "n", len(NAMES), names=NAMES, seed=3, task="pose", kpt_shape=KPT)
    m = ObjectDetector(metadata=dict(version='v0', model_name='yolov8', task='pose', framework='ultralytics'), weights=path)
    assert isinstance(m, Yolov8Pose) and m.device.type == "cuda"
    for conv in m.model.head.cv3:
        conv[-1].bias.data.normal_(-2.0, 0.5)
        conv[-1].weight.data.mul_(8.0)
    return m


def test_pose_predict_against_oracle(pose):
    rng = np.random.default_rng(6)
    frame = synth.frames_u8(rng, 1, 256, 256)[0]
    operators = [{"resize": [256, 144, 640, 360]}, {"pad": [0, 0, 56, 56]}]
    im = pose.preprocess(frame)
    pred = pose.forward(im)[0].cpu().numpy()
    s = pred[0, 4:4 + len(NAMES)].max(0)
    conf = float((s.min() + 3 * s.max()) / 4)
    (rd, ra), = ON.non_max_suppression(pred, conf, 0.4, nc=len(NAMES), max_det=50)
    assert len(rd) > 0
    boxes = OB.scale_boxes((256, 256), rd[:, :4], (256, 256, 3))
    M = OB.class_filter(rd[:, 4], rd[:, 5], np.full(len(NAMES), conf))
    kpts = OB.scale_coords((256, 256), rd[:, 6:].reshape(len(rd), *KPT), (256, 256, 3))[M]
    ref_pts = OG.revert_to_origin(kpts[:, :, :2].reshape(-1, 2), operators).reshape(-1, KPT[0], 2)
    ref_boxes = OG.revert_to_origin(boxes[M], operators)
    # postprocess keeps visibility; predict drops it and reverts
    res = pose.postprocess(pose.forward(im), im, frame, conf, 0.4, False, 50)
    assert res['points'][0].shape == (int(M.sum()), KPT[0], KPT[1]) and np.array_equal(res['points'][0], kpts)
    for as_tensor in (False, True):
        image = torch.from_numpy(frame).cuda() if as_tensor else frame
        out, tinfo = pose.predict(image, conf, operators=operators, iou=0.4, max_det=50)
        pts, bxs = out['points'], out['boxes']
        if as_tensor:
            assert pts.is_cuda and bxs.is_cuda
            pts, bxs = pts.cpu().numpy(), bxs.cpu().numpy()
        assert np.array_equal(pts, ref_pts) and np.array_equal(bxs, ref_boxes)
        assert out['classes'] == [NAMES[int(c)] for c in rd[M, 5]]
        assert set(tinfo) == {'preproc', 'proc', 'postproc'}
    none, _ = pose.predict(frame, 0.9999999, operators=operators)
    assert len(none['boxes']) == 0 and len(none['points']) == 0


@pytest.mark.parametrize("ndim,ragged,agnostic", [(3, False, False), (2, True, False), (3, True, True)])
def test_fused_keypoint_epilogue_equals_the_three_calls(ndim, ragged, agnostic):
    """`Yolov8Pose.postprocess` as ONE launch chain (nms_v8(finalize=dict(..., kpt=...))) against nms_v8 +
    finalize_detections + scale_coords (reference: object_detectors/yolov8_lmi/model.py:555-569) -- rows bit-equal,
    per-class thresholds, uniform and ragged original sizes, and against the oracle's scale_coords."""
    from object_detectors_amd import hip_ops as ops
    rng = np.random.default_rng(17 + ndim)
    B, nc, nk, S = 4, 3, 17, 320
    p = synth.head(rng, B, nc, nk * ndim, S, objects=12)
    p[:, 4 + nc:] = rng.uniform(-30, S + 30, (B, nk * ndim, p.shape[2])).astype(F32)
    x = torch.from_numpy(p).cuda()
    shapes = [(270, 480), (375, 500), (1080, 1920), (200, 90)] if ragged else [(270, 480)] * B
    ohw = torch.tensor(shapes, dtype=torch.int32).cuda() if ragged else shapes[0]
    thres = torch.tensor([0.3, 0.5, 0.26], dtype=torch.float64).cuda()
    det, count, anchor = ops.nms_v8(x, 0.25, 0.45, nc=nc, agnostic=agnostic)
    raw = det.clone()
    fin = ops.finalize_detections(det, count, nc, (S, S), ohw, thres, with_slots=False)
    ops.scale_coords(det, count, nk, ndim, (S, S), ohw)
    d2, c2, a2, f2 = ops.nms_v8(x, 0.25, 0.45, nc=nc, agnostic=agnostic,
                                finalize=dict(model_hw=(S, S), orig_hw=ohw, thres=thres, with_slots=False, kpt=(6, nk, ndim)))
    torch.cuda.synchronize()
    assert torch.equal(count, c2) and torch.equal(fin["keep_count"], f2["keep_count"]) and int(count.sum()) > 0
    for b, n in enumerate(count.tolist()):
        assert torch.equal(det[b, :n], d2[b, :n]) and torch.equal(anchor[b, :n], a2[b, :n]) and torch.equal(fin["keep"][b, :n], f2["keep"][b, :n]), b
        ref = OB.scale_coords((S, S), raw[b, :n, 6:].cpu().numpy().reshape(n, nk, ndim), shapes[b] + (3,))
        assert np.array_equal(d2[b, :n, 6:].cpu().numpy().reshape(n, nk, ndim), ref), b
    with pytest.raises(ValueError):
        ops.nms_v8(x, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=ohw, thres=thres, with_slots=False, kpt=(6, nk + 1, ndim)))
