"""The oracle against outputs of the REFERENCE's own code, captured in the build container:
tests/golden/yolov5_helpers.npz  (make_golden_yolov5_helpers.py: the torch-only method bodies of
    object_detectors/yolov5_lmi/trt/yolov5_trt.py:172-415) and
tests/golden/rescale_masks.npz   (make_golden_rescale_masks.py: lmi_utils/postprocess_utils/mask_utils.py:12-140).
Everything here is bit-exact unless the test says otherwise."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import inputs_yolov5_helpers as IN  # noqa: E402
import make_golden_rescale_masks as RM  # noqa: E402  (its `inputs`/CASES only; load_reference is never called here)
from oracle import boxes as OB, masks as OM, nms as ON, paste as OP  # noqa: E402

F32 = np.float32
Z = np.load(os.path.join(HERE, "golden", "yolov5_helpers.npz"))
ZR = np.load(os.path.join(HERE, "golden", "rescale_masks.npz"))


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a, F32).view(np.uint32), np.ascontiguousarray(b, F32).view(np.uint32))


def test_xywh2xyxy_and_box_iou():
    assert same_bits(ON.xywh2xyxy(IN.boxes_xywh(1, 256)), Z["xywh2xyxy"])
    a, b = IN.boxes_xyxy(2, 48, (640, 640)), IN.boxes_xyxy(3, 40, (640, 640))
    assert same_bits(ON.box_iou(a, b), Z["box_iou"])
    # the quotient the greedy pass thresholds is the same formula without eps
    assert np.abs(ON.box_iou(a, b, eps=0.0) - Z["box_iou"]).max() < 1e-6


def test_scale_boxes_clip_and_scale_segments():
    for i, (seed, n, s1, s0) in enumerate(IN.SCALE_CASES):
        assert same_bits(OB.scale_boxes(s1, IN.boxes_xyxy(seed, n, s1), s0, pad_mode="exact"), Z[f"scale_boxes{i}"]), i
        assert same_bits(OB.scale_coords(s1, IN.points_xy(seed, n, s1), s0), Z[f"scale_segments{i}"]), i
        b = IN.boxes_xyxy(seed, n, s1)
        clip = np.stack([np.clip(b[:, 0], 0, s0[1]), np.clip(b[:, 1], 0, s0[0]), np.clip(b[:, 2], 0, s0[1]), np.clip(b[:, 3], 0, s0[0])], 1)
        assert same_bits(clip, Z[f"clip_boxes{i}"])
    # with equal shapes (every reference caller) the rounded and the un-rounded pad are both 0
    seed, n, s1, s0 = IN.SCALE_CASES[-1]
    assert same_bits(OB.scale_boxes(s1, IN.boxes_xyxy(seed, n, s1), s0, pad_mode="round"), Z[f"scale_boxes{len(IN.SCALE_CASES) - 1}"])


def test_crop_mask_and_process_mask():
    for i, (seed, n, mh, mw) in enumerate(IN.MASK_CASES):
        proto, coef, boxes = IN.mask_inputs(seed, n, mh, mw)
        ind = np.unpackbits(Z[f"crop_mask{i}"], axis=-1, count=mw).astype(bool)
        assert np.array_equal(OM.crop_indicator(boxes, mh, mw), ind)
        assert int(Z[f"process_mask{i}_near_zero"]) == 0
        ref = np.unpackbits(Z[f"process_mask{i}"], axis=-1, count=mw).astype(bool)
        # v5 `sigmoid -> crop -> > 0.5` equals v8-native `crop -> > 0` when nothing is resampled
        assert np.array_equal(OM.process_mask_native(proto, coef, boxes, (mh, mw)).astype(bool), ref)
        assert np.array_equal(OM.process_mask_native_np(proto, coef, boxes, (mh, mw)).astype(bool), ref)
    # the integer-edge box: column 3 in, column 11 out, row 2 in, row 9 out
    ind = np.unpackbits(Z["crop_mask0"], axis=-1, count=40).astype(bool)[0]
    assert ind[2, 3] and not ind[2, 11] and not ind[9, 3] and not ind[1, 3] and not ind[2, 2] and ind[8, 10]


def test_nms_decode_matches_reference_arguments():
    for i, (seed, size, objects, nc, nm, conf) in enumerate(IN.NMS_CASES):
        head = IN.nms_head(seed, size, objects, nc, nm)
        for b in range(head.shape[0]):
            rows, anchors, offset = ON.decode_candidates(head[b], conf, nc)
            assert same_bits(rows, Z[f"nms{i}_rows{b}"]), (i, b)
            assert same_bits(offset, Z[f"nms{i}_offset_boxes{b}"])
            assert same_bits(rows[:, 4], Z[f"nms{i}_scores{b}"])
            if nc >= 80:
                assert (rows[:, 5] >= 69).any()     # the 1/16-px quantisation band is in the fixture
            # and the full oracle with nothing suppressed (IoU can never exceed 1) returns those rows
            det, kept = ON.non_max_suppression(head[b:b + 1], conf, 1.0, nc=nc, max_det=IN.NMS_MAX_DET)[0]
            assert same_bits(det, rows) and np.array_equal(kept, anchors)


def test_rescale_masks():
    for i, (seed, n, s, h, w, thr) in enumerate(RM.CASES):
        m, b = RM.inputs(seed, n, s, h, w)
        ref = np.unpackbits(ZR[f"out{i}"], axis=-1, count=w).astype(bool)
        assert np.array_equal(OP.rescale_masks(m, b, (h, w), thr), ref), i
        if h * w <= 101 * 67:
            v = OP.paste_np(m, b, (h, w))                 # independent numpy restatement of aten's sampling
            assert ((v >= F32(thr)) != ref).mean() < 1e-3
    assert tuple(ZR["empty_shape"]) == (0, 50, 60)
