"""HIP kernels (through the C ABI) against outputs of the REFERENCE's own code captured in the build container:
tests/golden/yolov5_helpers.npz (object_detectors/yolov5_lmi/trt/yolov5_trt.py:172-415, torch-only method bodies)
and tests/golden/rescale_masks.npz (lmi_utils/postprocess_utils/mask_utils.py:12-140).  Bit-exact throughout."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import inputs_yolov5_helpers as IN  # noqa: E402
import make_golden_rescale_masks as RM  # noqa: E402  (inputs / CASES only)

F32 = np.float32
Z = np.load(os.path.join(HERE, "golden", "yolov5_helpers.npz"))
ZR = np.load(os.path.join(HERE, "golden", "rescale_masks.npz"))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a):
    return np.ascontiguousarray(a, F32).view(np.uint32)


def test_finalize_unrounded_pad_vs_reference_scale_boxes(ops):
    """lmi_finalize_detections(pad_round=0) == the in-tree scale_boxes + clip_boxes (yolov5_trt.py:316-341)."""
    for i, (seed, n, s1, s0) in enumerate(IN.SCALE_CASES):
        det = np.zeros((1, n, 6), F32)
        det[0, :, :4] = IN.boxes_xyxy(seed, n, s1)
        det[0, :, 4] = 0.9
        d = dev(det)
        ops.finalize_detections(d, dev(np.array([n], np.int32)), 1, s1, s0[:2], 0.5, pad_round=False)
        assert np.array_equal(bits(d.cpu().numpy()[0, :, :4]), bits(Z[f"scale_boxes{i}"])), i


def test_scale_coords_vs_reference_scale_segments(ops):
    """lmi_scale_coords == the in-tree scale_segments (yolov5_trt.py:354-370)."""
    for i, (seed, n, s1, s0) in enumerate(IN.SCALE_CASES):
        det = np.zeros((1, n, 8), F32)
        det[0, :, 6:8] = IN.points_xy(seed, n, s1)
        d = dev(det)
        ops.scale_coords(d, dev(np.array([n], np.int32)), 1, 2, s1, s0[:2])
        assert np.array_equal(bits(d.cpu().numpy()[0, :, 6:8]), bits(Z[f"scale_segments{i}"])), i


def test_nms_decode_rows_vs_reference(ops):
    """lmi_nms_v8 with iou_thres = 1 (nothing can be suppressed) emits exactly the rows the reference's decode builds
    around its NMS call (yolov5_trt.py:232-266, :278): xywh->xyxy, first-argmax class, strict `>`, descending sort."""
    for i, (seed, size, objects, nc, nm, conf) in enumerate(IN.NMS_CASES):
        head = IN.nms_head(seed, size, objects, nc, nm)
        det, count, anchor = ops.nms_v8(dev(head), conf, 1.0, nc=nc, max_det=IN.NMS_MAX_DET)
        det, count = det.cpu().numpy(), count.cpu().numpy()
        for b in range(head.shape[0]):
            ref = Z[f"nms{i}_rows{b}"]
            assert count[b] == len(ref), (i, b)
            assert np.array_equal(bits(det[b, :count[b]]), bits(ref)), (i, b)


@pytest.mark.parametrize("out_dtype", [torch.uint8, torch.float32])
def test_process_mask_at_prototype_resolution_vs_reference(ops, out_dtype):
    """With the frame at the prototype resolution nothing is resampled, and the reference's in-tree
    `process_mask` (product -> sigmoid -> crop_mask -> > 0.5, yolov5_trt.py:391-415) must equal lmi_process_mask
    (product -> crop -> > 0) pixel for pixel; all-ones logits give `crop_mask`'s indicator itself (:373-388)."""
    for i, (seed, n, mh, mw) in enumerate(IN.MASK_CASES):
        proto, coef, boxes = IN.mask_inputs(seed, n, mh, mw)
        det = np.zeros((1, n, 38), F32)
        det[0, :, :4] = boxes
        det[0, :, 6:] = coef
        sm = np.arange(n, dtype=np.int32)
        sb = np.array([0, n], np.int32)
        got, ovf = ops.process_mask(dev(proto[None]), dev(det), dev(sm), dev(sb), (mh, mw), n, out_dtype)
        ref = np.unpackbits(Z[f"process_mask{i}"], axis=-1, count=mw)
        assert ovf.item() == 0 and np.array_equal(got.cpu().numpy() != 0, ref != 0), i
        ones = np.zeros_like(proto)
        ones[0] = 1.0
        det[0, :, 6:] = 0
        det[0, :, 6] = 1.0
        got, _ = ops.process_mask(dev(ones[None]), dev(det), dev(sm), dev(sb), (mh, mw), n, out_dtype)
        ind = np.unpackbits(Z[f"crop_mask{i}"], axis=-1, count=mw)
        assert np.array_equal(got.cpu().numpy() != 0, ind != 0), i


def test_paste_masks_vs_reference_rescale_masks(ops):
    """lmi_paste_masks == the reference's rescale_masks (mask_utils.py:66-140) on torch-CPU."""
    for i, (seed, n, s, h, w, thr) in enumerate(RM.CASES):
        m, b = RM.inputs(seed, n, s, h, w)
        ref = np.unpackbits(ZR[f"out{i}"], axis=-1, count=w).astype(bool)
        got = ops.paste_masks(dev(m), dev(b), (h, w), thr).cpu().numpy()
        assert got.dtype == bool and got.shape == ref.shape
        # aten's CPU grid_sample vectorises its weights differently from the scalar form: only pixels whose sampled
        # value sits within an ulp of the threshold may differ
        assert (got != ref).mean() < 1e-5, (i, int((got != ref).sum()))
