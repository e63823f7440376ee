"""End-to-end drop-in test on the GPU, modelled on the reference's own wrapper tests
(tests/object_detectors/yolov8_lmi/test_yolov8_model.py:72-126,169-178): build the wrapper
directly and through ObjectDetector, run predict on a resized frame with an operator list, check
the result dict -- but here every field is also compared with the CPU oracle fed the same head
output, which the reference's smoke-level assertions never do."""
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

from oracle import boxes as OB, cv2_resize as OC, geometry as OG, masks as OM, nms as ON, preprocess as OP  # noqa: E402
from object_detectors_amd import synth  # noqa: E402
from object_detectors_amd.nets.yolov8 import YoloV8  # noqa: E402

NAMES = {0: "person", 1: "car", 2: "dog", 3: "cat"}


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    from object_detector import ObjectDetector
    import od_models  # noqa: F401
    from yolov8_lmi.model import Yolov8
    d = tmp_path_factory.mktemp("w")
    out = {}
    for seg in (False, True):
        torch.manual_seed(7)
        m = YoloV8("n", 4, seg, NAMES)
        for conv in m.head.cv3:                      # spread the class scores so that NMS has work to do
            conv[-1].bias.data.normal_(-2.0, 0.5)
            conv[-1].weight.data.mul_(8.0)
        path = str(d / ("seg.pt" if seg else "det.pt"))
        torch.save({"format": "lmi-yolov8-torch", "scale": "n", "nc": 4, "seg": seg, "names": NAMES, "state_dict": m.state_dict()}, path)
        out[seg] = path
    det = Yolov8(out[False])
    seg = ObjectDetector(metadata=dict(version='v0', model_name='yolov8', task='seg', framework='ultralytics'), weights=out[True])
    assert det.device.type == "cuda" and isinstance(seg, Yolov8)
    return det, seg


def oracle_postprocess(pred, proto, model_hw, orig_hw, conf, names, iou, agnostic=False, max_det=300):
    nc = len(names)
    conf2 = OB.min_conf(conf, names)
    (rd, ra), = ON.non_max_suppression(pred, conf2, iou, agnostic=agnostic, max_det=max_det, nc=nc)
    boxes = OB.scale_boxes(model_hw, rd[:, :4], orig_hw + (3,))
    M = OB.class_filter(rd[:, 4], rd[:, 5], OB.class_thresholds(conf, names, nc))
    masks = OM.process_mask_native(proto, rd[:, 6:], boxes, orig_hw)[M] if proto is not None else None
    return boxes[M], rd[M, 4], [names[int(c)] for c in rd[M, 5]], masks


@pytest.mark.parametrize("conf", [0.3, {"person": 0.2, "car": 0.25, "dog": 0.3, "cat": 0.35, "unknown": 0.01}, {"cat": 0.2}])
def test_predict_det_numpy_and_tensor(models, conf):
    det, _ = models
    rng = np.random.default_rng(0)
    frame = synth.frames_u8(rng, 1, 240, 320)[0]
    # the reference's callers hand predict() a frame already at model size plus the operator list that
    # produced it (test_yolov8_model.py:84-88); here: a 448x640 original "resized" to 224x320
    resized = np.ascontiguousarray(frame[:224, :320])
    ops = [{'resize': (320, 224, 640, 448)}]
    det.warmup([224, 320])
    out, tinfo = det.predict(resized, conf, operators=ops, iou=0.4)
    assert set(tinfo) == {'preproc', 'proc', 'postproc'}
    # oracle on the same head output
    im = det.preprocess(resized)
    assert np.array_equal(im.cpu().numpy(), OP.preprocess(resized))
    pred = det.forward(im).cpu().numpy()
    boxes, scores, classes, _ = oracle_postprocess(pred, None, (224, 320), (224, 320), conf, NAMES, 0.4)
    assert len(boxes) > 0
    assert isinstance(out['boxes'], np.ndarray) and isinstance(out['classes'], list)
    assert np.array_equal(out['boxes'], OG.revert_to_origin(boxes, ops))
    assert np.array_equal(out['scores'], scores) and out['classes'] == classes
    # tensor in -> tensors out, on the GPU (test_yolov8_model.py:122-126)
    out_t, _ = det.predict(torch.from_numpy(resized).cuda(), conf, operators=ops, iou=0.4)
    assert out_t['boxes'].is_cuda and out_t['scores'].is_cuda
    assert np.array_equal(out_t['boxes'].cpu().numpy(), out['boxes'])
    img = det.annotate_image(out, resized)
    assert img.shape == resized.shape


def _conf_for(pred):
    """a threshold inside the score range of whatever the random-init head happens to output (many
    anchors share a score there, which also exercises the tie order)"""
    best = pred[:, 4:8].max(1)
    return float((best.min() + 3 * best.max()) / 4)


def test_predict_seg_masks_and_segments(models):
    _, seg = models
    rng = np.random.default_rng(1)
    frame = synth.frames_u8(rng, 1, 256, 256)[0]
    im = seg.preprocess(frame)
    y = seg.forward(im)
    pred, proto = y[0].cpu().numpy(), y[1][-1][0].cpu().numpy()
    conf = _conf_for(pred)
    out, _ = seg.predict(frame, conf, operators=[], iou=0.4, max_det=40)
    boxes, scores, classes, masks = oracle_postprocess(pred, proto, (256, 256), (256, 256), conf, NAMES, 0.4, max_det=40)
    assert len(boxes) > 0 and masks.shape[0] == len(boxes)
    assert np.array_equal(out['boxes'], OG.revert_to_origin(boxes, []))
    assert out['classes'] == classes
    assert out['masks'].dtype == np.float32 and out['masks'].shape == masks.shape
    assert OM.mask_iou(out['masks'], masks) >= 0.999
    assert len(out['segments']) == len(boxes) and all(s.shape[1] == 2 for s in out['segments'])
    none, _ = seg.predict(frame, 0.999999, operators=[])
    assert len(none['boxes']) == 0 and len(none['masks']) == 0


def test_predict_batch_ragged_frames(models):
    _, seg = models
    rng = np.random.default_rng(2)
    frames = [synth.frames_u8(rng, 1, h, w)[0] for h, w in ((270, 480), (300, 200), (270, 480))]
    im, ops = seg.letterbox(frames, (256, 256))
    y = seg.forward(im)
    pred, proto = y[0].cpu().numpy(), y[1][-1].cpu().numpy()
    conf = _conf_for(pred)
    res, _ = seg.predict_batch(frames, conf, imgsz=(256, 256), max_det=30)
    assert len(res['boxes']) == 3
    for i, f in enumerate(frames):
        boxes, scores, classes, masks = oracle_postprocess(pred[i:i + 1], proto[i], (256, 256), f.shape[:2], conf, NAMES, 0.4, max_det=30)
        assert np.abs(res['boxes'][i] - boxes).max() <= 1e-3 and res['classes'][i] == classes
        assert res['masks'][i].shape == masks.shape and OM.mask_iou(res['masks'][i], masks) >= 0.999


def test_predict_batch_grey_frames_equal_replicated_colour(models):
    """Grey planes (the runner's GRAY2BGR case, run_model.py:77-78) through predict_batch: same results as the frames
    replicated to three channels, for a list of planes and for one 2-D frame."""
    _, seg = models
    rng = np.random.default_rng(4)
    planes = [synth.frames_u8(rng, 1, h, w)[0, :, :, 0].copy() for h, w in ((270, 480), (300, 200))]
    colour = [np.repeat(p[..., None], 3, -1) for p in planes]
    im_g, ops_g = seg.letterbox(planes, (256, 256))
    im_c, ops_c = seg.letterbox(colour, (256, 256))
    assert ops_g == ops_c and torch.equal(im_g, im_c)
    conf = _conf_for(seg.forward(im_c)[0].cpu().numpy())
    res_g, _ = seg.predict_batch(planes, conf, imgsz=(256, 256), max_det=30)
    res_c, _ = seg.predict_batch(colour, conf, imgsz=(256, 256), max_det=30)
    assert len(res_g['boxes']) == len(res_c['boxes']) == 2
    for i in range(2):
        assert np.array_equal(res_g['boxes'][i], res_c['boxes'][i]) and res_g['classes'][i] == res_c['classes'][i]
        assert np.array_equal(res_g['masks'][i], res_c['masks'][i]) and res_g['masks'][i].shape[1:] == planes[i].shape
    one, _ = seg.predict_batch(planes[0], conf, imgsz=(256, 256), max_det=30)
    assert np.array_equal(one['boxes'][0], res_c['boxes'][0])


def test_predict_batch_eight_ragged_sizes_one_mask_launch(models):
    """Eight differently sized frames in one batch (the reference's postprocess loops over such a list,
    model.py:215-236): one NMS + finalize chain and ONE mask launch, checked per image against the oracle."""
    _, seg = models
    rng = np.random.default_rng(12)
    sizes = ((270, 480), (300, 200), (480, 640), (144, 256), (256, 256), (375, 500), (101, 67), (512, 384))
    frames = [synth.frames_u8(rng, 1, h, w)[0] for h, w in sizes]
    im, ops = seg.letterbox(frames, (256, 256))
    y = seg.forward(im)
    pred, proto = y[0].cpu().numpy(), y[1][-1].cpu().numpy()
    conf = _conf_for(pred)
    res, _ = seg.predict_batch(frames, conf, imgsz=(256, 256), max_det=20, _with_index=True)
    assert len(res['boxes']) >= 6
    for k, i in enumerate(res['image_index']):
        boxes, scores, classes, masks = oracle_postprocess(pred[i:i + 1], proto[i], (256, 256), sizes[i], conf, NAMES, 0.4, max_det=20)
        assert np.abs(res['boxes'][k] - boxes).max() <= 1e-3 and res['classes'][k] == classes
        assert np.array_equal(res['scores'][k], scores)
        assert res['masks'][k].shape == masks.shape and res['masks'][k].dtype == np.float32
        assert OM.mask_iou(res['masks'][k], masks) >= 0.999, i
    # uint8 masks through the same path
    seg.mask_dtype = torch.uint8
    try:
        res8, _ = seg.predict_batch(frames, conf, imgsz=(256, 256), max_det=20)
    finally:
        seg.mask_dtype = torch.float32
    for a, b in zip(res8['masks'], res['masks']):
        assert a.dtype == np.uint8 and np.array_equal(a, b.astype(np.uint8))


def test_predict_config1_640_det_and_seg(models):
    """BASELINE config 1: batch 1 at 640x640, conf 0.5-style call with the operator list of the reference's tests
    (test_yolov8_model.py:84-88: a frame stretched to 640x640, `{'resize': (640, 640, w, h)}`)."""
    det, seg = models
    rng = np.random.default_rng(13)
    frame = synth.frames_u8(rng, 1, 640, 640)[0]
    ops = [{'resize': (640, 640, 500, 375)}]
    det.warmup([640, 640])
    im = det.preprocess(frame)
    assert im.shape == (1, 3, 640, 640) and np.array_equal(im.cpu().numpy(), OP.preprocess(frame))
    pred = det.forward(im).cpu().numpy()
    assert pred.shape == (1, 8, 8400)
    conf = _conf_for(pred)
    out, tinfo = det.predict(frame, conf, operators=ops, iou=0.4)
    boxes, scores, classes, _ = oracle_postprocess(pred, None, (640, 640), (640, 640), conf, NAMES, 0.4)
    assert len(boxes) > 0 and np.array_equal(out['boxes'], OG.revert_to_origin(boxes, ops))
    assert np.array_equal(out['scores'], scores) and out['classes'] == classes
    # time_info: launch times like the reference's by default; real stage durations with sync_time_info (same results)
    assert set(tinfo) == {'preproc', 'proc', 'postproc'}
    det.sync_time_info = True
    try:
        out2, tsync = det.predict(frame, conf, operators=ops, iou=0.4)
    finally:
        det.sync_time_info = False
    assert np.array_equal(out2['boxes'], out['boxes']) and set(tsync) == set(tinfo) and all(v > 0 for v in tsync.values())
    y = seg.forward(seg.preprocess(frame))
    pred, proto = y[0].cpu().numpy(), y[1][-1][0].cpu().numpy()
    assert pred.shape == (1, 40, 8400) and proto.shape == (32, 160, 160)
    conf = _conf_for(pred)
    out, _ = seg.predict(frame, conf, operators=[], iou=0.4, max_det=60, return_segments=False)
    boxes, scores, classes, masks = oracle_postprocess(pred, proto, (640, 640), (640, 640), conf, NAMES, 0.4, max_det=60)
    assert len(boxes) > 0 and np.array_equal(out['boxes'], OG.revert_to_origin(boxes, []))
    assert out['masks'].shape == masks.shape and OM.mask_iou(out['masks'], masks) >= 0.999


def test_postprocess_reads_fp16_heads_natively(models):
    """`fp16=True` models hand postprocess an fp16 head (model.py:68,132): it is read as it is, values promoted in
    the kernels -- same result as the oracle on the promoted head."""
    det, _ = models
    rng = np.random.default_rng(14)
    head = torch.from_numpy(synth.head(rng, 2, 4, 0, 256, objects=10)).cuda().half()
    img = torch.empty((2, 3, 256, 256), device="cuda")
    origs = [np.zeros((270, 480, 3), np.uint8), np.zeros((270, 480, 3), np.uint8)]
    res = det.postprocess(head, img, origs, 0.3, iou=0.45)
    ref = head.float().cpu().numpy()
    for i in range(2):
        boxes, scores, classes, _ = oracle_postprocess(ref[i:i + 1], None, (256, 256), (270, 480), 0.3, NAMES, 0.45)
        assert np.array_equal(res['boxes'][i], boxes) and np.array_equal(res['scores'][i], scores) and res['classes'][i] == classes


def test_predict_seg_with_operators_reverts_on_device(models):
    """The reference's calling convention (test_yolov8_model.py:84-117): the caller resizes / pads the frame, passes
    the operator list, and predict() brings boxes, masks and segments back to sensor space.  Here the revert runs in
    the HIP kernels for numpy and tensor inputs alike; checked against the oracle's restatement of
    revert_to_origin / revert_masks_to_origin applied to the oracle's own post-processing result."""
    _, seg = models
    rng = np.random.default_rng(4)
    frame = synth.frames_u8(rng, 1, 256, 256)[0]            # stands for a 480x640 frame resized to 256x144 and padded
    operators = [{"resize": [256, 144, 640, 360]}, {"pad": [0, 0, 56, 56]}]
    im = seg.preprocess(frame)
    y = seg.forward(im)
    pred, proto = y[0].cpu().numpy(), y[1][-1][0].cpu().numpy()
    conf = _conf_for(pred)
    boxes, scores, classes, masks = oracle_postprocess(pred, proto, (256, 256), (256, 256), conf, NAMES, 0.4, max_det=40)
    assert len(boxes) > 0
    ref_boxes = OG.revert_to_origin(boxes, operators)
    ref_masks = OG.revert_masks_to_origin(masks, operators)
    assert ref_masks.shape[1:] == (360, 640)
    for as_tensor in (False, True):
        image = torch.from_numpy(frame).cuda() if as_tensor else frame
        out, _ = seg.predict(image, conf, operators=operators, iou=0.4, max_det=40)
        got_boxes, got_masks, got_segs = out['boxes'], out['masks'], out['segments']
        if as_tensor:
            assert got_boxes.is_cuda and got_masks.is_cuda and all(s.is_cuda for s in got_segs)
            got_boxes, got_masks, got_segs = got_boxes.cpu().numpy(), got_masks.cpu().numpy(), [s.cpu().numpy() for s in got_segs]
        assert np.array_equal(got_boxes, ref_boxes)
        assert out['classes'] == classes
        assert got_masks.shape == ref_masks.shape and got_masks.dtype == np.float32
        # interpolated 0/1 masks: equal wherever the model-space masks agree, i.e. almost everywhere (IoU >= 0.999 there)
        assert (np.abs(got_masks - ref_masks) > 1e-6).mean() < 2e-3
        assert len(got_segs) == len(boxes)
        for s in got_segs:
            assert s.shape[1] == 2 and (s >= 0).all() and (s == np.round(s)).all()
            if len(s):
                assert s[:, 0].max() <= 640 and s[:, 1].max() <= 360 + 140   # contour points of the padded canvas, un-padded then un-resized


def test_segments_from_device_masks_match_oracle():
    """masks2segments on GPU masks (bit-packed box regions -> host C++ tracer) against the numpy/scipy oracle run on
    the same masks copied whole, uint8 and fp32, with and without box hints."""
    from oracle import segments as OS
    from object_detectors_amd import segments as PSG
    rng = np.random.default_rng(31)
    n, h, w = 9, 270, 480
    yy, xx = np.mgrid[0:h, 0:w]
    masks = np.zeros((n, h, w), np.uint8)
    boxes = np.zeros((n, 4), np.float32)
    for i in range(n):
        x0, y0 = rng.uniform(0, w * 0.6), rng.uniform(0, h * 0.6)
        x1, y1 = min(x0 + rng.uniform(20, w * 0.5), w), min(y0 + rng.uniform(20, h * 0.5), h)
        boxes[i] = [x0, y0, x1, y1]
        m = np.zeros((h, w), bool)
        for _ in range(3):
            cy, cx, ry, rx = rng.uniform(y0, y1), rng.uniform(x0, x1), rng.uniform(4, 60), rng.uniform(4, 90)
            m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        m &= (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1)          # what crop_mask leaves
        masks[i] = m
    ref = OS.masks2segments(masks)
    for dt in (torch.uint8, torch.float32):
        dm = torch.from_numpy(masks).cuda().to(dt)
        for bx in (boxes, None, torch.from_numpy(boxes).cuda()):
            got = PSG.masks2segments(dm, boxes=bx)
            assert len(got) == n
            for g, r in zip(got, ref):
                assert np.array_equal(g, r)
    assert sum(len(r) for r in ref) > 50


def test_folder_harness_writes_the_reference_csv(models, tmp_path):
    """run_folder (the file-in / file-out loop of yolov8_lmi/run_model.py:72-183): PNG files of mixed sizes and a grey
    one -> preds.csv in the reference's format; rows equal what predict_batch gives frame by frame, in file order."""
    import csv as _csv
    from PIL import Image
    from object_detectors_amd import run_folder
    det, seg = models
    rng = np.random.default_rng(5)
    src = tmp_path / "imgs"
    src.mkdir()
    sizes = [(240, 320), (200, 200), (333, 251), (240, 320), (180, 320)]
    for i, (h, w) in enumerate(sizes):
        im = synth.frames_u8(rng, 1, h, w)[0]
        Image.fromarray(im[:, :, 0] if i == 2 else im).save(src / f"f{i:02d}.png")
    (src / "notes.txt").write_text("not an image")
    for model, tag in ((det, "det"), (seg, "seg")):
        out = tmp_path / tag
        im, _ = model.letterbox([run_folder.load_rgb(str(src / f"f{i:02d}.png")) for i in range(5)], (256, 256))
        y = model.forward(im)
        conf = min(_conf_for((y[0] if isinstance(y, (list, tuple)) else y)[i:i + 1].cpu().numpy()) for i in range(5))
        s = run_folder.run(model, str(src), str(out), (256, 256), conf, True, batch_size=2, geometry="letterbox")
        assert s["images"] == 5 and s["images_per_s"] > 0 and os.path.isfile(out / "summary.json")
        rows = list(_csv.reader(open(out / "preds.csv"), delimiter=";"))
        want = []
        for p in sorted(os.listdir(src)):
            if not p.endswith(".png"):
                continue
            res, _ = model.predict_batch([run_folder.load_rgb(str(src / p))], conf, imgsz=(256, 256), return_segments=True)
            if len(res["boxes"]):
                segs = res["segments"][0] if tag == "seg" else None
                want += run_folder.result_rows(p, res["boxes"][0], res["scores"][0], res["classes"][0], segs)
        assert s["detections"] > 0 and len(rows) == len(want) and len(rows) % 2 == 0
        assert rows == [[str(v) for v in r] for r in want]
        kinds = {r[3] for r in rows}
        assert kinds == ({"polygon"} if tag == "seg" else {"rect"})
        if tag == "det":
            assert rows[0][4] == "upper left" and rows[1][4] == "lower right" and rows[0][7] == "angle" and rows[0][8] == "0.00"
    # the reference runner's own geometry (the default): stretch to --sz, post-process against the stretched frame,
    # un-stretch with /rw, /rh in fp32, truncate (run_model.py:83-93,115-121) -- rebuilt here from the oracle's pieces
    out = tmp_path / "stretch"
    frames = [run_folder.load_rgb(str(src / f"f{i:02d}.png")) for i in range(5)]
    im, _ = det.letterbox(frames, (256, 256), stretch=True)
    for i, f in enumerate(frames):
        assert np.array_equal(im[i].cpu().numpy(), OP.preprocess(OC.resize_linear_u8(f, 256, 256))[0])
    pred = det.forward(im).cpu().numpy()
    conf = min(_conf_for(pred[i:i + 1]) for i in range(5))
    s = run_folder.run(det, str(src), str(out), (256, 256), conf, True, batch_size=3)
    assert s["geometry"] == "stretch"
    rows = list(_csv.reader(open(out / "preds.csv"), delimiter=";"))
    want = []
    for i, f in enumerate(frames):
        boxes, scores, classes, _ = oracle_postprocess(pred[i:i + 1], None, (256, 256), (256, 256), conf, NAMES, 0.45)
        rh, rw = 256 / f.shape[0], 256 / f.shape[1]
        b = boxes.copy()
        b[:, [0, 2]] /= rw
        b[:, [1, 3]] /= rh
        want += run_folder.result_rows(f"f{i:02d}.png", b, scores, classes, None)
    assert len(want) > 0 and rows == [[str(v) for v in r] for r in want]


def test_letterbox_uploads_host_frames_through_pinned_staging(models):
    """Host frames (numpy arrays, CPU tensors, a 4-D host batch, pinned tensors, mixed with device tensors) give the
    tensor the device-resident frames give; the staging buffer is reused by the next call."""
    m = models[0]
    rng = np.random.default_rng(31)
    sizes = [(270, 480), (375, 500), (270, 480), (1080, 1920), (64, 48), (720, 1280)]
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    ref, ref_ops = m.letterbox([torch.from_numpy(f).cuda() for f in frames], (320, 320))
    for trial in range(2):          # the second call reuses the staging buffer
        got, ops_ = m.letterbox(frames, (320, 320))
        assert torch.equal(got, ref) and ops_ == ref_ops
    mixed = [frames[0], torch.from_numpy(frames[1]), torch.from_numpy(frames[2]).cuda(), torch.from_numpy(frames[3]).pin_memory(),
             frames[4], torch.from_numpy(frames[5])]
    got, _ = m.letterbox(mixed, (320, 320))
    assert torch.equal(got, ref)
    batch = rng.integers(0, 256, (9, 375, 500, 3), dtype=np.uint8)     # frames 562500 bytes apart: packed, read in place
    ref4, _ = m.letterbox(torch.from_numpy(batch).cuda(), (320, 320))
    got4, _ = m.letterbox(batch, (320, 320))
    assert torch.equal(got4, ref4)
    grey = rng.integers(0, 256, (4, 120, 200), dtype=np.uint8)
    g_ref, _ = m.letterbox([torch.from_numpy(g).cuda() for g in grey], (160, 160))
    g_got, _ = m.letterbox([g for g in grey], (160, 160))
    assert torch.equal(g_got, g_ref)
