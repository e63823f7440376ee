"""Host logic of the drop-in boundary, on CPU: top-level module names, registry semantics
(object_detectors/object_detector.py:7-46), wrapper construction / thresholds / result-dict
assembly (object_detectors/yolov8_lmi/model.py:46-72, :156-174, :214-252), the product
pipeline_utils against the reference-generated fixtures, and contour tracing."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "object_detectors_amd", "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)

import gadget_utils.pipeline_utils as pipeline_utils  # noqa: E402
from object_detector import ObjectDetector  # noqa: E402
from od_base import ODBase  # noqa: E402
import od_models  # noqa: E402,F401
from yolov8_lmi.model import Yolov8  # noqa: E402
from ultralytics_lmi.yolo.model import Yolo  # noqa: E402
from object_detectors_amd.nets.yolov8 import save_checkpoint  # noqa: E402
from oracle import segments as SG  # noqa: E402
from object_detectors_amd import segments as PSG  # noqa: E402


@pytest.fixture(scope="module")
def weights(tmp_path_factory):
    d = tmp_path_factory.mktemp("w")
    det, seg = str(d / "y8n.pt"), str(d / "y8n-seg.pt")
    names = {i: f"c{i}" for i in range(6)}
    save_checkpoint(det, "n", 6, False, names)
    save_checkpoint(seg, "n", 6, True, names)
    return det, seg


def test_registry_semantics(weights):
    keys = set(ObjectDetector._registry)
    for v, names in (("v0", ["yolov8"]), ("v1", ["yolov8", "yolov11"])):
        for n in names:
            for t in ("od", "seg"):
                assert ("ultralytics", n, t, v, "{}") in keys
    m = ObjectDetector(metadata=dict(version='v0', model_name='yolov8', task='od', framework='ultralytics'), weights=weights[0], device='cpu')
    assert isinstance(m, Yolov8) and isinstance(m, ODBase) and not isinstance(m, ObjectDetector)
    m1 = ObjectDetector(metadata=dict(version='v1', model_name='yolov11', task='seg', framework='ultralytics'), model_path=weights[1], device='cpu')
    assert isinstance(m1, Yolo)
    with pytest.raises(ValueError):
        ObjectDetector(metadata=dict(version='v0', model_name='yolov8', task='od'), weights=weights[0])
    with pytest.raises(ValueError):
        ObjectDetector(metadata=dict(version='v9', model_name='yolov8', task='od', framework='ultralytics'), weights=weights[0])
    with pytest.raises(ValueError):
        ObjectDetector.register(metadata=dict(versions=['v0'], model_names=['yolov8'], tasks=['od'], frameworks=['ultralytics']))(Yolov8)
    with pytest.raises(TypeError):
        ODBase()


def test_wrapper_construction_thresholds_and_cpu_refusal(weights):
    with pytest.raises(FileNotFoundError):
        Yolov8("/nonexistent/weights.pt")
    m = Yolov8(weights[0], device='cpu')
    assert m.device.type == 'cpu' and m.names[3] == "c3"
    assert m.get_min_conf(0.5) == 0.5
    assert m.get_min_conf({"c1": 0.3, "c2": 0.6, "zzz": 0.01}) == 0.3
    assert m.get_min_conf({"zzz": 0.01}) == 1
    with pytest.raises(TypeError):
        m.get_min_conf(1)
    t = m._class_thresholds({"c1": 0.3, "c2": 0.6})
    assert t.dtype == torch.float64 and t.tolist() == [1, 0.3, 0.6, 1, 1, 1]
    m.warmup([64, 64])
    y = m.forward(torch.zeros(1, 3, 64, 64))
    assert y.shape == (1, 10, 84)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.preprocess(np.zeros((64, 64, 3), np.uint8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.postprocess(y, torch.zeros(1, 3, 64, 64), np.zeros((64, 64, 3), np.uint8), 0.5)
    with pytest.raises(TypeError):
        m.postprocess("bad", None, None, 0.5)


def test_result_dict_assembly_matches_reference_layout(weights):
    m = Yolov8(weights[0], device='cpu')
    det_h = np.zeros((3, 5, 6), np.float32)
    det_h[0, :3] = [[1, 2, 3, 4, 0.9, 0], [5, 6, 7, 8, 0.6, 2], [9, 10, 11, 12, 0.4, 5]]
    det_h[2, :1] = [[1, 1, 2, 2, 0.7, 1]]
    count_h = np.array([3, 0, 1])
    keep_h = np.zeros((3, 5), bool)
    keep_h[0, [0, 2]] = True            # the middle detection failed its per-class threshold
    origs = [np.zeros((8, 8, 3), np.uint8)] * 3
    r = m._assemble(None, None, det_h, count_h, keep_h, [None] * 3, origs, False, False)
    # image 1 had no NMS survivor: it contributes NO entry (model.py:219-220); image 2 has one row, filtered out
    assert len(r['boxes']) == 2 and set(r) == {'boxes', 'scores', 'classes'}
    assert isinstance(r['boxes'][0], np.ndarray) and r['boxes'][0].tolist() == [[1, 2, 3, 4], [9, 10, 11, 12]]
    assert r['classes'][0] == ['c0', 'c5'] and r['scores'][0].tolist() == [np.float32(0.9), np.float32(0.4)]
    assert r['boxes'][1].shape == (0, 4) and r['classes'][1] == []
    # tensor originals: per-image views of ONE gather of the survivors' rows (dense slot order) -- here built by hand
    kept = torch.from_numpy(np.concatenate([det_h[b, :count_h[b]][keep_h[b, :count_h[b]]] for b in range(3)]))
    rows_dev, slot_base = (kept[:, :4].contiguous(), kept[:, 4].contiguous()), np.array([0, 2, 2, 2])
    rt = m._assemble(rows_dev, slot_base, det_h, count_h, keep_h, [None] * 3, [torch.zeros(8, 8, 3)] * 3, False, False)
    assert isinstance(rt['boxes'][0], torch.Tensor) and isinstance(rt['classes'][0], list)
    assert rt['boxes'][0].tolist() == [[1, 2, 3, 4], [9, 10, 11, 12]] and rt['boxes'][1].shape == (0, 4) and rt['classes'][0] == ['c0', 'c5']


def test_pipeline_utils_matches_reference_fixtures(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "l1_geometry.json")))
    conv = lambda p, k: p if k == "list" else np.array(p, dtype={"f64": np.float64, "f32": np.float32}[k])
    for rec in gold["revert"]:
        x = conv(rec["pts"], rec["kind"])
        if rec["tensor"]:
            x = torch.as_tensor(x)
        got = pipeline_utils.revert_to_origin(x, rec["ops"])
        assert type(got) is (torch.Tensor if rec["tensor"] else (list if rec["kind"] == "list" else np.ndarray))
        assert np.array_equal(np.asarray(got.tolist() if hasattr(got, "tolist") else got, dtype=np.float64), np.asarray(rec["out"], dtype=np.float64))
    for rec in gold["apply"]:
        got = pipeline_utils.apply_operations(conv(rec["pts"], rec["kind"]), rec["ops"])
        assert np.array_equal(np.asarray(got, dtype=np.float64), np.asarray(rec["out"], dtype=np.float64))

    def pattern(h, w, c=3):
        y, x, ch = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
        return ((37 * y + 11 * x + 101 * ch) % 256).astype(np.uint8)
    for rec in gold["fit"]:
        im = pattern(rec["h"], rec["w"])
        for fn in (pipeline_utils.fit_im_to_size, pipeline_utils.fit_array_to_size):
            out, *pads = fn(im, W=rec["W"], H=rec["H"])
            assert pads == rec["pads"] and list(out.shape) == rec["shape"] and int(out.astype(np.int64).sum()) == rec["sum"]
        t, *pads = pipeline_utils.fit_im_to_size(torch.from_numpy(im), W=rec["W"], H=rec["H"])
        assert isinstance(t, torch.Tensor) and pads == rec["pads"]
    torch.set_num_threads(1)   # the fixtures were generated single-threaded (aten's CPU bilinear is thread-count dependent)
    for rec in gold["resize"]:
        out = pipeline_utils.resize_image(pattern(rec["h"], rec["w"]), **rec["kw"])
        assert list(out.shape) == rec["shape"] and str(out.dtype) == rec["dtype"]
        if "out" in rec:
            assert np.array_equal(out, np.array(rec["out"], dtype=np.uint8))
    for rec in gold["mask_revert"]:
        shape = rec["shape"]
        mk = pattern(*shape) if len(shape) == 3 else pattern(shape[0], shape[1], 1)[:, :, 0]
        out = pipeline_utils.revert_mask_to_origin(mk, rec["ops"])
        assert list(out.shape) == rec["out_shape"] and int(np.asarray(out).astype(np.int64).sum()) == rec["sum"]
    with pytest.raises(Exception):
        pipeline_utils.revert_to_origin(np.zeros((2, 3)), [])


def test_external_contours():
    m = np.zeros((12, 14), np.uint8)
    m[2:7, 3:9] = 1                     # rectangle 5x6 with a hole and a nested blob (not external)
    m[3:6, 4:8] = 0
    m[4, 5] = 1
    m[9, 11] = 1                        # isolated pixel
    cs = SG.find_external_contours(m)
    assert len(cs) == 2
    rect = {tuple(p) for p in cs[0].tolist()}
    assert rect == {(3, 2), (3, 6), (8, 6), (8, 2)}      # collinear runs collapsed to the corners
    assert cs[1].tolist() == [[11, 9]]
    segs = SG.masks2segments(np.stack([m, np.zeros_like(m)]))
    assert segs[0].shape == (5, 2) and segs[0].dtype == np.float32 and segs[1].shape == (0, 2)
    assert SG.masks2segments(np.stack([m]), strategy="largest")[0].shape == (4, 2)
    # a diagonal line: 8-connected, traced out and back
    d = np.eye(5, dtype=np.uint8)
    (c,) = SG.find_external_contours(d)
    assert c[0].tolist() == [0, 0] and [4, 4] in c.tolist()


def test_contour_tracer_matches_oracle_on_host_arrays():
    """The product's C++ tracer (packed bits, object_detectors_amd/csrc/contours.hip) against the numpy/scipy oracle,
    point for point: hand cases, random blobs, thin structures, nested components, both strategies, with and
    without box hints."""
    rng = np.random.default_rng(12)
    h, w = 70, 150
    yy, xx = np.mgrid[0:h, 0:w]
    masks = []
    m = np.zeros((h, w), np.uint8); m[2:7, 3:9] = 1; m[3:6, 4:8] = 0; m[4, 5] = 1; m[9, 11] = 1
    masks.append(m)
    masks.append(np.zeros((h, w), np.uint8))
    d = np.zeros((h, w), np.uint8); d[np.arange(40), np.arange(40) + 70] = 1      # diagonal line crossing a word boundary
    masks.append(d)
    for _ in range(8):                                                             # unions of ellipses, some touching the border
        m = np.zeros((h, w), bool)
        for _ in range(int(rng.integers(1, 5))):
            cy, cx, ry, rx = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(2, 25), rng.uniform(2, 50)
            m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        masks.append(m.astype(np.uint8))
    masks.append((rng.random((h, w)) < 0.5).astype(np.uint8))                      # salt and pepper: many tiny components
    ring = (((yy - 35) / 30.0) ** 2 + ((xx - 75) / 60.0) ** 2 < 1) & ~(((yy - 35) / 24.0) ** 2 + ((xx - 75) / 50.0) ** 2 < 1)
    ring |= ((yy - 35) ** 2 + (xx - 75) ** 2 < 30)                                 # a blob inside the ring's hole: not external
    masks.append(ring.astype(np.uint8))
    full = np.ones((h, w), np.uint8)
    masks.append(full)
    stack = np.stack(masks)
    for strategy in ("all", "largest"):
        ref = SG.masks2segments(stack, strategy=strategy)
        got = PSG.masks2segments(stack, strategy=strategy, threads=3)
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            assert g.dtype == np.float32 and g.shape == r.shape and np.array_equal(g, r)
    # box hints: any box that contains the mask's pixels gives the same points
    boxes = np.array([[0, 0, w, h]] * len(masks), np.float32)
    ys, xs = np.nonzero(masks[3])
    boxes[3] = [xs.min() - 0.5, ys.min() - 0.3, xs.max() + 0.2, ys.max() + 0.9]
    got = PSG.masks2segments(stack, boxes=boxes)
    for g, r in zip(got, SG.masks2segments(stack)):
        assert np.array_equal(g, r)
    assert PSG.masks2segments(stack[:0]) == []
    # float and torch CPU inputs
    got = PSG.masks2segments(torch.from_numpy(stack[:4].astype(np.float32)))
    for g, r in zip(got, SG.masks2segments(stack[:4])):
        assert np.array_equal(g, r)


def test_folder_harness_csv_rows_and_batches(tmp_path):
    """Host logic of object_detectors_amd/run_folder.py: the reference's csv rows (lmi_utils/label_utils/csv_utils.py:
    125-129), last detection first (yolov8_lmi/run_model.py:108), truncating integer coordinates, sorted batches."""
    import csv as _csv
    from object_detectors_amd import run_folder as RF
    rows = RF.result_rows("a.png", np.array([[1.9, 2.2, 30.7, 40.5], [5.0, 6.0, 7.0, 8.9]], np.float32), [0.91234, 0.5], ["cat", "dog"])
    assert rows == [["a.png", "dog", "0.5000", "rect", "upper left", 5, 6, "angle", "0.00"],
                    ["a.png", "dog", "0.5000", "rect", "lower right", 7, 8, "angle", "0.00"],
                    ["a.png", "cat", "0.9123", "rect", "upper left", 1, 2, "angle", "0.00"],
                    ["a.png", "cat", "0.9123", "rect", "lower right", 30, 40, "angle", "0.00"]]
    segs = [np.array([[1.5, 2.5], [3.5, 4.5], [5.0, 6.0]], np.float32), np.zeros((0, 2), np.float32)]
    rows = RF.result_rows("b.png", np.zeros((2, 4), np.float32), [0.7, 0.6], ["cat", "dog"], segs)
    assert rows == [["b.png", "cat", "0.7000", "polygon", "x values", 1, 3, 5], ["b.png", "cat", "0.7000", "polygon", "y values", 2, 4, 6]]
    RF.write_rows(rows, str(tmp_path / "p.csv"))
    assert list(_csv.reader(open(tmp_path / "p.csv"), delimiter=";"))[1] == ["b.png", "cat", "0.7000", "polygon", "y values", "2", "4", "6"]
    for name in ("b.png", "a.jpg", "c.txt", "d.BMP"):
        (tmp_path / name).write_bytes(b"x")
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "e.jpeg").write_bytes(b"x")
    got = RF.image_path_batches(2, str(tmp_path))
    assert [[os.path.basename(p) for p in b] for b in got] == [["a.jpg", "b.png"], ["d.BMP", "e.jpeg"]]
