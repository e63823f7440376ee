"""Generate tests/golden/yolov5_helpers.npz by running the REFERENCE's own in-tree statement of the box / mask
arithmetic: the torch-only method bodies of `YoLov5TRT` in object_detectors/yolov5_lmi/trt/yolov5_trt.py
(`xywh2xyxy` :306-313, `box_iou` :285-303, `clip_boxes` :316-325, `scale_boxes` :328-341, `scale_segments`
:354-370, `crop_mask` :373-388, `process_mask` :391-415 and the candidate decode of `non_max_suppression`
:172-282).

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden_yolov5_helpers.py

The module imports cv2, torchvision and tensorrt at the top; none of them is installed here.  Empty module objects
are registered under those names for the import, and the methods are called unbound on an object created without
`__init__` (which would open a TensorRT engine).  No cv2 / tensorrt function is called.  `non_max_suppression`
calls `torchvision.ops.nms` in its middle (:267): that ONE attribute is a recorder which stores the arguments the
reference passes (class-offset boxes, sorted scores) and returns `arange(n)` -- it implements no NMS and the greedy
suppression stays PARITY UNPINNED; what the recording pins is everything the reference computes BEFORE that call
(objectness x class score, first-argmax class, strict `>` threshold, xywh -> xyxy, descending sort, `box + cls*7680`
in fp32) and the row layout it emits after it.  The v5 layout (B, A, 5+nc+nm) is fed with objectness 1.0, which
makes `conf = obj*cls` equal to the v8 class score bit for bit.

Only inputs (seeds / shapes, see inputs_yolov5_helpers.py) and outputs are stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF_FILE = "/root/reference/object_detectors/yolov5_lmi/trt/yolov5_trt.py"

import inputs_yolov5_helpers as IN  # noqa: E402


def load_reference():
    recorded = []
    tv = types.ModuleType("torchvision")
    tv.ops = types.ModuleType("torchvision.ops")

    def record_nms_arguments(boxes, scores, iou_thres):
        recorded.append((boxes.clone(), scores.clone(), float(iou_thres)))
        return torch.arange(boxes.shape[0])
    tv.ops.nms = record_nms_arguments
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.modules.setdefault("tensorrt", types.ModuleType("tensorrt"))
    sys.modules["torchvision"] = tv
    spec = importlib.util.spec_from_file_location("ref_yolov5_trt", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cls = mod.YoLov5TRT
    obj = object.__new__(cls)
    obj.use_mask = False
    return cls, obj, recorded


def main():
    T, o, recorded = load_reference()
    torch.set_num_threads(1)
    out = {"torch_version": np.array(torch.__version__)}

    # xywh2xyxy, box_iou
    xywh = IN.boxes_xywh(1, 256)
    out["xywh2xyxy"] = T.xywh2xyxy(o, torch.from_numpy(xywh)).numpy()
    a, b = IN.boxes_xyxy(2, 48, (640, 640)), IN.boxes_xyxy(3, 40, (640, 640))
    out["box_iou"] = T.box_iou(o, torch.from_numpy(a), torch.from_numpy(b)).numpy()

    # scale_boxes (+ clip_boxes) and scale_segments, un-rounded pad
    for i, (seed, n, s1, s0) in enumerate(IN.SCALE_CASES):
        bx = torch.from_numpy(IN.boxes_xyxy(seed, n, s1))
        out[f"scale_boxes{i}"] = T.scale_boxes(o, s1, bx, s0).numpy()
        pts = torch.from_numpy(IN.points_xy(seed, n, s1))
        out[f"scale_segments{i}"] = T.scale_segments(o, s1, pts, s0).numpy()
        cl = torch.from_numpy(IN.boxes_xyxy(seed, n, s1))
        T.clip_boxes(o, cl, s0)
        out[f"clip_boxes{i}"] = cl.numpy()

    # crop_mask on all-ones masks (the crop indicator itself) and process_mask at the prototype resolution
    for i, (seed, n, mh, mw) in enumerate(IN.MASK_CASES):
        proto, coef, boxes = IN.mask_inputs(seed, n, mh, mw)
        ind = T.crop_mask(o, torch.ones((n, mh, mw)), torch.from_numpy(boxes))
        out[f"crop_mask{i}"] = np.packbits(ind.numpy() != 0, axis=-1)
        pm = T.process_mask(o, torch.from_numpy(proto), torch.from_numpy(coef), torch.from_numpy(boxes), (mh, mw))
        assert pm.shape == (n, mh, mw)
        out[f"process_mask{i}"] = np.packbits(pm.numpy() != 0, axis=-1)
        # pixels whose logit is too close to 0 for sigmoid(x) > 0.5 to agree with x > 0 (none expected)
        logits = (torch.from_numpy(coef) @ torch.from_numpy(proto).view(32, -1)).view(n, mh, mw)
        out[f"process_mask{i}_near_zero"] = np.array(int((logits.abs() < 1e-6).sum()))

    # non_max_suppression: the decode before torchvision.ops.nms and the rows after it
    for i, (seed, size, objects, nc, nm, conf) in enumerate(IN.NMS_CASES):
        head = IN.nms_head(seed, size, objects, nc, nm)                       # (B, 4+nc+nm, A), v8 layout
        B, _, A = head.shape
        v5 = np.concatenate([head[:, :4], np.ones((B, 1, A), np.float32), head[:, 4:]], 1).transpose(0, 2, 1)
        recorded.clear()
        o.use_mask = nm == 32
        rows = T.non_max_suppression(o, torch.from_numpy(np.ascontiguousarray(v5)), [conf] * nc, iou_thres=0.45,
                                     max_det=IN.NMS_MAX_DET, nm=nm)
        assert len(recorded) == B
        for b in range(B):
            r = rows[b].numpy()
            assert 0 < len(r) < IN.NMS_MAX_DET and len(np.unique(r[:, 4])) == len(r), "scores must be distinct"
            out[f"nms{i}_rows{b}"] = r
            out[f"nms{i}_offset_boxes{b}"] = recorded[b][0].numpy()
            out[f"nms{i}_scores{b}"] = recorded[b][1].numpy()

    path = os.path.join(HERE, "yolov5_helpers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
