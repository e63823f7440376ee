"""Per-call latency of the drop-in predict() on one 640x640 frame (random-init YOLOv8n heads), split as the wrapper reports it."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "object_detectors_amd", "dropin"))
import numpy as np, torch
from object_detectors_amd import synth
from object_detectors_amd.nets.yolov8 import YoloV8
from yolov8_lmi.model import Yolov8
NAMES = {i: f"c{i}" for i in range(80)}
d = tempfile.mkdtemp()
for seg in (False, True):
    torch.manual_seed(7)
    m = YoloV8("n", 80, seg, NAMES)
    for conv in m.head.cv3:
        conv[-1].bias.data.normal_(-2.0, 0.5); conv[-1].weight.data.mul_(8.0)
    path = os.path.join(d, "seg.pt" if seg else "det.pt")
    torch.save({"format": "lmi-yolov8-torch", "scale": "n", "nc": 80, "seg": seg, "names": NAMES, "state_dict": m.state_dict()}, path)
    model = Yolov8(path)
    model.warmup([640, 640])
    rng = np.random.default_rng(0)
    frame = synth.frames_u8(rng, 1, 640, 640)[0]
    ops = [{"resize": [640, 640, 1920, 1080]}]
    im = model.preprocess(frame); y = model.forward(im)
    pred = (y[0] if isinstance(y, (list, tuple)) else y)
    best = pred[:, 4:84].amax(1)
    conf = float((best.min() + 3 * best.max()) / 4)
    print('candidates', int((best > conf).sum()), 'conf', conf)
    for rs in (False, True):
        acc = {}
        n = 0
        for it in range(25):
            out, ti = model.predict(frame, conf, operators=ops, return_segments=rs)
            torch.cuda.synchronize()
            if it >= 5:
                for k, v in ti.items(): acc[k] = acc.get(k, 0) + v
                n += 1
        nb = len(out["boxes"]); nm_ = len(out["masks"]) if "masks" in out else 0
        print(("seg" if seg else "det"), "segments" if rs else "no segments", {k: round(v / n * 1e3, 3) for k, v in acc.items()}, "ms; boxes", nb, "masks", nm_, flush=True)
