"""Latency of the drop-in's own path (verdict r01 item 5): predict() on one 640x640 frame (BASELINE config 1) and
predict_batch() on 64 x 1080p frames (config 2 shape, seg), fp32 masks (the reference's dtype, the default) and uint8.
Random-init YOLOv8n (nets/yolov8.py); the forward is reported separately."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "object_detectors_amd", "dropin"))
import numpy as np, torch
from object_detectors_amd import synth
from object_detectors_amd.nets.yolov8 import YoloV8
from yolov8_lmi.model import Yolov8
NAMES = {i: f"c{i}" for i in range(80)}
d = tempfile.mkdtemp()
models = {}
for seg in (False, True):
    torch.manual_seed(7)
    m = YoloV8("n", 80, seg, NAMES)
    for conv in m.head.cv3:
        conv[-1].bias.data.normal_(-2.0, 0.5); conv[-1].weight.data.mul_(8.0)
    path = os.path.join(d, "seg.pt" if seg else "det.pt")
    torch.save({"format": "lmi-yolov8-torch", "scale": "n", "nc": 80, "seg": seg, "names": NAMES, "state_dict": m.state_dict()}, path)
    models[seg] = Yolov8(path)
    models[seg].warmup([640, 640])

def conf_for(pred, frac=0.75):
    best = pred[:, 4:84].amax(1)
    return float(best.min() + frac * (best.max() - best.min()))


def sync_time(fn, n=20, warm=5):
    ts = []
    for it in range(n + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        if it >= warm:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, out

print("== one 640x640 frame (config 1): stages timed separately, each bracketed by synchronize; median ms of 20")
rng = np.random.default_rng(0)
frame = synth.frames_u8(rng, 1, 640, 640)[0]
frame_dev = torch.from_numpy(frame).cuda()
ops = [{"resize": [640, 640, 1920, 1080]}]
for seg in (False, True):
    model = models[seg]
    t_pre, im = sync_time(lambda: model.preprocess(frame))
    t_pre_dev, _ = sync_time(lambda: model.preprocess(frame_dev))
    t_fwd, y = sync_time(lambda: model.forward(im))
    pred = y[0] if isinstance(y, (list, tuple)) else y
    conf = conf_for(pred)
    for inp, tag in ((frame, "numpy in/out"), (frame_dev, "tensor in/out")):
        for rs in ((False, True) if seg else (False,)):
            t_post, res = sync_time(lambda: model.postprocess(y, im, inp, conf, 0.4, False, 300, rs))
            t_all, (out, ti) = sync_time(lambda: model.predict(inp, conf, operators=ops, return_segments=rs))
            print(("seg" if seg else "det"), tag, "segments" if rs else "no segments",
                  f"preprocess {t_pre if inp is frame else t_pre_dev:.3f}  forward {t_fwd:.3f}  postprocess {t_post:.3f}  predict() with revert to 1080p {t_all:.3f}",
                  "| boxes", len(out["boxes"]), "masks", len(out["masks"]) if "masks" in out else 0, flush=True)

print("== 64 x 1080p frames -> 640 (config 2 shape), seg, tensors in/out, return_segments=False; median ms of 10")
model = models[True]
frames = torch.randint(0, 256, (64, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
t_lb, (im, _) = sync_time(lambda: model.letterbox(frames, (640, 640)), 10, 3)
t_fwd, y = sync_time(lambda: model.forward(im), 10, 3)
origs = list(frames)
for frac in (0.8, 0.6):
    conf = conf_for(y[0], frac)
    for dt in (torch.float32, torch.uint8):
        model.mask_dtype = dt
        t_post, res = sync_time(lambda: model.postprocess(y, im, origs, conf, 0.4, False, 300, False), 10, 3)
        nmask = sum(len(m) for m in res["masks"])
        print(f"conf {conf:.3f} mask dtype {dt}: letterbox {t_lb:.3f}  forward {t_fwd:.3f}  postprocess {t_post:.3f} ms; masks {nmask}"
              f" ({nmask * 1080 * 1920 * (4 if dt == torch.float32 else 1) / 1e9:.2f} GB) -> {64 / ((t_lb + t_post) * 1e-3):.0f} images/s pre+post", flush=True)
        del res
