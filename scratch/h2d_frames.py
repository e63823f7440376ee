"""Host frames -> device for the batched letterbox: what the upload of 64 x 1080p numpy frames costs by route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops
B, h, w = 64, 1080, 1920
rng = np.random.default_rng(0)
frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(B)]
dev = torch.device("cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
print("pageable, one .to() per frame: %.1f ms" % t(lambda: [torch.from_numpy(f).to(dev) for f in frames]))
pin = torch.empty((B, h, w, 3), dtype=torch.uint8, pin_memory=True)
dst = torch.empty((B, h, w, 3), dtype=torch.uint8, device=dev)
def staged():
    for i, f in enumerate(frames):
        pin[i].copy_(torch.from_numpy(f))
        dst[i].copy_(pin[i], non_blocking=True)
print("pinned staging, per-frame memcpy + async H2D: %.1f ms" % t(staged))
def staged_np():
    pn = pin.numpy()
    for i, f in enumerate(frames):
        np.copyto(pn[i], f)
    dst.copy_(pin, non_blocking=True)
print("pinned staging, numpy memcpy of all frames then one H2D: %.1f ms" % t(staged_np))
print("already pinned, one H2D: %.1f ms" % t(lambda: dst.copy_(pin, non_blocking=True)))
import concurrent.futures as cf
ex = cf.ThreadPoolExecutor(8)
def staged_mt():
    pn = pin.numpy()
    def one(i):
        np.copyto(pn[i], frames[i]); dst[i].copy_(pin[i], non_blocking=True)
    list(ex.map(one, range(B)))
print("pinned staging, 8 host threads (memcpy + async H2D per frame): %.1f ms" % t(staged_mt))
out = torch.empty((B, 3, 640, 640), device=dev)
print("letterbox kernel on the resident batch: %.3f ms" % t(lambda: ops.letterbox(dst, (640, 640), out=out)))

# the wrapper's own entry on host frames (Yolov8.letterbox / predict_batch)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "object_detectors_amd", "dropin"))
import tempfile
from object_detectors_amd.wrapper import Yolov8Core
from object_detectors_amd.nets.yolov8 import save_checkpoint
path = os.path.join(tempfile.mkdtemp(), "m.pt"); save_checkpoint(path, nc=80)
m = Yolov8Core(); m._init(path)
print("Yolov8.letterbox(list of 64 numpy 1080p frames): %.1f ms" % t(lambda: m.letterbox(frames, (640, 640))))
batch = np.stack(frames)
print("Yolov8.letterbox((64,1080,1920,3) numpy batch): %.1f ms" % t(lambda: m.letterbox(batch, (640, 640))))
small = [f[:480, :640].copy() for f in frames]
print("Yolov8.letterbox(list of 64 numpy 480x640 frames): %.2f ms" % t(lambda: m.letterbox(small, (640, 640))))
print("  the same through a pageable .to() per frame: %.2f ms" % t(lambda: ops.letterbox([torch.from_numpy(f).to(dev) for f in small], (640, 640))))
