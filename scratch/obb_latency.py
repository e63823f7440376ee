"""Rotated NMS (Yolov8Obb) latency: HIP events around nms_v8(rotated=True), head (B, 4+15+1, A)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]*1e3
rng = np.random.default_rng(4)
for S, B, objects in [(640, 1, 40), (640, 16, 40), (1024, 1, 80), (1024, 16, 80), (1024, 1, 300)]:
    head = synth.head(rng, B, 15, 1, S, objects=objects)
    head[:, -1] = (rng.random(head[:, -1].shape) - 0.5) * 1.5
    p = torch.from_numpy(head).cuda()
    d, c, a = ops.nms_v8(p, 0.25, 0.45, nc=15, rotated=True)
    t = timeit(lambda: ops.nms_v8(p, 0.25, 0.45, nc=15, rotated=True))
    d2, c2, a2 = ops.nms_v8(p[:, :19].contiguous(), 0.25, 0.45, nc=15)
    t2 = timeit(lambda: ops.nms_v8(p[:, :19].contiguous(), 0.25, 0.45, nc=15))
    print(f"S={S} B={B} objects={objects}: rotated {t:.1f} us (kept {c.float().mean():.0f}/img), axis-aligned {t2:.1f} us (kept {c2.float().mean():.0f})", flush=True)
