"""Suppressor lists (default where they apply) against the sorted sweep alone (sweep_only), HIP events around the call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, itertools
from object_detectors_amd import hip_ops as ops, synth
def med(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3
cases = [(16, 40, (5, 16), 80, True, 0.25, "typical agnostic"), (8, 110, (8, 16), 80, True, 0.25, "dense agnostic"), (4, 40, (5, 16), 80, True, 0.25, "typical agnostic"),
         (16, 40, (5, 16), 2, False, 0.25, "2 classes"), (16, 100, (5, 16), 2, False, 0.25, "2 classes dense"),
         (1, 40, (5, 16), 80, True, 0.25, "typical agnostic"), (1, 110, (8, 16), 80, True, 0.25, "dense agnostic"),
         (16, 110, (8, 16), 80, True, 0.25, "dense agnostic"), (64, 40, (5, 16), 80, True, 0.25, "typical agnostic"),
         (64, 110, (8, 16), 80, True, 0.25, "dense agnostic"),
         (1, 40, (5, 16), 80, True, 0.05, "conf 0.05 agnostic"), (16, 40, (5, 16), 80, True, 0.05, "conf 0.05 agnostic"),
         (64, 40, (5, 16), 2, False, 0.25, "2 classes"), (64, 100, (5, 16), 2, False, 0.25, "2 classes dense"), (1, 100, (5, 16), 2, False, 0.25, "2 classes dense"),
         (64, 40, (5, 16), 5, False, 0.25, "5 classes"), (32, 40, (5, 16), 1, False, 0.25, "1 class")]
for B, objects, cluster, nc, agn, conf, tag in cases:
    rng = np.random.default_rng(1)
    p = torch.from_numpy(synth.head(rng, B, nc, 0, 640, objects, cluster=cluster)).cuda()
    iou = 0.7 if conf < 0.2 else 0.45
    out = ops.nms_v8(p, conf, iou, nc=nc, agnostic=agn)
    res = []
    for so in (False, True):
        res.append(med(lambda: ops.nms_v8(p, conf, iou, nc=nc, agnostic=agn, out=out, sweep_only=so)))
    ncand = int((p[:, 4:].amax(1) > conf).sum().item()) // B
    print(f"B={B:3d} {tag:20s} cand/img {ncand:5d} kept {out[1].float().mean().item():6.1f}: lists {res[0]:7.1f} us, sweep only {res[1]:7.1f} us per call", flush=True)
