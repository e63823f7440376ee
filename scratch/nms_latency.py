import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
import itertools
for (B, objects, cluster, tag), agn in itertools.product([(1, 40, (5, 16), "typical"), (1, 110, (8, 16), "dense"), (64, 40, (5, 16), "typical"), (16, 110, (8, 16), "dense")], (False, True)):
    rng = np.random.default_rng(1)
    p = torch.from_numpy(synth.head(rng, B, 80, 0, 640, objects, cluster=cluster)).cuda()
    out = ops.nms_v8(p, 0.25, 0.45, nc=80, agnostic=agn)
    for _ in range(5): ops.nms_v8(p, 0.25, 0.45, nc=80, agnostic=agn, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for a, b in ev:
        a.record(); ops.nms_v8(p, 0.25, 0.45, nc=80, agnostic=agn, out=out); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    ncand = int((p[:, 4:].amax(1) > 0.25).sum().item()) // B
    print(f"B={B:3d} {tag:8s} {'agnostic ' if agn else 'per-class'} cand/img {ncand:5d} kept {out[1].float().mean().item():6.1f}: median {t[25]*1e3:7.1f} us per call, {t[25]*1e3/B:6.2f} us/image")
