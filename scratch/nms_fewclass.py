"""NMS with few classes (industrial models: 1-5 classes): candidate filter and suppression pass timed apart.
usage: python scratch/nms_fewclass.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth

rng = np.random.default_rng(4)
for (B, nc, obj) in [(64, 1, 40), (64, 2, 40), (64, 3, 40), (64, 5, 40), (64, 10, 40), (64, 80, 40), (1, 2, 40), (1, 80, 40), (64, 2, 15), (64, 2, 100), (1, 2, 100), (64, 4, 100)]:
    head_np = synth.head(rng, B, nc, 0, 640, obj)
    head = torch.from_numpy(head_np).cuda()
    sc = head_np[:, 4:4 + nc]
    cand = (sc.max(1) > 0.25)
    ncand = cand.sum(1)
    cls = sc.argmax(1)
    big = max(int(np.bincount(cls[b][cand[b]], minlength=nc).max()) for b in range(B))
    for agn in (False, True):
        n = 20
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for _ in range(3):
            ops.nms_v8(head, 0.25, 0.45, nc=nc, agnostic=agn)
        torch.cuda.synchronize()
        for a, m, e in ev:
            a.record(); out = ops.nms_v8(head, 0.25, 0.45, nc=nc, agnostic=agn, decode_done=m); e.record()
        torch.cuda.synchronize()
        td = sorted(a.elapsed_time(m) for a, m, e in ev)[n // 2] * 1e3
        ts = sorted(m.elapsed_time(e) for a, m, e in ev)[n // 2] * 1e3
        print(f"B={B} nc={nc} objects={obj} agnostic={agn}: candidates max {int(ncand.max())}/img, largest class {big}; kept {int(out[1].sum())}; filter {td:.1f} us, suppress {ts:.1f} us", flush=True)
