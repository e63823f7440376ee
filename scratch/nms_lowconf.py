"""NMS at validation-style thresholds (conf 0.001: nearly every anchor is a candidate), per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
rng = np.random.default_rng(1)
for B in (1, 16, 64):
    p = torch.from_numpy(synth.head(rng, B, 80, 0, 640, 40)).cuda()
    for conf in (0.25, 0.05, 0.01, 0.001):
        for agn in (False, True):
            n = 20
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for _ in range(3): ops.nms_v8(p, conf, 0.7, nc=80, agnostic=agn)
            torch.cuda.synchronize()
            for a, m, e in ev:
                a.record(); out = ops.nms_v8(p, conf, 0.7, nc=80, agnostic=agn, decode_done=m); e.record()
            torch.cuda.synchronize()
            td = sorted(a.elapsed_time(m) for a, m, e in ev)[n // 2] * 1e3
            ts = sorted(m.elapsed_time(e) for a, m, e in ev)[n // 2] * 1e3
            ncand = int((p[:, 4:].amax(1) > conf).sum().item()) // B
            print(f"B={B} conf={conf} agnostic={agn}: {ncand} candidates/img, kept {out[1].float().mean().item():.0f}/img: filter {td:.0f} us, suppress {ts:.0f} us", flush=True)
