"""revert_masks / paste_masks on widths that are / are not whole 16-byte chunks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]*1e3
m = (torch.rand(200, 640, 640, device="cuda") > 0.5).to(torch.uint8)
for (h, w) in [(1080, 1920), (1080, 1916), (375, 500), (375, 496)]:
    opsl = [{"resize": [640, 640, w, h]}]
    t = timeit(lambda: ops.revert_masks(m, opsl))
    print(f"revert 200 masks 640x640 -> {h}x{w}: {t:.0f} us = {200*h*w/t/1e6:.2f} TB/s written", flush=True)
soft = torch.rand(300, 28, 28, device="cuda")
bx = torch.rand(300, 4, device="cuda")
for (h, w) in [(1080, 1920), (1080, 1916), (375, 500)]:
    b = torch.stack([bx[:, 0] * w * 0.5, bx[:, 1] * h * 0.5, bx[:, 0] * w * 0.5 + 50 + bx[:, 2] * w * 0.4, bx[:, 1] * h * 0.5 + 50 + bx[:, 3] * h * 0.4], 1).contiguous()
    t = timeit(lambda: ops.paste_masks(soft, b, (h, w), 0.5))
    print(f"paste 300 masks -> {h}x{w}: {t:.0f} us = {300*h*w/t/1e6:.2f} TB/s written", flush=True)
