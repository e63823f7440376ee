"""paste_masks timing (n soft 28x28 masks into boxes of an image).  usage: python scratch/paste_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import hip_ops as ops
def timeit(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return t[len(t) // 2] * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for (n, H, W) in [(300, 1080, 1920), (30, 1080, 1920), (300, 480, 640), (100, 2160, 3840), (300, 1080, 1916), (300, 720, 1280)]:
    m = torch.rand((n, 28, 28), device="cuda", generator=g)
    x0 = torch.rand(n, device="cuda", generator=g) * W * 0.7; y0 = torch.rand(n, device="cuda", generator=g) * H * 0.7
    bw = torch.rand(n, device="cuda", generator=g) * W * 0.25 + 8; bh = torch.rand(n, device="cuda", generator=g) * H * 0.25 + 8
    boxes = torch.stack([x0, y0, x0 + bw, y0 + bh], 1)
    out = torch.empty((n, H, W), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: ops.paste_masks(m, boxes, (H, W)))
    tf = timeit(lambda: out.zero_())
    print(f"n={n} {H}x{W}: {t:.0f} us  {n*H*W/t/1e6:.2f} TB/s   (fill of the output: {tf:.0f} us)", flush=True)
