import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import hip_ops as ops
def timeit(fn, n=9):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return t[len(t) // 2] * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
r = []
for (m, h0, w0) in [(200, 1080, 1920), (200, 1920, 1080), (400, 720, 1280), (100, 2160, 3840), (100, 2048, 1024)]:
    gw, gh, L, R, T, B = ops.letterbox_geometry(h0, w0, 640, 640)
    opsl = [{"resize": [gw, gh, w0, h0]}, {"pad": [L, R, T, B]}]
    masks = (torch.rand((m, 640, 640), device="cuda", generator=g) > 0.5).to(torch.uint8)
    out = torch.empty((m, h0, w0), dtype=torch.uint8, device="cuda")
    r.append(f"{h0}x{w0}: {timeit(lambda: ops.revert_masks(masks, opsl)):.0f} (fill {timeit(lambda: out.zero_()):.0f})")
print(sys.argv[1] if len(sys.argv) > 1 else "", "  ".join(r))
