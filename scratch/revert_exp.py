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
cases = [(200, 1080, 1916), (200, 1920, 1080), (400, 720, 1276), (200, 375, 500), (400, 300, 100), (50, 2160, 3836), (100, 2048, 2448), (200, 1200, 1000)]
masks = (torch.rand((400, 640, 640), device="cuda", generator=g) > 0.5).float()
cfgs = [dict(), dict(LMI_RV_IT="9")]
for it in ("0",):
    for kb in (("12", "8"), ("12", "12"), ("16", "8"), ("16", "12"), ("16", "16"), ("20", "12"), ("20", "16"), ("24", "8"), ("24", "16"), ("24", "24"), ("32", "16")):
        cfgs.append(dict(LMI_RV_IT=it, LMI_RV_SRCKB=kb[0], LMI_RV_OUTKB=kb[1]))
for cfg in cfgs:
    for k in ("LMI_RV_IT", "LMI_RV_SRCKB", "LMI_RV_OUTKB"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    r = []
    for (m, h0, w0) in cases:
        gw, gh, L, R, T, B = ops.letterbox_geometry(h0, w0, 640, 640)
        opsl = [{"resize": [gw, gh, w0, h0]}, {"pad": [L, R, T, B]}]
        r.append(timeit(lambda: ops.revert_masks(masks[:m], opsl)))
    print(cfg, "  ".join(f"{t:.0f}" for t in r), flush=True)
