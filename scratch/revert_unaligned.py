"""revert_masks on output rows that are / are not whole 128-byte lines.  usage: python scratch/revert_unaligned.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import hip_ops as ops


def timeit(fn, n=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return t[len(t) // 2] * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for dt in (torch.float32, torch.uint8):
    for (m, h0, w0) in [(200, 1080, 1920), (200, 1080, 1916), (200, 1080, 1912), (200, 1080, 1918), (200, 1920, 1080), (200, 1920, 1088),
                        (200, 375, 500), (200, 375, 512), (50, 1000, 1504), (50, 1000, 1500), (100, 2160, 3840), (100, 2160, 3836),
                        (400, 720, 1280), (400, 720, 1276), (400, 300, 100), (400, 300, 96)]:
        gw, gh, L, R, T, B = ops.letterbox_geometry(h0, w0, 640, 640)
        opsl = [{"resize": [gw, gh, w0, h0]}, {"pad": [L, R, T, B]}]
        masks = (torch.rand((m, 640, 640), device="cuda", generator=g) > 0.5).to(dt)
        t = timeit(lambda: ops.revert_masks(masks, opsl), n=9)
        es = masks.element_size()
        by = m * (gw * gh + h0 * w0) * es
        print(f"{str(dt)[6:]} m={m} -> {h0}x{w0}: {t:.0f} us  {by/t/1e6:.2f} TB/s", flush=True)

# masks as the path produces them: a filled ellipse per mask (5-15 % of the frame), zeros elsewhere
print("-- object-like masks (one filled ellipse each)")
yy, xx = torch.meshgrid(torch.arange(640, device="cuda"), torch.arange(640, device="cuda"), indexing="ij")
for dt in (torch.float32, torch.uint8):
    for (m, h0, w0) in [(200, 1080, 1920), (200, 1920, 1080), (400, 720, 1280), (100, 2160, 3840), (200, 375, 500)]:
        gw, gh, L, R, T, B = ops.letterbox_geometry(h0, w0, 640, 640)
        opsl = [{"resize": [gw, gh, w0, h0]}, {"pad": [L, R, T, B]}]
        cy = T + torch.rand(m, device="cuda", generator=g) * gh; cx = L + torch.rand(m, device="cuda", generator=g) * gw
        ry = (0.1 + 0.2 * torch.rand(m, device="cuda", generator=g)) * gh; rx = (0.1 + 0.2 * torch.rand(m, device="cuda", generator=g)) * gw
        masks = ((((yy[None] - cy[:, None, None]) / ry[:, None, None]) ** 2 + ((xx[None] - cx[:, None, None]) / rx[:, None, None]) ** 2) < 1).to(dt)
        t = timeit(lambda: ops.revert_masks(masks, opsl), n=9)
        es = masks.element_size()
        by = m * (gw * gh + h0 * w0) * es
        print(f"{str(dt)[6:]} m={m} -> {h0}x{w0}: {t:.0f} us  {by/t/1e6:.2f} TB/s  (mask area {float(masks.float().mean()) * 640 * 640 / (gw * gh):.2f} of the frame)", flush=True)
