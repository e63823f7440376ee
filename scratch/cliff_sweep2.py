"""Second look for performance cliffs: letterbox flavours (torch, grey, fp16, ragged lists), preprocess, revert_masks,
paste_masks over sizes off the bench shapes.  usage: python scratch/cliff_sweep2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
LB = "nolb" not in sys.argv
print("-- letterbox flavours, 64 frames -> 640x640 (us per launch; bytes = source + output)")
for (h, w) in ([(1080, 1920), (750, 1000), (375, 500), (2160, 3840)] if LB else []):
    f = torch.randint(0, 256, (64, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    o32 = torch.empty((64, 3, 640, 640), device="cuda"); o16 = torch.empty((64, 3, 640, 640), device="cuda", dtype=torch.float16)
    fl = list(f.unbind(0))
    gray = f[..., :1].contiguous()
    r = {}
    r["cv2 uniform"] = timeit(lambda: ops.letterbox(f, (640, 640), out=o32))
    r["cv2 fp16"] = timeit(lambda: ops.letterbox(f, (640, 640), out=o16, fp16=True))
    r["cv2 list"] = timeit(lambda: ops.letterbox(fl, (640, 640), out=o32))
    r["cv2 stretch"] = timeit(lambda: ops.letterbox(f, (640, 640), out=o32, stretch=True))
    r["torch list"] = timeit(lambda: ops.letterbox(fl, (640, 640), out=o32, mode="torch", pad_value=0))
    r["grey"] = timeit(lambda: ops.letterbox(gray, (640, 640), out=o32))
    print(f"{h}x{w}: " + "  ".join(f"{k} {v:.0f}" for k, v in r.items()), flush=True)
    del f, fl, gray
sizes = [(480 + 7 * i, 640 + 13 * i) for i in range(64)]
fl = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda", generator=g) for h, w in sizes]
o32 = torch.empty((64, 3, 640, 640), device="cuda")
print(f"64 frames of 64 different sizes (480x640 .. 921x1459): cv2 {timeit(lambda: ops.letterbox(fl, (640, 640), out=o32)):.0f} us, torch {timeit(lambda: ops.letterbox(fl, (640, 640), out=o32, mode='torch')):.0f} us", flush=True)

print("-- preprocess_u8 (one image, us; bytes = in + fp32 out)")
for (h, w, c) in [(640, 640, 3), (1080, 1920, 3), (2160, 3840, 3), (1080, 1920, 1), (333, 1001, 3)]:
    im = torch.randint(0, 256, (h, w, c), dtype=torch.uint8, device="cuda", generator=g)
    t = timeit(lambda: ops.preprocess_u8(im))
    by = h * w * c + 3 * h * w * 4
    print(f"{h}x{w}x{c}: {t:.1f} us  {by/t/1e6:.2f} TB/s", flush=True)

print("-- revert_masks (m masks 640x640 letterboxed from the original; us; bytes = in + out)")
for dt in (torch.uint8, torch.float32):
    for (m, h0, w0) in [(200, 1080, 1920), (20, 1080, 1920), (200, 375, 500), (200, 2160, 3840), (50, 1000, 1504), (200, 1080, 1916), (200, 1920, 1080)]:
        gw, gh, L, R, T, B = ops.letterbox_geometry(h0, w0, 640, 640)
        opsl = [{"resize": [gw, gh, w0, h0]}, {"pad": [L, R, T, B]}]
        masks = (torch.rand((m, 640, 640), device="cuda", generator=g) > 0.5).to(dt)
        t = timeit(lambda: ops.revert_masks(masks, opsl), n=8)
        es = masks.element_size()
        by = m * (gw * gh + h0 * w0) * es
        print(f"{str(dt)[6:]} m={m} -> {h0}x{w0}: {t:.0f} us  {by/t/1e6:.2f} TB/s", flush=True)
    opsl = [{"resize": [640, 640, 1920, 1080]}, {"flip": [True, False, 1920, 1080]}]
    masks = (torch.rand((100, 640, 640), device="cuda", generator=g) > 0.5).to(dt)
    t = timeit(lambda: ops.revert_masks(masks, opsl), n=8)
    print(f"{str(dt)[6:]} m=100 stretch 640x640 -> 1080x1920 + flip: {t:.0f} us", flush=True)

print("-- paste_masks (n soft 28x28 masks into boxes of an image; us; bytes = n*H*W out)")
for (n, H, W) in [(300, 1080, 1920), (30, 1080, 1920), (300, 480, 640), (100, 2160, 3840), (300, 1080, 1916)]:
    m = torch.rand((n, 28, 28), device="cuda", generator=g)
    x0 = torch.rand(n, device="cuda", generator=g) * W * 0.7; y0 = torch.rand(n, device="cuda", generator=g) * H * 0.7
    bw = torch.rand(n, device="cuda", generator=g) * W * 0.25 + 8; bh = torch.rand(n, device="cuda", generator=g) * H * 0.25 + 8
    boxes = torch.stack([x0, y0, x0 + bw, y0 + bh], 1)
    t = timeit(lambda: ops.paste_masks(m, boxes, (H, W)), n=8)
    print(f"n={n} {H}x{W}: {t:.0f} us  {n*H*W/t/1e6:.2f} TB/s", flush=True)
