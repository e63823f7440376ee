"""Letterbox (cv2 flavour) over general geometries, not only the exact x3 / x6 scales of the bench workloads:
time per launch of a batch, bytes that must move (whole source rows with a non-zero tap + the output) and the rate.
usage: python scratch/lb_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import hip_ops as ops


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return t[len(t) // 2] * 1e3


CASES = [(1080, 1920, 640, 64), (2160, 3840, 640, 64), (720, 1280, 640, 64), (960, 1280, 640, 64), (750, 1000, 640, 64),
         (800, 1333, 640, 64), (480, 640, 640, 64), (375, 500, 640, 64), (1200, 1600, 640, 64), (1080, 1920, 1280, 32),
         (1536, 2048, 1280, 32), (2160, 3840, 1280, 32), (1000, 1504, 640, 64), (1080, 1920, 960, 32)]
for (h, w, S, B) in CASES:
    fs = [torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device='cuda') for _ in range(3)]
    o = torch.empty((B, 3, S, S), device='cuda')
    k = [0]

    def rot():
        k[0] = (k[0] + 1) % 3
        ops.letterbox(fs[k[0]], (S, S), out=o, pad_value=114, swap_rb=True)
    t = timeit(rot)
    nw, nh, *_ = ops.letterbox_geometry(h, w, S, S)
    # source rows with a tap: all rows when the scale is below 2, else ~2 rows per output row (1 when the second weight is 0)
    sy = h / nh
    rows = set()
    import math
    for dy in range(nh):
        fy = (dy + 0.5) * sy - 0.5
        iy = math.floor(fy)
        a = round((fy - iy) * 2048)
        if a != 2048:
            rows.add(min(max(iy, 0), h - 1))
        if a != 0:
            rows.add(min(max(iy + 1, 0), h - 1))
    src = len(rows) * w * 3
    out = 3 * S * S * 4
    tot = B * (src + out)
    print(f"{h}x{w} -> {S} (content {nw}x{nh}) B={B}: {t:7.1f} us  rows {len(rows)}/{h}  bytes {tot/1e6:7.1f} MB  {tot/t/1e6:5.2f} TB/s", flush=True)
    del fs, o
