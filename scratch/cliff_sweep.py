"""Looks for performance cliffs off the bench shapes: process_mask over original-frame sizes / dtypes / model sizes,
NMS over class counts, anchor counts, extra channels and batch sizes.  usage: python scratch/cliff_sweep.py [mask|nms]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in e)
    return t[len(t) // 2] * 1e3


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("mask", "all"):
    rng = np.random.default_rng(3)
    for S, B in ((640, 32), (1280, 8), (320, 32)):
        nc, nm = 80, 32
        head = torch.from_numpy(synth.head(rng, B, nc, nm, S, 40)).cuda()
        proto = torch.from_numpy(synth.proto(rng, B, nm, S // 4, S // 4)).cuda()
        for (h0, w0) in [(1080, 1920), (1080, 1916), (375, 500), (480, 640), (2160, 3840), (720, 1280), (100, 100), (64, 48), (1200, 1600), (640, 640),
                         (1920, 1080), (3000, 4000), (333, 1001)]:
            for dt in (torch.uint8, torch.float32):
                det, count, anchor, fin = ops.nms_v8(head, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(h0, w0), thres=0.25))
                total = int(fin["slot_base"][-1].item())
                es = 1 if dt == torch.uint8 else 4
                if total * h0 * w0 * es > 40e9:
                    total_cap = int(40e9 // (h0 * w0 * es))
                else:
                    total_cap = total
                out = torch.empty((total_cap, h0, w0), dtype=dt, device="cuda")
                try:
                    t = timeit(lambda: ops.process_mask(proto, det, fin["slot_map"], fin["slot_base"], (h0, w0), total_cap, dt, out=out))
                except Exception as e:
                    print(f"S={S} {h0}x{w0} {dt}: FAILED {e}", flush=True); continue
                by = total_cap * h0 * w0 * es
                print(f"S={S} B={B} {h0}x{w0} {str(dt)[6:]}: {total_cap} masks, {by/1e6:.0f} MB, {t:.0f} us = {by/t/1e6:.2f} TB/s", flush=True)
                del out
if what in ("nms", "all"):
    rng = np.random.default_rng(4)
    for (B, nc, nm, S, obj) in [(64, 80, 0, 640, 40), (1, 80, 0, 640, 40), (64, 1, 0, 640, 40), (64, 2, 0, 640, 40), (64, 3, 32, 640, 40), (64, 91, 0, 640, 40),
                                (16, 1000, 0, 640, 40), (64, 80, 51, 640, 20), (64, 80, 32, 320, 20), (16, 80, 32, 1280, 60), (64, 80, 0, 608, 40),
                                (64, 80, 0, 416, 40), (8, 80, 0, 640, 300), (64, 15, 1, 640, 40)]:
        A = synth.anchors_for(S)
        head = torch.from_numpy(synth.head(rng, B, nc, nm, S, obj)).cuda()
        for agn in (False, True):
            t = timeit(lambda: ops.nms_v8(head, 0.25, 0.45, nc=nc, agnostic=agn))
            det, count, anchor = ops.nms_v8(head, 0.25, 0.45, nc=nc, agnostic=agn)
            by = head.numel() * 4
            print(f"B={B} nc={nc} nm={nm} A={A} objects={obj} agnostic={agn}: kept {int(count.sum())}  {t:.1f} us  ({by/t/1e6:.2f} TB/s of the head, {t/B:.2f} us/image)", flush=True)
