"""process_mask on frame widths that are / are not whole 16-byte chunks (uint8 masks): the one-pass kernel's byte-store path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
rng = np.random.default_rng(3)
B, nc, nm, S = 64, 80, 32, 640
head = torch.from_numpy(synth.head(rng, B, nc, nm, S, 40)).cuda()
proto = torch.from_numpy(synth.proto(rng, B, nm, S // 4, S // 4)).cuda()
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]*1e3
for (h0, w0) in [(375, 500), (375, 496), (375, 512), (427, 640), (1080, 1920), (1080, 1916), (1080, 1924)]:
    det, count, anchor, fin = ops.nms_v8(head, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(h0, w0), thres=0.25))
    total = int(fin["slot_base"][-1].item())
    out = torch.empty((total, h0, w0), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: ops.process_mask(proto, det, fin["slot_map"], fin["slot_base"], (h0, w0), total, torch.uint8, out=out))
    print(f"{h0}x{w0}: {total} masks, {total*h0*w0/1e6:.0f} MB, logits+upsample {t:.0f} us = {total*h0*w0/t/1e6:.2f} TB/s", flush=True)
