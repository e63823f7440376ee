"""process_mask on a ragged batch (64 originals of 64 different sizes) against the uniform call on the same volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in e)
    return t[len(t)//2]*1e3
rng = np.random.default_rng(3)
B, nc, nm, S = 64, 80, 32, 640
head = torch.from_numpy(synth.head(rng, B, nc, nm, S, 40)).cuda()
proto = torch.from_numpy(synth.proto(rng, B, nm, S // 4, S // 4)).cuda()
for name, sizes in [("64 sizes 720x960 .. 1098x1464, widths multiples of 16", [(720 + 6 * i, 960 + 16 * (i // 2)) for i in range(B)]),
                    ("64 sizes 480x640 .. 921x1459, odd widths", [(480 + 7 * i, 640 + 13 * i) for i in range(B)]),
                    ("uniform 1080x1920 through the ragged entry", [(1080, 1920)] * B)]:
    ohw = torch.tensor(sizes, dtype=torch.int32, device="cuda")
    det, count, anchor, fin = ops.nms_v8(head, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=ohw, thres=0.25))
    sb = fin["slot_base"].cpu().numpy()
    counts = [int(sb[i + 1] - sb[i]) for i in range(B)]
    by = sum(c * h * w for c, (h, w) in zip(counts, sizes))
    t = timeit(lambda: ops.process_mask_ragged(proto, det, fin["slot_map"], fin["slot_base"], sizes, counts))
    print(f"{name}: {sum(counts)} masks, {by/1e6:.0f} MB: {t:.0f} us = {by/t/1e6:.2f} TB/s (includes the output allocation and the descriptor upload)", flush=True)
det, count, anchor, fin = ops.nms_v8(head, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(1080, 1920), thres=0.25))
total = int(fin["slot_base"][-1].item())
out = torch.empty((total, 1080, 1920), dtype=torch.uint8, device="cuda")
t = timeit(lambda: ops.process_mask(proto, det, fin["slot_map"], fin["slot_base"], (1080, 1920), total, out=out))
print(f"uniform entry 1080x1920: {total} masks: {t:.0f} us = {total*1080*1920/t/1e6:.2f} TB/s")
