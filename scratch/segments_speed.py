"""masks2segments on the seg640 bench volume: ~2900 1080p masks -> contours (device bit-pack + host tracing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth, segments as PSG
rng = np.random.default_rng(2)
B, nc, nm, S, h0, w0 = 64, 80, 32, 640, 1080, 1920
head = torch.from_numpy(synth.head(rng, B, nc, nm, S, 40)).cuda()
proto = torch.from_numpy(synth.proto(rng, B, nm, S // 4, S // 4)).cuda()
det, count, anchor, fin = ops.nms_v8(head, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(h0, w0), thres=0.25))
total = int(fin['slot_base'][-1].item())
masks, _ = ops.process_mask(proto, det, fin['slot_map'], fin['slot_base'], (h0, w0), total)
rows = det.reshape(-1, det.shape[-1])[fin['slot_map'][:total].long()]
boxes = rows[:, :4].contiguous()
for _ in range(2):
    t0 = time.perf_counter(); segs = PSG.masks2segments(masks[:total], boxes=boxes); t1 = time.perf_counter()
print(f"{total} masks 1080p -> segments in {(t1-t0)*1e3:.1f} ms ({sum(len(s) for s in segs)} points), {(t1-t0)/total*1e6:.1f} us/mask, threads={min(16, os.cpu_count())}")
t0 = time.perf_counter(); h = masks[:200].cpu(); t1 = time.perf_counter()
print(f"for scale: D2H of 200 whole masks {(t1-t0)*1e3:.1f} ms -> {(t1-t0)/200*total*1e3:.0f} ms for all")
