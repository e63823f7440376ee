import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from object_detectors_amd import segments as PSG
cfg = bench.WORKLOADS['seg640']
dev = torch.device('cuda', 0)
frames, head, proto = bench.make_inputs(cfg, 2, dev)
pipe = bench.Pipeline(cfg, frames, head, proto)
det, count, anchor, fin, masks = pipe.step(); torch.cuda.synchronize()
total = int(fin['slot_base'][-1].item())
rows = det.reshape(-1, det.shape[-1])[fin['slot_map'][:total].long()]
boxes = rows[:, :4].contiguous()
for _ in range(2):
    t0 = time.perf_counter(); segs = PSG.masks2segments(masks[:total], boxes=boxes); t1 = time.perf_counter()
print(f"{total} masks 1080p -> segments in {(t1-t0)*1e3:.1f} ms ({sum(len(s) for s in segs)} points), {(t1-t0)/total*1e6:.1f} us/mask, threads={min(16, os.cpu_count())}")
t0 = time.perf_counter(); h = masks[:200].cpu(); t1 = time.perf_counter()
print(f"for scale: D2H of 200 whole masks {(t1-t0)*1e3:.1f} ms -> {(t1-t0)/200*total*1e3:.0f} ms for all")
