import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from object_detectors_amd import hip_ops as ops
cfg = dict(bench.WORKLOADS['seg640']); cfg['batch'] = 16
dev = torch.device('cuda', 0)
frames, head, proto = bench.make_inputs(cfg, 2, dev)
det, count, anchor = ops.nms_v8(head, 0.25, 0.45, nc=80)
fin = ops.finalize_detections(det, count, 80, (640, 640), (1080, 1920), 0.25)
total = int(fin['slot_base'][-1].item())
for dt in (torch.float32, torch.uint8):
    out = torch.empty((total, 1080, 1920), dtype=dt, device=dev)
    evs = []
    for _ in range(8):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ops.process_mask(proto, det, fin['slot_map'], fin['slot_base'], (1080, 1920), total, dt, out=out, split_events=ev); evs.append(ev)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)[4]
    print(dt, total, 'masks', f'{t:.3f} ms', f'{out.numel()*out.element_size()/t/1e6:.0f} GB/s')
