import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
cfg = bench.WORKLOADS['seg640']
dev = torch.device('cuda', 0)
frames, head, proto = bench.make_inputs(cfg, 2, dev)
pipe = bench.Pipeline(cfg, frames, head, proto)
det, count, anchor, fin, masks = pipe.step(); torch.cuda.synchronize()
total = int(fin['slot_base'][-1].item())
rows = det.reshape(-1, det.shape[-1])[fin['slot_map'][:total].long()].cpu().numpy()
w = np.ceil(rows[:, 2]) - np.ceil(rows[:, 0]); h = np.ceil(rows[:, 3]) - np.ceil(rows[:, 1])
print('slots', total, 'mean box w %.0f h %.0f' % (w.mean(), h.mean()), 'area fraction %.3f' % ((w * h).sum() / (total * 1080 * 1920)), 'rows fraction %.3f' % (h.sum() / (total * 1080)))
print('empty boxes', int(((w <= 0) | (h <= 0)).sum()), ' mask pixels set fraction %.4f' % (masks[:total].float().mean().item()))
