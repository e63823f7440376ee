import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from object_detectors_amd import hip_ops as ops, synth
B = int(sys.argv[1]); objects = int(sys.argv[2])
rng = np.random.default_rng(1)
p = torch.from_numpy(synth.head(rng, B, 80, 0, 640, objects, cluster=(8, 16) if objects > 50 else (5, 16))).cuda()
out = ops.nms_v8(p, 0.25, 0.45, nc=80, agnostic=True)
for _ in range(30):
    ops.nms_v8(p, 0.25, 0.45, nc=80, agnostic=True, out=out)
torch.cuda.synchronize()
