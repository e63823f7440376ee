"""A few revert_masks calls for a PMC pass.  usage: python3 scratch/revert_pmc_run.py [uint8|float32] [h0 w0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import hip_ops as ops
dt = torch.uint8 if (len(sys.argv) < 2 or sys.argv[1] == "uint8") else torch.float32
h0, w0 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
g = torch.Generator(device="cuda").manual_seed(0)
gw, gh, L, R, T, B = ops.letterbox_geometry(h0, w0, 640, 640)
opsl = [{"resize": [gw, gh, w0, h0]}, {"pad": [L, R, T, B]}]
masks = (torch.rand((200, 640, 640), device="cuda", generator=g) > 0.5).to(dt)
for _ in range(4):
    out = ops.revert_masks(masks, opsl)
torch.cuda.synchronize()
print("ok", tuple(out.shape))
