"""cProfile of Yolov8.postprocess on one 640x640 frame (detection, tensor in/out)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "object_detectors_amd", "dropin"))
import numpy as np, torch
from object_detectors_amd import synth
from object_detectors_amd.wrapper import Yolov8Core
from object_detectors_amd.nets.yolov8 import save_checkpoint
import tempfile
d = tempfile.mkdtemp()
path = os.path.join(d, "m.pt")
save_checkpoint(path, nc=80)
m = Yolov8Core(); m._init(path)
rng = np.random.default_rng(0)
head = torch.from_numpy(synth.head(rng, 1, 80, 0, 640, 40)).cuda()
img = torch.zeros(1, 3, 640, 640, device="cuda")
orig = torch.zeros(640, 640, 3, dtype=torch.uint8, device="cuda")
for _ in range(5): r = m.postprocess(head.clone(), img, orig, 0.25)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): r = m.postprocess(head, img, orig, 0.25)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
print("boxes", len(r["boxes"][0]))
