"""cProfile of the segments step of Yolov8.postprocess (return_segments=True, the reference's default) on one seg frame."""
import cProfile, pstats, os, sys, io, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "object_detectors_amd", "dropin"))
import numpy as np, torch, tempfile
from object_detectors_amd import synth
from object_detectors_amd.wrapper import Yolov8Core
from object_detectors_amd.nets.yolov8 import save_checkpoint
path = os.path.join(tempfile.mkdtemp(), "m.pt"); save_checkpoint(path, nc=80, seg=True)
m = Yolov8Core(); m._init(path); m.mask_dtype = torch.uint8
rng = np.random.default_rng(0)
head = torch.from_numpy(synth.head(rng, 1, 80, 32, 640, 14)).cuda()
proto = torch.from_numpy(synth.proto(rng, 1, 32, 160, 160)).cuda()
img = torch.zeros(1, 3, 640, 640, device="cuda")
for name, orig in (("tensor", torch.zeros(640, 640, 3, dtype=torch.uint8, device="cuda")), ("numpy", np.zeros((640, 640, 3), np.uint8))):
    for seg in (False, True):
        for _ in range(5): r = m.postprocess((head.clone(), proto), img, orig, 0.25, return_segments=seg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): r = m.postprocess((head.clone(), proto), img, orig, 0.25, return_segments=seg)
        torch.cuda.synchronize()
        print(f"{name} in/out, return_segments={seg}: {(time.perf_counter()-t0)*10:.3f} ms per call, {len(r['boxes'][0])} masks")
orig = torch.zeros(640, 640, 3, dtype=torch.uint8, device="cuda")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): r = m.postprocess((head.clone(), proto), img, orig, 0.25, return_segments=True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5500])
