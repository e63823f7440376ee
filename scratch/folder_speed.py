"""End-to-end folder harness on synthetic 1080p JPEGs: images/s and where the time goes (run_folder.run)."""
import os, sys, time, tempfile, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "object_detectors_amd", "dropin"))
import numpy as np, torch
from PIL import Image
from object_detectors_amd import run_folder
from object_detectors_amd.nets.yolov8 import save_checkpoint
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
base = (np.kron(rng.integers(0, 256, (68, 120, 3), dtype=np.uint8), np.ones((16, 16, 1), np.uint8)))[:1080, :1920]
N = 256
for i in range(N):
    Image.fromarray(np.roll(base, 7 * i, axis=1)).save(os.path.join(d, f"im{i:04d}.jpg"), quality=90)
t0 = time.time(); frames = [run_folder.load_rgb(os.path.join(d, f"im{i:04d}.jpg")) for i in range(64)]; ser = (time.time() - t0) / 64
print(f"serial PIL decode: {ser*1e3:.1f} ms per 1080p JPEG")
path = os.path.join(d, "m.pt"); save_checkpoint(path, nc=80)
from yolov8_lmi.model import Yolov8
m = Yolov8(path)
for bs in (16, 64):
    out = os.path.join(d, f"out{bs}")
    s = run_folder.run(m, d, out, (640, 640), 0.25, True, bs, "stretch")
    print(json.dumps(s))
