import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import obb as OO
from object_detectors_amd import hip_ops as ops
F32=np.float32
seed, agnostic = 11, False
rng = np.random.default_rng(seed)
nc, A, size = 6, 768, 640
p = np.zeros((1, 4 + nc + 1, A), F32)
p[0, 0:2] = rng.uniform(0, size, (2, A))
kind = rng.integers(0, 6, A)
w = rng.uniform(8, 120, A); h = rng.uniform(8, 120, A)
w[kind == 1], h[kind == 1] = 0.0, rng.uniform(0, 50, (kind == 1).sum())
w[kind == 2], h[kind == 2] = rng.uniform(100, 400, (kind == 2).sum()), rng.uniform(0.1, 0.4, (kind == 2).sum())
w[kind == 3], h[kind == 3] = rng.uniform(1, 3, (kind == 3).sum()), rng.uniform(1, 3, (kind == 3).sum())
w[kind == 4], h[kind == 4] = rng.uniform(400, 900, (kind == 4).sum()), rng.uniform(300, 800, (kind == 4).sum())
p[0, 2], p[0, 3] = w, h
p[0, 4:4 + nc] = 0.01
score = rng.permutation(A).astype(F32) / A * 0.6 + 0.3
cls = rng.integers(0, nc, A)
p[0, 4 + cls, np.arange(A)] = score
p[0, -1] = rng.uniform(-math.pi / 4, 3 * math.pi / 4, A)
det, count, anchor = ops.nms_v8(torch.from_numpy(p).cuda(), 0.25, 0.45, agnostic=agnostic, max_det=1000, nc=nc, rotated=True)
(rows, ra, margin), = OO.non_max_suppression_rotated(p, 0.25, 0.45, agnostic=agnostic, max_det=1000, nc=nc)
g = set(anchor[0, :int(count[0])].cpu().tolist()); o = set(ra.tolist())
print("gpu only", sorted(g - o), "oracle only", sorted(o - g))
def probiou_t(b1, b2):   # torch fp32, ultralytics' formula
    def cov(b):
        a = b[:, 2].pow(2) / 12; bb = b[:, 3].pow(2) / 12; c = b[:, 4]
        cos, sin = c.cos(), c.sin()
        return a * cos.pow(2) + bb * sin.pow(2), a * sin.pow(2) + bb * cos.pow(2), (a - bb) * cos * sin
    x1, y1 = b1[:, 0:1], b1[:, 1:2]; x2, y2 = b2[None, :, 0], b2[None, :, 1]
    a1, b1_, c1 = (v[:, None] for v in cov(b1)); a2, b2_, c2 = (v[None] for v in cov(b2))
    eps = 1e-7
    t1 = (((a1 + a2) * (y1 - y2).pow(2) + (b1_ + b2_) * (x1 - x2).pow(2)) / ((a1 + a2) * (b1_ + b2_) - (c1 + c2).pow(2) + eps)) * 0.25
    t2 = (((c1 + c2) * (x2 - x1) * (y1 - y2)) / ((a1 + a2) * (b1_ + b2_) - (c1 + c2).pow(2) + eps)) * 0.5
    t3 = (((a1 + a2) * (b1_ + b2_) - (c1 + c2).pow(2)) / (4 * ((a1 * b1_ - c1.pow(2)).clamp_(0) * (a2 * b2_ - c2.pow(2)).clamp_(0)).sqrt() + eps) + eps).log() * 0.5
    bd = (t1 + t2 + t3).clamp(eps, 100.0)
    return 1 - (1.0 - (-bd).exp() + eps).sqrt()
boxes = np.stack([p[0,0]+cls*7680.0, p[0,1]+cls*7680.0, p[0,2], p[0,3], p[0,-1]],1).astype(F32)
for a in sorted((o - g) | (g - o)):
    hi = np.nonzero(score > score[a])[0]
    for devname in ("cpu", "cuda"):
        tb = torch.from_numpy(boxes).to(devname)
        iou = probiou_t(tb[hi], tb[a:a+1])[:, 0].cpu().numpy()
        npi = OO.batch_probiou(boxes[hi], boxes[a:a+1])[:, 0]
        bad = np.nonzero(~(iou < 0.45))[0]
        print("anchor", a, "kind", int(kind[a]), "w,h", float(w[a]), float(h[a]), "ang", float(p[0,-1,a]), devname, "hits:", [(int(hi[k]), int(kind[hi[k]]), int(cls[hi[k]]), float(iou[k]), float(npi[k]), float(w[hi[k]]), float(h[hi[k]]), float(p[0,-1,hi[k]]), float(np.hypot(p[0,0,a]-p[0,0,hi[k]], p[0,1,a]-p[0,1,hi[k]]))) for k in bad][:6])
