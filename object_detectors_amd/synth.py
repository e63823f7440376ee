"""Seeded synthetic inputs shaped like the BASELINE configs: frames, YOLOv8
head outputs with planted object clusters, prototype tensors.  numpy on the host; callers move
the arrays to the device.  Used by bench.py, __graft_entry__.smoke() and the tests."""
from __future__ import annotations

import numpy as np

F32 = np.float32


def anchors_for(size: int) -> int:
    return (size // 8) ** 2 + (size // 16) ** 2 + (size // 32) ** 2


def frames_u8(rng: np.random.Generator, batch: int, h: int, w: int) -> np.ndarray:
    """uint8 (B,H,W,3): uniform noise over a smooth gradient, so resampling sees both regimes."""
    out = np.empty((batch, h, w, 3), np.uint8)
    gy = np.linspace(0, 1, h, dtype=F32)[:, None, None]
    gx = np.linspace(0, 1, w, dtype=F32)[None, :, None]
    for b in range(batch):
        noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        grad = (255 * (0.5 * gy + 0.5 * gx) * np.array([1.0, 0.7, 0.4], F32)).astype(np.uint8)
        out[b] = np.where(rng.random((h, w, 1)) < 0.5, noise, grad)
    return out


def head(rng: np.random.Generator, batch: int, nc: int, nm: int, size: int, objects: int = 40,
         cluster=(5, 16), classes=None, anchors: int = 0) -> np.ndarray:
    """(B, 4+nc+nm, A) fp32: low background scores, `objects` planted clusters of jittered anchors
    per image with peak scores U(0.3,0.95).  Scores are made distinct per image so that the kept
    order cannot depend on tie-breaking."""
    A = anchors or anchors_for(size)
    p = np.empty((batch, 4 + nc + nm, A), F32)
    p[:, 0:2] = rng.random((batch, 2, A), dtype=F32) * size
    p[:, 2:4] = np.exp(rng.uniform(np.log(8), np.log(size / 2), (batch, 2, A))).astype(F32)
    p[:, 4:4 + nc] = (1.0 / (1.0 + np.exp(-rng.normal(-6.0, 1.0, (batch, nc, A))))).astype(F32)
    if nm:
        p[:, 4 + nc:] = rng.normal(0, 0.5, (batch, nm, A)).astype(F32)
    for b in range(batch):
        for _ in range(objects):
            k = int(rng.integers(cluster[0], cluster[1]))
            a = rng.choice(A, k, replace=False)
            cx, cy = rng.random(2) * size
            w, h = np.exp(rng.uniform(np.log(16), np.log(size / 2), 2))
            cls = int(rng.choice(classes)) if classes is not None else int(rng.integers(0, nc))
            p[b, 0, a] = cx + rng.normal(0, 2, k)
            p[b, 1, a] = cy + rng.normal(0, 2, k)
            p[b, 2, a] = w * np.exp(rng.normal(0, 0.05, k))
            p[b, 3, a] = h * np.exp(rng.normal(0, 0.05, k))
            p[b, 4 + cls, a] = rng.uniform(0.3, 0.95) * rng.uniform(0.7, 1.0, k)
            if nm:  # clustered anchors share a mask
                p[b, 4 + nc:, a] = (rng.normal(0, 0.5, (nm, 1)) + rng.normal(0, 0.02, (nm, k))).astype(F32).T
        # distinct best scores per image: nudge duplicates by one ulp steps
        best = p[b, 4:4 + nc].max(0)
        arg = p[b, 4:4 + nc].argmax(0)
        order = np.argsort(best, kind="stable")
        sb = best[order]
        for i in range(1, A):
            if sb[i] <= sb[i - 1]:
                sb[i] = np.nextafter(sb[i - 1], F32(2))
        best[order] = sb
        p[b, 4 + arg, np.arange(A)] = best
    return p


def proto(rng: np.random.Generator, batch: int, nm: int, mh: int, mw: int) -> np.ndarray:
    """(B, nm, mh, mw) fp32 ~N(0,1), box-smoothed 5x5 so masks have coherent regions."""
    x = rng.normal(0, 1, (batch, nm, mh + 4, mw + 4)).astype(F32)
    c = np.cumsum(np.cumsum(np.pad(x, ((0, 0), (0, 0), (1, 0), (1, 0))), axis=2), axis=3)
    s = c[:, :, 5:, 5:] - c[:, :, :-5, 5:] - c[:, :, 5:, :-5] + c[:, :, :-5, :-5]
    return np.ascontiguousarray((s / 5.0).astype(F32))
