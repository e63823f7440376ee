"""Generate tests/golden/rescale_masks.npz by running the REFERENCE's own `rescale_masks`
(lmi_utils/postprocess_utils/mask_utils.py:12-140, the detectron2-style paste used by detectron2_lmi/model.py:257,489).

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden_rescale_masks.py

The module imports numba (`njit`, `numba.np.extensions.cross2d`) and cv2 at the top; neither is installed here.
Placeholders are registered for the import: `njit(...)` hands the decorated function back unchanged and `cross2d`
/ `cv2` are empty -- they only decorate or serve the convex-hull / contour helpers further down the file
(:142-306), which are never called here.  `rescale_masks` and `rescale_mask_func` are torch-only
(`@torch.jit.script`) and run as the reference wrote them, on torch-CPU.  Only inputs (seeds / shapes) and outputs
are stored; the boolean masks are bit-packed along the last axis.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/lmi_utils/postprocess_utils/mask_utils.py"
F32 = np.float32

# (seed, n, mask side, image h, image w, threshold)
CASES = [(41, 7, 28, 480, 640, 0.5), (42, 3, 28, 101, 67, 0.5), (43, 4, 14, 64, 64, 0.3), (44, 5, 28, 1080, 1920, 0.5),
         (45, 6, 56, 375, 500, 0.7)]


def inputs(seed, n, s, h, w):
    """Soft blobs in [0,1] and xyxy boxes, some leaving the image."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:s, 0:s]
    m = np.empty((n, s, s), F32)
    for i in range(n):
        cy, cx, r = rng.uniform(s * 0.3, s * 0.7), rng.uniform(s * 0.3, s * 0.7), rng.uniform(s * 0.2, s * 0.5)
        m[i] = 1.0 / (1.0 + np.exp((np.sqrt((y - cy) ** 2 + (x - cx) ** 2) - r) * 1.5)) + rng.normal(0, 0.02, (s, s))
    m = np.clip(m, 0, 1).astype(F32)
    x0, y0 = rng.uniform(-20, w * 0.6, n), rng.uniform(-20, h * 0.6, n)
    b = np.stack([x0, y0, x0 + rng.uniform(8, w * 0.5, n), y0 + rng.uniform(8, h * 0.5, n)], 1).astype(F32)
    return m, b


def load_reference():
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (lambda f: f)
    nb.np = types.ModuleType("numba.np")
    nb.np.extensions = types.ModuleType("numba.np.extensions")
    nb.np.extensions.cross2d = None
    sys.modules.update({"numba": nb, "numba.np": nb.np, "numba.np.extensions": nb.np.extensions})
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_mask_utils", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    torch.set_num_threads(1)
    out = {"torch_version": np.array(torch.__version__), "cases": np.array(CASES, dtype=np.float64)}
    for i, (seed, n, s, h, w, thr) in enumerate(CASES):
        m, b = inputs(seed, n, s, h, w)
        r = ref.rescale_masks(torch.from_numpy(m), torch.from_numpy(b), (h, w), thr)
        assert r.dtype == torch.bool and tuple(r.shape) == (n, h, w) and r.any()
        out[f"out{i}"] = np.packbits(r.numpy(), axis=-1)
    empty = ref.rescale_masks(torch.zeros((0, 28, 28)), torch.zeros((0, 4)), (50, 60), 0.5)
    out["empty_shape"] = np.array(empty.shape, dtype=np.int64)
    path = os.path.join(HERE, "rescale_masks.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
