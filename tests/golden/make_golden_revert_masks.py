"""Generate tests/golden/revert_masks.npz by running the REFERENCE's revert_masks_to_origin.

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden_revert_masks.py

As in make_golden_l1.py an empty module object is registered as `cv2` for the import of
lmi_utils/gadget_utils/pipeline_utils.py (the functions used here are torch-only).  Only inputs and outputs
are stored.  Inputs are seeded blobs of 0/1 like the detector's masks, as fp32 (what the reference returns)
and as uint8.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/lmi_utils"
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [  # (m, h, w, dtype, operator list)
    (3, 40, 40, "float32", [{"resize": [40, 24, 120, 72]}, {"pad": [0, 0, 8, 8]}]),            # letterbox undo, x3
    (2, 40, 40, "uint8", [{"resize": [40, 24, 120, 72]}, {"pad": [0, 0, 8, 8]}]),
    (2, 48, 64, "float32", [{"resize": [64, 48, 100, 75]}]),                                      # stretch undo, x1.5625
    (2, 36, 50, "float32", [{"pad": [3, 4, -2, -5]}, {"flip": [True, False, 50, 36]}]),           # crop + grow, flip
    (2, 32, 32, "float32", [{"flip": [False, True, 64, 48]}, {"resize": [32, 24, 64, 48]}, {"pad": [0, 0, 4, 4]}]),
    (2, 30, 44, "uint8", [{"resize": [44, 30, 21, 17]}, {"flip": [True, True, 44, 30]}]),        # shrink, both flips
    (1, 33, 47, "float32", [{"resize": [47, 33, 130, 91]}, {"pad": [5, 6, 1, 2]}]),               # odd sizes, asymmetric pad
]


def blobs(rng, m, h, w):
    y, x = np.mgrid[0:h, 0:w]
    out = np.zeros((m, h, w), np.float32)
    for i in range(m):
        for _ in range(3):
            cy, cx = rng.uniform(0, h), rng.uniform(0, w)
            ry, rx = rng.uniform(3, h / 2), rng.uniform(3, w / 2)
            out[i] = np.maximum(out[i], (((y - cy) / ry) ** 2 + ((x - cx) / rx) ** 2 < 1).astype(np.float32))
    return out


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import gadget_utils.pipeline_utils as pu  # the reference module

    torch.set_num_threads(1)
    rng = np.random.default_rng(11)
    arrays, meta = {}, []
    for i, (m, h, w, dt, ops) in enumerate(CASES):
        src = blobs(rng, m, h, w).astype(dt)
        out = pu.revert_masks_to_origin(torch.from_numpy(src), ops)
        out = out.numpy()
        arrays[f"in{i}"] = src
        arrays[f"out{i}"] = out
        meta.append({"ops": ops, "dtype": dt, "in_shape": list(src.shape), "out_shape": list(out.shape)})
    arrays["meta"] = np.frombuffer(json.dumps({"torch": torch.__version__, "cases": meta}).encode(), np.uint8)
    path = os.path.join(HERE, "revert_masks.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
