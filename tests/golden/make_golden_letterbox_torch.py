"""Generate tests/golden/letterbox_torch.npz by running the REFERENCE's own helpers for the LMI-flavoured
caller-side resize: `resize_image` then `fit_im_to_size`
(lmi_utils/gadget_utils/pipeline_utils.py:23-142, as chained by ultralytics_lmi/yolo/run_model.py:87-100).

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden_letterbox_torch.py

The module does `import cv2` at the top; the two functions used here are torch-only, so an empty module object is
registered under that name for the import (no cv2 function is called, no cv2 behaviour is faked).  Only inputs
(seeds / shapes) and outputs are stored.  aten's CPU bilinear picks a different kernel, with different rounding,
when torch runs on ONE thread (29-80 of 518 k pixels differ on 480x640 -> 360x480); the fixture records the result
for >= 2 threads (2, 3 and 8 threads were verified to agree), which is what a default torch install computes.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/lmi_utils"
HERE = os.path.dirname(os.path.abspath(__file__))

# (seed, h, w, resize kwargs {H, W}, pad target (H, W), pad value)
CASES = [
    (1, 5, 7, {"W": 4, "H": 3}, (6, 6), 0),
    (2, 48, 64, {"W": 40, "H": 30}, (40, 40), 0),
    (3, 37, 53, {"H": 64}, (64, 96), 0),           # aspect rule: W = int(w*H/h)
    (4, 90, 50, {"W": 32}, (64, 64), 114),          # H = int(h*W/w); ultralytics' pad value
    (5, 108, 192, {"W": 64, "H": 36}, (64, 64), 0),  # exact x3 reduction like 1080p -> 640x360
    (6, 75, 100, {"W": 128, "H": 96}, (128, 128), 0),  # enlargement
    (7, 61, 83, {"W": 96, "H": 70}, (64, 80), 0),   # pad target smaller than the content: centred crop
    (8, 120, 160, {"W": 107, "H": 80}, (96, 112), 7),
]


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import gadget_utils.pipeline_utils as pu  # the reference module

    torch.set_num_threads(8)
    out = {"torch_version": np.array(torch.__version__), "threads": np.array(8), "n": np.array(len(CASES))}
    for i, (seed, h, w, kw, (ph, pw), value) in enumerate(CASES):
        im0 = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        im1 = pu.resize_image(im=im0, **kw)
        im2, pl, pr, pt, pb = pu.fit_im_to_size(im=im1, H=ph, W=pw, value=value)
        out[f"case{i}_meta"] = np.array([seed, h, w, kw.get("H") or -1, kw.get("W") or -1, ph, pw, value], dtype=np.int64)
        out[f"case{i}_resized_shape"] = np.array(im1.shape, dtype=np.int64)
        out[f"case{i}_pads"] = np.array([pl, pr, pt, pb], dtype=np.int64)
        out[f"case{i}_out"] = np.asarray(im2)
        # the same chain with torch on ONE thread (aten then runs its other CPU kernel on 3-channel frames)
        torch.set_num_threads(1)
        im1t = pu.resize_image(im=im0, **kw)
        im2t, *_ = pu.fit_im_to_size(im=im1t, H=ph, W=pw, value=value)
        out[f"case{i}_out_1t"] = np.asarray(im2t)
        torch.set_num_threads(8)
    np.savez_compressed(os.path.join(HERE, "letterbox_torch.npz"), **out)
    print("wrote letterbox_torch.npz", os.path.getsize(os.path.join(HERE, "letterbox_torch.npz")), "bytes")


if __name__ == "__main__":
    main()
