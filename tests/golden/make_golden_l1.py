"""Generate tests/golden/l1_geometry.json by running the REFERENCE's own L1 helpers.

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden_l1.py

`lmi_utils/gadget_utils/pipeline_utils.py` does `import cv2` at module level but the
functions exercised here (revert_to_origin, apply_operations, resize_image,
fit_im_to_size, revert_mask_to_origin) are torch-only, so an empty module object is
registered under the name `cv2` for the import; no cv2 function is ever called and no
cv2 behaviour is faked.  Only inputs and outputs are stored (data, not source).
"""

import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/lmi_utils"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import gadget_utils.pipeline_utils as pu  # the reference module

    torch.set_num_threads(1)
    out = {"torch": torch.__version__, "revert": [], "apply": [], "fit": [], "resize": [], "mask_revert": []}

    revert_cases = [
        ([[10.1, 20.0], [30.2, 40.4], [50.5, 60.4], [70.2, 80.1]], "list",
         [{"resize": [100, 100, 200, 300]}, {"pad": [8, 9, 10, 11]}, {"stretch": [1.5, 2]}]),
        ([[15.3, 25.8], [35.7, 45], [55.6, 65], [75, 85.3]], "f64",
         [{"resize": [200, 300, 100, 100]}, {"pad": [-8, -9, 10, 11]}, {"stretch": [1.3, 1.5]}]),
        ([[15, 25, 35, 45], [55, 65, 75, 85]], "list",
         [{"resize": [200, 300, 100, 100]}, {"pad": [8, 9, 10, 11]}, {"stretch": [1.5, 2]}]),
        ([[100, 200, 300, 400]], "f32",
         [{"resize": [640, 360, 1920, 1080]}, {"pad": [0, 0, 140, 140]}]),
        ([[0.5, 1.5, 2.5, 3.5], [4.5, 5.5, 6.5, 7.5]], "f32", []),
        ([[12.25, 300.75, 630.5, 359.5], [0.0, 141.0, 5.0, 150.0]], "f32",
         [{"resize": [640, 360, 3840, 2160]}, {"pad": [0, 0, 140, 140]}]),
        ([[10.0, 20.0], [630.0, 400.0]], "f64",
         [{"resize": [640, 480, 1280, 960]}, {"pad": [0, 0, 80, 80]}, {"flip": [True, True, 640, 640]}]),
    ]
    rng = np.random.default_rng(7)
    big = (rng.random((64, 4)) * 640).astype(np.float32)
    revert_cases.append((big.tolist(), "f32", [{"resize": [640, 427, 1333, 889]}, {"pad": [0, 0, 106, 107]}]))

    def conv(p, kind):
        if kind == "list":
            return p
        return np.array(p, dtype={"f64": np.float64, "f32": np.float32}[kind])

    for pts, kind, ops in revert_cases:
        for as_tensor in (False, True):
            x = conv(pts, kind)
            if as_tensor:
                x = torch.as_tensor(x)
            r = pu.revert_to_origin(x, ops)
            r = r.tolist() if hasattr(r, "tolist") else r
            out["revert"].append({"pts": pts, "kind": kind, "tensor": as_tensor, "ops": ops, "out": r})

    apply_cases = [
        ([[10, 20], [30, 40], [50, 60], [70, 80]], "list",
         [{"resize": [100, 100, 200, 300]}, {"pad": [8, 9, 10, 11]}, {"stretch": [1.5, 2]}, {"flip": [True, False, 100, 100]}]),
        ([[15.0, 25.2], [35.1, 45.0], [55.0, 65], [75.5, 85]], "f64",
         [{"resize": [200, 300, 100, 100]}, {"pad": [6, 7, 9, 10]}, {"stretch": [1.0, 2]}]),
        ([[15, 25, 35, 45], [55, 65, 75, 85]], "list",
         [{"resize": [200, 300, 100, 100]}, {"pad": [11, 9, -2, -3]}, {"flip": [False, True, 200, 300]}]),
    ]
    for pts, kind, ops in apply_cases:
        r = pu.apply_operations(conv(pts, kind), ops)
        r = r.tolist() if hasattr(r, "tolist") else r
        out["apply"].append({"pts": pts, "kind": kind, "ops": ops, "out": r})

    def pattern(h, w, c=3):
        y, x, ch = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
        return ((37 * y + 11 * x + 101 * ch) % 256).astype(np.uint8)

    for (h, w), (W, H) in [((12, 10), (16, 8)), ((100, 100), (121, None)), ((100, 100), (None, 131)),
                           ((100, 100), (151, 131)), ((100, 100), (71, 75)), ((100, 100), (None, 89)),
                           ((100, 100), (89, None)), ((3, 4), (6, 6)), ((360, 640), (640, 640)),
                           ((427, 640), (640, 640))]:
        im = pattern(h, w)
        r, pl, pr, pt, pb = pu.fit_im_to_size(im, W=W, H=H)
        rec = {"h": h, "w": w, "W": W, "H": H, "pads": [pl, pr, pt, pb], "shape": list(r.shape),
               "sum": int(r.astype(np.int64).sum())}
        if r.size <= 200:
            rec["out"] = r.tolist()
        out["fit"].append(rec)

    for (h, w), kw in [((5, 7), {"W": 4, "H": 3}), ((5, 7), {"W": 10}), ((12, 10), {"W": 20, "H": 24}),
                       ((150, 100), {"H": 300}), ((100, 150), {"W": 30}), ((9, 6), {"W": 3, "H": 5})]:
        im = pattern(h, w)
        r = pu.resize_image(im, **kw)
        rec = {"h": h, "w": w, "kw": kw, "shape": list(r.shape), "dtype": str(r.dtype)}
        if r.size <= 900:
            rec["out"] = r.tolist()
        out["resize"].append(rec)

    for shape, ops in [((100, 100, 3), [{"pad": [8, 9, 10, 11]}, {"flip": [True, False, 100, 100]}]),
                       ((90, 100), [{"flip": [True, False, 100, 90]}]),
                       ((64, 64), [{"resize": [64, 36, 192, 108]}, {"pad": [0, 0, 14, 14]}])]:
        m = pattern(*shape) if len(shape) == 3 else pattern(shape[0], shape[1], 1)[:, :, 0]
        r = pu.revert_mask_to_origin(m, ops)
        out["mask_revert"].append({"shape": list(shape), "ops": ops, "out_shape": list(r.shape),
                                   "sum": int(np.asarray(r).astype(np.int64).sum())})

    with open(os.path.join(HERE, "l1_geometry.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "l1_geometry.json"))


if __name__ == "__main__":
    main()
