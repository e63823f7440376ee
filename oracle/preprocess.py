"""Oracle: `Yolov8.preprocess` (object_detectors/yolov8_lmi/model.py:112-134).
TEST INFRASTRUCTURE ONLY.

One image, (H,W) / (H,W,1) / (H,W,3) -> (1,3,H,W) fp32|fp16 in [0,1]:
2-D -> unsqueeze(-1) (:123-124); 1 channel -> expand to 3 (:125-126);
unsqueeze(0) + permute(0,3,1,2) + contiguous (:128-130); cast (:132); `/= 255`
(:133, IEEE division, not a reciprocal multiply).  No resize, pad or channel
swap.  Pinned by exhaustive check of all 256 byte values against torch-CPU in
tests/test_oracle_preprocess.py.
"""

from __future__ import annotations

import numpy as np

from .cv2_resize import normalize_u8


def preprocess(im: np.ndarray, fp16: bool = False) -> np.ndarray:
    if im.ndim == 2:
        im = im[:, :, None]
    if im.shape[-1] == 1:
        im = np.broadcast_to(im, im.shape[:2] + (3,))
    chw = np.ascontiguousarray(im.transpose(2, 0, 1))[None]
    if chw.dtype == np.uint8:
        return normalize_u8(chw, fp16)
    # non-uint8 inputs are cast first, then divided in the target precision
    if fp16:
        return (chw.astype(np.float16).astype(np.float32) / np.float32(255.0)).astype(np.float16)
    return chw.astype(np.float32) / np.float32(255.0)
