"""Oracle `preprocess` vs torch-CPU running the reference's exact op sequence
(object_detectors/yolov8_lmi/model.py:119-133), on every byte value."""
import numpy as np
import pytest
import torch

from oracle import preprocess as P


def _reference_ops(im: np.ndarray, fp16: bool):
    t = torch.tensor(im)
    if t.ndim == 2:
        t = t.unsqueeze(-1)
    if t.shape[-1] == 1:
        t = t.expand(-1, -1, 3)
    t = t.unsqueeze(0).permute((0, 3, 1, 2)).contiguous()
    t = t.half() if fp16 else t.float()
    t /= 255
    return t.numpy()


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("shape", [(16, 16, 3), (16, 16), (16, 16, 1)])
def test_all_byte_values(shape, fp16):
    im = np.arange(256, dtype=np.uint8).reshape(16, 16)
    if len(shape) == 3:
        im = np.stack([im, im[::-1], im.T], -1) if shape[2] == 3 else im[:, :, None]
    got = P.preprocess(im, fp16)
    ref = _reference_ops(im, fp16)
    assert got.dtype == ref.dtype and got.shape == ref.shape == (1, 3, 16, 16)
    assert np.array_equal(got.view(np.uint16 if fp16 else np.uint32), ref.view(np.uint16 if fp16 else np.uint32))
