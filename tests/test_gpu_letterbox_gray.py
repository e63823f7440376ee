"""Grey planes through the letterbox (mode | LMI_SRC_GRAY).  The reference's callers make a grey image three equal
channels BEFORE they resize it (cv2.cvtColor(im0, COLOR_GRAY2BGR), object_detectors/yolov8_lmi/run_model.py:77-78),
and the resize treats channels independently: the tensor must equal, bit for bit, the colour letterbox of the
replicated frame -- checked against the oracle and against this library's own 3-channel path, both resamplers,
uniform and ragged batches, fp32 and fp16, strided rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cv2_resize as OC
from object_detectors_amd import synth


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


CASES = [(54, 96, 64, 64), (1080, 1920, 640, 640), (960, 1280, 640, 640), (375, 500, 640, 640), (33, 77, 96, 128),
         (641, 333, 640, 640), (100, 1000, 640, 640), (2160, 3840, 640, 640), (7, 5, 32, 32), (300, 300, 316, 316),
         (192, 192, 64, 64), (128, 128, 64, 64), (21, 21, 63, 63), (720, 1280, 1280, 1280)]


@pytest.mark.parametrize("h,w,sh,sw", CASES)
def test_gray_uniform_equals_replicated_colour(ops, h, w, sh, sw):
    rng = np.random.default_rng(h * 7919 + w)
    gray = synth.frames_u8(rng, 2, h, w)[..., 0].copy()                      # (2, h, w)
    got, gops = ops.letterbox(dev(gray[..., None]), (sh, sw), pad_value=114, swap_rb=True)
    col, cops = ops.letterbox(dev(np.repeat(gray[..., None], 3, -1)), (sh, sw), pad_value=114, swap_rb=True)
    assert gops == cops and torch.equal(got, col)
    for b in range(2):
        ref, rops = OC.letterbox_u8(np.repeat(gray[b][..., None], 3, -1), sh, sw, pad_value=114, swap_rb=True)
        assert gops[b] == rops and np.array_equal(got[b].cpu().numpy().view(np.uint32), ref.view(np.uint32)), b


@pytest.mark.parametrize("mode", ["cv2", "torch"])
@pytest.mark.parametrize("fp16", [False, True])
def test_gray_ragged_batch_both_resamplers(ops, mode, fp16):
    rng = np.random.default_rng(17)
    sizes = [(375, 500), (1080, 1920), (427, 640), (64, 48), (640, 640), (33, 1000), (480, 640)]
    planes = [synth.frames_u8(rng, 1, h, w)[0, :, :, 1].copy() for h, w in sizes]
    kw = dict(pad_value=0 if mode == "torch" else 114, fp16=fp16, mode=mode)
    got, gops = ops.letterbox([dev(p) for p in planes], (640, 640), **kw)                       # (H,W) planes
    col, cops = ops.letterbox([dev(np.repeat(p[..., None], 3, -1)) for p in planes], (640, 640), **kw)
    assert gops == cops and torch.equal(got, col)
    one, _ = ops.letterbox(dev(planes[2]), (640, 640), **kw)                                    # a single 2-D plane
    assert torch.equal(one[0], got[2])
    # rows wider than the image (a view into a larger buffer)
    big = dev(np.pad(planes[0], ((0, 0), (0, 37))))
    view, _ = ops.letterbox([big[:, :500].unsqueeze(-1)], (640, 640), **kw)
    assert torch.equal(view[0], got[0])
    with pytest.raises(ValueError):
        ops.letterbox([dev(planes[0]), dev(np.repeat(planes[1][..., None], 3, -1))], (640, 640))


def test_gray_torch_flavour_stretch_and_crop(ops):
    """the helpers' geometry with a grey plane: resize=(H, W) given, content larger than the destination (centred crop)"""
    rng = np.random.default_rng(18)
    p = synth.frames_u8(rng, 1, 300, 420)[0, :, :, 0].copy()
    for resize, dst in (((200, 150), (128, 128)), ((None, 96), (160, 80)), ((64, None), (64, 64))):
        a, oa = ops.letterbox([dev(p)], dst, pad_value=0, mode="torch", resize=resize)
        b, ob = ops.letterbox([dev(np.repeat(p[..., None], 3, -1))], dst, pad_value=0, mode="torch", resize=resize)
        assert oa == ob and torch.equal(a, b)
        a1, _ = ops.letterbox([dev(p)], dst, pad_value=0, mode="torch", resize=resize, torch_threads=1)
        b1, _ = ops.letterbox([dev(np.repeat(p[..., None], 3, -1))], dst, pad_value=0, mode="torch", resize=resize, torch_threads=1)
        assert torch.equal(a1, b1)


def _seeds(first):
    import os
    return [first] + [10_000 + first + k for k in range(int(os.environ.get("LMI_FUZZ_EXTRA", "0")))]


@pytest.mark.parametrize("seed", _seeds(31))
def test_gray_random_geometries(ops, seed):
    """random sizes, strides, pads and destinations, both resamplers: grey plane == replicated colour frame"""
    rng = np.random.default_rng(seed)
    for it in range(16):
        n = int(rng.integers(1, 5))
        sizes = [(int(rng.integers(2, 500)), int(rng.integers(2, 700))) for _ in range(n)]
        dst = (int(rng.integers(8, 400)), int(rng.integers(8, 400)))
        planes = [rng.integers(0, 256, s, dtype=np.uint8) for s in sizes]
        kw = dict(pad_value=int(rng.integers(0, 256)), swap_rb=bool(it & 1), fp16=bool(it & 2), stretch=bool(it & 4))
        for mode in ("cv2", "torch"):
            colour = [dev(np.repeat(p[..., None], 3, -1)) for p in planes]
            try:
                a, oa = ops.letterbox([dev(p) for p in planes], dst, mode=mode, **kw)
            except ValueError:   # a frame so thin that its content rounds to zero rows or columns: refused either way
                with pytest.raises(ValueError):
                    ops.letterbox(colour, dst, mode=mode, **kw)
                continue
            b, ob = ops.letterbox(colour, dst, mode=mode, **kw)
            assert oa == ob and torch.equal(a, b), (seed, it, mode, sizes, dst)
