"""The LMI-helper flavour of the caller-side resize through the C ABI:
lmi_letterbox_u8(mode=LMI_RESIZE_TORCH_BILINEAR) against the reference's own outputs
(tests/golden/letterbox_torch.npz: resize_image + fit_im_to_size, lmi_utils/gadget_utils/pipeline_utils.py:23-142)
and against the oracle on seeded inputs.  Bar: bit-exact (integer image, one IEEE division)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import geometry as OG

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ops():
    from object_detectors_amd import hip_ops
    return hip_ops


def test_golden_reference_outputs(ops):
    z = np.load(os.path.join(HERE, "golden", "letterbox_torch.npz"))
    for i in range(int(z["n"])):
        seed, h, w, H, W, ph, pw, val = (int(v) for v in z[f"case{i}_meta"])
        im = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        resize = (None if H < 0 else H, None if W < 0 else W)
        out, oplist = ops.letterbox([torch.from_numpy(im).cuda()], (ph, pw), pad_value=val, mode="torch", resize=resize)
        ref = z[f"case{i}_out"]                                   # (ph, pw, 3) uint8 from the reference
        assert out.shape == (1, 3, ph, pw) and out.dtype == torch.float32
        assert np.array_equal(out[0].cpu().numpy(), ref.transpose(2, 0, 1).astype(np.float32) / np.float32(255)), f"case {i}"
        assert oplist[0][1]["pad"] == [int(v) for v in z[f"case{i}_pads"]]
        one, _ = ops.letterbox([torch.from_numpy(im).cuda()], (ph, pw), pad_value=val, mode="torch", resize=resize, torch_threads=1)
        ref1 = z[f"case{i}_out_1t"]                               # the reference with torch.set_num_threads(1)
        assert np.array_equal(one[0].cpu().numpy(), ref1.transpose(2, 0, 1).astype(np.float32) / np.float32(255)), f"case {i} (1 thread)"
        assert oplist[0][0]["resize"] == [int(z[f"case{i}_resized_shape"][1]), int(z[f"case{i}_resized_shape"][0]), w, h]


def _seeds(first):
    return [first] + [10_000 + first + k for k in range(int(os.environ.get("LMI_FUZZ_EXTRA", "0")))]


@pytest.mark.parametrize("seed", _seeds(21))
@pytest.mark.parametrize("fp16", [False, True])
def test_random_geometries_against_oracle(ops, fp16, seed):
    rng = np.random.default_rng(seed)
    for it in range(24):
        h, w = int(rng.integers(2, 400)), int(rng.integers(2, 500))
        dh, dw = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        kind = it % 4
        resize = [None, (int(rng.integers(4, 300)), None), (None, int(rng.integers(4, 300))),
                  (int(rng.integers(4, 300)), int(rng.integers(4, 300)))][kind]
        g = ops.torch_letterbox_geometry(h, w, dh, dw, resize)
        if g[0] < 1 or g[1] < 1:
            continue
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        val, swap = int(rng.integers(0, 256)), bool(it & 1)
        rh, rw = (None, None) if resize is None else resize
        want, want_ops = OG.letterbox_torch_u8(im, dh, dw, rw, rh, val, swap_rb=swap, fp16=fp16)
        got, got_ops = ops.letterbox([torch.from_numpy(im).cuda()], (dh, dw), pad_value=val, swap_rb=swap, fp16=fp16,
                                     mode="torch", resize=resize)
        assert got_ops[0] == want_ops
        assert np.array_equal(got[0].cpu().numpy(), want), (h, w, dh, dw, resize)


def test_batch_1080p_to_640_and_tensor_input(ops):
    rng = np.random.default_rng(22)
    frames = torch.from_numpy(rng.integers(0, 256, (3, 1080, 1920, 3), dtype=np.uint8)).cuda()
    out, oplist = ops.letterbox(frames, (640, 640), pad_value=0, swap_rb=True, mode="torch")
    assert out.shape == (3, 3, 640, 640)
    assert oplist[0] == [{"resize": [640, 360, 1920, 1080]}, {"pad": [0, 0, 140, 140]}]
    for b in (0, 2):
        want, _ = OG.letterbox_torch_u8(frames[b].cpu().numpy(), 640, 640, pad_value=0, swap_rb=True)
        assert np.array_equal(out[b].cpu().numpy(), want)
    # a non-integer scale at full size (the x3 case above only picks pixels): 1080p -> 1280 wide
    out2, _ = ops.letterbox(frames[:1], (1280, 1280), pad_value=114, mode="torch")
    want2, _ = OG.letterbox_torch_u8(frames[0].cpu().numpy(), 1280, 1280, pad_value=114)
    assert np.array_equal(out2[0].cpu().numpy(), want2)
    # ragged batch, mixed sizes
    small = [torch.from_numpy(rng.integers(0, 256, s, dtype=np.uint8)).cuda() for s in ((480, 640, 3), (375, 500, 3), (427, 640, 3))]
    out3, ops3 = ops.letterbox(small, (640, 640), pad_value=0, mode="torch")
    for b, f in enumerate(small):
        want3, wops = OG.letterbox_torch_u8(f.cpu().numpy(), 640, 640, pad_value=0)
        assert ops3[b] == wops and np.array_equal(out3[b].cpu().numpy(), want3)


def test_single_thread_variant_against_oracle(ops):
    """torch_threads=1: aten's 4-weight kernel for every 3-channel frame, whatever the output size."""
    rng = np.random.default_rng(24)
    differs = 0
    for h, w, dh, dw in ((480, 640, 360, 480), (270, 480, 256, 256), (1080, 1920, 1280, 1280), (61, 83, 64, 80)):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want, _ = OG.letterbox_torch_u8(im, dh, dw, pad_value=9, one_thread=True)
        many, _ = OG.letterbox_torch_u8(im, dh, dw, pad_value=9)
        got, _ = ops.letterbox([torch.from_numpy(im).cuda()], (dh, dw), pad_value=9, mode="torch", torch_threads=1)
        assert np.array_equal(got[0].cpu().numpy(), want), (h, w, dh, dw)
        differs += int((want != many).sum())
    assert differs > 0   # the two thread regimes really are different results on these shapes


def test_mode_validation(ops):
    f = torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        ops.letterbox([f], (16, 16), mode="lanczos")
    # contents on both sides of aten's kernel switch (new_h + new_w = 128) in one launch
    rng = np.random.default_rng(23)
    a = rng.integers(0, 256, (8, 80, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (80, 80, 3), dtype=np.uint8)
    out, oplist = ops.letterbox([torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()], (100, 100), pad_value=3, mode="torch", resize=(None, 100))
    assert oplist[0][0]["resize"] == [100, 10, 80, 8] and oplist[1][0]["resize"] == [100, 100, 80, 80]
    for i, im in enumerate((a, b)):
        want, _ = OG.letterbox_torch_u8(im, 100, 100, 100, None, 3)
        assert np.array_equal(out[i].cpu().numpy(), want)
    with pytest.raises(Exception):   # the cv2 flavour does not crop
        ops.letterbox([f], (4, 4), mode="cv2", resize=(8, 8))
