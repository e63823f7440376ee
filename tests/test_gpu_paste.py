"""GPU parity for the mask paste (lmi_utils/postprocess_utils/mask_utils.py:66-140):
lmi_paste_masks against torch-CPU executing the reference's op sequence with its own F.grid_sample
(oracle/paste.py; PARITY UNPINNED against the reference module itself, which needs numba + cv2 to import)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "object_detectors_amd", "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)

from oracle import masks as OM, paste as OP  # noqa: E402

F32 = np.float32


def soft_masks(rng, n, s):
    y, x = np.mgrid[0:s, 0:s]
    out = np.empty((n, s, s), F32)
    for i in range(n):
        cy, cx, r = rng.uniform(s * 0.3, s * 0.7), rng.uniform(s * 0.3, s * 0.7), rng.uniform(s * 0.2, s * 0.5)
        d = np.sqrt((y - cy) ** 2 + (x - cx) ** 2)
        out[i] = 1.0 / (1.0 + np.exp((d - r) * 1.5)) + rng.normal(0, 0.02, (s, s))
    return np.clip(out, 0, 1).astype(F32)


@pytest.mark.parametrize("n,s,h,w", [(7, 28, 480, 640), (3, 28, 101, 67), (40, 28, 1080, 1920), (2, 14, 64, 64),
                                       (9, 28, 270, 479), (5, 28, 53, 17), (4, 28, 48, 20), (6, 28, 33, 16), (12, 28, 1080, 1916)])
def test_paste_masks_vs_torch_grid_sample(n, s, h, w):
    from postprocess_utils.mask_utils import rescale_masks
    rng = np.random.default_rng(n * 7 + s)
    m = soft_masks(rng, n, s)
    x0, y0 = rng.uniform(-20, w * 0.6, n), rng.uniform(-20, h * 0.6, n)
    b = np.stack([x0, y0, x0 + rng.uniform(8, w * 0.5, n), y0 + rng.uniform(8, h * 0.5, n)], 1).astype(F32)   # some boxes leave the image
    got = rescale_masks(torch.from_numpy(m).cuda(), torch.from_numpy(b).cuda(), (h, w), 0.5)
    assert got.dtype == torch.bool and got.shape == (n, h, w)
    got = got.cpu().numpy()
    k = min(n, 6)                                   # the CPU oracle samples whole images: a few masks are enough
    ref = OP.rescale_masks(m[:k], b[:k], (h, w), 0.5)
    assert ref.sum() > 0 and OM.mask_iou(got[:k], ref) >= 0.999
    assert (got[:k] != ref).mean() < 1e-5           # only pixels whose sampled value sits on the threshold may differ
    # nothing outside the box grown by one mask texel (zero padding of grid_sample)
    for i in range(n):
        tx, ty = (b[i, 2] - b[i, 0]) / s, (b[i, 3] - b[i, 1]) / s
        ys, xs = np.nonzero(got[i])
        if len(ys):
            assert xs.min() + 0.5 >= b[i, 0] - tx and xs.max() + 0.5 <= b[i, 2] + tx
            assert ys.min() + 0.5 >= b[i, 1] - ty and ys.max() + 0.5 <= b[i, 3] + ty
    # batch invariance
    one = rescale_masks(torch.from_numpy(m[1:2]).cuda(), torch.from_numpy(b[1:2]).cuda(), (h, w), 0.5)
    assert np.array_equal(one[0].cpu().numpy(), got[1])


def test_paste_masks_edges():
    from postprocess_utils.mask_utils import rescale_masks
    empty = rescale_masks(torch.zeros((0, 28, 28), device="cuda"), torch.zeros((0, 4), device="cuda"), (50, 60))
    assert empty.shape == (0, 50, 60)
    with pytest.raises(AssertionError):
        rescale_masks(torch.zeros((1, 28, 20), device="cuda"), torch.zeros((1, 4), device="cuda"), (50, 60))
    with pytest.raises(RuntimeError):
        rescale_masks(torch.zeros((1, 28, 28)), torch.zeros((1, 4)), (50, 60))
    # a mask of ones fills its box interior; a degenerate (zero-width) box yields nothing (0/0 -> NaN >= thr is False)
    m = torch.ones((2, 28, 28), device="cuda")
    b = torch.tensor([[10.0, 5.0, 30.0, 25.0], [40.0, 5.0, 40.0, 25.0]], device="cuda")
    out = rescale_masks(m, b, (40, 60), 0.5).cpu().numpy()
    assert out[0, 6:24, 11:29].all() and out[0].sum() <= 22 * 22 and out[1].sum() == 0
    ref = OP.rescale_masks(m.cpu().numpy(), b.cpu().numpy(), (40, 60), 0.5)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("h,w", [(96, 160), (90, 150)])
def test_paste_masks_irregular_boxes_thresholds_and_large_masks(h, w):
    """The two-pass form (constant fill + paste_box_kernel) and what it hands to the per-pixel kernels: boxes that are
    degenerate, reversed, not finite or astronomically large (paste_box_kernel's general path), a box far outside the
    image, threshold 0 (the constant outside the boxes is then 1), and masks above 60 x 60 (paste_masks_kernel /
    paste_masks_flat_kernel): every output equals the oracle's."""
    from postprocess_utils.mask_utils import rescale_masks
    rng = np.random.default_rng(h + w)
    nan, inf = float("nan"), float("inf")
    b = np.array([[10, 8, 70, 60], [30, 20, 30, 50], [60, 50, 20, 10], [nan, 5, 40, 30], [5, 5, inf, 40], [-3e9, -3e9, 3e9, 3e9],
                  [-500, -400, -300, -200], [0, 0, w, h], [w - 9.5, h - 7.25, w + 30, h + 40], [12.5, 3.25, 13.0, 80.0]], F32)
    for s in (28, 64):
        m = soft_masks(rng, len(b), s)
        for thr in (0.5, 0.0):
            got = rescale_masks(torch.from_numpy(m).cuda(), torch.from_numpy(b).cuda(), (h, w), thr).cpu().numpy()
            with np.errstate(all="ignore"):
                ref = OP.rescale_masks(m, b, (h, w), thr)
            assert got.shape == ref.shape
            for i in range(len(b)):
                assert (got[i] != ref[i]).mean() < 1e-4, (s, thr, i)   # only pixels whose sampled value sits on the threshold may differ
            assert np.array_equal(got[1], ref[1]) and np.array_equal(got[3], ref[3]) and np.array_equal(got[6], ref[6])


def test_paste_masks_visualisation_mode():
    """threshold < 0 (mask_utils.py:131-133): the sampled values times 255 as uint8 instead of a bool mask; equal to the
    oracle wherever the sampled value times 255 is not within rounding of an integer (the oracle's grid_sample sums its
    four taps in aten's vectorised order)."""
    from postprocess_utils.mask_utils import rescale_masks
    rng = np.random.default_rng(77)
    for (n, s, h, w) in [(6, 28, 120, 200), (3, 28, 75, 131), (2, 64, 90, 160)]:
        m = soft_masks(rng, n, s)
        x0, y0 = rng.uniform(-10, w * 0.5, n), rng.uniform(-10, h * 0.5, n)
        b = np.stack([x0, y0, x0 + rng.uniform(12, w * 0.5, n), y0 + rng.uniform(12, h * 0.5, n)], 1).astype(F32)
        got = rescale_masks(torch.from_numpy(m).cuda(), torch.from_numpy(b).cuda(), (h, w), -1.0)
        assert got.dtype == torch.uint8 and got.shape == (n, h, w)
        got = got.cpu().numpy().astype(np.int32)
        ref = OP.rescale_masks(m, b, (h, w), -1.0).astype(np.int32)
        assert ref.max() > 100 and np.abs(got - ref).max() <= 1 and (got != ref).mean() < 1e-3
        assert np.array_equal(got == 0, ref == 0) or (np.logical_xor(got == 0, ref == 0)).mean() < 1e-4
