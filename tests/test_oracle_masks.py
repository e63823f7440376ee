"""Oracle `masks`: torch-CPU path (what the reference's CPU path executes through ultralytics)
vs an independent numpy restatement of the aten bilinear arithmetic, and analytic cases."""
import numpy as np
import pytest

from oracle import masks as M

F32 = np.float32


def test_crop_window_values():
    assert M.mask_crop_window(160, 160, 1080, 1920) == (35, 125, 0, 160)
    assert M.mask_crop_window(160, 160, 640, 640) == (0, 160, 0, 160)
    assert M.mask_crop_window(320, 320, 1080, 1920) == (70, 250, 0, 320)
    assert M.mask_crop_window(160, 160, 500, 375) == (0, 160, 20, 140)


@pytest.mark.parametrize("h0,w0,mh", [(108, 192, 40), (75, 50, 40), (64, 64, 16), (270, 480, 80)])
def test_torch_path_vs_numpy_restatement(h0, w0, mh):
    rng = np.random.default_rng(h0 * 7 + w0)
    n = 6
    proto = rng.normal(0, 1, (32, mh, mh)).astype(F32)
    coef = rng.normal(0, 0.5, (n, 32)).astype(F32)
    x1 = rng.random(n) * w0 * 0.6
    y1 = rng.random(n) * h0 * 0.6
    boxes = np.stack([x1, y1, x1 + w0 * 0.35, y1 + h0 * 0.35], 1).astype(F32)
    a = M.process_mask_native(proto, coef, boxes, (h0, w0))
    b = M.process_mask_native_np(proto, coef, boxes, (h0, w0))
    assert a.shape == b.shape == (n, h0, w0) and a.dtype == np.float32
    assert set(np.unique(a)) <= {0.0, 1.0}
    assert M.mask_iou(a, b) >= 0.999
    assert a.sum() > 0
    # nothing outside the (half-open) boxes
    for i in range(n):
        ys, xs = np.nonzero(a[i])
        assert np.all(xs >= boxes[i, 0]) and np.all(xs < boxes[i, 2])
        assert np.all(ys >= boxes[i, 1]) and np.all(ys < boxes[i, 3])


def test_analytic_half_plane():
    # proto channel 0 = column ramp crossing zero at the centre; coef picks it -> right half is 1
    mh = 32
    proto = np.zeros((32, mh, mh), F32)
    proto[0] = (np.arange(mh, dtype=F32) - 15.5)[None, :]
    coef = np.zeros((1, 32), F32)
    coef[0, 0] = 1
    boxes = np.array([[0, 0, 128, 128]], F32)
    m = M.process_mask_native(proto, coef, boxes, (128, 128))
    assert np.all(m[0, :, :64] == 0) and np.all(m[0, :, 64:] == 1)
    m = M.process_mask_native(proto, -coef, boxes, (128, 128))
    assert np.all(m[0, :, :64] == 1) and np.all(m[0, :, 64:] == 0)


def test_empty():
    assert M.process_mask_native(np.zeros((32, 8, 8), F32), np.zeros((0, 32), F32), np.zeros((0, 4), F32), (20, 30)).shape == (0, 20, 30)
