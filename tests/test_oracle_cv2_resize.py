"""Oracle `cv2_resize` cross-checks.  opencv-python is not installed: PARITY UNPINNED against
the binary.  What is checked here: (1) an independent per-pixel scalar implementation of the
published fixed-point scheme (the reference's own testing idiom,
tests/lmi_utils/gadget_utils/test_pipeline_utils.py:118-152), (2) hand-derived closed forms
(exact x3 picks a pixel, x6 and x2 are 2x2 box means, identity, constants, saturation,
both border clamps), (3) the letterbox geometry against the reference helpers' pad rule."""
import math
import struct

import numpy as np
import pytest

from oracle import cv2_resize as C
from oracle import geometry as G


def _f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def _round_half_even(x):
    return int(round(x))  # python round is half-to-even on exact binary values


def _axis_scalar(src_n, dst_n, d, is_x):
    scale = 1.0 / (dst_n / src_n)
    f = _f32((d + 0.5) * scale - 0.5)
    s = math.floor(f)
    f = _f32(f - s)
    if is_x:
        if s < 0:
            f, s = 0.0, 0
        if s >= src_n - 1:
            f, s = 0.0, src_n - 1
    w0 = _round_half_even(_f32(_f32(1.0 - f) * 2048.0))
    w1 = _round_half_even(_f32(f * 2048.0))
    s0 = min(max(s, 0), src_n - 1)
    s1 = min(max(s + 1, 0), src_n - 1)
    return s0, s1, w0, w1


def resize_scalar(src, dst_w, dst_h):
    h, w, c = src.shape
    out = np.zeros((dst_h, dst_w, c), np.uint8)
    xs = [_axis_scalar(w, dst_w, d, True) for d in range(dst_w)]
    ys = [_axis_scalar(h, dst_h, d, False) for d in range(dst_h)]
    for dy, (y0, y1, b0, b1) in enumerate(ys):
        for dx, (x0, x1, a0, a1) in enumerate(xs):
            for k in range(c):
                r0 = int(src[y0, x0, k]) * a0 + int(src[y0, x1, k]) * a1
                r1 = int(src[y1, x0, k]) * a0 + int(src[y1, x1, k]) * a1
                v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
                assert 0 <= v <= 255
                out[dy, dx, k] = v
    return out


@pytest.mark.parametrize("shape,dst", [((5, 7), (4, 3)), ((7, 5), (3, 4)), ((24, 32), (13, 9)), ((9, 6), (17, 23)),
                                        ((15, 20), (32, 24)), ((31, 17), (16, 16)), ((2, 2), (9, 9)), ((1, 5), (3, 4)),
                                        ((40, 3), (2, 33))])
def test_vectorised_equals_scalar(shape, dst):
    rng = np.random.default_rng(hash((shape, dst)) % 2**32)
    src = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    assert np.array_equal(C.resize_linear_u8(src, dst[0], dst[1]), resize_scalar(src, dst[0], dst[1]))


def test_exact_x3_picks_pixel():
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (27, 48, 3), dtype=np.uint8)
    out = C.resize_linear_u8(src, 16, 9)
    assert np.array_equal(out, src[1::3, 1::3])


@pytest.mark.parametrize("k", [2, 6])
def test_even_integer_scale_is_box_mean_of_centre_2x2(k):
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (6 * k, 8 * k, 3), dtype=np.uint8).astype(np.int32)
    out = C.resize_linear_u8(src.astype(np.uint8), 8, 6)
    o = k // 2 - 1
    a, b = src[o::k, o::k], src[o::k, o + 1::k]
    c, d = src[o + 1::k, o::k], src[o + 1::k, o + 1::k]
    assert np.array_equal(out, ((a + b + c + d + 2) >> 2).astype(np.uint8))


def test_identity_constant_and_saturation():
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (11, 13, 3), dtype=np.uint8)
    assert np.array_equal(C.resize_linear_u8(src, 13, 11), src)
    for v in (0, 1, 127, 254, 255):
        const = np.full((9, 14, 3), v, np.uint8)
        for dst in [(5, 4), (31, 20), (14, 3)]:
            assert np.all(C.resize_linear_u8(const, *dst) == v)


def test_border_clamps_on_upscale():
    # upscale: first columns have floor(fx) = -1 (clamped to pixel 0, weight reset);
    # last columns have sx >= w-1 (pixel w-1).  Rows keep their weights but read a clamped row.
    src = np.zeros((4, 4, 3), np.uint8)
    src[:, 0] = 200
    src[:, -1] = 50
    src[0, :] = 90
    out = C.resize_linear_u8(src, 16, 16)
    assert np.all(out[6:, 0] == 200) and np.all(out[6:, -1] == 50)
    assert np.all(out[0, :] == 90)
    s0, s1, w0, w1 = C.linear_coeffs(4, 16, clamp_taps=True)
    assert s0[0] == 0 and w0[0] == 2048 and w1[0] == 0
    assert s0[-1] == 3 and w0[-1] == 2048 and w1[-1] == 0
    t0, t1, v0, v1 = C.linear_coeffs(4, 16, clamp_taps=False)
    assert t0[0] == 0 and t1[0] == 0 and v0[0] + v1[0] == 2048 and v1[0] != 0


def test_weights_sum_to_one():
    for src_n, dst_n in [(1920, 640), (1333, 640), (500, 640), (375, 480), (7, 1000), (4096, 33)]:
        for clamp in (True, False):
            _, _, w0, w1 = C.linear_coeffs(src_n, dst_n, clamp)
            assert np.all(w0 + w1 == 2048)
            assert w0.min() >= 0 and w1.min() >= 0


@pytest.mark.parametrize("h,w,S", [(1080, 1920, 640), (2160, 3840, 640), (1080, 1920, 1280), (375, 500, 640),
                                   (427, 640, 640), (800, 1333, 640), (641, 333, 640), (100, 1000, 640),
                                   (480, 640, 640), (333, 777, 512)])
def test_letterbox_geometry_matches_reference_pad_rule(h, w, S):
    nw, nh, left, right, top, bottom = C.letterbox_geometry(h, w, S, S)
    assert left + nw + right == S and top + nh + bottom == S
    # reference helper rule (pipeline_utils.py:110-131): pad_L = (W-w)//2
    pl, pr, pt, pb = G.fit_pads(nh, nw, W=S, H=S)
    assert (left, right, top, bottom) == (pl, pr, pt, pb)
    assert max(nw, nh) == S


def test_letterbox_composition():
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (54, 96, 3), dtype=np.uint8)
    chw, ops = C.letterbox_u8(frame, 64, 64, pad_value=114, swap_rb=True)
    assert chw.shape == (3, 64, 64) and chw.dtype == np.float32
    assert ops == [{"resize": [64, 36, 96, 54]}, {"pad": [0, 0, 14, 14]}]
    small = C.resize_linear_u8(frame, 64, 36)
    assert np.array_equal(chw[:, 14:50, :], (small[:, :, ::-1].transpose(2, 0, 1).astype(np.float32) / np.float32(255)))
    assert np.all(chw[:, :14] == np.float32(114) / np.float32(255))
    assert np.all(chw[:, 50:] == np.float32(114) / np.float32(255))
    h16, _ = C.letterbox_u8(frame, 64, 64, fp16=True)
    assert h16.dtype == np.float16
