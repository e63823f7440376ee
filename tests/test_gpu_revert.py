"""GPU parity for the device-side operator-list revert: lmi_revert_points_f32 and
lmi_revert_masks_stage against (a) outputs of the reference's own revert_to_origin / revert_masks_to_origin
(tests/golden/l1_geometry.json, tests/golden/revert_masks.npz) and (b) the numpy oracle on larger seeded inputs.
Bars: points bit-exact; fp32 masks bit-exact for enlargements (aten's CPU result for reductions depends on its
vectorisation path: <= 1 level on <= 2 % of the pixels there)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import geometry as OG

HERE = os.path.dirname(os.path.abspath(__file__))
F32 = np.float32


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def test_revert_points_golden_and_random(ops):
    gold = json.load(open(os.path.join(HERE, "golden", "l1_geometry.json")))
    n = 0
    for case in gold["revert"]:
        if case["kind"] != "f32" or not case["tensor"]:
            continue
        pts = torch.tensor(case["pts"], dtype=torch.float32).cuda()
        got = ops.revert_points(pts, case["ops"]).cpu().numpy()
        assert np.array_equal(got, np.array(case["out"], F32))
        n += 1
    assert n >= 4
    rng = np.random.default_rng(21)
    for cols, oplist in [(4, [{"resize": [640, 427, 1333, 889]}, {"pad": [0, 0, 106, 107]}]),
                         (2, [{"resize": [640, 360, 3840, 2160]}, {"pad": [0, 0, 140, 140]}, {"flip": [True, True, 640, 640]}]),
                         (4, [{"stretch": [1.3, 0.7]}, {"pad": [-5, -6, 7, 8]}, {"flip": [False, True, 100, 200]}, {"resize": [300, 200, 77, 91]}]),
                         (2, [])]:
        p = (rng.random((5000, cols)) * 700 - 30).astype(F32)
        got = ops.revert_points(torch.from_numpy(p).cuda(), oplist).cpu().numpy()
        ref = OG.revert_to_origin(p, oplist)
        assert got.dtype == np.float32 and np.array_equal(got, np.asarray(ref, F32))
    with pytest.raises(Exception):
        ops.revert_points(torch.zeros((3, 3), device="cuda"), [])
    with pytest.raises(Exception):
        ops.revert_points(torch.zeros((3, 2), device="cuda"), [{"rotate": [1]}])


def test_revert_masks_golden(ops):
    z = np.load(os.path.join(HERE, "golden", "revert_masks.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    for i, c in enumerate(meta["cases"]):
        src, ref = z[f"in{i}"], z[f"out{i}"]
        got = ops.revert_masks(torch.from_numpy(src).cuda(), c["ops"]).cpu().numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype
        assert np.array_equal(got, OG.revert_masks_to_origin(src, c["ops"]))     # the oracle, bit for bit
        assert np.array_equal(got, ref), f"case {i}"                              # the reference's own output


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_revert_masks_letterbox_undo_1080p(ops, dtype):
    """The shape the path produces: masks at the 640x640 model input, undone to 1080x1920."""
    rng = np.random.default_rng(8)
    m = 12
    y, x = np.mgrid[0:640, 0:640]
    src = np.zeros((m, 640, 640), dtype)
    for i in range(m):
        cy, cx, ry, rx = rng.uniform(150, 490), rng.uniform(0, 640), rng.uniform(10, 150), rng.uniform(10, 300)
        src[i] = (((y - cy) / ry) ** 2 + ((x - cx) / rx) ** 2 < 1).astype(dtype)
    oplist = [{"resize": [640, 360, 1920, 1080]}, {"pad": [0, 0, 140, 140]}]
    got = ops.revert_masks(torch.from_numpy(src).cuda(), oplist)
    assert got.shape == (m, 1080, 1920)
    for i in (0, 5, 11):
        assert np.array_equal(got[i].cpu().numpy(), OG.revert_mask_to_origin(src[i], oplist))
    # batch invariance and flip involution
    one = ops.revert_masks(torch.from_numpy(src[3:4]).cuda(), oplist)
    assert torch.equal(one[0], got[3])
    f = [{"flip": [True, True, 1920, 1080]}]
    assert torch.equal(ops.revert_masks(ops.revert_masks(got, f), f), got)
    assert torch.equal(ops.revert_masks(got, f), torch.flip(got, [1, 2]))
    # pad only: centred zero pad and centred crop
    padded = ops.revert_masks(got[:2], [{"pad": [-3, -4, -5, -6]}])
    assert padded.shape == (2, 1091, 1927)
    assert torch.equal(padded[:, 5:5 + 1080, 3:3 + 1920], got[:2])
    border = padded.clone()
    border[:, 5:5 + 1080, 3:3 + 1920] = 0
    assert not border.any()   # (comparing fp32 sums of the two shapes would depend on the reduction order)
    cropped = ops.revert_masks(got[:2], [{"pad": [10, 11, 20, 21]}])
    assert torch.equal(cropped, got[:2, 20:1080 - 21, 10:1920 - 11])
    with pytest.raises(ValueError):
        ops.revert_masks(got[:1], [{"pad": [1000, 1000, 0, 0]}])
    assert ops.revert_masks(got[:0], oplist).shape == (0, 1080, 1920)


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
@pytest.mark.parametrize("w,h,flip", [
    (1916, 1080, None), (1080, 1920, (True, False)), (500, 375, (False, True)), (333, 250, (True, True)),
    (36, 200, None), (100, 60, (True, False)), (4100, 300, None), (1000, 750, None), (2044, 500, (True, True)),
    (1917, 1080, (True, False)), (2448, 2048, None), (3836, 600, (False, True)), (516, 1300, None), (1080, 1920, None),
    (1280, 720, (True, False)), (2050, 1100, None), (3836, 2160, None),
])
def test_revert_masks_rows_that_are_not_whole_lines(ops, dtype, w, h, flip):
    """Output rows that are not whole 128-byte lines (fp32: revert_resize_lines_kernel, results staged in LDS and
    written out in chunks cut by absolute address; uint8: revert_resize_u8_kernel with four runs of 4 pixels per lane,
    narrow last groups sharing rows between their lanes, the all-background shortcut, or the generic kernel for very
    large enlargements), with the flips folded into the stage, every lane split the host
    can choose (1-4 chunk columns per lane), widths that are not whole 16-byte chunks, a width beyond the kernel's
    limit and shapes below its threshold (revert_resize_kernel): each mask equals the oracle bit for bit."""
    rng = np.random.default_rng(w * 7 + h)
    g = min(640 / h, 640 / w)
    nw, nh = int(round(w * g)), int(round(h * g))
    pl, pt = (640 - nw) // 2, (640 - nh) // 2
    m = 5
    src = np.zeros((m, 640, 640), dtype)
    y, x = np.mgrid[0:640, 0:640]
    for i in range(m):
        cy, cx = rng.uniform(pt, pt + nh), rng.uniform(pl, pl + nw)
        ry, rx = rng.uniform(5, nh / 2 + 5), rng.uniform(5, nw / 2 + 5)
        inside = ((y - cy) / ry) ** 2 + ((x - cx) / rx) ** 2 < 1
        src[i] = inside.astype(dtype) if dtype == np.uint8 else inside * rng.random((640, 640), dtype=np.float32)
    oplist = [{"resize": [nw, nh, w, h]}, {"pad": [pl, 640 - nw - pl, pt, 640 - nh - pt]}]
    if flip:
        oplist.insert(0, {"flip": [flip[0], flip[1], w, h]})
    got = ops.revert_masks(torch.from_numpy(src).cuda(), oplist)
    assert got.shape == (m, h, w)
    gc = got.cpu().numpy()
    for i in range(m):
        assert np.array_equal(gc[i], OG.revert_mask_to_origin(src[i], oplist)), i
    # a plane that does not start on a 16-byte boundary (a view into a larger buffer, one float / byte further)
    if dtype == np.float32:
        big = torch.zeros(m * 640 * 640 + 1, dtype=torch.float32, device="cuda")
        view = big[1:].view(m, 640, 640)
        view.copy_(torch.from_numpy(src))
        assert torch.equal(ops.revert_masks(view, oplist), got)
