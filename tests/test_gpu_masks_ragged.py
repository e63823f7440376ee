"""Masks for a batch of DIFFERENTLY sized originals in one launch (lmi_process_mask_ex with a table of images;
object_detectors/yolov8_lmi/model.py:215-236 loops over such a list), and the interpolation arithmetic itself
bit for bit against torch-CPU's F.interpolate."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from object_detectors_amd import synth
from oracle import masks as OM

F32 = np.float32


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def scene(rng, sizes, counts, max_det, nm=32):
    B = len(sizes)
    det = np.zeros((B, max_det, 6 + nm), F32)
    for b, ((h0, w0), n) in enumerate(zip(sizes, counts)):
        x1 = rng.random(n) * w0 * 0.6 - 2
        y1 = rng.random(n) * h0 * 0.6 - 2
        det[b, :n, 0], det[b, :n, 1] = x1, y1
        det[b, :n, 2] = np.minimum(x1 + rng.uniform(0.05, 0.5, n) * w0, w0)
        det[b, :n, 3] = np.minimum(y1 + rng.uniform(0.05, 0.5, n) * h0, h0)
        det[b, :n, 4] = 0.9
        det[b, :n, 6:] = rng.normal(0, 0.5, (n, nm))
    sm = np.full(B * max_det, -1, np.int32)
    sb = [0]
    for b, n in enumerate(counts):
        sm[sb[-1]:sb[-1] + n] = b * max_det + np.arange(n)
        sb.append(sb[-1] + n)
    return det, sm, np.array(sb, np.int32)


@pytest.mark.parametrize("out_dtype", [torch.uint8, torch.float32])
def test_ragged_batch_equals_per_image_calls_and_oracle(ops, out_dtype):
    rng = np.random.default_rng(61)
    # eight sizes: landscape, portrait, 4K, tiny, a width that is not a whole 16-byte chunk is in the second test
    sizes = [(1080, 1920), (480, 640), (640, 480), (2160, 3840), (720, 1280), (368, 496), (1200, 1600), (144, 256)]
    counts = [7, 3, 0, 2, 9, 4, 1, 5]
    B, mh, mw, max_det = len(sizes), 160, 160, 16
    proto = synth.proto(rng, B, 32, mh, mw)
    det, sm, sb = scene(rng, sizes, counts, max_det)
    views = ops.process_mask_ragged(dev(proto), dev(det), dev(sm), dev(sb), sizes, counts, out_dtype)
    assert [tuple(v.shape) for v in views] == [(n, h, w) for (h, w), n in zip(sizes, counts)]
    for b, ((h0, w0), n) in enumerate(zip(sizes, counts)):
        if n == 0:
            continue
        # the same image alone through the uniform entry: bit for bit
        sm1 = np.full(max_det, -1, np.int32)
        sm1[:n] = np.arange(n)
        one, _ = ops.process_mask(dev(proto[b:b + 1]), dev(det[b:b + 1]), dev(sm1), dev(np.array([0, n], np.int32)), (h0, w0), n, out_dtype)
        assert torch.equal(views[b], one), b
        ref = OM.process_mask_native(proto[b], det[b, :n, 6:], det[b, :n, :4], (h0, w0))
        assert ref.sum() > 0 and OM.mask_iou(views[b].cpu().numpy(), ref) >= 0.999, b


@pytest.mark.parametrize("sizes", [
    [(375, 500), (101, 67), (480, 640), (60, 60)],     # 500 and 67 are not whole chunks; 60 + 60 <= 128: aten's 4-weight kernel, the one-pass form for the batch
    [(375, 500), (101, 67), (480, 640), (211, 333)],   # every frame separable: two passes, stores at any address
])
def test_ragged_batch_with_unaligned_widths(ops, sizes):
    rng = np.random.default_rng(62)
    counts = [4, 3, 2, 3]
    B, mh, mw, max_det = len(sizes), 40, 40, 8
    proto = synth.proto(rng, B, 32, mh, mw)
    det, sm, sb = scene(rng, sizes, counts, max_det)
    both = {}
    for out_dtype in (torch.uint8, torch.float32):
        for upsample in ("split", "single"):
            views = ops.process_mask_ragged(dev(proto), dev(det), dev(sm), dev(sb), sizes, counts, out_dtype, upsample=upsample)
            both[upsample] = views
            for b, ((h0, w0), n) in enumerate(zip(sizes, counts)):
                ref = OM.process_mask_native(proto[b], det[b, :n, 6:], det[b, :n, :4], (h0, w0))
                assert OM.mask_iou(views[b].cpu().numpy(), ref) >= 0.999, b
        for a, c in zip(both["split"], both["single"]):
            assert torch.equal(a, c)
    with pytest.raises(ValueError):
        ops.process_mask_ragged(dev(proto), dev(det), dev(sm), dev(sb), sizes[:2], counts[:2])
    none = ops.process_mask_ragged(dev(proto), dev(det), dev(sm), dev(np.zeros(B + 1, np.int32)), sizes, [0] * B)
    assert [tuple(v.shape) for v in none] == [(0, h, w) for h, w in sizes]


@pytest.mark.parametrize("h0,w0,mh,mw", [(1080, 1920, 160, 160), (375, 500, 160, 160), (270, 480, 40, 40), (60, 60, 40, 40),
                                         (2160, 3840, 160, 160), (640, 640, 160, 160), (97, 131, 52, 40)])
def test_interpolation_is_bit_exact_against_torch_cpu(ops, h0, w0, mh, mw):
    """With one-hot coefficients the logits ARE prototype planes (1*p + 0*... is exact), so every difference left
    would come from the crop window, the bilinear arithmetic, the box crop or the threshold: none is allowed."""
    rng = np.random.default_rng(h0 + w0)
    n = 6
    proto = synth.proto(rng, 1, 32, mh, mw)
    det = np.zeros((1, n, 38), F32)
    det[0, :, 0] = rng.uniform(-3, w0 * 0.3, n)
    det[0, :, 1] = rng.uniform(-3, h0 * 0.3, n)
    det[0, :, 2] = w0 - rng.uniform(-3, w0 * 0.3, n)
    det[0, :, 3] = h0 - rng.uniform(-3, h0 * 0.3, n)
    det[0, np.arange(n), 6 + rng.integers(0, 32, n)] = 1.0
    sm, sb = np.arange(n, dtype=np.int32), np.array([0, n], np.int32)
    ref = OM.process_mask_native(proto[0], det[0, :, 6:], det[0, :, :4], (h0, w0)).astype(bool)
    assert 0.05 < ref.mean() < 0.95
    for dtype in (torch.uint8, torch.float32):
        for upsample in ("split", "single"):
            got, _ = ops.process_mask(dev(proto), dev(det), dev(sm), dev(sb), (h0, w0), n, dtype, upsample=upsample)
            assert np.array_equal(got.cpu().numpy() != 0, ref), (dtype, upsample, int((got.cpu().numpy().astype(bool) != ref).sum()))


def test_fp16_prototypes_equal_the_converted_tensor(ops):
    """`fp16=True` models (object_detectors/yolov8_lmi/model.py:68,132) hand fp16 prototypes to process_mask_native,
    which multiplies `protos.float()`: the kernel reads the halves as they are (LMI_MASK_PROTO_F16) and must give the
    masks of the converted tensor bit for bit -- uniform and ragged originals, and against the oracle on the rounded
    values; an odd prototype width (no row kernel) is converted on the way in."""
    from object_detectors_amd import _lib
    rng = np.random.default_rng(63)
    sizes = [(1080, 1920), (480, 640), (375, 500), (720, 1280)]
    counts = [6, 3, 4, 5]
    B, mh, mw, max_det = len(sizes), 160, 160, 8
    proto16 = dev(synth.proto(rng, B, 32, mh, mw)).half()
    proto32 = proto16.float()
    det, sm, sb = scene(rng, sizes, counts, max_det)
    a = ops.process_mask_ragged(proto16, dev(det), dev(sm), dev(sb), sizes, counts, torch.uint8)
    b = ops.process_mask_ragged(proto32, dev(det), dev(sm), dev(sb), sizes, counts, torch.uint8)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y) and int(x.sum()) > 0, i
        ref = OM.process_mask_native(proto32[i].cpu().numpy(), det[i, :counts[i], 6:], det[i, :counts[i], :4], sizes[i])
        assert OM.mask_iou(x.cpu().numpy(), ref) >= 0.999, i
    # uniform originals, the whole batch at the first size
    sm_u = np.full(B * max_det, -1, np.int32)
    sm_u[:sb[-1]] = np.concatenate([i * max_det + np.arange(n) for i, n in enumerate(counts)])
    u16, _ = ops.process_mask(proto16, dev(det), dev(sm_u), dev(sb), sizes[0], int(sb[-1]), torch.uint8)
    u32, _ = ops.process_mask(proto32, dev(det), dev(sm_u), dev(sb), sizes[0], int(sb[-1]), torch.uint8)
    assert torch.equal(u16, u32)
    # odd width: hip_ops converts (the MFMA tiles read fp32 only) ...
    p_odd = dev(synth.proto(rng, 1, 32, 41, 41)).half()
    o16, _ = ops.process_mask(p_odd, dev(det[:1]), dev(sm_u[:max_det]), dev(np.array([0, counts[0]], np.int32)), (270, 480), counts[0], torch.uint8)
    o32, _ = ops.process_mask(p_odd.float(), dev(det[:1]), dev(sm_u[:max_det]), dev(np.array([0, counts[0]], np.int32)), (270, 480), counts[0], torch.uint8)
    assert torch.equal(o16, o32)
    # ... and the C ABI refuses the flag where the row kernel cannot serve it
    L = _lib.load()
    ws = torch.empty((int(L.lmi_mask_workspace_bytes(8, 41, 41)),), dtype=torch.uint8, device="cuda")
    d, m, s = dev(det[:1]), dev(sm_u[:max_det]), dev(np.array([0, counts[0]], np.int32))
    rc = L.lmi_mask_logits_ex(p_odd.data_ptr(), 1, 32, 41, 41, d.data_ptr(), max_det, 38, m.data_ptr(), s.data_ptr(), 270, 480, None, None,
                              8, ws.data_ptr(), ws.numel(), _lib.LMI_MASK_PROTO_F16, None)
    assert rc == -1 and b"fp16 prototypes" in L.lmi_last_error()


def test_bool_masks_share_the_uint8_bytes(ops):
    """out_dtype=torch.bool: the kernels' 0/1 bytes are torch bools (same storage), uniform and ragged."""
    rng = np.random.default_rng(64)
    sizes, counts = [(480, 640), (375, 500)], [3, 4]
    proto = synth.proto(rng, 2, 32, 160, 160)
    det, sm, sb = scene(rng, sizes, counts, 8)
    a = ops.process_mask_ragged(dev(proto), dev(det), dev(sm), dev(sb), sizes, counts, torch.bool)
    b = ops.process_mask_ragged(dev(proto), dev(det), dev(sm), dev(sb), sizes, counts, torch.uint8)
    for x, y in zip(a, b):
        assert x.dtype == torch.bool and torch.equal(x, y.bool()) and torch.equal(x.view(torch.uint8), y)
    u, _ = ops.process_mask(dev(proto), dev(det), dev(sm), dev(sb), sizes[0], 7, torch.bool)
    v, _ = ops.process_mask(dev(proto), dev(det), dev(sm), dev(sb), sizes[0], 7, torch.uint8)
    assert u.dtype == torch.bool and torch.equal(u.view(torch.uint8), v)
