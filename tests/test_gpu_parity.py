"""GPU parity tests: the HIP path (called through the C ABI of liblmi_hip.so via ctypes) against
the CPU oracle on the same seeded inputs.  Bars (BASELINE.json north_star): letterbox tensor
bit-exact, NMS kept indices identical, boxes within 1e-3 px (they are in fact bit-exact), masks
IoU >= 0.999.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import boxes as OB
from oracle import cv2_resize as OC
from oracle import masks as OM
from oracle import nms as ON
from oracle import preprocess as OP
from object_detectors_amd import synth

F32 = np.float32


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({2: np.uint16, 4: np.uint32}[a.dtype.itemsize])


# ---------------------------------------------------------------- preprocess (a1)
@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("shape", [(16, 16, 3), (16, 16), (16, 16, 1), (37, 53, 3), (5, 7)])
def test_preprocess_bit_exact(ops, shape, fp16):
    rng = np.random.default_rng(1)
    if shape[:2] == (16, 16):
        im = np.arange(256, dtype=np.uint8).reshape(16, 16)
        im = np.stack([im, im[::-1], im.T], -1) if shape == (16, 16, 3) else im.reshape(shape)
    else:
        im = rng.integers(0, 256, shape, dtype=np.uint8)
    got = ops.preprocess_u8(dev(im), fp16).cpu().numpy()
    ref = OP.preprocess(im, fp16)
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.array_equal(bits(got), bits(ref))


# ---------------------------------------------------------------- letterbox (a2+a3+a1)
LB_CASES = [  # (h, w, S_h, S_w)
    (54, 96, 64, 64), (1080, 1920, 640, 640), (960, 1280, 640, 640), (750, 1000, 640, 640), (800, 1333, 640, 640),
    (375, 500, 640, 640), (427, 640, 640, 640), (33, 77, 96, 128), (720, 1280, 1280, 1280), (641, 333, 640, 640),
    (100, 1000, 640, 640), (2160, 3840, 640, 640), (7, 5, 32, 32), (480, 640, 640, 640), (300, 300, 316, 316),
    # exact integer scales: odd ones have a zero second weight (those source rows are not fetched), even ones 1/2, 1/2
    (64, 64, 64, 64), (128, 128, 64, 64), (192, 192, 64, 64), (256, 256, 64, 64), (320, 320, 64, 64), (448, 448, 64, 64),
    (21, 21, 63, 63), (180, 320, 64, 64),
]


@pytest.mark.parametrize("h,w,sh,sw", LB_CASES)
def test_letterbox_bit_exact_uniform(ops, h, w, sh, sw):
    rng = np.random.default_rng(h * 10007 + w)
    frames = synth.frames_u8(rng, 2, h, w)
    got, gops = ops.letterbox(dev(frames), (sh, sw), pad_value=114, swap_rb=True)
    got = got.cpu().numpy()
    for b in range(2):
        ref, rops = OC.letterbox_u8(frames[b], sh, sw, pad_value=114, swap_rb=True)
        assert gops[b] == rops
        bad = np.argwhere(bits(got[b]) != bits(ref))
        assert bad.shape[0] == 0, f"{bad.shape[0]} mismatching elements, first (c,y,x): {bad[:8].tolist()}"


def test_letterbox_plan_execute_tables_built_once(ops):
    """LMI_LB_TABLES_ONLY / LMI_LB_TABLES_READY: the tables of a geometry are built once into a caller-owned buffer; calls
    that leave the table pass out equal the oracle (and the ordinary call) whatever the pixels, channel order, pad value,
    dtype or grey / colour -- and a buffer built for another geometry is refused."""
    rng = np.random.default_rng(77)
    for h, w, sh, sw, stretch in [(375, 500, 640, 640, False), (1080, 1920, 640, 640, False), (33, 77, 96, 128, True)]:
        tabs = ops.letterbox_tables((h, w), (sh, sw), "cuda" if stretch else "cuda:0", stretch=stretch)   # "cuda": the current device
        for swap, pad, fp16, grey in [(True, 114, False, False), (False, 0, True, False), (True, 7, False, True)]:
            frames = synth.frames_u8(rng, 3, h, w)
            f = frames[..., :1] if grey else frames
            got, gops = ops.letterbox(dev(f), (sh, sw), pad_value=pad, swap_rb=swap, fp16=fp16, stretch=stretch, tables=tabs)
            plain, pops = ops.letterbox(dev(f), (sh, sw), pad_value=pad, swap_rb=swap, fp16=fp16, stretch=stretch)
            assert gops == pops and torch.equal(got, plain)
            got = got.cpu().numpy()
            for b in range(3):
                src = np.repeat(f[b], 3, axis=2) if grey else f[b]
                ref, rops = OC.letterbox_u8(src, sh, sw, pad_value=pad, swap_rb=swap, fp16=fp16, stretch=stretch)
                assert gops[b] == rops and np.array_equal(bits(got[b]), bits(ref))
        with pytest.raises(ValueError):
            ops.letterbox(dev(synth.frames_u8(rng, 1, h + 1, w)), (sh, sw), stretch=stretch, tables=tabs)
    with pytest.raises(ValueError):   # both flags at once
        from object_detectors_amd import _lib
        t = ops.letterbox_tables((64, 64), (32, 32), "cuda:0")
        o = torch.empty((1, 3, 32, 32), device="cuda")
        _lib.check(_lib.load().lmi_letterbox_u8_uniform(dev(synth.frames_u8(rng, 1, 64, 64)).data_ptr(), 1, 64, 64, 192, 64 * 192, o.data_ptr(), 32, 32, 0,
                                                        0, 0, 0, _lib.LMI_LB_TABLES_ONLY | _lib.LMI_LB_TABLES_READY, t.data_ptr(), t.numel(), 0))


def test_letterbox_ragged_batch_fp16_stretch_and_rgb(ops):
    rng = np.random.default_rng(5)
    sizes = [(375, 500), (1080, 1920), (427, 640), (64, 48), (640, 640), (33, 1000)]
    frames = [synth.frames_u8(rng, 1, h, w)[0] for h, w in sizes]
    for fp16, stretch, swap, pad in [(False, False, False, 0), (True, False, True, 114), (False, True, True, 7)]:
        got, gops = ops.letterbox([dev(f) for f in frames], (640, 640), pad_value=pad, swap_rb=swap, fp16=fp16, stretch=stretch)
        got = got.cpu().numpy()
        for b, f in enumerate(frames):
            ref, rops = OC.letterbox_u8(f, 640, 640, pad_value=pad, swap_rb=swap, fp16=fp16, stretch=stretch)
            assert gops[b] == rops
            bad = np.argwhere(bits(got[b]) != bits(ref))
            assert bad.shape[0] == 0, f"frame {b} fp16={fp16} stretch={stretch}: {bad.shape[0]} mismatches, first {bad[:8].tolist()}"


def test_letterbox_strided_rows_and_bad_args(ops):
    rng = np.random.default_rng(6)
    big = synth.frames_u8(rng, 1, 120, 200)[0]
    view = dev(big)[:, :160]  # row stride 600 bytes > 3*160
    got, _ = ops.letterbox([view], (96, 96))
    ref, _ = OC.letterbox_u8(big[:, :160], 96, 96)
    assert np.array_equal(bits(got[0].cpu().numpy()), bits(ref))
    with pytest.raises(ValueError):
        ops.letterbox(dev(big)[None], (96, 96), pad_value=300)
    with pytest.raises(TypeError):
        ops.letterbox(torch.from_numpy(big)[None], (96, 96))


def test_letterbox_frames_at_any_byte_address(ops):
    """Frames are read where they lie: every start address modulo 16 (views into one buffer, no copy), image strides
    that are not multiples of 16 (a packed batch of 375x500 frames), colour and grey, cv2 and torch resamplers."""
    rng = np.random.default_rng(16)
    h, w = 45, 61
    for ch in (3, 1):
        frame = synth.frames_u8(rng, 1, h, w)[0][:, :, :ch]
        n = h * w * ch
        buf = torch.zeros(16 * (n + 64), dtype=torch.uint8, device="cuda")
        views = []
        for k in range(16):
            start = k * (n + 64) // 16 * 16 + k           # start address = k (mod 16)
            buf[start:start + n] = dev(frame).reshape(-1)
            v = buf[start:start + n].view(h, w, ch)
            assert v.data_ptr() % 16 == k and v.data_ptr() == buf.data_ptr() + start
            views.append(v)
        for mode in ("cv2", "torch"):
            got, _ = ops.letterbox(views, (96, 128), pad_value=9, swap_rb=True, mode=mode)
            assert all(torch.equal(got[k], got[0]) for k in range(16)), mode
        col = np.repeat(frame, 3, axis=2) if ch == 1 else frame
        ref, _ = OC.letterbox_u8(col, 96, 128, pad_value=9, swap_rb=True)
        got, _ = ops.letterbox(views, (96, 128), pad_value=9, swap_rb=True)
        assert np.array_equal(bits(got[5].cpu().numpy()), bits(ref))
    # a packed uniform batch whose frames lie 375*500*3 = 562500 bytes apart (4 mod 16), read in place by the uniform entry
    frames = synth.frames_u8(rng, 5, 375, 500)
    d = dev(frames)
    before = torch.cuda.memory_allocated()
    got, _ = ops.letterbox(d, (640, 640), pad_value=114)
    assert torch.cuda.memory_allocated() - before <= got.numel() * 4 + (1 << 20)   # the output (+ a table workspace): no copy of the frames
    for b in (0, 1, 3):
        ref, _ = OC.letterbox_u8(frames[b], 640, 640, pad_value=114)
        assert np.array_equal(bits(got[b].cpu().numpy()), bits(ref)), b


def test_letterbox_full_size_closed_forms(ops):
    """BASELINE sizes, checked through size-independent closed forms instead of the slow oracle:
    1080p->640 is an exact x3 (pixel pick), 4K->640 an exact x6 (2x2 box mean)."""
    rng = np.random.default_rng(8)
    f = torch.from_numpy(rng.integers(0, 256, (8, 1080, 1920, 3), dtype=np.uint8)).cuda()
    out, ops_ = ops.letterbox(f, (640, 640), pad_value=114)
    assert ops_[0] == [{"resize": [640, 360, 1920, 1080]}, {"pad": [0, 0, 140, 140]}]
    lut = (torch.arange(256, dtype=torch.float32) / 255).cuda()   # IEEE division on the CPU (GPU torch multiplies by 1/255)
    ref = lut[f[:, 1::3, 1::3, :].long()].permute(0, 3, 1, 2)
    assert torch.equal(out[:, :, 140:500, :], ref)
    assert torch.all(out[:, :, :140] == lut[114]) and torch.all(out[:, :, 500:] == lut[114])
    # C4's per-step batch at its literal size: 64 4K frames (1.6 GB) in one launch
    g = torch.randint(0, 256, (64, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    out, _ = ops.letterbox(g, (640, 640), pad_value=0)
    for a in range(0, 64, 8):
        r0, r1 = g[a:a + 8, 2::6].int(), g[a:a + 8, 3::6].int()   # rows 6dy+2 and 6dy+3: the two taps of the x6 reduction
        box = (r0[:, :, 2::6] + r0[:, :, 3::6] + r1[:, :, 2::6] + r1[:, :, 3::6] + 2) >> 2
        assert torch.equal(out[a:a + 8, :, 140:500, :], lut[box.long()].permute(0, 3, 1, 2)), a
    assert torch.all(out[:, :, :140] == 0) and torch.all(out[:, :, 500:] == 0)


# ---------------------------------------------------------------- NMS (a4)
def run_nms(ops, p, **kw):
    det, count, anchor = ops.nms_v8(dev(p), **kw)
    torch.cuda.synchronize()
    det, count, anchor = det.cpu().numpy(), count.cpu().numpy(), anchor.cpu().numpy()
    return [(det[b, :count[b]], anchor[b, :count[b]]) for b in range(p.shape[0])]


def assert_nms_equal(got, ref):
    assert len(got) == len(ref)
    for b, ((gd, ga), (rd, ra)) in enumerate(zip(got, ref)):
        assert ga.tolist() == ra.tolist(), f"image {b}: kept anchors differ"
        assert np.array_equal(bits(gd), bits(rd.astype(F32))), f"image {b}: rows differ"


def _hand_head(boxes_xywh, cls_scores, nc, nm=0, A=None):
    n = len(boxes_xywh)
    A = A or n
    p = np.zeros((1, 4 + nc + nm, A), F32)
    p[0, :4, :n] = np.asarray(boxes_xywh, F32).T
    p[0, 2:4, n:] = 1.0
    for i, sc in enumerate(cls_scores):
        for k, v in sc.items():
            p[0, 4 + k, i] = v
    if nm:
        p[0, 4 + nc:, :] = np.arange(nm * A, dtype=F32).reshape(nm, A)
    return p


def test_nms_hand_cases(ops):
    cases = []
    cases.append((_hand_head([[5, 5, 10, 10], [5, 2.5, 10, 5]], [{0: 0.9}, {0: 0.8}], 2), dict(conf_thres=0.25, iou_thres=0.5, nc=2)))
    cases.append((_hand_head([[5, 5, 10, 10], [5, 2.5, 10, 5]], [{0: 0.9}, {0: 0.8}], 2), dict(conf_thres=0.25, iou_thres=0.49, nc=2)))
    p3 = _hand_head([[50, 50, 20, 20]] * 3, [{0: 0.9}, {1: 0.8}, {0: 0.7}], 3)
    cases.append((p3, dict(conf_thres=0.25, iou_thres=0.45, nc=3, agnostic=False)))
    cases.append((p3, dict(conf_thres=0.25, iou_thres=0.45, nc=3, agnostic=True)))
    thr = 0.25
    cases.append((_hand_head([[10, 10, 4, 4], [30, 30, 4, 4], [60, 60, 4, 4]],
                             [{0: thr}, {1: 0.5, 2: 0.5}, {2: float(np.nextafter(F32(thr), F32(1)))}], 3), dict(conf_thres=thr, iou_thres=0.45, nc=3)))
    boxes = [[20 * i + 5, 5, 8, 8] for i in range(12)]
    scores = [{0: 0.5}] * 4 + [{0: 0.9 - 0.01 * i} for i in range(8)]
    cases.append((_hand_head(boxes, scores, 1), dict(conf_thres=0.25, iou_thres=0.45, nc=1, max_det=6)))
    cases.append((_hand_head(boxes, scores, 1), dict(conf_thres=0.25, iou_thres=0.45, nc=1, max_det=300)))   # ties keep anchor order
    for k in (0, 79):
        cases.append((_hand_head([[100.02, 100.02, 0.05, 0.05], [100.05, 100.05, 0.05, 0.05]], [{k: 0.9}, {k: 0.8}], 80),
                      dict(conf_thres=0.25, iou_thres=0.45, nc=80)))
    cases.append((_hand_head([[10, 10, 4, 4], [50, 50, 4, 4]], [{1: 0.6}, {0: 0.7}], 2, nm=4, A=7), dict(conf_thres=0.25, iou_thres=0.45, nc=2)))
    cases.append((_hand_head([[5, 5, 2, 2]], [{0: 0.1}], 2, nm=3), dict(conf_thres=0.25, iou_thres=0.45, nc=2)))   # no candidates
    for p, kw in cases:
        assert_nms_equal(run_nms(ops, p, **kw), ON.non_max_suppression(p, **kw))


@pytest.mark.parametrize("nc,nm,size,objects,agnostic,conf", [
    (80, 0, 640, 40, False, 0.25), (80, 0, 640, 40, True, 0.25), (80, 32, 640, 40, False, 0.25),
    (80, 32, 320, 120, False, 0.05), (80, 32, 320, 120, True, 0.05), (3, 0, 160, 10, False, 0.25), (1, 32, 320, 30, False, 0.25),
])
def test_nms_synthetic_heads(ops, nc, nm, size, objects, agnostic, conf):
    rng = np.random.default_rng(nc * 1000 + nm + size + objects)
    p = synth.head(rng, 3, nc, nm, size, objects)
    kw = dict(conf_thres=conf, iou_thres=0.45, agnostic=agnostic, nc=nc, max_det=300)
    ref = ON.non_max_suppression(p, **kw)
    assert max(len(r[1]) for r in ref) > 5
    assert_nms_equal(run_nms(ops, p, **kw), ref)


def test_nms_dense_scene_high_classes_and_max_det(ops):
    """config 5 flavour: >= 1000 candidates/image, classes >= 69 (offset quantisation), max_det cap."""
    rng = np.random.default_rng(55)
    p = synth.head(rng, 2, 80, 32, 640, objects=400, cluster=(3, 6), classes=np.arange(60, 80))
    ncand = (p[:, 4:84].max(1) > F32(0.25)).sum(1)
    assert ncand.min() >= 1000
    for agnostic in (False, True):
        for max_det in (300, 50):
            kw = dict(conf_thres=0.25, iou_thres=0.45, agnostic=agnostic, nc=80, max_det=max_det)
            ref = ON.non_max_suppression(p, **kw)
            assert_nms_equal(run_nms(ops, p, **kw), ref)
    assert len(ref[0][1]) == 50


def test_nms_odd_anchor_count_and_many_candidates(ops):
    """A % 4 != 0 takes the scalar load path; a very low threshold makes every anchor a candidate
    (n > 16384 exercises the global-memory sort, n > max_nms the top-k cut)."""
    rng = np.random.default_rng(77)
    p = synth.head(rng, 2, 5, 0, 160, objects=8, anchors=523)
    kw = dict(conf_thres=0.25, iou_thres=0.45, nc=5)
    assert_nms_equal(run_nms(ops, p, **kw), ON.non_max_suppression(p, **kw))
    p = synth.head(rng, 1, 4, 0, 640, objects=30, anchors=20000)
    p[:, 2:4] = np.minimum(p[:, 2:4], 12.0)   # small boxes keep the CPU oracle's greedy loop short-ish
    kw = dict(conf_thres=1e-6, iou_thres=0.45, nc=4, max_det=300)
    assert (p[0, 4:].max(0) > F32(1e-6)).sum() > 16384
    assert_nms_equal(run_nms(ops, p, **kw), ON.non_max_suppression(p, **kw))
    kw = dict(conf_thres=1e-6, iou_thres=0.45, nc=4, max_det=300, max_nms=5000)
    assert_nms_equal(run_nms(ops, p, **kw), ON.non_max_suppression(p, **kw))


def test_nms_argument_errors(ops):
    p = dev(np.zeros((1, 6, 8), F32))
    with pytest.raises(ValueError):
        ops.nms_v8(p, 0.25, 1.5, nc=2)
    with pytest.raises(Exception):
        ops.nms_v8(p, 0.25, 0.45, nc=2, max_det=5000)
    with pytest.raises(TypeError):
        ops.nms_v8(torch.zeros((1, 6, 8)), 0.25, 0.45, nc=2)


# ---------------------------------------------------------------- finalize (a5, a6)
@pytest.mark.parametrize("h0,w0,S", [(1080, 1920, 640), (2160, 3840, 640), (427, 640, 640), (375, 500, 640), (720, 1280, 1280), (640, 640, 640)])
def test_finalize_scale_clip_threshold(ops, h0, w0, S):
    rng = np.random.default_rng(h0 + S)
    B, max_det, nc = 3, 40, 6
    det = np.zeros((B, max_det, 6), F32)
    det[..., :4] = rng.random((B, max_det, 4), dtype=F32) * (S + 60) - 30
    det[..., 4] = rng.random((B, max_det), dtype=F32)
    det[..., 5] = rng.integers(0, nc, (B, max_det)).astype(F32)
    det[0, 0, 4] = F32(0.3)            # fp32(0.3) > 0.3 as doubles -> kept
    det[0, 0, 5] = 1
    count = np.array([40, 17, 0], np.int32)
    thr = np.array([0.5, 0.3, 0.6, 1.0, 0.05, 0.9], np.float64)
    for th_arg, th_ref in ((dev(thr), thr), (0.5, np.full(nc, 0.5))):
        d = dev(det.copy())
        r = ops.finalize_detections(d, dev(count), nc, (S, S), (h0, w0), th_arg)
        torch.cuda.synchronize()
        d = d.cpu().numpy()
        keep = r["keep"].cpu().numpy().astype(bool)
        slot_base = r["slot_base"].cpu().numpy()
        slot_map = r["slot_map"].cpu().numpy()
        mask_slot = r["mask_slot"].cpu().numpy()
        nxt = 0
        for b in range(B):
            n = count[b]
            ref = OB.scale_boxes((S, S), det[b, :n, :4], (h0, w0, 3))
            assert np.array_equal(bits(d[b, :n, :4]), bits(ref))
            m = OB.class_filter(det[b, :n, 4], det[b, :n, 5], th_ref)
            assert keep[b, :n].tolist() == m.tolist() and not keep[b, n:].any()
            assert r["keep_count"].cpu().numpy()[b] == m.sum()
            assert slot_base[b] == nxt
            for i in np.nonzero(m)[0]:
                assert mask_slot[b, i] == nxt and slot_map[nxt] == b * max_det + i
                nxt += 1
            assert (mask_slot[b][~keep[b]] == -1).all()
        assert slot_base[B] == nxt
        if isinstance(th_arg, torch.Tensor):
            assert keep[0, 0]   # fp32(0.3) vs the double 0.3: kept


# ---------------------------------------------------------------- masks (a7)
@pytest.mark.parametrize("h0,w0,S,out_dtype", [(270, 480, 160, torch.uint8), (1080, 1920, 640, torch.uint8), (375, 500, 640, torch.float32),
                                                 (101, 67, 160, torch.uint8), (640, 640, 640, torch.uint8),
                                                 (90, 120, 100, torch.uint8),      # 25x25 prototypes: odd width -> MFMA tile kernel
                                                 (240, 320, 320, torch.uint8)])    # 200 + 150 survivors: several staging batches
def test_process_mask_iou(ops, h0, w0, S, out_dtype):
    rng = np.random.default_rng(h0 * 3 + w0)
    many = (h0, w0) == (240, 320)
    B, nc, nm, max_det = 2, 4, 32, (256 if many else 24)
    mh = mw = S // 4
    proto = synth.proto(rng, B, nm, mh, mw)
    n = np.array([200, 150] if many else [9, 5], np.int32)
    det = np.zeros((B, max_det, 6 + nm), F32)
    for b in range(B):
        x1 = rng.random(n[b]) * w0 * 0.6
        y1 = rng.random(n[b]) * h0 * 0.6
        det[b, :n[b], 0] = x1
        det[b, :n[b], 1] = y1
        det[b, :n[b], 2] = np.minimum(x1 + rng.uniform(0.05, 0.4, n[b]) * w0, w0)
        det[b, :n[b], 3] = np.minimum(y1 + rng.uniform(0.05, 0.4, n[b]) * h0, h0)
        det[b, :n[b], 4] = rng.uniform(0.2, 0.9, n[b])
        det[b, :n[b], 5] = rng.integers(0, nc, n[b])
        det[b, :n[b], 6:] = rng.normal(0, 0.5, (n[b], nm))
    keep = [det[b, :n[b], 4] > 0.4 for b in range(B)]
    # hand-built slot tables (boxes are already in original-image space)
    slot_map, slot_base = [], [0]
    for b in range(B):
        slot_map += [b * max_det + i for i in np.nonzero(keep[b])[0]]
        slot_base.append(len(slot_map))
    total = slot_base[-1]
    assert total > 3
    sm = np.full(B * max_det, -1, np.int32)
    sm[:total] = slot_map
    masks, overflow = ops.process_mask(dev(proto), dev(det), dev(sm), dev(np.array(slot_base, np.int32)), (h0, w0), total, out_dtype)
    torch.cuda.synchronize()
    assert overflow.item() == 0
    masks = masks.cpu().numpy()
    assert masks.shape == (total, h0, w0)
    assert set(np.unique(masks)) <= {0, 1}
    k = 0
    for b in range(B):
        ref = OM.process_mask_native(proto[b], det[b, :n[b], 6:], det[b, :n[b], :4], (h0, w0))[keep[b]]
        got = masks[k:k + ref.shape[0]]
        k += ref.shape[0]
        assert ref.sum() > 0
        assert OM.mask_iou(got, ref) >= 0.999
        for g, r in zip(got, ref):   # per-mask bar as well, for masks of meaningful size
            if r.sum() >= 2000:
                assert OM.mask_iou(g, r) >= 0.999
    # capacity smaller than the survivor count: extra slots are skipped and reported
    masks2, overflow2 = ops.process_mask(dev(proto), dev(det), dev(sm), dev(np.array(slot_base, np.int32)), (h0, w0), total - 2, out_dtype)
    torch.cuda.synchronize()
    assert overflow2.item() == 2
    assert np.array_equal(masks2.cpu().numpy(), masks[:total - 2])


# ---------------------------------------------------------------- end to end post-processing
def test_postprocess_chain_seg(ops):
    """NMS -> finalize -> masks on one synthetic seg batch, vs the oracle chained the way
    object_detectors/yolov8_lmi/model.py:212-236 chains the reference calls."""
    rng = np.random.default_rng(99)
    B, nc, nm, S, h0, w0 = 2, 80, 32, 320, 270, 480
    p = synth.head(rng, B, nc, nm, S, objects=12)
    proto = synth.proto(rng, B, nm, S // 4, S // 4)
    conf = 0.25
    det, count, anchor = ops.nms_v8(dev(p), conf, 0.45, nc=nc)
    r = ops.finalize_detections(det, count, nc, (S, S), (h0, w0), 0.4)
    total = int(r["slot_base"][-1].item())
    masks, overflow = ops.process_mask(dev(proto), det, r["slot_map"], r["slot_base"], (h0, w0), total)
    torch.cuda.synchronize()
    det_h, count_h, keep_h, masks_h = det.cpu().numpy(), count.cpu().numpy(), r["keep"].cpu().numpy().astype(bool), masks.cpu().numpy()
    ref = ON.non_max_suppression(p, conf, 0.45, nc=nc)
    k = 0
    for b in range(B):
        rd, ra = ref[b]
        assert anchor.cpu().numpy()[b, :count_h[b]].tolist() == ra.tolist()
        boxes = OB.scale_boxes((S, S), rd[:, :4], (h0, w0, 3))
        assert np.abs(det_h[b, :count_h[b], :4] - boxes).max() <= 1e-3
        M = OB.class_filter(rd[:, 4], rd[:, 5], np.full(nc, 0.4))
        assert keep_h[b, :count_h[b]].tolist() == M.tolist()
        rm = OM.process_mask_native(proto[b], rd[:, 6:], boxes, (h0, w0))[M]
        gm = masks_h[k:k + rm.shape[0]]
        k += rm.shape[0]
        assert OM.mask_iou(gm, rm) >= 0.999
    assert k == total
