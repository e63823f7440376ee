"""Edge cases on the GPU path that the reference's domain has: nothing detected, more survivors than the caller's
mask buffer, degenerate frames, single-pixel sources, boxes on the frame border."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import boxes as OB
from oracle import masks as OM
from oracle import nms as ON
from oracle import cv2_resize as OC
from object_detectors_amd import synth

F32 = np.float32


def sweep_seeds(first):
    """The committed seed of a differential sweep, plus LMI_FUZZ_EXTRA more (a one-off soak on the GPU box:
    `LMI_FUZZ_EXTRA=40 python -m pytest tests/test_gpu_edges.py -m gpu -k "random or boundaries"`)."""
    import os
    return [first] + [10_000 + first + k for k in range(int(os.environ.get("LMI_FUZZ_EXTRA", "0")))]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def test_nothing_detected_anywhere(ops):
    rng = np.random.default_rng(0)
    p = synth.head(rng, 3, 8, 32, 160, objects=0)
    assert p[:, 4:12].max() < 0.5
    det, count, anchor = ops.nms_v8(dev(p), 0.5, 0.45, nc=8)
    r = ops.finalize_detections(det, count, 8, (160, 160), (90, 160), 0.5)
    torch.cuda.synchronize()
    assert count.cpu().tolist() == [0, 0, 0] and r["keep_count"].cpu().tolist() == [0, 0, 0]
    assert r["slot_base"].cpu().tolist() == [0, 0, 0, 0] and not r["keep"].any()
    # one image of the batch with detections, the others without
    p2 = p.copy()
    p2[1] = synth.head(rng, 1, 8, 32, 160, objects=6)[0]
    det, count, anchor = ops.nms_v8(dev(p2), 0.25, 0.45, nc=8)
    ref = ON.non_max_suppression(p2, 0.25, 0.45, nc=8)
    c = count.cpu().tolist()
    assert c[0] == 0 and c[2] == 0 and c[1] == len(ref[1][1]) > 0
    assert anchor.cpu().numpy()[1, :c[1]].tolist() == ref[1][1].tolist()


def test_mask_buffer_smaller_than_survivors(ops):
    rng = np.random.default_rng(1)
    B, nc, nm, S, h0, w0 = 2, 4, 32, 160, 90, 160
    p = synth.head(rng, B, nc, nm, S, objects=10)
    proto = synth.proto(rng, B, nm, S // 4, S // 4)
    det, count, _ = ops.nms_v8(dev(p), 0.25, 0.45, nc=nc)
    r = ops.finalize_detections(det, count, nc, (S, S), (h0, w0), 0.25)
    total = int(r["slot_base"][-1].item())
    assert total > 6
    full, ov = ops.process_mask(dev(proto), det, r["slot_map"], r["slot_base"], (h0, w0), total)
    cap = total - 5
    part, ov2 = ops.process_mask(dev(proto), det, r["slot_map"], r["slot_base"], (h0, w0), cap)
    torch.cuda.synchronize()
    assert ov.item() == 0 and ov2.item() == 5 and part.shape[0] == cap
    assert torch.equal(part, full[:cap])           # the slots that fit are complete and identical


@pytest.mark.parametrize("h,w,S", [(1, 1, 32), (2, 3, 32), (1, 64, 64), (64, 1, 64), (3, 700, 64)])
def test_letterbox_degenerate_frames(ops, h, w, S):
    rng = np.random.default_rng(h * 100 + w)
    f = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    out, geo = ops.letterbox(dev(f), (S, S))
    torch.cuda.synchronize()
    ref, ops_ref = OC.letterbox_u8(f[0], S, S, pad_value=114, swap_rb=False)
    got = out[0].cpu().numpy()
    assert geo[0] == ops_ref
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_boxes_on_the_frame_border_and_outside(ops):
    """scale_boxes clips to the frame; masks of boxes that touch or leave the frame stay inside it and agree with the oracle."""
    rng = np.random.default_rng(5)
    nm, S, h0, w0 = 32, 160, 90, 160
    proto = synth.proto(rng, 1, nm, S // 4, S // 4)
    max_det = 8
    det = np.zeros((1, max_det, 6 + nm), F32)
    boxes = np.array([[-30, -20, 40, 50], [100, 40, 400, 300], [0, 35, 160, 125], [60, 80, 60, 90], [10.5, 40.25, 10.75, 41.0]], F32)
    n = len(boxes)
    det[0, :n, :4] = boxes            # model space, some outside the 160x160 canvas
    det[0, :n, 4] = 0.9
    det[0, :n, 6:] = rng.normal(0, 0.5, (n, nm))
    d = dev(det)
    count = torch.tensor([n], dtype=torch.int32).cuda()
    r = ops.finalize_detections(d, count, 4, (S, S), (h0, w0), 0.5)
    total = int(r["slot_base"][-1].item())
    assert total == n
    masks, _ = ops.process_mask(dev(proto), d, r["slot_map"], r["slot_base"], (h0, w0), total)
    torch.cuda.synchronize()
    sb = OB.scale_boxes((S, S), boxes, (h0, w0, 3))
    assert np.abs(d.cpu().numpy()[0, :n, :4] - sb).max() <= 1e-3
    assert sb.min() >= 0 and sb[:, [0, 2]].max() <= w0 and sb[:, [1, 3]].max() <= h0
    ref = OM.process_mask_native(proto[0], det[0, :n, 6:], sb, (h0, w0))
    got = masks.cpu().numpy()
    assert got.shape == ref.shape
    for g, rr in zip(got, ref):
        assert (g != rr).sum() <= max(2, 0.001 * rr.sum())
    assert got[3].sum() == 0            # zero-width box


@pytest.mark.parametrize("seed", sweep_seeds(2024))
def test_letterbox_random_geometries_one_ragged_batch(ops, seed):
    """Differential sweep: 48 random frame sizes (tiny to wide, up- and down-scales, odd strides of the scale factor) in
    ragged batches against the oracle, bit for bit, over the option combinations."""
    rng = np.random.default_rng(seed)
    sizes = [(int(rng.integers(1, 300)), int(rng.integers(1, 400))) for _ in range(40)]
    sizes += [(719, 1279), (1081, 1921), (97, 1003), (1003, 97), (640, 640), (641, 639), (320, 320), (159, 161)]
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    dframes = [dev(f) for f in frames]
    for (dh, dw), fp16, stretch, swap, pad in [((160, 160), False, False, True, 114), ((96, 224), True, False, False, 0),
                                               ((224, 96), False, True, True, 255), ((320, 320), False, False, False, 114)]:
        got, gops = ops.letterbox(dframes, (dh, dw), pad_value=pad, swap_rb=swap, fp16=fp16, stretch=stretch)
        got = got.cpu().numpy()
        view = np.uint16 if fp16 else np.uint32
        for b, f in enumerate(frames):
            ref, rops = OC.letterbox_u8(f, dh, dw, pad_value=pad, swap_rb=swap, fp16=fp16, stretch=stretch)
            assert gops[b] == rops, (sizes[b], dh, dw)
            bad = np.argwhere(got[b].view(view) != ref.view(view))
            assert bad.shape[0] == 0, f"frame {sizes[b]} -> {(dh, dw)} stretch={stretch}: {bad.shape[0]} mismatches, first {bad[:4].tolist()}"


@pytest.mark.parametrize("sweep_only", [False, True])   # suppressor lists first (agnostic / few classes) or the sorted sweep alone
@pytest.mark.parametrize("seed", sweep_seeds(77))
def test_nms_random_configurations_with_score_ties(ops, seed, sweep_only):
    """Differential sweep over head shapes and NMS settings; scores quantised to 1/128 so that many candidates tie
    and the order rule (equal scores keep anchor order, torch.sort stable) decides the kept set."""
    rng = np.random.default_rng(seed)
    for trial in range(24):
        nc = int(rng.choice([1, 2, 7, 80]))
        nm = int(rng.choice([0, 0, 32, 5]))
        A = int(rng.choice([64, 333, 1000, 2100, 4116]))
        B = int(rng.integers(1, 4))
        p = synth.head(rng, B, nc, nm, 320, objects=int(rng.integers(0, 30)), anchors=A)
        p[:, 4:4 + nc] = np.round(p[:, 4:4 + nc] * 128) / 128            # ties, including exact threshold hits
        conf = float(rng.choice([0.25, 0.5, 1 / 128, 0.0]))
        iou = float(rng.choice([0.45, 0.3, 0.7, 0.0, 1.0]))
        agnostic = bool(rng.integers(0, 2))
        max_det = int(rng.choice([300, 5, 1, 64]))
        det, count, anchor = ops.nms_v8(dev(p), conf, iou, agnostic=agnostic, max_det=max_det, nc=nc, sweep_only=sweep_only)
        torch.cuda.synchronize()
        ref = ON.non_max_suppression(p, conf, iou, agnostic=agnostic, max_det=max_det, nc=nc)
        for b in range(B):
            rd, ra = ref[b]
            n = int(count[b])
            tag = f"trial {trial}: nc={nc} nm={nm} A={A} conf={conf} iou={iou} agnostic={agnostic} max_det={max_det} image {b}"
            assert anchor[b, :n].cpu().numpy().tolist() == ra.tolist(), tag
            assert n == 0 or np.abs(det[b, :n].cpu().numpy() - rd).max() <= 1e-3, tag


@pytest.mark.parametrize("seed", sweep_seeds(303))
def test_process_mask_random_shapes(ops, seed):
    """Differential sweep over prototype sizes (odd widths and non-square included), frame sizes and box layouts."""
    rng = np.random.default_rng(seed)
    nm = 32
    for trial in range(14):
        mh, mw = int(rng.choice([20, 25, 40, 41, 80])), int(rng.choice([20, 25, 40, 52, 80]))
        h0, w0 = int(rng.integers(17, 400)), int(rng.integers(17, 500))
        B, max_det = int(rng.integers(1, 4)), 16
        proto = synth.proto(rng, B, nm, mh, mw)
        det = np.zeros((B, max_det, 6 + nm), F32)
        n = rng.integers(0, max_det + 1, B)
        slot_map, slot_base = [], [0]
        for b in range(B):
            x1, y1 = rng.random(n[b]) * w0 * 0.8, rng.random(n[b]) * h0 * 0.8
            det[b, :n[b], 0], det[b, :n[b], 1] = x1, y1
            det[b, :n[b], 2] = np.minimum(x1 + rng.uniform(0.5, 0.6 * w0, n[b]), w0)
            det[b, :n[b], 3] = np.minimum(y1 + rng.uniform(0.5, 0.6 * h0, n[b]), h0)
            det[b, :n[b], 6:] = rng.normal(0, 0.5, (n[b], nm))
            slot_map += [b * max_det + i for i in range(n[b])]
            slot_base.append(len(slot_map))
        total = slot_base[-1]
        if total == 0:
            continue
        sm = np.full(B * max_det, -1, np.int32)
        sm[:total] = slot_map
        dtype = torch.uint8 if trial % 2 else torch.float32
        gain = min(mh / h0, mw / w0)
        if int(mh - (mh - h0 * gain) / 2) <= int((mh - h0 * gain) / 2) or int(mw - (mw - w0 * gain) / 2) <= int((mw - w0 * gain) / 2):
            # a frame so thin that scale_masks' crop of the prototype window is empty: the reference path raises
            # (F.interpolate on an empty input), and so does this one, with the same exception type
            with pytest.raises(RuntimeError):
                OM.process_mask_native(proto[0], det[0, :1, 6:], det[0, :1, :4], (h0, w0))
            with pytest.raises(RuntimeError):
                ops.process_mask(dev(proto), dev(det), dev(sm), dev(np.array(slot_base, np.int32)), (h0, w0), total, dtype)
            continue
        masks, ov = ops.process_mask(dev(proto), dev(det), dev(sm), dev(np.array(slot_base, np.int32)), (h0, w0), total, dtype)
        torch.cuda.synchronize()
        got = masks.cpu().numpy()
        k = 0
        for b in range(B):
            ref = OM.process_mask_native(proto[b], det[b, :n[b], 6:], det[b, :n[b], :4], (h0, w0))
            g = got[k:k + n[b]]
            k += n[b]
            tag = f"trial {trial}: proto {mh}x{mw} frame {h0}x{w0} image {b}"
            assert g.shape == ref.shape, tag
            diff = (g != ref).sum()
            assert diff <= max(3, 0.001 * ref.sum()), f"{tag}: {diff} pixels differ of {int(ref.sum())} set"


@pytest.mark.parametrize("rotated", [False, True])
def test_nms_halves_equal_the_one_call_form(ops, rotated):
    """lmi_nms_decode + lmi_nms_suppress (hip_ops.nms_v8(decode_done=event)) against lmi_nms_v8[_rotated], with work
    on a second stream started from the event between the halves, as bench.py schedules the letterbox."""
    rng = np.random.default_rng(31)
    nc, nm = (15, 1) if rotated else (80, 32)
    head = synth.head(rng, 4, nc, nm, 640, 30)
    if rotated:
        head[:, 4 + nc] = rng.uniform(-0.7, 2.2, head[:, 4 + nc].shape)
    pred = torch.from_numpy(head).cuda()
    one = ops.nms_v8(pred.clone(), 0.25, 0.45, nc=nc, rotated=rotated)
    ev, side = torch.cuda.Event(), torch.cuda.Stream()
    frames = torch.from_numpy(rng.integers(0, 256, (2, 270, 480, 3), dtype=np.uint8)).cuda()
    torch.cuda.synchronize()
    two = ops.nms_v8(pred.clone(), 0.25, 0.45, nc=nc, rotated=rotated, decode_done=ev)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        lb, _ = ops.letterbox(frames, (160, 160))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(one[1], two[1]) and int(one[1].sum()) > 0
    for b, c in enumerate(one[1].tolist()):
        assert torch.equal(one[0][b, :c], two[0][b, :c]) and torch.equal(one[2][b, :c], two[2][b, :c])
    want, _ = OC.letterbox_u8(frames[1].cpu().numpy(), 160, 160)
    assert np.array_equal(lb[1].cpu().numpy(), want)


def test_mask_upsample_two_pass_form_equals_single_pass_and_oracle(ops):
    """Frames with h0 + w0 > 128 take the zero-fill + box-pass form of lmi_mask_upsample: against the one-kernel
    form (flag LMI_MASK_UPSAMPLE_SINGLE) bit for bit and against the oracle, on the layouts that stress
    it: full-frame and out-of-frame boxes, empty boxes, > 256 chunk columns (fp32, 4K-wide rows), frames smaller
    than the prototype window, and a buffer with more capacity than survivors (dead slots stay untouched)."""
    import os
    rng = np.random.default_rng(404)
    nm = 32
    cases = [  # (mh, mw, h0, w0, dtype)
        (40, 40, 270, 480, torch.uint8), (40, 40, 270, 480, torch.float32), (80, 80, 1080, 1920, torch.uint8),
        (40, 40, 135, 3840, torch.float32), (40, 52, 33, 48, torch.uint8), (80, 80, 2160, 3840, torch.uint8),
        (160, 160, 64, 96, torch.uint8), (25, 40, 200, 16, torch.uint8), (20, 20, 17, 4, torch.float32),
        # rows that are not whole 16-byte chunks: stores at any address, the row's last chunk pixel by pixel, a fill with a tail
        (40, 40, 270, 481, torch.uint8), (80, 80, 375, 500, torch.uint8), (40, 40, 133, 203, torch.float32), (40, 40, 129, 7, torch.uint8),
    ]
    for ci, (mh, mw, h0, w0, dtype) in enumerate(cases):
        B, max_det = 2, 8
        proto = synth.proto(rng, B, nm, mh, mw)
        det = np.zeros((B, max_det, 6 + nm), F32)
        boxes = np.array([[0, 0, w0, h0], [-50, -30, w0 * 0.4, h0 * 0.5], [w0 * 0.3, h0 * 0.2, w0 + 40, h0 + 9],
                          [w0 * 0.5, h0 * 0.5, w0 * 0.5, h0 * 0.9], [w0 + 5, 3, w0 + 50, 30], [w0 * 0.1, h0 * 0.6, w0 * 0.12, h0 * 0.61],
                          [1.5, 2.5, 17.25, h0 - 0.5], [w0 * 0.7, h0 * 0.1, w0 * 0.95, h0 * 0.35]], F32)
        n = [8, 5]
        slot_map, slot_base = [], [0]
        for b in range(B):
            det[b, :n[b], :4] = boxes[rng.permutation(8)[:n[b]]]
            det[b, :n[b], 6:] = rng.normal(0, 0.5, (n[b], nm))
            slot_map += [b * max_det + i for i in range(n[b])]
            slot_base.append(len(slot_map))
        total, capacity = slot_base[-1], slot_base[-1] + 3
        sm = np.full(B * max_det, -1, np.int32)
        sm[:total] = slot_map
        args = (dev(proto), dev(det), dev(sm), dev(np.array(slot_base, np.int32)), (h0, w0), capacity, dtype)
        fill = 171 if dtype == torch.uint8 else 171.0
        two = torch.full((capacity, h0, w0), fill, dtype=dtype, device="cuda")
        one = torch.full((capacity, h0, w0), fill, dtype=dtype, device="cuda")
        ops.process_mask(*args, out=two)
        ops.process_mask(*args, out=one, upsample="single")
        torch.cuda.synchronize()
        assert torch.equal(two, one), f"case {ci}"
        assert bool((two[total:] == fill).all()), f"case {ci}: dead slots were written"
        got, k = two.cpu().numpy(), 0
        for b in range(B):
            ref = OM.process_mask_native(proto[b], det[b, :n[b], 6:], det[b, :n[b], :4], (h0, w0))
            g = got[k:k + n[b]]
            k += n[b]
            assert (g != ref).sum() <= max(3, 0.001 * ref.sum()), f"case {ci} image {b}"


@pytest.mark.parametrize("seed", sweep_seeds(909))
def test_nms_class_parallel_path_boundaries(ops, seed):
    """The per-class decomposition of the greedy pass (class_parallel_nms in nms.hip) and its fallbacks, against the
    oracle's plain greedy NMS on the class-offset boxes: classes of exactly 64 / 65 / 256 / 257 candidates (one,
    two, four words per bit row, fallback), 128 and 129 classes, boxes spread wider than the class offset (the
    decomposition is not valid there: chunked path), max_det cutting the kept list, heavy overlap inside a class."""
    rng = np.random.default_rng(seed)

    def head(class_sizes, A, spread=600.0, nc=None, big=None):
        nc = nc or len(class_sizes)
        p = np.zeros((1, 4 + nc, A), F32)
        p[0, 4:] = 0.01
        n = sum(class_sizes)
        idx = rng.permutation(A)[:n]
        centres = rng.uniform(40, spread, (max(1, n // 6), 2))
        k = 0
        for c, m in enumerate(class_sizes):
            for _ in range(m):
                a = idx[k]; k += 1
                ctr = centres[rng.integers(0, len(centres))] + rng.normal(0, 3.0, 2)
                p[0, 0, a], p[0, 1, a] = ctr
                p[0, 2, a], p[0, 3, a] = rng.uniform(20, 60, 2)
                p[0, 4 + c, a] = rng.uniform(0.3, 0.95)
        if big is not None:   # one candidate far outside the max_wh band
            p[0, 0, idx[0]] = big
        return p

    cases = [
        dict(sizes=[64, 3, 10]), dict(sizes=[65, 64, 1]), dict(sizes=[256, 20]), dict(sizes=[257, 20]),
        dict(sizes=[128, 129, 130, 1, 0, 7]), dict(sizes=[5] * 128), dict(sizes=[5] * 129),
        dict(sizes=[40, 30, 20], big=9000.0), dict(sizes=[40, 30, 20], spread=7700.0),
        dict(sizes=[200, 100, 50], max_det=7), dict(sizes=[300, 300, 300, 300], max_det=300),
        dict(sizes=[63, 1, 64, 2, 65]), dict(sizes=[1]), dict(sizes=[2, 0, 0, 1]),
        # the kernel leaves the decomposition to the sorted sweep when the largest class exceeds max(48, 0.3 n - 50)
        # (few, large classes: cheaper there): enough other candidates keep the 2-, 3- and 4-word bit rows in use
        dict(sizes=[65] + [20] * 20), dict(sizes=[130] + [24] * 20), dict(sizes=[256] + [26] * 30), dict(sizes=[257] + [26] * 30),
        dict(sizes=[89] + [20] * 20), dict(sizes=[90] + [20] * 20),   # n = 489 / 490: the bound is 96 / 97
    ]
    for ci, cs in enumerate(cases):
        sizes = cs["sizes"]
        A = 2100
        p = head(sizes, A, spread=cs.get("spread", 600.0), big=cs.get("big"))
        nc = len(sizes)
        for iou in (0.45, 0.2):
            md = cs.get("max_det", 300)
            det, count, anchor = ops.nms_v8(dev(p), 0.25, iou, max_det=md, nc=nc)
            torch.cuda.synchronize()
            (rd, ra), = ON.non_max_suppression(p, 0.25, iou, max_det=md, nc=nc)
            n = int(count[0])
            tag = f"case {ci} sizes={sizes[:6]} iou={iou}"
            assert anchor[0, :n].cpu().numpy().tolist() == ra.tolist(), tag
            assert n == 0 or np.abs(det[0, :n].cpu().numpy() - rd).max() <= 1e-3, tag


@pytest.mark.parametrize("sweep_only", [True, False])
@pytest.mark.parametrize("seed", sweep_seeds(1212))
def test_nms_sorted_sweep_windows(ops, seed, sweep_only):
    """The sweep over the sorted list (agnostic NMS; per-class NMS with few, large classes, which the class-parallel
    path hands over): windows of 256 candidates, of which the first 64 still alive are resolved per step.  Scenes that
    leave fewer / more than 64 alive in a window, more than one window, a max_det cut inside a step, one class only."""
    rng = np.random.default_rng(seed)
    for nc, objects, agn, md in [(1, 40, False, 300), (2, 100, False, 300), (3, 15, False, 300), (80, 110, True, 300),
                                 (5, 60, True, 300), (2, 100, True, 37), (4, 150, False, 64)]:
        p = synth.head(rng, 2, nc, 0, 320, objects, cluster=(6, 16))
        det, count, anchor = ops.nms_v8(dev(p), 0.25, 0.45, agnostic=agn, max_det=md, nc=nc, sweep_only=sweep_only)
        torch.cuda.synchronize()
        ref = ON.non_max_suppression(p, 0.25, 0.45, agnostic=agn, max_det=md, nc=nc)
        for b, (rd, ra) in enumerate(ref):
            n = int(count[b])
            tag = f"nc={nc} objects={objects} agnostic={agn} max_det={md} image {b}"
            assert anchor[b, :n].cpu().numpy().tolist() == ra.tolist(), tag
            assert n == 0 or np.array_equal(det[b, :n].cpu().numpy(), rd.astype(F32)), tag


@pytest.mark.parametrize("seed", sweep_seeds(4097))
def test_nms_more_candidates_than_the_lds_buffers_hold(ops, seed):
    """Validation-style thresholds: (nearly) every one of 8400 anchors is a candidate.  The kernel first takes only the
    head of the sorted list (the 4096 largest keys, by a radix histogram) and comes back for the whole list when that
    ends short of max_det keeps: random boxes (max_det keeps come from the head), boxes in a few tight clusters (far
    fewer keeps than max_det: the whole list), many equal scores at the cut, a tiny max_det, agnostic and per class."""
    rng = np.random.default_rng(seed)
    A = 8400

    def head(nc, clusters, quant):
        p = np.zeros((2, 4 + nc, A), F32)
        if clusters:
            ctr = rng.uniform(60, 580, (clusters, 2)); wh = rng.uniform(30, 120, (clusters, 2))
            k = rng.integers(0, clusters, (2, A))
            p[:, 0:2] = np.moveaxis(ctr[k], -1, 1) + rng.normal(0, 0.7, (2, 2, A))
            p[:, 2:4] = np.moveaxis(wh[k], -1, 1) * np.exp(rng.normal(0, 0.01, (2, 2, A)))
        else:
            p[:, 0:2] = rng.uniform(0, 640, (2, 2, A))
            p[:, 2:4] = rng.uniform(10, 200, (2, 2, A))
        sc = rng.uniform(0.011, 0.95, (2, nc, A)).astype(F32) * (rng.random((2, nc, A)) < 0.6)
        if quant:
            sc = np.round(sc * quant) / quant     # long runs of equal scores, also across the 4096-key cut
        p[:, 4:] = sc
        return p

    for nc, clusters, quant, agn, md, iou in [(3, 0, 0, False, 300, 0.7), (3, 40, 0, False, 300, 0.45), (80, 25, 0, True, 300, 0.45),
                                              (2, 0, 64, True, 300, 0.6), (1, 60, 16, False, 5, 0.45), (5, 0, 0, False, 1000, 0.3)]:
        p = head(nc, clusters, quant)
        assert (p[:, 4:].max(1) > 0.01).sum(1).min() > 4096
        det, count, anchor = ops.nms_v8(dev(p), 0.01, iou, agnostic=agn, max_det=md, nc=nc)
        torch.cuda.synchronize()
        ref = ON.non_max_suppression(p, 0.01, iou, agnostic=agn, max_det=md, nc=nc)
        for b, (rd, ra) in enumerate(ref):
            n = int(count[b])
            tag = f"nc={nc} clusters={clusters} quant={quant} agnostic={agn} max_det={md} image {b}"
            assert anchor[b, :n].cpu().numpy().tolist() == ra.tolist(), tag
            assert n == 0 or np.array_equal(det[b, :n].cpu().numpy(), rd.astype(F32)), tag
