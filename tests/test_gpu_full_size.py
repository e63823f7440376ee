"""GPU parity at BASELINE.json's full sizes.

The vectorised oracle finishes whole 64-image NMS batches in well under a second, so kept indices and boxes are
compared for every image; full-resolution masks are compared with the oracle on a few images and, for the whole
batch, through size-independent properties: the greedy-NMS fixpoint (checked by an independent scalar-formula
restatement in this file, not by the oracle), zeros outside the box, batch invariance (an image processed alone
gives the same bytes as inside the batch) and idempotence of a second run into the same buffers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import boxes as OB
from oracle import masks as OM
from oracle import nms as ON
from object_detectors_amd import synth

F32 = np.float32


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def iou_matrix(a, b):
    """fp32 IoU of every box in a against every box in b with torchvision's operation order."""
    ax1, ay1, ax2, ay2 = (a[:, i, None] for i in range(4))
    bx1, by1, bx2, by2 = (b[None, :, i] for i in range(4))
    aa = ((ax2 - ax1) * (ay2 - ay1)).astype(F32)
    ba = ((bx2 - bx1) * (by2 - by1)).astype(F32)
    w = np.maximum(F32(0), np.minimum(ax2, bx2) - np.maximum(ax1, bx1)).astype(F32)
    h = np.maximum(F32(0), np.minimum(ay2, by2) - np.maximum(ay1, by1)).astype(F32)
    inter = (w * h).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(inter > 0, inter / ((aa + ba).astype(F32) - inter).astype(F32), F32(0)).astype(F32)


def assert_greedy_fixpoint(p, kept, conf, iou, nc, agnostic, max_det, max_wh=7680.0):
    """kept (anchor indices in output order) is THE greedy NMS answer for head p (C, A) iff: it is sorted by score,
    no kept box is suppressed by an earlier kept box, and every candidate that was not kept is suppressed by a kept
    box of higher score (or lies beyond the max_det-th keep)."""
    cls_scores = p[4:4 + nc]
    score = cls_scores.max(0)
    cls = cls_scores.argmax(0)
    cand = np.nonzero(score > F32(conf))[0]
    kept = np.asarray(kept)
    assert len(set(kept.tolist())) == len(kept) and set(kept.tolist()) <= set(cand.tolist())
    assert len(kept) <= max_det
    ks = score[kept]
    assert (np.diff(ks) < 0).all()                      # scores are distinct by construction
    half_w, half_h = (p[2] / F32(2)).astype(F32), (p[3] / F32(2)).astype(F32)
    xyxy = np.stack([p[0] - half_w, p[1] - half_h, p[0] + half_w, p[1] + half_h], 1).astype(F32)
    off = np.zeros_like(score) if agnostic else (cls.astype(F32) * F32(max_wh)).astype(F32)
    ob = (xyxy + off[:, None]).astype(F32)
    kk = iou_matrix(ob[kept], ob[kept]).astype(np.float64) > iou
    assert not np.triu(kk, 1).any()                     # kept boxes do not suppress each other
    rest = np.setdiff1d(cand, kept)
    if len(kept) == max_det:                            # greedy stopped there: later candidates were never examined
        rest = rest[score[rest] > ks[-1]]
    if len(rest):
        s = iou_matrix(ob[rest], ob[kept]).astype(np.float64) > iou
        s &= score[kept][None, :] > score[rest][:, None]
        assert s.any(1).all()                           # every dropped candidate has a higher-scoring suppressor


def test_c2_det_bs64_nms_all_images(ops):
    """C2: B=64, head (64,84,8400), conf 0.25, iou 0.45: every image against the oracle + fixpoint + batch invariance."""
    rng = np.random.default_rng(2)
    B, nc, S = 64, 80, 640
    p = synth.head(rng, B, nc, 0, S, objects=40)
    pd = dev(p)
    det, count, anchor = ops.nms_v8(pd, 0.25, 0.45, nc=nc)
    torch.cuda.synchronize()
    det_h, count_h, anchor_h = det.cpu().numpy(), count.cpu().numpy(), anchor.cpu().numpy()
    ref = ON.non_max_suppression(p, 0.25, 0.45, nc=nc)
    for b in range(B):
        rd, ra = ref[b]
        n = int(count_h[b])
        assert anchor_h[b, :n].tolist() == ra.tolist()
        assert np.abs(det_h[b, :n, :6] - rd[:, :6]).max() <= 1e-3
        assert_greedy_fixpoint(p[b], anchor_h[b, :n], 0.25, 0.45, nc, False, 300)
    # an image processed alone gives the same rows as inside the batch
    for b in (0, 17, 63):
        d1, c1, a1 = ops.nms_v8(pd[b:b + 1].contiguous(), 0.25, 0.45, nc=nc)
        n = int(c1.item())
        assert n == count_h[b] and torch.equal(a1[0, :n], anchor[b, :n]) and torch.equal(d1[0, :n], det[b, :n])
    # scale/clip + threshold at full batch: boxes against the oracle for all images
    r = ops.finalize_detections(det, count, nc, (S, S), (1080, 1920), 0.25)
    torch.cuda.synchronize()
    det2 = det.cpu().numpy()
    for b in range(B):
        rd, _ = ref[b]
        boxes = OB.scale_boxes((S, S), rd[:, :4], (1080, 1920, 3))
        assert np.abs(det2[b, :len(rd), :4] - boxes).max() <= 1e-3
    assert r["keep_count"].cpu().numpy().tolist() == count_h.tolist()   # same threshold as NMS: everything kept


@pytest.mark.parametrize("agnostic", [False, True])
def test_c5_dense_seg_nms(ops, agnostic):
    """C5 shard: 16 images, (16,116,8400), >= 1000 candidates per image over all 80 classes (classes >= 69
    exercise the fp32 class-offset quantisation), max_det 300, both NMS modes."""
    rng = np.random.default_rng(5)
    B, nc, nm, S = 16, 80, 32, 640
    p = synth.head(rng, B, nc, nm, S, objects=110, cluster=(8, 16))
    # move a third of the planted objects' classes to 69..79 by swapping class rows
    p[:, 4 + np.arange(0, 11)], p[:, 4 + np.arange(69, 80)] = p[:, 4 + np.arange(69, 80)].copy(), p[:, 4 + np.arange(0, 11)].copy()
    ncand = (p[:, 4:4 + nc].max(1) > F32(0.25)).sum(1)
    assert ncand.min() >= 1000
    det, count, anchor = ops.nms_v8(dev(p), 0.25, 0.45, nc=nc, agnostic=agnostic, max_det=300)
    torch.cuda.synchronize()
    det_h, count_h, anchor_h = det.cpu().numpy(), count.cpu().numpy(), anchor.cpu().numpy()
    ref = ON.non_max_suppression(p, 0.25, 0.45, nc=nc, agnostic=agnostic, max_det=300)
    hi = 0
    for b in range(B):
        rd, ra = ref[b]
        n = int(count_h[b])
        assert anchor_h[b, :n].tolist() == ra.tolist()
        assert np.abs(det_h[b, :n] - rd).max() <= 1e-3          # box, conf, class and the 32 coefficients
        assert_greedy_fixpoint(p[b], anchor_h[b, :n], 0.25, 0.45, nc, agnostic, 300)
        hi += int((rd[:, 5] >= 69).sum())
    assert hi > 0


def test_c3_seg1280_chain(ops):
    """C3 at its literal batch: S=1280, head (32,116,33600), proto (32,32,320,320), 1080p originals, ~60 objects per
    image.  Kept indices, boxes and class filter against the oracle for all 32 images, full-resolution oracle masks for
    two, the properties for all ~1900 masks."""
    rng = np.random.default_rng(3)
    B, nc, nm, S, h0, w0 = 32, 80, 32, 1280, 1080, 1920
    p = synth.head(rng, B, nc, nm, S, objects=60)
    proto = synth.proto(rng, B, nm, S // 4, S // 4)
    det, count, anchor = ops.nms_v8(dev(p), 0.25, 0.45, nc=nc)
    r = ops.finalize_detections(det, count, nc, (S, S), (h0, w0), 0.3)
    total = int(r["slot_base"][-1].item())
    protod = dev(proto)
    masks, overflow = ops.process_mask(protod, det, r["slot_map"], r["slot_base"], (h0, w0), total)
    torch.cuda.synchronize()
    assert overflow.item() == 0 and masks.shape == (total, h0, w0)
    det_h, count_h, keep_h = det.cpu().numpy(), count.cpu().numpy(), r["keep"].cpu().numpy().astype(bool)
    slot_base = r["slot_base"].cpu().numpy()
    ref = ON.non_max_suppression(p, 0.25, 0.45, nc=nc)
    for b in range(B):
        rd, ra = ref[b]
        n = int(count_h[b])
        assert anchor.cpu().numpy()[b, :n].tolist() == ra.tolist()
        boxes = OB.scale_boxes((S, S), rd[:, :4], (h0, w0, 3))
        assert np.abs(det_h[b, :n, :4] - boxes).max() <= 1e-3
        M = OB.class_filter(rd[:, 4], rd[:, 5], np.full(nc, 0.3))
        assert keep_h[b, :n].tolist() == M.tolist()
        if b < 2:   # full-resolution oracle masks for two images
            rm = OM.process_mask_native(proto[b], rd[:, 6:], boxes, (h0, w0))[M]
            gm = masks[slot_base[b]:slot_base[b + 1]].cpu().numpy()
            assert gm.shape == rm.shape and OM.mask_iou(gm, rm) >= 0.999
    # property, all masks: nothing outside the box (x1 <= col < x2, y1 <= row < y2)
    sm = r["slot_map"][:total].long()
    bx = det.reshape(-1, det.shape[-1])[sm][:, :4]
    ys = torch.arange(h0, device="cuda", dtype=torch.float32)[None, :, None]
    xs = torch.arange(w0, device="cuda", dtype=torch.float32)[None, None, :]
    nonempty = 0
    for a in range(0, total, 256):
        bb, mm = bx[a:a + 256], masks[a:a + 256].bool()
        inside = (xs >= bb[:, 0, None, None]) & (xs < bb[:, 2, None, None]) & (ys >= bb[:, 1, None, None]) & (ys < bb[:, 3, None, None])
        assert not (mm & ~inside).any()
        nonempty += int(mm.flatten(1).any(1).sum().item())
    assert total > 1000 and nonempty / total > 0.5      # and most masks are not empty (random logits: some are)
    # idempotence: a second run into the same buffer leaves the same bytes
    before = masks.clone()
    ops.process_mask(protod, det, r["slot_map"], r["slot_base"], (h0, w0), total, out=masks)
    torch.cuda.synchronize()
    assert torch.equal(before, masks)


def test_c2_seg_bs64_masks_properties(ops):
    """The bench workload itself (seg640, B=64, 1080p): masks of 4 images against the oracle, the rest through
    properties; one image processed alone gives the same bytes as in the batch."""
    rng = np.random.default_rng(2)
    B, nc, nm, S, h0, w0 = 64, 80, 32, 640, 1080, 1920
    p = synth.head(rng, B, nc, nm, S, objects=40)
    proto = synth.proto(rng, B, nm, S // 4, S // 4)
    protod = dev(proto)
    det, count, anchor = ops.nms_v8(dev(p), 0.25, 0.45, nc=nc)
    r = ops.finalize_detections(det, count, nc, (S, S), (h0, w0), 0.25)
    total = int(r["slot_base"][-1].item())
    masks, overflow = ops.process_mask(protod, det, r["slot_map"], r["slot_base"], (h0, w0), total)
    torch.cuda.synchronize()
    assert overflow.item() == 0 and total == int(count.sum().item())
    slot_base = r["slot_base"].cpu().numpy()
    ref = ON.non_max_suppression(p, 0.25, 0.45, nc=nc)
    for b in (0, 1, 31, 63):
        rd, _ = ref[b]
        boxes = OB.scale_boxes((S, S), rd[:, :4], (h0, w0, 3))
        rm = OM.process_mask_native(proto[b], rd[:, 6:], boxes, (h0, w0))
        gm = masks[slot_base[b]:slot_base[b + 1]].cpu().numpy()
        assert gm.shape == rm.shape and OM.mask_iou(gm, rm) >= 0.999
    # batch invariance of the mask path: image 40 alone
    b = 40
    d1, c1, a1 = ops.nms_v8(dev(p[b:b + 1]), 0.25, 0.45, nc=nc)
    r1 = ops.finalize_detections(d1, c1, nc, (S, S), (h0, w0), 0.25)
    t1 = int(r1["slot_base"][-1].item())
    m1, _ = ops.process_mask(protod[b:b + 1].contiguous(), d1, r1["slot_map"], r1["slot_base"], (h0, w0), t1)
    torch.cuda.synchronize()
    assert t1 == slot_base[b + 1] - slot_base[b] and torch.equal(m1, masks[slot_base[b]:slot_base[b + 1]])
    # checksum of checksums: per-mask pixel counts survive a reshuffle of the batch order (images are independent)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    d2, c2, a2 = ops.nms_v8(dev(p[perm.numpy()]), 0.25, 0.45, nc=nc)
    r2 = ops.finalize_detections(d2, c2, nc, (S, S), (h0, w0), 0.25)
    t2 = int(r2["slot_base"][-1].item())
    m2, _ = ops.process_mask(protod[perm.cuda()].contiguous(), d2, r2["slot_map"], r2["slot_base"], (h0, w0), t2)
    torch.cuda.synchronize()
    assert t2 == total
    sums = masks.flatten(1).sum(1, dtype=torch.int64).cpu().numpy()
    sums2 = m2.flatten(1).sum(1, dtype=torch.int64).cpu().numpy()
    sb2 = r2["slot_base"].cpu().numpy()
    for j, b in enumerate(perm.tolist()):
        assert sums2[sb2[j]:sb2[j + 1]].tolist() == sums[slot_base[b]:slot_base[b + 1]].tolist()
