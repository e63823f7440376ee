"""lmi_nms_v8_ex: (a) the finalize stage fused into the suppression kernel's epilogue must give exactly what
lmi_nms_v8 followed by lmi_finalize_detections gives (rows, keep flags, counts, dense slots); (b) an fp16 head read
natively must give exactly what `pred.float()` + the fp32 path gives, and what the oracle gives on the promoted head."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from object_detectors_amd import synth
from oracle import boxes as OB, nms as ON

F32 = np.float32


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from object_detectors_amd import hip_ops
    return hip_ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def assert_same_finalize(ops, pred, conf, iou, nc, model_hw, orig_hw, thres, agnostic=False, max_det=300, pad_round=True, split=False):
    d0, c0, a0 = ops.nms_v8(pred, conf, iou, agnostic=agnostic, max_det=max_det, nc=nc)
    r0 = ops.finalize_detections(d0, c0, nc, model_hw, orig_hw, thres, pad_round=pad_round)
    ev = torch.cuda.Event() if split else None
    d1, c1, a1, r1 = ops.nms_v8(pred, conf, iou, agnostic=agnostic, max_det=max_det, nc=nc, decode_done=ev,
                                finalize=dict(model_hw=model_hw, orig_hw=orig_hw, thres=thres, pad_round=pad_round))
    torch.cuda.synchronize()
    cnt = c0.cpu().numpy()
    assert np.array_equal(cnt, c1.cpu().numpy())
    for b in range(pred.shape[0]):
        n = cnt[b]
        assert torch.equal(d0[b, :n].view(torch.int32), d1[b, :n].view(torch.int32)), b
        assert torch.equal(a0[b, :n], a1[b, :n])
        assert torch.equal(r0["keep"][b, :n], r1["keep"][b, :n]) and not r1["keep"][b, n:].any()
        assert torch.equal(r0["mask_slot"][b, :n], r1["mask_slot"][b, :n]) and (r1["mask_slot"][b, n:] == -1).all()
    assert torch.equal(r0["keep_count"], r1["keep_count"])
    assert torch.equal(r0["slot_base"], r1["slot_base"])
    total = int(r0["slot_base"][-1].item())
    assert torch.equal(r0["slot_map"][:total], r1["slot_map"][:total])
    return total, cnt


def test_fused_finalize_equals_two_calls(ops):
    rng = np.random.default_rng(5)
    B, nc, nm, S = 6, 80, 32, 640
    head = synth.head(rng, B, nc, nm, S, objects=40)
    head[2, 4:4 + nc] *= 0.01                       # an image without candidates
    dense = synth.head(rng, 1, nc, nm, S, objects=150, cluster=(8, 20))
    head[4] = dense[0]
    p = dev(head)
    thr = np.full(nc, 0.3)
    thr[::3] = 0.55
    thr[7] = 1.0
    ragged = torch.tensor([[1080, 1920], [2160, 3840], [480, 640], [375, 500], [720, 1280], [640, 640]], dtype=torch.int32, device="cuda")
    for _ in range(2):                              # the same workspace twice
        total, cnt = assert_same_finalize(ops, p, 0.25, 0.45, nc, (S, S), (1080, 1920), 0.4)
        assert total > 50 and cnt[2] == 0
    assert_same_finalize(ops, p, 0.25, 0.45, nc, (S, S), ragged, dev(thr))
    assert_same_finalize(ops, p, 0.25, 0.6, nc, (S, S), (2160, 3840), 0.25, agnostic=True, max_det=50, pad_round=False)
    assert_same_finalize(ops, p, 0.25, 0.45, nc, (S, S), (1080, 1920), 0.4, split=True)
    assert_same_finalize(ops, p[2:3], 0.25, 0.45, nc, (S, S), (1080, 1920), 0.4)     # a batch with nothing at all
    # and against the oracle chain, image 0
    d, c, a, r = ops.nms_v8(p, 0.25, 0.45, nc=nc, finalize=dict(model_hw=(S, S), orig_hw=(1080, 1920), thres=0.4))
    rd, ra = ON.non_max_suppression(head[:1], 0.25, 0.45, nc=nc)[0]
    n = int(c[0].item())
    assert a[0, :n].cpu().tolist() == ra.tolist()
    ref = OB.scale_boxes((S, S), rd[:, :4], (1080, 1920, 3))
    assert np.array_equal(d[0, :n, :4].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert r["keep"][0, :n].cpu().numpy().astype(bool).tolist() == OB.class_filter(rd[:, 4], rd[:, 5], np.full(nc, 0.4)).tolist()


def test_fused_finalize_batch_above_1024(ops):
    """More images than the suppression kernel has lanes."""
    rng = np.random.default_rng(6)
    B, nc, A = 1100, 4, 64
    head = np.zeros((B, 4 + nc, A), F32)
    head[:, 0:2] = rng.random((B, 2, A), dtype=F32) * 64
    head[:, 2:4] = rng.uniform(4, 20, (B, 2, A)).astype(F32)
    head[:, 4:] = rng.random((B, nc, A), dtype=F32) * 0.6
    total, cnt = assert_same_finalize(ops, dev(head), 0.4, 0.45, nc, (64, 64), (48, 64), 0.5, max_det=16)
    assert total > B


@pytest.mark.parametrize("agnostic", [False, True])
def test_fp16_head_is_promoted_not_recomputed(ops, agnostic):
    rng = np.random.default_rng(8)
    B, nc, nm, S = 3, 80, 32, 640
    head16 = torch.from_numpy(synth.head(rng, B, nc, nm, S, objects=40)).cuda().half()
    d16, c16, a16 = ops.nms_v8(head16, 0.25, 0.45, agnostic=agnostic, nc=nc)
    d32, c32, a32 = ops.nms_v8(head16.float(), 0.25, 0.45, agnostic=agnostic, nc=nc)
    cnt = c32.cpu().numpy()
    assert np.array_equal(cnt, c16.cpu().numpy()) and cnt.min() > 5
    ref = ON.non_max_suppression(head16.float().cpu().numpy(), 0.25, 0.45, agnostic=agnostic, nc=nc)
    for b in range(B):
        n = cnt[b]
        assert torch.equal(d16[b, :n].view(torch.int32), d32[b, :n].view(torch.int32))
        assert torch.equal(a16[b, :n], a32[b, :n])
        # fp16 scores tie often: the oracle's stable order (score desc, anchor asc) is the kernels' order
        assert a16[b, :n].cpu().tolist() == ref[b][1].tolist()
    # odd anchor count: the scalar-load variant of the candidate filter
    odd = head16[:, :, :8399].contiguous()
    do, co, ao = ops.nms_v8(odd, 0.25, 0.45, agnostic=agnostic, nc=nc)
    dr, cr, ar = ops.nms_v8(odd.float(), 0.25, 0.45, agnostic=agnostic, nc=nc)
    assert torch.equal(co, cr)
    for b in range(B):
        assert torch.equal(ao[b, :int(cr[b])], ar[b, :int(cr[b])])
    # rotated path reads fp16 too
    obb = torch.from_numpy(synth.head(rng, 2, 15, 1, 640, objects=30)).cuda()
    obb[:, -1] = (torch.rand_like(obb[:, -1]) - 0.5) * 1.5
    h = obb.half()
    x, cx, ax = ops.nms_v8(h, 0.25, 0.45, nc=15, rotated=True)
    y, cy, ay = ops.nms_v8(h.float(), 0.25, 0.45, nc=15, rotated=True)
    assert torch.equal(cx, cy)
    for b in range(2):
        assert torch.equal(x[b, :int(cy[b])].view(torch.int32), y[b, :int(cy[b])].view(torch.int32))
