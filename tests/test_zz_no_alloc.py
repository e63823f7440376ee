"""lmi_nms_v8_static -- the entry point that allocates nothing (its workspace is a device array built into the
library) -- against lmi_nms_v8 with a caller-allocated workspace, through the C ABI: identical rows, counts and
kept anchor indices, on calls in a row that share the built-in array (the second with a smaller input, the third
larger again, so stale contents of a previous call are in place each time)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32 = np.float32


def run_both(L, ops, check, head, conf, iou, nc, agnostic, max_det=300):
    pred = torch.from_numpy(head).cuda()
    b, ch, a = pred.shape
    nm = ch - 4 - nc
    ref_det, ref_count, ref_anchor = ops.nms_v8(pred, conf, iou, agnostic=agnostic, max_det=max_det, nc=nc)
    det = torch.full((b, max_det, 6 + nm), -7.0, dtype=torch.float32, device="cuda")
    count = torch.full((b,), -1, dtype=torch.int32, device="cuda")
    anchor = torch.full((b, max_det), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    # "allocates nothing", pinned at the driver: torch's own counter only sees torch's caching allocator, a hipMalloc
    # inside the library would not move it.  The per-device set-up (LDS limits, the array's address) ran with the
    # reference call above.
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    check(L.lmi_nms_v8_static(pred.data_ptr(), b, nc, nm, a, conf, iou, int(agnostic), max_det, 30000, 7680.0,
                              det.data_ptr(), count.data_ptr(), anchor.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before and torch.cuda.mem_get_info()[0] == free_before
    rc, c = ref_count.cpu().numpy(), count.cpu().numpy()
    assert np.array_equal(rc, c) and c.sum() > 0
    for i in range(b):
        assert torch.equal(det[i, :c[i]].view(torch.int32), ref_det[i, :c[i]].view(torch.int32)), i
        assert torch.equal(anchor[i, :c[i]], ref_anchor[i, :c[i]]), i
        assert (det[i, c[i]:] == -7.0).all()          # rows beyond the count stay untouched, as in lmi_nms_v8
    return int(c.sum())


def test_nms_static_matches_nms_v8_on_consecutive_calls():
    assert torch.cuda.is_available(), "this test needs the GPU"
    from object_detectors_amd import _lib, hip_ops as ops, synth
    L = _lib.load()
    big = synth.head(np.random.default_rng(71), 4, 80, 32, 640, objects=40)        # (4, 116, 8400)
    small = synth.head(np.random.default_rng(72), 2, 80, 32, 320, objects=12)      # (2, 116, 2100)
    dense = synth.head(np.random.default_rng(73), 3, 80, 0, 640, objects=120, cluster=(8, 20))
    n1 = run_both(L, ops, _lib.check, big, 0.25, 0.45, 80, False)
    n2 = run_both(L, ops, _lib.check, small, 0.25, 0.45, 80, False)               # smaller input on the same array
    n3 = run_both(L, ops, _lib.check, dense, 0.25, 0.6, 80, True)                 # agnostic path, larger again
    n4 = run_both(L, ops, _lib.check, big, 0.3, 0.45, 80, False, max_det=20)
    assert n1 > n2 > 0 and n3 > 0 and n4 == 80
    # under HIP graph capture, where an allocation is an error: captured once, replayed on fresh contents
    pred = torch.from_numpy(small).cuda()
    ref = ops.nms_v8(pred, 0.25, 0.45, nc=80)
    det = torch.empty((2, 300, 38), dtype=torch.float32, device="cuda")
    count = torch.empty((2,), dtype=torch.int32, device="cuda")
    anchor = torch.empty((2, 300), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(L.lmi_nms_v8_static(pred.data_ptr(), 2, 80, 32, 2100, 0.25, 0.45, 0, 300, 30000, 7680.0, det.data_ptr(),
                                       count.data_ptr(), anchor.data_ptr(), torch.cuda.current_stream().cuda_stream))
    det.fill_(-1.0); count.fill_(-1)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(count, ref[1])
    for i, n in enumerate(count.tolist()):
        assert torch.equal(det[i, :n], ref[0][i, :n]) and torch.equal(anchor[i, :n], ref[2][i, :n])
    # beyond the built-in workspace: rejected before anything is launched
    pred = torch.zeros((12, 84, 8400), device="cuda")
    out = torch.zeros((12, 300, 6), device="cuda")
    cnt = torch.zeros((12,), dtype=torch.int32, device="cuda")
    anc = torch.zeros((12, 300), dtype=torch.int32, device="cuda")
    rc = L.lmi_nms_v8_static(pred.data_ptr(), 12, 80, 0, 8400, 0.25, 0.45, 0, 300, 30000, 7680.0, out.data_ptr(), cnt.data_ptr(),
                             anc.data_ptr(), None)
    assert rc == -4 and b"built-in one holds" in L.lmi_last_error()
    assert L.lmi_nms_workspace_bytes(11, 8400) <= _lib.LMI_NMS_STATIC_BYTES < L.lmi_nms_workspace_bytes(12, 8400)
