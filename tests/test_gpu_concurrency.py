"""The header's claim (include/lmi_hip.h, Conventions): the library keeps no per-call global state, so calls on different
streams -- from different host threads -- are independent.  Two host threads, each with its own torch stream (hence its own
workspaces in hip_ops), run the whole path side by side on different inputs: letterbox (uniform and ragged: the descriptor
cache is shared) -> nms_v8 with the fused finalize (per class and agnostic: the suppressor-list kernels use the workspace
hardest) -> process_mask; every result must be bit-equal to the same calls run back to back on one stream.  Then
lmi_nms_v8_static, whose one built-in workspace per GPU makes overlapping calls race by contract: two threads that
serialise their calls on ONE stream must still get the right rows.  One process, run once."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from object_detectors_amd import synth


def _inputs(seed, B, h, w, S, nc, nm, objects):
    rng = np.random.default_rng(seed)
    dev = lambda a: torch.from_numpy(a).cuda()
    frames = dev(synth.frames_u8(rng, B, h, w))
    ragged = [dev(synth.frames_u8(rng, 1, h - 7 * i, w - 11 * i)[0]) for i in range(3)]
    head = dev(synth.head(rng, B, nc, nm, S, objects=objects))
    proto = dev(synth.proto(rng, B, nm, S // 4, S // 4))
    return frames, ragged, head, proto


def _path(ops, inp, S, nc, h, w, agnostic):
    frames, ragged, head, proto = inp
    lb, _ = ops.letterbox(frames, (S, S), pad_value=114, swap_rb=True)
    lr, _ = ops.letterbox(ragged, (S, S), pad_value=7)
    det, count, anchor, fin = ops.nms_v8(head, 0.25, 0.45, nc=nc, agnostic=agnostic,
                                         finalize=dict(model_hw=(S, S), orig_hw=(h, w), thres=0.3))
    cap = det.shape[0] * det.shape[1]
    masks, ovf = ops.process_mask(proto, det, fin["slot_map"], fin["slot_base"], (h, w), cap)
    return lb, lr, det, count, anchor, fin["keep"], fin["slot_base"], masks


def _same(a, b, tag):
    lb, lr, det, count, anchor, keep, base, masks = a
    lb2, lr2, det2, count2, anchor2, keep2, base2, masks2 = b
    assert torch.equal(lb, lb2) and torch.equal(lr, lr2), tag
    assert torch.equal(count, count2) and torch.equal(base, base2), tag
    total = int(base[-1])
    for i, n in enumerate(count.tolist()):
        assert torch.equal(det[i, :n], det2[i, :n]) and torch.equal(anchor[i, :n], anchor2[i, :n]) and torch.equal(keep[i, :n], keep2[i, :n]), (tag, i)
    assert total > 0 and torch.equal(masks[:total], masks2[:total]), tag


def test_two_threads_two_streams_whole_path():
    from object_detectors_amd import hip_ops as ops
    S, nc, nm, B, h, w = 320, 6, 32, 5, 270, 480
    jobs = {0: [(_inputs(100 + k, B, h, w, S, nc, nm, 12 + k), k % 2 == 0) for k in range(4)],
            1: [(_inputs(200 + k, B, h - 30, w - 40, S, nc, nm, 25 + k), k % 2 == 1) for k in range(4)]}
    dims = {0: (h, w), 1: (h - 30, w - 40)}
    ref = {t: [_path(ops, inp, S, nc, *dims[t], agn) for inp, agn in jobs[t]] for t in jobs}
    torch.cuda.synchronize()
    got, errs = {0: [], 1: []}, []
    gate = threading.Barrier(2)

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            gate.wait()
            with torch.cuda.stream(stream):
                for rep in range(6):                       # the streams' kernels interleave on the chip for a while
                    for inp, agn in jobs[t]:
                        out = _path(ops, inp, S, nc, *dims[t], agn)
                        if rep == 5:
                            got[t].append(out)
            stream.synchronize()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not errs, errs
    for t in jobs:
        assert len(got[t]) == len(ref[t])
        for k, (a, b) in enumerate(zip(got[t], ref[t])):
            _same(a, b, (t, k))


def test_static_nms_entry_from_two_threads_serialised_on_one_stream():
    from object_detectors_amd import _lib
    L = _lib.load()
    nc, A = 5, 2100
    heads = [torch.from_numpy(synth.head(np.random.default_rng(300 + k), 2, nc, 0, 320, objects=10 + 3 * k)).cuda() for k in range(4)]

    def call(p, static):
        det = torch.empty((2, 300, 6), dtype=torch.float32, device="cuda")
        count = torch.empty((2,), dtype=torch.int32, device="cuda")
        anchor = torch.empty((2, 300), dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        if static:
            _lib.check(L.lmi_nms_v8_static(p.data_ptr(), 2, nc, 0, A, 0.25, 0.45, 0, 300, 30000, 7680.0, det.data_ptr(), count.data_ptr(),
                                           anchor.data_ptr(), st))
        else:
            ws = torch.empty((int(L.lmi_nms_workspace_bytes(2, A)),), dtype=torch.uint8, device="cuda")
            _lib.check(L.lmi_nms_v8(p.data_ptr(), 2, nc, 0, A, 0.25, 0.45, 0, 300, 30000, 7680.0, det.data_ptr(), count.data_ptr(),
                                    anchor.data_ptr(), ws.data_ptr(), ws.numel(), st))
        return det, count, anchor

    ref = [call(p, False) for p in heads]
    torch.cuda.synchronize()
    shared = torch.cuda.Stream()
    lock, got, errs = threading.Lock(), {}, []

    def worker(t):
        try:
            for rep in range(8):
                for k in range(t, 4, 2):
                    with lock, torch.cuda.stream(shared):   # one call at a time, all on one stream: the contract of the static entry
                        got[(k, rep)] = call(heads[k], True)
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in (0, 1)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    shared.synchronize()
    assert not errs, errs
    for (k, rep), (det, count, anchor) in got.items():
        rd, rc, ra = ref[k]
        assert torch.equal(count, rc), (k, rep)
        for i, n in enumerate(rc.tolist()):
            assert torch.equal(det[i, :n], rd[i, :n]) and torch.equal(anchor[i, :n], ra[i, :n]), (k, rep, i)
