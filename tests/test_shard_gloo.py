"""N>1 path on CPU: two gloo ranks shard a set of images, each runs the (oracle) post-processing
on its own block with no communication, then the end-of-run gather reproduces the single-process
totals.  Mirrors what bench.py does over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_detectors_amd import shard, synth
from oracle import nms as ON


def test_shard_bounds_cover_everything():
    for n, w in [(8192, 1), (8192, 2), (8192, 8), (128, 8), (10, 4), (3, 8), (0, 2)]:
        spans = [shard.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) == -(-n // w) if n else True
    with pytest.raises(ValueError):
        shard.shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _work(head, lo, hi):
    kept, chk = 0, 0
    for b in range(lo, hi):
        (det, idx), = ON.non_max_suppression(head[b:b + 1], 0.25, 0.45, nc=8)
        kept += len(idx)
        chk = (chk + shard.checksum_indices(torch.from_numpy(idx)) * (b + 1)) % (2**61 - 1)
    return kept, chk


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    head = synth.head(np.random.default_rng(4), 6, 8, 0, 160, objects=6)
    lo, hi = shard.shard_bounds(head.shape[0], world, rank)
    kept, chk = _work(head, lo, hi)
    rep = shard.gather_report({"images": hi - lo, "kept": kept, "chk_hi": chk >> 31, "chk_lo": chk & (2**31 - 1), "t": 0.1 * (rank + 1)})
    # per-block timings as bench.py gathers them: the slowest rank per block, then the median block
    blocks = shard.gather_vector(torch.tensor([1.0, 3.0, 2.0]) * (rank + 1))
    rep[0]["median_of_max"] = float(torch.stack(blocks).max(dim=0).values.median())
    dist.barrier()
    if rank == 0:
        q.put(rep)
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rep = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    head = synth.head(np.random.default_rng(4), 6, 8, 0, 160, objects=6)
    kept, chk = _work(head, 0, 6)
    assert len(rep) == 2 and sum(r["images"] for r in rep) == 6
    assert rep[0].pop("median_of_max") == 4.0      # max over ranks = [2, 6, 4] -> median 4
    assert sum(r["kept"] for r in rep) == kept and kept > 0
    total = sum((int(r["chk_hi"]) << 31) | int(r["chk_lo"]) for r in rep) % (2**61 - 1)
    assert total == chk
    assert max(r["t"] for r in rep) == pytest.approx(0.2)


def test_gather_report_without_group():
    assert shard.gather_report({"a": 1.0}) == [{"a": 1.0}]


def test_gather_vector_without_a_process_group():
    out = shard.gather_vector(torch.tensor([0.5, 0.25]))
    assert len(out) == 1 and out[0].dtype == torch.float64 and out[0].tolist() == [0.5, 0.25]


def test_roofline_byte_counts():
    from object_detectors_amd import roofline as R
    assert R.letterbox_bytes(1080, 1920, 640, 640) == 7_680_000 == R.letterbox_bytes(2160, 3840, 640, 640)
    assert R.letterbox_row_granular_bytes(2160, 3840, 640, 640) == 720 * 11520 + 4_915_200
    # exact upsample bytes: the whole mask once + only the logits rows x columns under the box
    whole = R.mask_upsample_bytes(1, 1080, 1920, 160, 160)
    assert whole == 1080 * 1920 + 90 * 160 * 4
    full = R.mask_upsample_bytes_exact(np.array([[0.0, 0.0, 1920.0, 1080.0]]), 1080, 1920, 160, 160)
    assert full == whole
    small = R.mask_upsample_bytes_exact(np.array([[100.0, 100.0, 400.0, 300.0]]), 1080, 1920, 160, 160)
    assert 1080 * 1920 < small < 1080 * 1920 + 30 * 30 * 4
    assert R.mask_upsample_bytes_exact(np.array([[5.0, 5.0, 5.0, 9.0]]), 1080, 1920, 160, 160) == 1080 * 1920   # empty box: zeros only
