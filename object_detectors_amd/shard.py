"""Image sharding across the GPUs of a node and the end-of-run report gather.

Every stage of the path is per image, so the batch shards by image with NO collective on the hot
path: rank r of W owns a contiguous block of ceil(N/W) images.  The only communication is one
all_gather of a few counters per rank after the timed region (RCCL over xGMI with backend
'nccl', gloo on CPU for tests): latency-bound, a few hundred bytes.
"""
from __future__ import annotations

from typing import Dict, List

import torch


def shard_bounds(n_items: int, world: int, rank: int):
    """[start, stop) of rank's contiguous block; blocks are ceil(n/world) long, the tail may be short or empty."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    per = -(-n_items // world)
    start = min(rank * per, n_items)
    return start, min(start + per, n_items)


def checksum_indices(idx: torch.Tensor) -> int:
    """Order-sensitive 62-bit checksum of kept indices (position-weighted sum modulo a prime), used
    to compare shards' detections across runs / rank counts without gathering the detections."""
    if idx.numel() == 0:
        return 0
    v = idx.to(torch.int64).flatten().cpu()
    pos = torch.arange(1, v.numel() + 1, dtype=torch.int64)
    p = 2305843009213693951  # 2^61 - 1
    acc = 0
    for a, b in zip(v.tolist(), pos.tolist()):
        acc = (acc + (a + 1) * (b % 1000003 + 1)) % p
    return acc


def gather_report(local: Dict[str, float], device=None) -> List[Dict[str, float]]:
    """all_gather of a flat dict of numbers; returns one dict per rank (on every rank).  Without an
    initialised process group it returns [local]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [dict(local)]
    keys = sorted(local)
    t = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [dict(zip(keys, o.tolist())) for o in out]


def gather_vector(local: torch.Tensor, device=None) -> List[torch.Tensor]:
    """all_gather of a 1-D float64 tensor of equal length on every rank (e.g. per-block timings); returns the list of
    per-rank CPU tensors.  Without an initialised process group it returns [local]."""
    import torch.distributed as dist
    local = local.to(torch.float64).flatten()
    if not (dist.is_available() and dist.is_initialized()):
        return [local.cpu()]
    t = local.to(device) if device is not None else local
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu() for o in out]
