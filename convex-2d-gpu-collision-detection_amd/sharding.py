"""Multi-GPU sharding of the hot path: one process per GPU, contiguous ranges, one reduce.

Every pair, scene and Monte-Carlo sample is independent, and the random
stream is keyed by (seed, scene, sample), so a rank only needs to know WHICH units are its
own; the only exchange is a sum of a few 64-bit counters (hit counts, sample counts) — one
all-reduce (RCCL on GPUs via torch.distributed's "nccl" backend, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Sequence, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` units for `rank` of `world`; the first
    total % world ranks take one extra unit; the union over ranks is [0, total)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def all_reduce_counters(values: Sequence[int], device=None):
    """Sum a few non-negative 64-bit counters over all ranks (a single all-reduce).
    Returns a list of Python ints.  Without an initialised process group it is the identity."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def max_over_ranks(seconds: float, device=None) -> float:
    """Slowest rank's time (the contract's whole-job time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return float(seconds)
    if dist.get_backend() == "gloo":
        device = None  # rehearsal backend reduces on the host
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
