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


def gather_rows(values: Sequence[float], device=None):
    """Every rank's row of a few doubles, in rank order: ONE all_gather (always issued AFTER a timed region, never inside one).
    Without an initialised process group it is the one row of this process."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [[float(v) for v in values]]
    if dist.get_backend() == "gloo":
        device = None  # rehearsal backend gathers on the host
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    rows = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, t)
    return [r.tolist() for r in rows]


def kernel_time_spread(kernel_ms: float, units: float, frac=None, work=None, device=None):
    """What makes an N > 1 bench line readable: kernel_ms = THIS rank's HIP-event time of one launch (or of a leg's one
    device loop), units = what that launch processed on this rank, work = what its roofline counts when that is not the unit
    (the drawn samples of an adaptive loop), frac = this rank's roofline fraction (or None).

    Returns (spread, kernels_only, frac_slowest_rank):
      spread            {min, median, max, ranks, slowest_rank} of kernel_ms over ranks,
      kernels_only      all ranks' units per launch / the SLOWEST rank's launch time, units per second,
      frac_slowest_rank the roofline fraction of the slowest rank: this rank's `frac` rescaled by the two ranks' work and time
                        (a curve is bounded by the slowest rank; `frac` in the line is rank 0's), None without `frac`."""
    import statistics

    import torch.distributed as dist

    rows = gather_rows([kernel_ms, units, units if work is None else work], device)
    me = rows[dist.get_rank()] if (dist.is_available() and dist.is_initialized()) else rows[0]
    ms = [r[0] for r in rows]
    slow = max(range(len(ms)), key=lambda k: ms[k])
    spread = {"min": round(min(ms), 5), "median": round(statistics.median(ms), 5), "max": round(ms[slow], 5), "ranks": len(rows), "slowest_rank": slow}
    kernels_only = sum(r[1] for r in rows) / (ms[slow] * 1e-3) if ms[slow] > 0 else None
    frac_slowest = None
    if frac is not None and me[2] and ms[slow] > 0:
        frac_slowest = round(frac * (rows[slow][2] / me[2]) * (me[0] / ms[slow]), 4)
    return spread, kernels_only, frac_slowest
