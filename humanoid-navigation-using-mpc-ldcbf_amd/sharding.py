"""Batch sharding over the GPUs of a node: every (state, goal, obstacle-set) instance is
independent, so rank r owns a contiguous slice and there is no data-path collective.  The only
collective is an all-gather of a three-number counter record after the timed region (RCCL on GPUs,
gloo in the CPU tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of rank ``rank``; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# how the last gather_counters call of this process moved its record (bench.py prints it; the RCCL regression test reads it)
last_gather = {"collective": None, "backend": None, "device": None, "world": 1}


def gather_counters(elapsed_s: float, n_problems: int, n_solved: int, device=None):
    """All ranks -> (max elapsed, total problems, total solved, per-rank table [world,3])."""
    rec = torch.tensor([float(elapsed_s), float(n_problems), float(n_solved)], dtype=torch.float64,
                       device=device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized():
        parts = [torch.zeros_like(rec) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, rec)
        table = torch.stack(parts).cpu()
        last_gather.update(collective="all_gather", backend=str(dist.get_backend()), device=str(rec.device),
                           world=dist.get_world_size())
    else:
        table = rec.cpu()[None, :]
        last_gather.update(collective=None, backend=None, device=str(rec.device), world=1)
    return float(table[:, 0].max()), int(table[:, 1].sum()), int(table[:, 2].sum()), table
