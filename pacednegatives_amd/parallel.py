"""Data-parallel helpers: one process per GPU, collectives through ``torch.distributed`` (backend
"nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The lceT5 step has exactly two exchange points per step (what Lightning's DDP strategy does for the
reference, ``train/train_lce.py:89-91``): the 1-float gradient of eta after the meta backward, and the
parameter gradients after the main backward.  Both are summed here and divided by the world size
inside the AdamW kernel (``grad_scale``), so each is ONE all-reduce over a contiguous fp32 buffer.
"""
from __future__ import annotations

import torch


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def world_size() -> int:
    d = _dist()
    return d.get_world_size() if d else 1


def rank() -> int:
    d = _dist()
    return d.get_rank() if d else 0


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (no-op for a single process)."""
    d = _dist()
    if d and d.get_world_size() > 1:
        d.all_reduce(t, op=d.ReduceOp.SUM)
    return t


def grad_scale() -> float:
    """Factor that turns a summed gradient into the mean over ranks."""
    return 1.0 / world_size()


def shard_indices(n_items: int, rank_: int, world: int):
    """Items of rank r under a non-shuffling distributed sampler: r, r+W, ... (tail dropped so that every
    rank sees the same number of items)."""
    usable = n_items - n_items % world
    return list(range(rank_, usable, world))
