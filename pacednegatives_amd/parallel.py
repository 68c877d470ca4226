"""Data-parallel helpers: one process per GPU, collectives through ``torch.distributed`` (backend
"nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The lceT5 step exchanges what Lightning's DDP strategy exchanges for the reference
(``train/train_lce.py:89-91``): the 1-float gradient of eta after the meta backward, and the parameter
gradients of the main backward.  Both are summed here and divided by the world size inside the AdamW
kernel (``grad_scale``).  The parameter gradients live in one contiguous fp32 buffer: the first step
exchanges it in ONE all-reduce after the backward pass, later steps in contiguous ranges as their
gradients become final (``LCEModel.training_step``).
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


def _host_staged(d) -> bool:
    """gloo (CPU tests, rehearsals with several ranks on one GPU) is given host tensors only: its paths for device
    tensors are little used upstream, and one such call hung here.  RCCL works on device memory directly."""
    return d.get_backend() != "nccl"


def exchange_forced() -> bool:
    """PACED_FORCE_EXCHANGE=1: a process group of ONE rank still issues every all-reduce, and the step takes the
    schedule of several ranks.  A rehearsal of the RCCL call sites (stream order, ranges of the gradient buffer as
    collective operands) on a single GPU; the sums are those of one rank."""
    import os
    return os.environ.get("PACED_FORCE_EXCHANGE") == "1" and _dist() is not None


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (no-op for a single process)."""
    d = _dist()
    if d and (d.get_world_size() > 1 or exchange_forced()):
        if t.is_cuda and _host_staged(d):
            host = t.detach().cpu()              # also orders the exchange after the kernels that produced t
            d.all_reduce(host, op=d.ReduceOp.SUM)
            t.copy_(host)
        else:
            d.all_reduce(t, op=d.ReduceOp.SUM)
    return t


def max_over_ranks(value: float, device=None) -> float:
    """Largest ``value`` of any rank (the step time a multi-rank benchmark reports)."""
    d = _dist()
    if not d or d.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if _host_staged(d) or device is None else device)
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())


def all_gather_stack(t: torch.Tensor) -> torch.Tensor:
    """[W, *t.shape]: every rank's ``t`` (same shape and dtype on all ranks), on ``t``'s device.

    RCCL gathers straight into one preallocated device buffer.  The gloo backend (CPU tests, and rehearsals in which
    several ranks share one GPU) is given host tensors: the device tensor is copied to the host first, which also
    orders the exchange after the kernels that produce ``t`` on the calling stream."""
    d = _dist()
    if not d or d.get_world_size() == 1:
        return t.unsqueeze(0)
    W = d.get_world_size()
    t = t.contiguous()
    if not _host_staged(d):
        out = torch.empty((W,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        d.all_gather_into_tensor(out, t)
        return out
    host = t.detach().cpu()
    parts = [torch.empty_like(host) for _ in range(W)]
    d.all_gather(parts, host)
    return torch.stack(parts, 0).to(t.device)


def grad_scale() -> float:
    """Factor that turns a summed gradient into the mean over ranks."""
    return 1.0 / world_size()


def shard_indices(n_items: int, rank_: int, world: int):
    """Items of rank r under ``torch.utils.data.DistributedSampler(shuffle=False, drop_last=False)``, the sampler Lightning
    puts in front of the reference's sequential DataLoader when ``strategy='ddp'`` (``train/train_lce.py:89-91``,
    ``lceT5.py:66-67``): the index list 0 .. n-1 is padded with its own head up to a multiple of the world size, then rank r
    takes positions r, r+W, ...  Every rank sees ceil(n / W) items; with n % W != 0 the first items are seen twice."""
    if world <= 1:
        return list(range(n_items))
    if n_items == 0:
        return []
    per_rank = -(-n_items // world)
    order = list(range(n_items))
    pad = per_rank * world - n_items
    while pad > 0:                       # (a list shorter than the padding is repeated, as the sampler does)
        order += order[:pad]
        pad = per_rank * world - len(order)
    return order[rank_:per_rank * world:world]
