"""User-batch data parallelism: one process per GPU, a full weight replica per
GPU, users sharded across ranks.  Lanes never interact in the forward (model_hier.py:39-93 is
per-row of B), so the evaluation path has exactly one collective: a sum all-reduce of the 8 metric
accumulators (lane-mean SUMS + user count + valid steps, include/hiertcn.h `hiertcn_metrics`).
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, device: torch.device | None = None) -> Tuple[int, int, int]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run) -> (rank, world, local_rank).
    world == 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def user_shard(n_users: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard of the user axis: (first_user, n_users_of_rank).  The first
    n_users % world ranks hold one user more."""
    base, extra = divmod(int(n_users), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def all_reduce_metrics(m8: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of the 8 accumulators; afterwards out[k] / out[6] is the metric
    over the GLOBAL batch (a mean over all lanes of all ranks, model.py:116-117 / loss.py:215-219)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if m8.is_cuda and dist.get_backend() == "gloo":   # (tests / smoke runs: gloo reduces host tensors)
            host = m8.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            m8.copy_(host)
        else:
            dist.all_reduce(m8, op=dist.ReduceOp.SUM)
    return m8


def max_over_ranks(seconds: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_gradients(flat: torch.Tensor, bucket_bytes: int = 256 << 20) -> int:
    """Sum the flat fp32 gradient arena over ranks, in place, in buckets (so that a bucket's RCCL all-reduce can
    overlap the next bucket's launch; the arena is dominated by the two [N, d] item tables -- 2 x 307 MB at
    cfg-3).  Returns the world size: the caller divides by it (equal user counts per rank: weak scaling), which
    makes the update the gradient of the GLOBAL batch's lane-mean loss (model.py:116-117)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return 1
    n = max(1, bucket_bytes // 4)
    works = [dist.all_reduce(flat[i:i + n], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, flat.numel(), n)]
    for wk in works:
        wk.wait()
    return dist.get_world_size()
