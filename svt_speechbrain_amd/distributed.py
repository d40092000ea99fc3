"""Multi-GPU driver for the forward path: clips are independent units, so the utterance
batch is split contiguously over ranks (one process per GPU), every rank runs the reference semantics on
its own shard (exactly what ``nn.DataParallel`` / DDP do in the reference, ``speechbrain/core.py:1150-1169``:
the two whole-batch layer norms are per device shard), and the only collective is one all-gather of the
final logits over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU for tests."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """-> (rank, local_rank, world_size); initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Static contiguous split; the first ``n_items % world`` ranks take one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local: torch.Tensor, n_total: int, world: int) -> torch.Tensor:
    """Gather per-rank (n_r, ...) shards (possibly uneven) into (n_total, ...) on every rank."""
    if world == 1:
        return local
    base, rem = divmod(n_total, world)
    cap = base + (1 if rem else 0)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(out[r * cap: r * cap + (hi - lo)])
    return torch.cat(parts, dim=0)


def max_over_ranks(value: float, world: int, device) -> float:
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(world: int) -> None:
    if world > 1:
        dist.barrier()
