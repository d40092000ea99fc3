"""Multi-GPU driver for the forward path: clips are independent units, so the utterance
batch is split contiguously over ranks (one process per GPU), every rank runs the reference semantics on
its own shard (exactly what ``nn.DataParallel`` / DDP do in the reference, ``speechbrain/core.py:1150-1169``:
the two whole-batch layer norms are per device shard), and the only collective is one all-gather of the
final logits -- or of the compact decoded frames, 16 bytes per frame instead of 80 -- over RCCL/xGMI
(backend "nccl" on ROCm) or gloo on CPU for tests.

``run_sharded`` is the step / shard / gather loop itself (warm-up, barrier + device sync on both sides of exactly
K timed steps, max over ranks): ``bench.py`` times the GPU path with it, ``tests/test_distributed_cpu.py`` runs the same
function on two gloo ranks with a deterministic CPU forward, so the N > 1 control flow is exercised without GPUs."""
from __future__ import annotations

import contextlib
import os
import time
from typing import Callable, List, Optional, Sequence, Tuple

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
        if backend is None:   # SVT_DIST_BACKEND=gloo: test mode for one-GPU boxes (collectives staged through the host, see _gather_into)
            backend = os.environ.get("SVT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Static contiguous split; the first ``n_items % world`` ranks take one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _host_staged() -> bool:
    """gloo has no all-gather for device tensors: with it (CPU tests, or several ranks sharing ONE GPU in the GPU test of the
    N > 1 control flow) device tensors go through the host.  RCCL (`nccl`) takes the device tensors as they are."""
    return dist.get_backend() == "gloo"


def _gather_into(out: torch.Tensor, inp: torch.Tensor) -> None:
    if inp.is_cuda and _host_staged():
        tmp = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(tmp, inp.cpu())
        out.copy_(tmp)
    else:
        dist.all_gather_into_tensor(out, inp)


class RowGatherer:
    """All-gather of per-rank ``(n_r, *row_shape)`` shards (possibly uneven) into ``(n_total, *row_shape)`` on every rank,
    with every buffer allocated ONCE: a timed step allocates nothing.  Even split: one ``all_gather_into_tensor`` straight
    from the caller's tensor into the result.  Uneven split: shards are padded to the largest one and the result is
    compacted with a precomputed row index."""

    def __init__(self, n_total: int, world: int, rank: int, row_shape: Sequence[int], dtype: torch.dtype, device):
        self.n_total, self.world, self.rank = int(n_total), int(world), int(rank)
        self.row_shape = tuple(int(d) for d in row_shape)
        self.lo, self.hi = shard_bounds(n_total, rank, world)
        base, rem = divmod(n_total, world)
        self.cap = base + (1 if rem else 0)
        self.even = rem == 0
        self.pad = self.idx = self.raw = None
        if world > 1:
            self.out = torch.empty((n_total,) + self.row_shape, dtype=dtype, device=device)
            if not self.even:
                self.pad = torch.zeros((self.cap,) + self.row_shape, dtype=dtype, device=device)
                self.raw = torch.empty((world * self.cap,) + self.row_shape, dtype=dtype, device=device)
                idx = []
                for r in range(world):
                    lo, hi = shard_bounds(n_total, r, world)
                    idx.extend(range(r * self.cap, r * self.cap + (hi - lo)))
                self.idx = torch.tensor(idx, dtype=torch.long, device=device)

    def bytes_per_rank(self) -> int:
        n = self.cap
        for d in self.row_shape:
            n *= d
        return n * torch.empty((), dtype=self.out.dtype).element_size() if self.world > 1 else 0

    def __call__(self, local: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return local
        if tuple(local.shape) != (self.hi - self.lo,) + self.row_shape:
            raise ValueError(f"rank {self.rank}: expected a {(self.hi - self.lo,) + self.row_shape} shard, got {tuple(local.shape)}")
        if self.even:
            _gather_into(self.out, local.contiguous())
            return self.out
        self.pad[: local.shape[0]].copy_(local)
        _gather_into(self.raw, self.pad)
        torch.index_select(self.raw, 0, self.idx, out=self.out)
        return self.out


def all_gather_rows(local: torch.Tensor, n_total: int, world: int) -> torch.Tensor:
    """One-off form of ``RowGatherer`` (allocates its buffers on every call: use the class inside a loop)."""
    if world == 1:
        return local
    g = RowGatherer(n_total, world, dist.get_rank(), local.shape[1:], local.dtype, local.device)
    return g(local)


def max_over_ranks(value: float, world: int, device) -> float:
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, world: int, device) -> List[float]:
    """value of every rank, in rank order, on every rank."""
    if world == 1:
        return [float(value)]
    dev = "cpu" if _host_staged() else device
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    out = torch.empty(world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.tolist()]


def barrier(world: int) -> None:
    if world > 1:
        dist.barrier()


def run_sharded(forward_fns: Sequence[Callable[[], torch.Tensor]], n_total: int, rank: int, world: int, steps: int,
                warmup: int, device, lanes: Optional[Sequence] = None, gatherers: Optional[Sequence[RowGatherer]] = None,
                sync: Optional[Callable[[], None]] = None, clock: Callable[[], float] = time.perf_counter) -> dict:
    """The sharded step loop.

    ``forward_fns[i]()`` runs ONE step of this rank's shard on lane ``i`` (its own stream, encoder object and workspace)
    and returns the rows to gather (``(n_local, ...)``: logits or decoded frames).  Successive steps go round-robin over
    the lanes; lane ``i``'s step is issued inside ``lanes[i]`` (a context manager factory such as
    ``lambda: torch.cuda.stream(s)``; None = no context) and its all-gather -- ``gatherers[i]``, one preallocated
    ``RowGatherer`` per lane because two steps are in flight -- is enqueued from the same context, so the collectives are
    issued in the same order on every rank.  ``warmup`` untimed steps, then a barrier + ``sync()`` (device
    synchronisation), EXACTLY ``steps`` timed steps, ``sync()`` + barrier, and the MAX of the elapsed time over ranks.

    Returns ``{"elapsed": max over ranks, "elapsed_local", "elapsed_per_rank": [...], "out": last gathered tensor,
    "ranks": dist.get_world_size() as seen after init}``."""
    nl = len(forward_fns)
    if nl < 1:
        raise ValueError("run_sharded needs at least one lane")
    if lanes is None:
        lanes = [None] * nl
    if gatherers is None:
        gatherers = [None] * nl
    if len(lanes) != nl or len(gatherers) != nl:
        raise ValueError("one lane context and one gatherer per forward function")
    if sync is None:
        def sync():
            return None
    counter = 0
    out = None

    def step():
        nonlocal counter, out
        i = counter % nl
        counter += 1
        with (lanes[i]() if lanes[i] is not None else contextlib.nullcontext()):
            rows = forward_fns[i]()
            out = gatherers[i](rows) if (gatherers[i] is not None and world > 1) else rows
        return out

    for _ in range(warmup):
        step()
    sync()
    barrier(world)
    sync()
    t0 = clock()
    for _ in range(steps):
        step()
    sync()
    barrier(world)
    local_elapsed = clock() - t0
    per_rank = gather_floats(local_elapsed, world, device)
    return {"elapsed": max(per_rank), "elapsed_local": local_elapsed, "elapsed_per_rank": per_rank, "out": out,
            "ranks": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1, "step": step}
