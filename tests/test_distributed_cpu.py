"""world_size-2 gloo test of the N>1 path: contiguous sharding + all-gather of per-rank logits (uneven
shards included).  The per-rank forward is replaced by a deterministic function of the clip index, since the
HIP path needs a GPU; what is under test is the distributed plumbing bench.py uses."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from svt_speechbrain_amd import distributed as D
    r, l, w = D.init_from_env(backend="gloo")
    lo, hi = D.shard_bounds(n_total, r, w)
    local = torch.stack([torch.full((5, 20), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 5, 20)
    full = D.all_gather_rows(local, n_total, w)
    D.barrier(w)
    t = D.max_over_ranks(float(rank + 1), w, "cpu")
    ok = full.shape == (n_total, 5, 20) and all(float(full[i, 0, 0]) == float(i) for i in range(n_total)) and t == float(world)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    for n_total in (8, 7):
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
        for p in ps:
            p.start()
        res = [q.get(timeout=120) for _ in ps]
        for p in ps:
            p.join(timeout=60)
        assert sorted(res) == [(0, True), (1, True)]
