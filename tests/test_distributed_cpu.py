"""world_size-2 (and one world_size-8, uneven: 509 clips) gloo tests of the N>1 path.  The per-rank forward is replaced by a deterministic function of the clip
index (the HIP path needs a GPU); what is under test is the distributed plumbing itself -- contiguous sharding, the
preallocated all-gather of per-rank logits / compact frames (even and uneven shards), and ``distributed.run_sharded``,
the very loop ``bench.py --gpus N`` times (lane alternation, warm-up, exactly K timed steps, max over ranks)."""
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
    ok = ok and D.gather_floats(float(rank) + 0.5, w, "cpu") == [i + 0.5 for i in range(world)]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _spawn(target, n_ranks, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, n_ranks, port) + args + (q,)) for r in range(n_ranks)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    return sorted(res)


def test_two_rank_shard_and_gather():
    for n_total in (8, 7):
        assert _spawn(_worker, 2, n_total) == [(0, True), (1, True)]


def _sharded_worker(rank, world, port, n_total, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from svt_speechbrain_amd import distributed as D
    r, l, w = D.init_from_env(backend="gloo")
    lo, hi = D.shard_bounds(n_total, r, w)
    calls = {"lane": [], "ctx": []}

    # deterministic "forward" of this rank's shard: logits[c, t, j] = 1000 c + t + j / 32 + 1e6 * step parity of the lane
    def make_forward(lane):
        def fwd():
            calls["lane"].append(lane)
            c = torch.arange(lo, hi, dtype=torch.float32)[:, None, None]
            t = torch.arange(T, dtype=torch.float32)[None, :, None]
            j = torch.arange(20, dtype=torch.float32)[None, None, :]
            return 1000.0 * c + t + j / 32 + 1e6 * lane
        return fwd

    def make_frames(lane):  # the compact decoded frames (int32 x 4 per frame) of §8(e)
        def fwd():
            c = torch.arange(lo, hi, dtype=torch.int32)[:, None, None]
            return (c * 100 + torch.arange(T, dtype=torch.int32)[None, :, None] + torch.arange(4, dtype=torch.int32)[None, None, :]).contiguous()
        return fwd

    class Lane:  # stands for `torch.cuda.stream(s)`
        def __init__(self, i):
            self.i = i

        def __call__(self):
            return self

        def __enter__(self):
            calls["ctx"].append(self.i)

        def __exit__(self, *a):
            return False

    ok = True
    for kind, make, shape, dtype in (("logits", make_forward, (T, 20), torch.float32), ("frames", make_frames, (T, 4), torch.int32)):
        calls["lane"].clear(); calls["ctx"].clear()
        gs = [D.RowGatherer(n_total, w, r, shape, dtype, "cpu") for _ in range(2)]
        res = D.run_sharded([make(0), make(1)], n_total, r, w, steps=5, warmup=2, device="cpu", lanes=[Lane(0), Lane(1)], gatherers=gs)
        out = res["out"]
        ok = ok and out.shape == (n_total,) + shape and res["ranks"] == world and len(res["elapsed_per_rank"]) == world
        ok = ok and res["elapsed"] == max(res["elapsed_per_rank"]) and res["elapsed"] > 0
        ok = ok and calls["ctx"] == [0, 1, 0, 1, 0, 1, 0]          # 2 warm-up + exactly 5 timed steps, lanes alternate
        if kind == "logits":
            ok = ok and calls["lane"] == calls["ctx"]
            last_lane = calls["lane"][-1]
            want = 1000.0 * torch.arange(n_total, dtype=torch.float32)[:, None, None] + torch.arange(T, dtype=torch.float32)[None, :, None] \
                + torch.arange(20, dtype=torch.float32)[None, None, :] / 32 + 1e6 * last_lane
            ok = ok and torch.equal(out, want)
            ok = ok and out.data_ptr() == gs[last_lane].out.data_ptr()     # the preallocated buffer, no per-step allocation
        else:
            want = torch.arange(n_total, dtype=torch.int32)[:, None, None] * 100 + torch.arange(T, dtype=torch.int32)[None, :, None] \
                + torch.arange(4, dtype=torch.int32)[None, None, :]
            ok = ok and torch.equal(out, want)
    # a wrong shard shape is an error on the rank that produced it, not a hang in the collective
    try:
        D.RowGatherer(n_total, w, r, (T, 20), torch.float32, "cpu")(torch.zeros(hi - lo + 1, T, 20))
        ok = False
    except ValueError:
        pass
    D.barrier(w)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_run_sharded_two_ranks_uneven_and_even():
    for n_total in (7, 8):
        assert _spawn(_sharded_worker, 2, n_total, 6) == [(0, True), (1, True)]


def test_run_sharded_eight_ranks_uneven_509_clips():
    """SURVEY.md §8(e) at the node's real rank count, without the hardware: eight gloo ranks, 509 clips (five ranks own 64, three own
    63), the same run_sharded loop and preallocated gathers as `bench.py --gpus 8`; every gathered row is checked on every rank."""
    from svt_speechbrain_amd.distributed import shard_bounds
    sizes = [b - a for a, b in (shard_bounds(509, r, 8) for r in range(8))]
    assert sum(sizes) == 509 and sorted(set(sizes)) == [63, 64]
    assert _spawn(_worker, 8, 509) == [(r, True) for r in range(8)]
    assert _spawn(_sharded_worker, 8, 509, 3) == [(r, True) for r in range(8)]


def test_run_sharded_single_rank_needs_no_process_group():
    from svt_speechbrain_amd import distributed as D
    seen = []
    res = D.run_sharded([lambda: (seen.append(1), torch.ones(3, 2))[1]], 3, 0, 1, steps=4, warmup=1, device="cpu")
    assert len(seen) == 5 and res["ranks"] == 1 and res["out"].shape == (3, 2) and res["elapsed_per_rank"] == [res["elapsed_local"]]
