"""RCCL sanity on the one GPU a test box has: a single-rank `nccl` process group (torch.distributed's nccl backend IS RCCL on
ROCm) through the very calls `distributed.run_sharded` / `bench.py` make at N > 1 -- `all_gather_into_tensor` of logits and of
compact frames from a side stream into preallocated buffers, `all_reduce(MAX)` of the elapsed time, `all_gather` of per-rank
rates, `barrier` -- so that communicator creation, the collective kernels and the stream ordering are exercised before the first
multi-GPU run.  The multi-rank data path itself (shards, uneven splits, ordering across lanes) is covered by the 2-rank gloo tests
(tests/test_distributed_cpu.py).  Runs in a child process: a process group is process-wide state."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SVT_ROOT"])
from svt_speechbrain_amd import distributed as D
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
logits = torch.randn(32, 499, 20, device=dev)
frames = torch.randint(0, 100, (32 * 499, 4), dtype=torch.int32, device=dev)
out_l = torch.empty_like(logits)
out_f = torch.empty_like(frames)
for _ in range(3):
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
        dist.all_gather_into_tensor(out_l, logits)
        dist.all_gather_into_tensor(out_f, frames)
torch.cuda.synchronize()
assert torch.equal(out_l, logits) and torch.equal(out_f, frames)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
g = torch.empty(1, dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(g, t)
assert g.tolist() == [1.25]
dist.barrier()
# the package's helpers at world 1 (no collective) still agree
assert D.gather_floats(2.5, 1, dev) == [2.5] and D.max_over_ranks(3.0, 1, dev) == 3.0
# the optional global-batch norms through RCCL: with one rank the two 16-byte all-reduces are identities
import svt_speechbrain_amd as S
cfg = S.PRESETS["tiny-group"]
enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, normalize_wav=True, precision="fp32", seed=1).to(dev)
wav = 0.1 * torch.randn(3, 4000, device=dev)
want = enc(wav)
enc.set_global_batch_norm(3)
with torch.cuda.stream(side):
    side.wait_stream(torch.cuda.current_stream())
    got = enc(wav)
torch.cuda.synchronize()
assert torch.equal(got, want)
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_single_rank_rccl_collectives():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", SVT_ROOT=root,
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
