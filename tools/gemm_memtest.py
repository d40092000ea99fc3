"""Is a GEMM shape memory-side bound?  Times the LDS-DMA kernels with every A row (row stride 0) and / or every W row
(ldw 0) aliased to one row, which removes that operand's L2 / fabric traffic (and, unavoidably, its bit toggling).
Measured: QKV 65.9 -> 59.6 us, FFN-1 84.1 -> 74.1, FFN-2 69.1 -> 60.4, out-proj 24.9 -> 23.0 with both aliased: <= 10 %."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from svt_speechbrain_amd import _lib
lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
def run(name, M, N, K, a_same, w_same, iters=30):
    g = torch.Generator().manual_seed(1)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev, torch.bfloat16)
    W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dev, torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rstr = 0 if a_same else K
    ldw = 0 if w_same else K
    def call():
        _lib.check(lib.svt_debug_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, M, N, K, M, 0, rstr, ldw, 0, 0, 0, st), "g")
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:8s} A_same={a_same} W_same={w_same}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
for name, M, N, K in [("qkv", 15968, 2304, 768), ("ffn1", 15968, 3072, 768), ("ffn2", 15968, 768, 3072), ("out", 15968, 768, 768)]:
    for a_same, w_same in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        run(name, M, N, K, a_same, w_same)
