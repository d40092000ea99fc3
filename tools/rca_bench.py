"""Timing of the RCA fusion forward (BASELINE config C4 shape: B=16, T1=499, T2=500, d_model 1024)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svt_speechbrain_amd as S
dev = "cuda:0"
for prec in ("bf16", "fp32"):
    fus = S.FusionRCA(precision=prec).to(dev)
    g = torch.Generator().manual_seed(0)
    a = torch.randn(16, 499, 1024, generator=g).to(dev); v = torch.randn(16, 500, 1024, generator=g).to(dev)
    for _ in range(3): out = fus(a, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 20
    for _ in range(n): out = fus(a, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"FusionRCA {prec}: {dt*1e3:.3f} ms per batch of 16 -> {16/dt:.0f} clips/s, {16*33.38e9/dt/1e12:.0f} TFLOP/s (reference FLOP count)")
