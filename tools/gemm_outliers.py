"""Launch-time outliers of a contraction shape: batches of back-to-back launches timed with events, the slowest batches listed
(round 3 saw single yardstick lines 5-10x off: was that the kernel or the box?).
Usage: python tools/gemm_outliers.py [--name ffn2_b] [--variant 0|49] [--batches 400] [--per 25]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402
from gemm_bench import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--name", default="ffn2_b")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--batches", type=int, default=400)
    ap.add_argument("--per", type=int, default=25)
    a = ap.parse_args()
    lib = _lib.load()
    lib.svt_debug_set(3, a.variant)
    dev = torch.device("cuda:0")
    name, M, N, K, conv, act, out_f32, resid = [s for s in SHAPES if s[0] == a.name][0]
    g = torch.Generator().manual_seed(1)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev, torch.bfloat16)
    W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dev, torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(lib.svt_debug_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), None, M, N, K, M, 0, K, K, act, 0, 0, st),
                   "svt_debug_gemm")
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.batches)]
    for e0, e1 in ev:
        e0.record()
        for _ in range(a.per):
            call()
        e1.record()
    torch.cuda.synchronize()
    t = torch.tensor([e0.elapsed_time(e1) / a.per * 1e3 for e0, e1 in ev])
    srt = t.sort().values
    print(f"{name} variant {a.variant}: {a.batches} batches x {a.per} launches: median {t.median():.1f} us, p99 {srt[int(0.99 * len(srt))]:.1f}, "
          f"max {t.max():.1f} (batch {int(t.argmax())}); batches above 1.5 x median: {(t > 1.5 * t.median()).sum().item()}")


if __name__ == "__main__":
    main()
