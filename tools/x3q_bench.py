"""Micro-benchmark of the pair-row split-operand product (csrc/gemm_x3q.hip) on the encoder's shapes: us per launch and algorithmic
TFLOP/s (peak of the mode: 2.5 PF / 3 = 833), via svt_debug_gemm_pairs' event timing.  usage: python tools/x3q_bench.py [--bm 0|256|192|128]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svt_speechbrain_amd import _lib

SHAPES = [
    # name, M, N, K, conv (T_in, T_out, stride, cin) or None, act, out_kind
    ("conv1", 32 * 15999, 512, 1536, (31999, 15999, 2, 512), 1, 1),
    ("conv2", 32 * 7999, 512, 1536, (15999, 7999, 2, 512), 1, 1),
    ("conv4", 32 * 1999, 512, 1536, (3999, 1999, 2, 512), 1, 1),
    ("proj", 15968, 768, 512, None, 0, 0),
    ("qkv", 15968, 2304, 768, None, 0, 2),
    ("outproj", 15968, 768, 768, None, 0, 0),
    ("ffn1_gelu", 15968, 3072, 768, None, 1, 1),
    ("ffn1_noact", 15968, 3072, 768, None, 0, 1),
    ("ffn1_f32out", 15968, 3072, 768, None, 0, 0),
    ("ffn2", 15968, 768, 3072, None, 0, 0),
    ("large_qkv", 31936, 3072, 1024, None, 0, 2),
    ("large_ffn1", 31936, 4096, 1024, None, 1, 1),
    ("large_ffn2", 31936, 1024, 4096, None, 0, 0),
    ("sq4096", 4096, 4096, 4096, None, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bm", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--prec", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--dbg", type=int, default=0, help="svt_debug_set(0, n): timing ablations of gemm_x3q_kernel (make DIAG=1; results are wrong): "
                    "1 no LDS-DMA after the head, 2 no fragment reads, 4 no MFMAs, 8 no barriers between slots (sums combine)")
    a = ap.parse_args()
    lib = _lib.load()
    _lib.require_gpu()
    lib.svt_debug_set(1, a.bm)
    lib.svt_debug_set(0, a.dbg)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, conv, act, ok in SHAPES:
        if a.only and a.only not in name:
            continue
        if conv:
            T_in, T_out, st, cin = conv
            B = M // T_out
            A = (torch.rand(B, T_in, cin, generator=g) * 2 - 1).to(dev)
            rpb, bstr, rstr = T_out, T_in * cin, st * cin
        else:
            A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev)
            rpb, bstr, rstr = M, 0, K
        W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        Cc = torch.empty((M, N), device=dev)
        ms = C.c_float(0)
        _lib.check(lib.svt_debug_gemm_pairs(a.prec, A.data_ptr(), A.numel(), W.data_ptr(), Cc.data_ptr(), b.data_ptr(), M, N, K, rpb, bstr, rstr,
                                            act, ok, 0, torch.cuda.current_stream().cuda_stream, a.iters, C.byref(ms)), "svt_debug_gemm_pairs")
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12
        print(f"{name:14s} M {M:7d} N {N:5d} K {K:5d} act {act} out {ok} bm {a.bm:3d}: {ms.value * 1e3:8.1f} us  {tf:6.1f} TFLOP/s  ({tf / 833.3:.3f} of 833)", flush=True)
        del A, W, Cc


if __name__ == "__main__":
    main()
