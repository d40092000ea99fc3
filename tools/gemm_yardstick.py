"""Yardstick for the bf16 dense-contraction kernels: the same (M, N, K) problems through the vendor library (hipBLASLt / rocBLAS
behind torch.nn.functional.linear, bf16 in / bf16 out, bias) and through this build's kernels (svt_debug_gemm, bias + the
layer's real epilogue).  PyTorch is only the harness here; nothing in the product path calls the library.
  python tools/gemm_yardstick.py [--iters 30]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import gemm_bench  # noqa: E402

NAMES = ["conv1", "conv2", "conv3", "conv4", "conv5", "proj", "qkv", "out_b", "ffn1", "ffn2_b", "large_qkv", "large_out_b", "large_ffn1",
         "large_ffn2_b", "s35_qkv", "s35_ffn1", "s35_ffn2", "sq4096", "sq8192"]


def library(M, N, K, iters):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev, torch.bfloat16)
    W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dev, torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev, torch.bfloat16)
    for _ in range(3):
        torch.nn.functional.linear(A, W, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.nn.functional.linear(A, W, b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--variants", default="0", help="comma-separated svt_debug_set key-3 values to time on this build's side (0 = default dispatch)")
    ap.add_argument("--names", default=None)
    ap.add_argument("--no-library", action="store_true")
    a = ap.parse_args()
    from svt_speechbrain_amd import _lib
    _lib.load().svt_debug_set(12, 1)
    print(f"torch {torch.__version__}  blas backend: {torch.backends.cuda.preferred_blas_library()}")
    for s in gemm_bench.SHAPES:
        if s[0] not in (a.names.split(",") if a.names else NAMES):
            continue
        name, M, N, K = s[:4]
        ms = 0.0 if a.no_library else library(M, N, K, a.iters)
        if not a.no_library:
            print(f"{name:12s} M={M:7d} N={N:5d} K={K:5d} library (plain GEMM + bias)      {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)
        for v in a.variants.split(","):
            _lib.load().svt_debug_set(3, int(v))
            print(f"  variant {v:>3s}: ", end="")
            gemm_bench.run(*s, 1, False, a.iters)
        _lib.load().svt_debug_set(3, 0)
