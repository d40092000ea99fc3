"""Probe of the persistent staggered GEMM with exact integer data, by feature: bias, long K, several tiles per workgroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from svt_speechbrain_amd import _lib
lib = _lib.load()
lib.svt_debug_set(3, int(sys.argv[1]) if len(sys.argv) > 1 else 50)
lib.svt_debug_set(1, int(sys.argv[2]) if len(sys.argv) > 2 else 256)
dev = torch.device("cuda:0")


def probe(tag, M, N, K, use_bias, act=0):
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    W = torch.randint(-2, 3, (N, K), generator=g).float()
    bias = (torch.arange(N) % 7).float() if use_bias else None
    A_ = A.to(dev, torch.bfloat16); W_ = W.to(dev, torch.bfloat16)
    b_ = bias.to(dev) if use_bias else None
    C = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.svt_debug_gemm(1, A_.data_ptr(), W_.data_ptr(), C.data_ptr(), b_.data_ptr() if use_bias else None, None, M, N, K, M, 0, K, K,
                                  act, 0, 0, st), "gemm")
    torch.cuda.synchronize()
    ref = A_.float() @ W_.float().t()
    if use_bias:
        ref = ref + b_
    if act:
        ref = torch.nn.functional.gelu(ref)
    ref = ref.to(torch.bfloat16).float()
    Cf = C.float()
    bad = ((Cf - ref).abs() > 0.02 * (1 + ref.abs())) | torch.isnan(Cf)
    print(f"{tag:28s} M={M} N={N} K={K} bias={use_bias} act={act}: mismatches {int(bad.sum())} of {M * N}, nan {int(torch.isnan(Cf).sum())}")
    if bad.any():
        idx = bad.nonzero()
        rows = idx[:, 0]; cols = idx[:, 1]
        print("   bad rows: min", int(rows.min()), "max", int(rows.max()), " distinct row%256:", sorted(set((rows % 256).tolist()))[:40])
        print("   bad cols: min", int(cols.min()), "max", int(cols.max()), " distinct col%256:", sorted(set((cols % 256).tolist()))[:40])
        r0, c0 = int(rows[0]), int(cols[0])
        print("   first bad at", r0, c0, "got", Cf[r0, c0:c0 + 8].tolist(), "ref", ref[r0, c0:c0 + 8].tolist())


probe("N=256, 400 m-tiles", 256 * 400, 256, 128, False)
probe("N=256, 400 m-tiles, K=768", 256 * 400, 256, 768, False)
probe("M=256, 400 n-tiles", 256, 256 * 400, 128, False)
probe("M=512, 200 n-tiles", 512, 256 * 200, 128, False)
probe("many tiles (3 per wg)", 256 * 96, 2048, 128, False)
probe("many tiles K=768", 256 * 96, 2048, 768, False)
