"""Micro-benchmark + correctness check of the dense contraction kernel on the encoder's real shapes
(random data — zero-filled operands read high on this chip).  Usage: python tools/gemm_bench.py [--check]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402

SHAPES = [  # name, M, N, K, conv(T_in, T_out, stride, Cin) or None, act, out_f32, resid
    ("conv1", 32 * 15999, 512, 1536, (31999, 15999, 2, 512), 1, 0, 0),
    ("conv2", 32 * 7999, 512, 1536, (15999, 7999, 2, 512), 1, 0, 0),
    ("conv3", 32 * 3999, 512, 1536, (7999, 3999, 2, 512), 1, 0, 0),
    ("conv4", 32 * 1999, 512, 1536, (3999, 1999, 2, 512), 1, 0, 0),
    ("conv5", 32 * 999, 512, 1024, (1999, 999, 2, 512), 1, 0, 0),
    ("proj", 15968, 768, 512, None, 0, 1, 0),
    ("qkv", 15968, 2304, 768, None, 0, 0, 0),
    ("out_proj", 15968, 768, 768, None, 0, 1, 0),
    ("ffn1", 15968, 3072, 768, None, 1, 0, 0),
    ("ffn2", 15968, 768, 3072, None, 0, 1, 0),
    ("out_b", 15968, 768, 768, None, 0, 0, 0),        # bf16 branch outputs (what the bf16-mode encoder asks for)
    ("ffn2_b", 15968, 768, 3072, None, 0, 0, 0),
    ("conv4_plain", 32 * 1999, 512, 1536, None, 1, 0, 0),   # conv4's product with non-overlapping rows (pitch 3072 B)
    ("large_ffn1", 31936, 4096, 1024, None, 1, 0, 0),
    ("large_ffn2_b", 31936, 1024, 4096, None, 0, 0, 0),
    ("large_out_b", 31936, 1024, 1024, None, 0, 0, 0),
    ("large_conv1", 64 * 15999, 512, 1536, (31999, 15999, 2, 512), 0, 0, 0),   # layer-norm conv stack: no activation in the GEMM
    ("large_ffn2", 31936, 1024, 4096, None, 0, 1, 0),
    ("large_qkv", 31936, 3072, 1024, None, 0, 0, 0),
    ("large_out", 31936, 1024, 1024, None, 0, 1, 0),
    ("conv1_noact", 32 * 15999, 512, 1536, (31999, 15999, 2, 512), 0, 0, 0),   # what the GELU epilogue costs: same products without it
    ("ffn1_noact", 15968, 3072, 768, None, 0, 0, 0),
    ("large_ffn1_noact", 31936, 4096, 1024, None, 0, 0, 0),
    ("b1_qkv", 249, 2304, 768, None, 0, 0, 0),       # one 5 s utterance (the recipes' evaluation batch)
    ("b1_out", 249, 768, 768, None, 0, 0, 0),
    ("b1_ffn1", 249, 3072, 768, None, 1, 0, 0),
    ("b1_ffn2", 249, 768, 3072, None, 0, 0, 0),
    ("b1_ffn2r", 249, 768, 3072, None, 0, 1, 1),
    ("b1_conv5", 499, 512, 1024, (999, 499, 2, 512), 1, 0, 0),
    ("b2_ffn2", 998, 768, 3072, None, 0, 0, 0),
    ("b2_ffn1", 998, 3072, 768, None, 1, 0, 0),
    ("b4_ffn2", 1996, 768, 3072, None, 0, 0, 0),
    ("b4_ffn1", 1996, 3072, 768, None, 1, 0, 0),
    ("b4_out", 1996, 768, 768, None, 0, 0, 0),
    ("b8_ffn2", 3992, 768, 3072, None, 0, 0, 0),
    ("b8_out", 3992, 768, 768, None, 0, 0, 0),
    ("s35_qkv", 8715, 2304, 768, None, 0, 0, 0),      # a 3-minute song as one batch: 35 x 5 s
    ("s35_out", 8715, 768, 768, None, 0, 0, 0),
    ("s35_ffn1", 8715, 3072, 768, None, 1, 0, 0),
    ("s35_ffn2", 8715, 768, 3072, None, 0, 0, 0),
    ("sq4096", 4096, 4096, 4096, None, 0, 0, 0),
    ("sq8192", 8192, 8192, 8192, None, 0, 0, 0),
]


def full_check(A, W, bias, C, conv, M, N, K, act, chunk=16384):
    """every row of C against an fp32 torch product (checker only; chunks keep the fp32 copies small)"""
    worst = 0.0
    dev = A.device
    Wf = W[:, :K].float().t().contiguous()
    if conv:
        T_in, T_out, st, cin = conv
        k = K // cin
        B = M // T_out
        idx = (torch.arange(T_out, device=dev) * st)[:, None] + torch.arange(k, device=dev)[None, :]
        for b0 in range(B):
            ref = A[b0, idx].reshape(T_out, K).float() @ Wf + bias
            if act == 1:
                ref = torch.nn.functional.gelu(ref)
            got = C[b0 * T_out:(b0 + 1) * T_out].float()
            worst = max(worst, ((got - ref).abs() / (1.0 + ref.abs())).max().item())
    else:
        for r0 in range(0, M, chunk):
            ref = A[r0:r0 + chunk, :K].float() @ Wf + bias
            if act == 1:
                ref = torch.nn.functional.gelu(ref)
            got = C[r0:r0 + chunk].float()
            worst = max(worst, ((got - ref).abs() / (1.0 + ref.abs())).max().item())
    return worst


def run(name, M, N, K, conv, act, out_f32, resid, prec, check, iters, pad=0, nobias=False, fullcheck=False):
    lib = _lib.load()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if prec == 1 else torch.float32  # prec 2 / 3: split-operand engine, fp32 operands in memory
    g = torch.Generator(device="cpu").manual_seed(1)
    if conv:
        T_in, T_out, st, cin = conv
        B = M // T_out
        A = (torch.rand(B, T_in, cin, generator=g) * 2 - 1).to(dev, dt)
        rpb, bstr, rstr = T_out, T_in * cin, st * cin
    else:
        A = (torch.rand(M, K + pad, generator=g) * 2 - 1).to(dev, dt)
        rpb, bstr, rstr = M, 0, K + pad
    W = ((torch.rand(N, K + (0 if conv else pad), generator=g) * 2 - 1) / K ** 0.5).to(dev, dt)
    ldw = W.shape[1]
    bias = torch.randn(N, generator=g).to(dev)
    if nobias:
        bias.zero_()
    R = torch.randn(M, N, generator=g).to(dev) if resid else None
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if (out_f32 or prec != 1) else dt)
    st_ = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(lib.svt_debug_gemm(prec, A.data_ptr(), W.data_ptr(), C.data_ptr(), None if nobias else bias.data_ptr(),
                                      R.data_ptr() if resid else None, M, N, K, rpb, bstr, rstr, ldw, act, out_f32, 0, st_),
                   "svt_debug_gemm")

    C.fill_(float("nan"))
    call()
    torch.cuda.synchronize()
    err = None
    if fullcheck and not resid:
        err = full_check(A, W, bias, C, conv, M, N, K, act)
    elif check:
        if conv:
            T_in, T_out, st, cin = conv
            k = K // cin
            idx = (torch.arange(T_out, device=dev) * st)[:, None] + torch.arange(k, device=dev)[None, :]
            nb = min(2, B)
            A2 = A[:nb, idx].reshape(nb * T_out, K).float()
            ref = A2 @ W[:, :K].float().t() + bias
            got = C[: nb * T_out].float()
        else:
            rows = min(M, 4096)
            ref = A[:rows, :K].float() @ W[:, :K].float().t() + bias
            got = C[:rows].float()
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        if resid:
            ref = ref + R[: ref.shape[0]]
        err = (got - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    print(f"{name:12s} M={M:7d} N={N:5d} K={K:5d} prec={('fp32', 'bf16', 'bf16x3', 'fp16x3')[prec]} {ms * 1e3:9.1f} us  {tf:8.1f} TFLOP/s"
          + (f"  max|err|={err:.3e}" if err is not None else ""), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--prec", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--names", default=None, help="comma-separated exact shape names")
    ap.add_argument("--fullcheck", action="store_true", help="compare EVERY output row with an fp32 torch product (relative to 1 + |ref|)")
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--bm", type=int, default=0)
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--no-skinny", action="store_true", help="small problems on the large-tile kernels (A/B)")
    ap.add_argument("--skinny-max-tiles", type=int, default=32)
    ap.add_argument("--variant", type=int, default=0, help="svt_debug_set key 3 (experimental kernel variants)")
    ap.add_argument("--nobias", action="store_true")
    ap.add_argument("--no-x3-dma", action="store_true", help="split-operand modes: register-staged kernel instead of the LDS-DMA one (A/B)")
    ap.add_argument("--pad", type=int, default=0, help="extra elements of row pitch for A and W (plain GEMM shapes)")
    ap.add_argument("--set", action="append", default=[], help="key=value for svt_debug_set (repeatable), e.g. --set 28=1: two-slot schedule of gemm_pps_kernel")
    ap.add_argument("--lib-suffix", default=None, help="load libsvt_mi355_<suffix>.so (an experimental build of the bf16 library) instead")
    a = ap.parse_args()
    if a.lib_suffix:
        _lib.LIB_PATH = _lib.LIB_PATH.replace("libsvt_mi355.so", f"libsvt_mi355_{a.lib_suffix}.so")
    _lib.load().svt_debug_set(0, a.dbg)
    _lib.load().svt_debug_set(1, a.bm)
    _lib.load().svt_debug_set(2, a.ring)
    _lib.load().svt_debug_set(3, a.variant)
    _lib.load().svt_debug_set(6, 0 if a.no_skinny else 1)
    _lib.load().svt_debug_set(7, a.skinny_max_tiles)
    _lib.load().svt_debug_set(11, 0 if a.no_x3_dma else 1)
    _lib.load().svt_debug_set(12, 1)
    for kv in a.set:
        _lib.load().svt_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
    for s in SHAPES:
        if a.only and a.only not in s[0]:
            continue
        if a.names and s[0] not in a.names.split(","):
            continue
        run(*s, a.prec, a.check, a.iters, a.pad, a.nobias, a.fullcheck)
