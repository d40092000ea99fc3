"""Micro-benchmark + check of the fused attention kernel on the encoder's shapes.
Usage: python tools/attn_bench.py [--check]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402

SHAPES = [("base", 32, 499, 12, 64), ("large", 64, 499, 16, 64), ("c1", 1, 249, 12, 64), ("rca", 16, 499, 8, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--wide", type=int, default=None, help="svt_debug_set key 8 (workgroup shape / occupancy experiments of the fused attention)")
    ap.add_argument("--H", type=int, default=None, help="override the number of heads (workgroups per launch: occupancy / rounds experiments)")
    ap.add_argument("--T", type=int, default=None, help="override the sequence length of every shape (fixed cost vs per-key-tile cost)")
    ap.add_argument("--variant", type=int, default=None, help="svt_debug_set key 21 (A/B of the softmax arithmetic of the 8-wave head_dim-64 kernel)")
    ap.add_argument("--stamps", action="store_true", help="tile stamps of the 8-wave head_dim-64 kernel (svt_debug_set key 18): where a wave's "
                    "time goes inside key tile 4")
    a = ap.parse_args()
    lib = _lib.load()
    if a.wide is not None:
        lib.svt_debug_set(8, a.wide)
    if a.variant is not None:
        lib.svt_debug_set(21, a.variant)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for name, B, T, H, dh in SHAPES:
        if a.only and a.only not in name:
            continue
        if a.T:
            T = a.T
        if a.H:
            H = a.H
        D = H * dh
        g = torch.Generator().manual_seed(3)
        qkv = (torch.randn(B, T, 3 * D, generator=g) * 1.5).to(dev, torch.bfloat16)
        out = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16)
        scale = dh ** -0.5

        def call():
            _lib.check(lib.svt_debug_attention(1, qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, out.data_ptr(),
                                               B, T, H, dh, 3 * D, 3 * D, D, scale, 0, st), "svt_debug_attention")
        call()
        torch.cuda.synchronize()
        err = None
        if a.check:
            nb = min(B, 2)
            q, k, v = [x.float().view(nb, T, H, dh).transpose(1, 2) for x in qkv[:nb].split(D, dim=-1)]
            ref = torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v
            err = (out[:nb].float().view(nb, T, H, dh).transpose(1, 2) - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        tf = 4.0 * B * H * T * T * dh / us / 1e6
        print(f"{name:6s} B={B:3d} T={T} H={H:2d} dh={dh:3d}: {us:8.1f} us  {tf:7.1f} TFLOP/s"
              + (f"  max|err|={err:.3e}" if err is not None else ""), flush=True)
        if a.stamps and dh == 64 and B * H * ((T + 255) // 256) >= 512:
            lib.svt_debug_set(18, 1)
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            lib.svt_debug_set(18, 0)
            nwg = B * H * ((T + 255) // 256)
            rec = out.view(-1).view(torch.int32)[: nwg * 8 * 16].view(nwg * 8, 16).cpu().to(torch.int64) & 0xFFFFFFFF
            ok = (rec[:, 8] >> 16) == 0x5A5A
            rec = rec[ok]

            def d(i, j):
                return ((rec[:, j] - rec[:, i]) & 0xFFFFFFFF).double()
            names = ["wait for the tile's fills (vmcnt)  + barrier", "issue the next tile's LDS-DMA", "K fragment reads + 8 S MFMAs (issue)",
                     "mask / row max / exchange / rescale", "exp2 + row sums", "P conversion, V transposing reads, 8 PV MFMAs (issue)",
                     "loop back to the next tile's top"]
            print(f"   key tile 4, core cycles per phase, median over {rec.shape[0]} waves (a tile = 512 cycles of MFMA per wave, four waves per SIMD):")
            tot = 0.0
            for k in range(6):
                v = d(k, k + 1).median().item()
                tot += v
                print(f"     {names[k]:55s} {v:7.0f}")
            v = d(6, 7).median().item()
            print(f"     {names[6]:55s} {v:7.0f}")
            print(f"     tile total {tot + v:.0f} cycles")


if __name__ == "__main__":
    main()
