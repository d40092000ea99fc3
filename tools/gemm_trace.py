"""Per-workgroup phase timeline of the LDS-DMA dense-contraction kernels (dbg = 9: the kernels stamp s_memrealtime,
100 MHz, at entry / first slab / tile ends / exit; the `resid` argument carries the trace buffer).
Usage: python tools/gemm_trace.py [--only qkv] [--ring 2|4] [--bm N]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402
from gemm_bench import SHAPES  # noqa: E402


def x3_slots(a):
    lib = _lib.load()
    lib.svt_debug_set(12, 1)   # keep the registered split weights between calls
    lib.svt_debug_set(3, 32 if a.one_tile else 34)   # never / always the persistent form
    dev = torch.device("cuda:0")
    for name, M, N, K, conv, act, out_f32, resid in SHAPES:
        if a.only != name:
            continue
        g = torch.Generator().manual_seed(1)
        if conv:
            T_in, T_out, st, cin = conv
            B = M // T_out
            A = (torch.rand(B, T_in, cin, generator=g) * 2 - 1).to(dev)
            rpb, bstr, rstr = T_out, T_in * cin, st * cin
        else:
            A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev)
            rpb, bstr, rstr = M, 0, K
        W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        C = torch.empty(M, N, device=dev, dtype=torch.float32)
        trace = torch.zeros(65536 * 16, dtype=torch.int64, device=dev)
        st_ = torch.cuda.current_stream().cuda_stream

        def call(dbg):
            lib.svt_debug_set(0, dbg)
            _lib.check(lib.svt_debug_gemm(3, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(),
                                          trace.data_ptr() if dbg == 9 else None, M, N, K, rpb, bstr, rstr, W.shape[1], act,
                                          1, 0, st_), "svt_debug_gemm")
        import time
        for _ in range(3):
            call(0)
        torch.cuda.synchronize()
        t_end = time.time() + max(a.load_seconds, 0.5)
        while time.time() < t_end:
            for _ in range(20):
                call(0)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call(0)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: fp16x3 {'one-tile' if a.one_tile else 'persistent'} kernel {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
        recs = []
        for mode in (1, 2, 3, 4):
            lib.svt_debug_set(15, mode)
            trace.zero_()
            for _ in range(5):
                call(0)
            call(9)
            torch.cuda.synchronize()
            r = trace[65536:65536 + 256 * 8 * 32].view(256, 8, 32).cpu()
            recs.append(r[r[:, 0, 9] > 0])
        lib.svt_debug_set(15, 0)
        lib.svt_debug_set(0, 0)

        def d(x, y):
            return ((y - x) & 0xFFFFFFFF).double()
        names0 = ["LOAD0", "MMA0", "LOAD1", "MMA1", "LOAD2", "MMA2", "LOAD3", "MMA3+retire"]
        names1 = ["LOAD0", "MMA0", "LOAD1", "MMA1", "LOAD2", "MMA2", "LOAD3+retire", "MMA3(+epi)"]
        print(f"  slab {int(recs[0][0,0,10])} of {int(recs[0][0,0,9])}: core cycles per slot (work = start .. arrival at the barrier, wait = arrival .. "
              f"release), median over {recs[0].shape[0]} workgroups x 4 waves; 24 MFMAs = 384 cycles per MMA slot")
        for grp, names in ((0, names0), (1, names1)):
            w = slice(4 * grp, 4 * grp + 4)
            tot = 0.0
            print(f"   waves {4*grp}-{4*grp+3}:")
            for k in range(8):
                src = recs[k // 2]
                b = 2 * (k & 1)
                work = d(src[:, w, b], src[:, w, b + 1]).median().item()
                wait = d(src[:, w, b + 1], src[:, w, b + 2]).median().item()
                tot += work + wait
                print(f"     {names[k]:14s} work {work:6.0f}   wait {wait:6.0f}")
            print(f"     slab total {tot:.0f} cycles (3072 = the MFMA pipe's share)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="qkv")
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--bm", type=int, default=0)
    ap.add_argument("--nobias", action="store_true")
    ap.add_argument("--load-seconds", type=float, default=0.0,
                    help="run the kernel back to back for this long before the stamped launch (the clock the chip holds under "
                         "load: MI355X_MICROARCH.md 'DVFS give-back' item 6 asks for >= 2 s)")
    ap.add_argument("--variant", type=int, default=0, help="extra diagnostic variant: 10 = no stores, 11 = no pointer setup")
    ap.add_argument("--force-variant", type=int, default=0, help="svt_debug_set key 3 for the whole run (50 = the persistent staggered kernel, "
                    "51 / 53 / 54 = without LDS-DMA / epilogue / both)")
    ap.add_argument("--slots", type=int, default=0, help="gemm_pps_kernel slot stamps of one slab: 5 = middle of the second tile, 6 = last slab of "
                    "the first tile, 7 = first slab of the second tile (uses --force-variant 70 + this digit)")
    ap.add_argument("--one-tile", action="store_true", help="with --x3-slots: gemm_x3s_kernel (one tile per workgroup) instead of the persistent kernel")
    ap.add_argument("--x3-slots", action="store_true", help="slot stamps of gemm_x3p_kernel (fp16x3, the persistent split-operand kernel): four "
                    "launches, two slots each")
    ap.add_argument("--p1w", action="store_true", help="trace gemm_p1w_kernel (svt_debug_set key 29 = 2) instead of gemm_pps_kernel; adds the "
                    "cycles its waves spend at the slab barriers")
    ap.add_argument("--set", action="append", default=[], help="key=value for svt_debug_set (repeatable), e.g. --set 37=128: persistent launches of 128 workgroups")
    a = ap.parse_args()
    for kv in a.set:
        _lib.load().svt_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
    if a.x3_slots:
        return x3_slots(a)
    if a.p1w:
        a.force_variant = a.force_variant or 70   # the persistent kernels' record layout
    if a.slots:
        a.force_variant = 70 + a.slots
    lib = _lib.load()
    lib.svt_debug_set(1, a.bm)
    lib.svt_debug_set(2, a.ring)
    lib.svt_debug_set(3, 0 if a.p1w else a.force_variant)
    if a.p1w:
        lib.svt_debug_set(29, 2)
    dev = torch.device("cuda:0")
    for name, M, N, K, conv, act, out_f32, resid in SHAPES:
        if a.only != name and not (a.only not in [x[0] for x in SHAPES] and a.only in name):
            continue
        g = torch.Generator().manual_seed(1)
        if conv:
            T_in, T_out, st, cin = conv
            B = M // T_out
            A = (torch.rand(B, T_in, cin, generator=g) * 2 - 1).to(dev, torch.bfloat16)
            rpb, bstr, rstr = T_out, T_in * cin, st * cin
        else:
            A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev, torch.bfloat16)
            rpb, bstr, rstr = M, 0, K
        W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dev, torch.bfloat16)
        bias = torch.randn(N, generator=g).to(dev)
        C = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
        trace = torch.zeros(65536 * 16, dtype=torch.int64, device=dev)
        st_ = torch.cuda.current_stream().cuda_stream

        def call(dbg):
            lib.svt_debug_set(0, dbg)
            _lib.check(lib.svt_debug_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), None if a.nobias else bias.data_ptr(),
                                          trace.data_ptr() if dbg == 9 else None, M, N, K, rpb, bstr, rstr, W.shape[1], act,
                                          out_f32, 0, st_), "svt_debug_gemm")
        for _ in range(3):
            call(0)
        torch.cuda.synchronize()
        if a.load_seconds > 0:
            import time
            t_end = time.time() + a.load_seconds
            while time.time() < t_end:
                for _ in range(200):
                    call(0)
                torch.cuda.synchronize()
            for _ in range(200):
                call(0)
        call(9)
        torch.cuda.synchronize()
        if a.variant:
            # the trace pointer travels in `resid`, the variant in dbg: dbg 9 is needed for the trace, so the library
            # treats key 3 as "extra variant while tracing"
            lib.svt_debug_set(3, a.variant)
            call(9)
            torch.cuda.synchronize()
            lib.svt_debug_set(3, 0)
        t = trace[:65536].view(-1, 8).cpu()
        t = t[t[:, 5] > 0].double()
        t0 = t[:, 0].min()
        us = 0.01
        span = (t[:, 4].max() - t0) * us
        print(f"{name}: {t.shape[0]} wave-records, kernel span {span:.1f} us")
        print(f"  entry skew  (begin - first begin): mean {((t[:,0]-t0)*us).mean():.2f}  max {((t[:,0]-t0)*us).max():.2f} us")
        print(f"  prologue    (first slab ready)   : mean {((t[:,1]-t[:,0])*us).mean():.2f}  max {((t[:,1]-t[:,0])*us).max():.2f} us")
        print(f"  main loop   (sum over tiles)     : mean {(t[:,2]*us).mean():.2f}  max {(t[:,2]*us).max():.2f} us   per tile {(t[:,2]/t[:,5]*us).mean():.2f}")
        print(f"  epilogue    (sum over tiles)     : mean {(t[:,3]*us).mean():.2f}  max {(t[:,3]*us).max():.2f} us   per tile {(t[:,3]/t[:,5]*us).mean():.2f}")
        print(f"  exit        (end - first begin)  : mean {((t[:,4]-t0)*us).mean():.2f}  min {((t[:,4]-t0)*us).min():.2f} max {((t[:,4]-t0)*us).max():.2f} us")
        if (t[:, 6] > 0).any() and a.force_variant >= 50:
            # persistent staggered kernel: core clocks over the whole stream (first slab ready .. exit), all tiles of the workgroup
            mhz = (t[:, 6] / (t[:, 4] - t[:, 1]).clamp(min=1)) * 100.0
            clk = mhz.median().item() * 1e6
            busy = (2.0 * t[:, 7] * 256 * K * t[:, 5]) / (t[:, 2] * 1e-8) / (4 * 1024 * clk)
            nk = K // 64
            print(f"  per K slab (main loop / slabs): mean {(t[:,2] / (t[:,5] * nk) * us).mean():.3f} us   per epilogue: mean {(t[:,3] / t[:,5] * us).mean():.2f} us")
        elif (t[:, 6] > 0).any():
            mhz = (t[:, 6] / t[:, 2].clamp(min=1)) * 100.0
            clk = mhz.median().item() * 1e6
            busy = (2.0 * t[:, 7] * 256 * K) / (t[:, 2] * 1e-8) / (4 * 1024 * clk)   # per workgroup = per CU
        if (t[:, 6] > 0).any():
            print(f"  core clock over the main loop (s_memtime / s_memrealtime): median {mhz.median():.0f} MHz "
                  f"(min {mhz.min():.0f}, max {mhz.max():.0f}); MFMA pipe busy for {busy.mean():.2f} of the main-loop cycles "
                  f"(1024 bf16 FLOP/clk/SIMD)")
        print(f"  tiles per workgroup: min {int(t[:,5].min())} max {int(t[:,5].max())}")
        if a.p1w:
            cw = trace[524288:524288 + 1024].cpu().double()
            cw = cw[cw > 0]
            slabs = (t[:, 5] * (K // 64)).mean().item()
            print(f"  gemm_p1w_kernel: core cycles between reaching a slab's counted wait and leaving its barrier: mean {cw.mean().item() / slabs:.0f} per slab "
                  f"(the matrix pipe idles for them; {int(t[0, 7].item()) // 32 * 8 * 2 * 16} = its own cycles per slab)")
        if a.slots:
            def stamps():
                r = trace[65536:65536 + 256 * 8 * 32].view(256, 8, 32).cpu()
                return r[r[:, 0, 9] > 0]
            starts = stamps()
            lib.svt_debug_set(15, 1)
            trace.zero_()
            call(9)
            torch.cuda.synchronize()
            ends = stamps()
            lib.svt_debug_set(15, 0)
            # launch 1 holds stamps 0..8 (both ends of slots 0-3), launch 2 stamps 8..16 (slots 4-7 and the next slab's start)
            def d(x, y):
                return ((y - x) & 0xFFFFFFFF).double()
            names0 = ["LOAD0", "MMA0", "LOAD1", "MMA1", "LOAD2", "MMA2", "LOAD3", "MMA3+retire"]
            names1 = ["LOAD0", "MMA0", "LOAD1", "MMA1", "LOAD2", "MMA2", "LOAD3+retire", "MMA3(+epi)"]
            print(f"  slab {int(starts[0,0,10])} of {int(starts[0,0,9])}: core cycles per slot, work = slot start .. the wave's arrival at the barrier, "
                  f"wait = arrival .. release; median over {starts.shape[0]} workgroups x 4 waves")
            for grp, names in ((0, names0), (1, names1)):
                w = slice(4 * grp, 4 * grp + 4)
                tot = 0.0
                print(f"   waves {4*grp}-{4*grp+3}:")
                for k in range(8):
                    src = starts if k < 4 else ends
                    b = 2 * (k & 3)
                    work = d(src[:, w, b], src[:, w, b + 1]).median().item()
                    wait = d(src[:, w, b + 1], src[:, w, b + 2]).median().item()
                    tot += work + wait
                    print(f"     {names[k]:14s} work {work:6.0f}   wait {wait:6.0f}")
                print(f"     slab total {tot:.0f} cycles (2048 = the MFMA pipe's share)")


if __name__ == "__main__":
    main()
