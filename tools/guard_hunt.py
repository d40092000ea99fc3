#!/usr/bin/env python3
"""Out-of-bounds hunt driver: runs encoder forwards of the fuzz geometries with page-guarded buffers (svt_debug_set key 13),
ONE child process per (seed, precision, guard mode, guarded part), because a fault aborts the process.  For every child that
dies it re-runs the case with AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=3 and prints the last kernel the runtime launched.
  python tools/guard_hunt.py [--seeds 0-23] [--precisions bf16,fp32,...]"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARTS = {"weights": (1, 0, 0), "workspace": (0, 1, 0), "input": (0, 0, 1)}


def child(seed, prec, mode, part):
    import logging
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    logging.disable(logging.WARNING)
    import svt_speechbrain_amd as S
    from svt_speechbrain_amd import _device, _lib
    from test_gpu_fuzz import random_case
    import test_gpu_guard as G
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cfg, B, L = random_case(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(dev)
    wg, sg, ig = PARTS[part]

    def build():
        return S.HuggingFaceWav2Vec2(cfg.name, None, config=cfg, normalize_wav=True, precision=prec, seed=seed).to(dev)

    want = build()(wav).cpu()
    lib.svt_debug_set(13, mode if wg else 0)
    if sg:
        lib.svt_debug_set(13, mode)
        _device.DeviceSlot.workspace = G._guarded_workspace
    enc = build()
    if ig:
        lib.svt_debug_set(13, mode)
    got = enc(G.guarded_like(wav) if ig else wav).cpu()
    torch.cuda.synchronize()
    print("EQUAL" if torch.equal(got, want) else "DIFFERENT", flush=True)
    os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0-23")
    ap.add_argument("--precisions", default="fp32,fp16x3,bf16x3,bf16")
    ap.add_argument("--child", nargs=4)
    a = ap.parse_args()
    if a.child:
        return child(int(a.child[0]), a.child[1], int(a.child[2]), a.child[3])
    lo, _, hi = a.seeds.partition("-")
    bad = 0
    for seed in range(int(lo), int(hi or lo) + 1):
        for prec in a.precisions.split(","):
            for mode in (1, 2):
                for part in PARTS:
                    cmd = [sys.executable, os.path.abspath(__file__), "--child", str(seed), prec, str(mode), part]
                    r = subprocess.run(cmd, capture_output=True, text=True)
                    if r.returncode == 0 and "EQUAL" in r.stdout:
                        continue
                    bad += 1
                    fault = [ln for ln in r.stderr.splitlines() if "fault" in ln.lower()]
                    print(f"seed {seed} {prec} mode {mode} {part}: rc {r.returncode} {r.stdout.strip()} {fault[:1]}", flush=True)
                    if r.returncode != 0:
                        env = dict(os.environ, AMD_SERIALIZE_KERNEL="3", AMD_LOG_LEVEL="3")
                        t = subprocess.run(cmd, capture_output=True, text=True, env=env)
                        names = re.findall(r"ShaderName : (\S+)", t.stderr)
                        print("   last kernels:", names[-3:], flush=True)
    print(f"guard hunt: {bad} failing case(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
