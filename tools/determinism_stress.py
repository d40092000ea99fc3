#!/usr/bin/env python3
"""Is a forward reproducible bit for bit while OTHER processes use the same GPU?

`--procs N` starts N copies of itself on device 0 (as `bench.py --gpus 8` under SVT_SHARE_GPU=1 does).  Every copy forwards `--shards` different
inputs (`--model`, `--batch` x `--seconds`, `--precision`) back to back on one stream, `--iters` sweeps, and compares the logits and EVERY byte
of the encoder's workspace after each forward with the first sweep's: the workspace holds all intermediate tensors, so the first region that differs (region
names: svt_debug_encoder_layout, include/svt_mi355.h) names the kernel that was not reproducible.  Regions that later stages overwrite
show their last writer only.  Exit code 1 if anything differed.

    python tools/determinism_stress.py --procs 8 --iters 40
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

REGIONS = ["moments", "conv0_coef", "conv0_tables", "conv_act0", "conv_act1", "conv_f32", "proj_in", "hF", "preF", "layer_in", "resid_f32",
           "resid_lo", "posconv_in", "posconv_out", "qkv", "scores", "probs", "v_t", "qkv_planes", "attn_out", "ffn_hidden", "gate", "relpos",
           "head_dots"]


def child(args):
    import torch
    import svt_speechbrain_amd as S
    from svt_speechbrain_amd import _lib, weights as W
    from svt_speechbrain_amd.config import PRESETS
    dev = torch.device("cuda:0")
    cfg = PRESETS[args.model]
    enc = S.HuggingFaceWav2Vec2(args.model, None, config=cfg, precision=args.precision, normalize_wav=True, seed=1986).to(dev)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=2986))
    head = head.to(dev)
    L = int(args.seconds * 16000)
    T = cfg.frames(L)
    # `--shards` different inputs, forwarded back to back on ONE stream without a host synchronisation in between (bench.py --verify does
    # exactly this when it recomputes every rank's shard): the workspace then holds ANOTHER input's tensors when a forward starts
    wavs = []
    for sh in range(args.shards):
        g = torch.Generator().manual_seed(1986 + 100 * args.rank_seed + sh)
        wavs.append((0.1 * torch.randn(args.batch, L, generator=g)).clamp_(-1, 1).to(dev))
    frames = [torch.empty((args.batch * T, 4), dtype=torch.int32, device=dev) for _ in wavs]
    lib = enc._lib()
    if args.encoders > 1:
        # upload stress: `--encoders` device objects created one after the other from the same parameters (each creation uploads all
        # weights); every object's forward of input 0 must give the same bits
        import collections
        import hashlib
        objs = [enc] + [enc.replica() for _ in range(args.encoders - 1)]
        hs = []
        for o_ in objs:
            lg = o_.forward_head(wavs[0], head, frames=frames[0])
            hs.append(hashlib.sha1(lg.cpu().numpy().tobytes()).hexdigest()[:12])
        common, n_common = collections.Counter(hs).most_common(1)[0]
        odd = [(i, h) for i, h in enumerate(hs) if h != common]
        print(f"[proc {args.proc_id}] {len(objs)} encoder objects: {n_common} agree on {common}" + (f"; DIFFERENT: {odd}" if odd else ""), flush=True)
        print(f"[proc {args.proc_id}] first-sweep logits sha1 {common}", flush=True)
        return 1 if odd else 0
    first = None
    bad = 0
    offs = None
    for it in range(args.iters):
        cur = []
        for sh, wav in enumerate(wavs):
            logits = enc.forward_head(wav, head, frames=frames[sh])
            ws = enc._sync_device(dev).ws
            cur.append((logits, ws.clone(), frames[sh].clone()))     # stream-ordered copies: no host synchronisation
        torch.cuda.synchronize()
        if first is None:
            slot = enc._sync_device(dev)
            o = (C.c_int64 * 25)()
            n = lib.svt_debug_encoder_layout(slot.handle, args.batch, L, o, 25)
            assert n == 25, _lib.last_error(lib)
            offs = list(o)
            first = cur
            import hashlib   # across processes (same inputs with --same-input): the parent compares these
            print(f"[proc {args.proc_id}] first-sweep logits sha1 " + " ".join(
                hashlib.sha1(c[0].cpu().numpy().tobytes()).hexdigest()[:12] for c in cur), flush=True)
            continue
        for sh, (logits, ws, fr) in enumerate(cur):
            dl = (logits - first[sh][0]).abs().max().item()
            if dl == 0 and torch.equal(ws, first[sh][1]) and torch.equal(fr, first[sh][2]):
                continue
            bad += 1
            diff = (ws != first[sh][1])
            names = []
            live = [(o_, nm) for o_, nm in zip(offs[:24], REGIONS) if o_ >= 0] + [(offs[24], "end")]
            live.sort()
            for (o_, nm), (o2, _) in zip(live[:-1], live[1:]):
                seg = diff[o_:o2]
                if seg.numel() and bool(seg.any()):
                    idx = int(seg.nonzero()[0])
                    names.append(f"{nm}: {int(seg.sum())} of {o2 - o_} bytes, first at +{idx}")
                    if nm == "moments":   # the fp64 statistic itself, both runs
                        k = idx // 8
                        a = ws[o_:o2].view(torch.float64)[k].item()
                        b = first[sh][1][o_:o2].view(torch.float64)[k].item()
                        names[-1] += f" (fp64 #{k}: {a!r} vs first sweep {b!r}, rel {abs(a - b) / max(abs(b), 1e-300):.3e})"
            print(f"[proc {args.proc_id} pid {os.getpid()}] sweep {it} input {sh}: max |dlogit| {dl:.3e}; regions that differ: "
                  + ("; ".join(names) or "none"), flush=True)
    print(f"[proc {args.proc_id}] {args.iters} sweeps over {args.shards} inputs, {bad} forwards differed from the first sweep", flush=True)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--model", default="wav2vec2-base")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--shards", type=int, default=8, help="different inputs forwarded back to back through the same workspace")
    ap.add_argument("--same-input", action="store_true", help="every process gets the same clips")
    ap.add_argument("--rank-seed", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--proc-id", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--encoders", type=int, default=1, help="> 1: upload stress -- this many device objects per process, each must give the same bits")
    args = ap.parse_args()
    if args.rank_seed >= 0:
        sys.exit(child(args))
    procs = []
    for r in range(args.procs):
        cmd = [sys.executable, os.path.abspath(__file__), "--iters", str(args.iters), "--model", args.model, "--batch", str(args.batch),
               "--seconds", str(args.seconds), "--precision", args.precision, "--shards", str(args.shards), "--encoders", str(args.encoders), "--rank-seed", str(0 if args.same_input else r), "--proc-id", str(r)]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True))
    rc = 0
    shas = {}
    for r, p in enumerate(procs):
        out, _ = p.communicate()
        sys.stdout.write(out)
        for ln in out.splitlines():
            if "first-sweep logits sha1" in ln:
                shas[r] = ln.split("sha1", 1)[1].strip()
        rc |= p.returncode
    if args.same_input and len(set(shas.values())) > 1:
        print("PROCESSES DISAGREE on the first sweep (same inputs, same weights):")
        for r, v in sorted(shas.items()):
            print(f"  proc {r}: {v}")
        rc |= 1
    print("REPRODUCIBLE" if rc == 0 else "NOT REPRODUCIBLE")
    sys.exit(1 if rc else 0)


if __name__ == "__main__":
    main()
