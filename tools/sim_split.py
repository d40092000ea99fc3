"""CPU simulation (build container, test infrastructure): what does a split-operand MFMA mode cost in accuracy?

Every dense product of the oracle (F.linear / F.conv1d / torch.matmul) is replaced by a sum of products of
low-precision pieces of its operands, accumulated in fp32 -- the arithmetic a `precision="bf16x3"` GEMM performs on
v_mfma_f32_16x16x32_bf16 -- and the result is compared with the reference golden of the same case.

    python tools/sim_split.py base_c1 bf16x1 bf16x3 bf16x6 f16x1 f16x3
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import svt_oracle as O  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402
from svt_speechbrain_amd.config import PRESETS  # noqa: E402


def pieces(x, dt, n):
    out, r = [], x.float()
    for _ in range(n):
        p = r.to(dt).float()
        out.append(p)
        r = r - p
    return out


def make_ops(mode):
    """Replacements for (F.linear, F.conv1d, torch.matmul, F.gelu).  A trailing "s" (``bf16x1s`` / ``f16x1s``: the STORED-activation
    simulation, round 6) also rounds what the 16-bit kernels STORE in 16 bits beside the operands: the result of every F.linear (the
    GEMM epilogue writes 16-bit rows; an out-projection / FFN-2 / projection result then meets the fp32 residual arithmetic already
    rounded) and of every GELU (conv activations, positional embedding) -- with a linear -> GELU pair rounded ONCE, behind the GELU,
    as the fused epilogue does.  Convolution and matmul results are not rounded by themselves: their consumers are products (operand
    rounding, already simulated), a GELU, or -- layer-norm extractors -- an fp32 buffer."""
    stored = mode.endswith("s")
    if stored:
        mode = mode[:-1]
    dt = torch.bfloat16 if mode.startswith("bf16") else torch.float16
    nprod = int(mode.split("x")[1])
    # product list: (index of A piece, index of W piece), smallest terms dropped
    plan = {1: [(0, 0)], 3: [(0, 0), (1, 0), (0, 1)], 6: [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (0, 2)]}[nprod]
    npieces = max(max(a, b) for a, b in plan) + 1
    lin, conv, mm, act = F.linear, F.conv1d, torch.matmul, F.gelu
    raw = {}   # id(rounded F.linear result) -> (rounded result, exact result): what a GELU right behind the product starts from

    def linear(x, w, b=None):
        xs, ws = pieces(x, dt, npieces), pieces(w, dt, npieces)
        y = sum(lin(xs[i], ws[j]) for i, j in reversed(plan))
        y = y if b is None else y + b
        if not stored:
            return y
        r = y.to(dt).float()
        raw.clear()
        raw[id(r)] = (r, y)
        return r

    def gelu(x, *a, **kw):
        if not stored:
            return act(x, *a, **kw)
        hit = raw.get(id(x))
        src = hit[1] if hit is not None and hit[0] is x else x
        return act(src, *a, **kw).to(dt).float()

    def conv1d(x, w, b=None, **kw):
        if x.shape[1] == 1:  # conv0 is not an MFMA product in the kernels (fp32 VALU)
            return conv(x, w, b, **kw)
        xs, ws = pieces(x, dt, npieces), pieces(w, dt, npieces)
        y = sum(conv(xs[i], ws[j], None, **kw) for i, j in reversed(plan))
        return y if b is None else y + b[None, :, None]

    def matmul(a, b):
        xs, ws = pieces(a, dt, npieces), pieces(b, dt, npieces)
        return sum(mm(xs[i], ws[j]) for i, j in reversed(plan))

    return linear, conv1d, matmul, gelu


def simulate(fx, mode):
    """One golden case `fx` (tests/golden/<name>.pt, loaded) through the oracle with every dense product replaced by `mode`'s
    sum of piece products -> (max |dlogit|, mean |dlogit|, frames whose octave / pitch-class argmax differs from the reference,
    over all clips, frames in total, then the note level: notes of the reference, and the COnPOff / COnP / COn F1 of the simulated
    notes against the reference's -- svt_speechbrain_amd/agreement.py::note_agreement).  CPU; a few seconds per case."""
    cfg = PRESETS[fx["cfg"]]
    sd = W.seeded_encoder_state_dict(cfg, seed=fx["weight_seed"])
    hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=fx["head_seed"])
    g = torch.Generator().manual_seed(fx["wav_seed"])
    wav = (0.1 * torch.randn(fx["B"], fx["L"], generator=g)).clamp_(-1, 1)
    keep = (F.linear, F.conv1d, torch.matmul, F.gelu)
    F.linear, F.conv1d, torch.matmul, F.gelu = make_ops(mode)
    try:
        with torch.no_grad():
            feats = O.encoder_forward(sd, cfg, wav)
    finally:
        F.linear, F.conv1d, torch.matmul, F.gelu = keep
    with torch.no_grad():
        logits = O.head_forward(feats, hd["w.weight"], hd["w.bias"])
    err = (logits - fx["logits"]).abs()
    p_on, p_off, octv, pc = O.decode_frames(logits)
    mism = 0
    notes, ref_notes = [], []
    for b, d in enumerate(fx["decode"]):
        mism += int(((octv[b] != d["oct"]) | (pc[b] != d["pc"])).sum())
        info = list(zip(p_on[b].numpy(), p_off[b].numpy(), octv[b].tolist(), pc[b].tolist()))
        notes.append(O.frame2note(info, 0.4, 0.5))
        ref_notes.append(d["notes"])
    from svt_speechbrain_amd.agreement import note_agreement
    na = note_agreement(notes, ref_notes)
    return (float(err.max()), float(err.mean()), mism, int(octv.shape[0] * octv.shape[1]), na["reference_notes"], na["COnPOff_f1"],
            na["COnP_f1"], na["COn_f1"])


def main():
    name = sys.argv[1]
    fx = torch.load(os.path.join(ROOT, "tests", "golden", f"{name}.pt"), weights_only=False)
    torch.set_num_threads(8)
    for mode in sys.argv[2:]:
        mx, mn, mism, total, n_ref, f_full, f_nooff, f_on = simulate(fx, mode)
        print(f"{name} {mode}: max|dlogit| {mx:.3e} mean {mn:.3e} argmax mismatches {mism}/{total}; {n_ref} reference notes, "
              f"F1 COnPOff {f_full:.3f} COnP {f_nooff:.3f} COn {f_on:.3f}", flush=True)


if __name__ == "__main__":
    main()
