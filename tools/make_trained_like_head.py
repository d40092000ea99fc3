"""Fit a TRAINED-LIKE frame head for the parity study (VERDICT r05 "next" #3) -> tests/golden/trained_like_head.pt.

No checkpoint can be downloaded here, so the bench's head is random and every per-frame decision is a near tie.  This script trains the
20-way head the way the recipes do -- frozen encoder, the recipes' loss (BCE on onset with positive weight 15 and on offset, NLL on
the octave and pitch-class log-softmax: MIR_ST500/train_audio_ssl.py:50-75, hparams onset_positive_weight / offset_positive_weight) --
on the exact fp32 features (the CPU oracle) of seeded synthetic singing (svt_speechbrain_amd/synth.py), with L-BFGS and weight decay.
The encoder stays the seeded random one of bench.py (seed 1986): only the head is fitted, which is all a frozen-encoder recipe trains.

Also recorded, from the CPU alone: what the operand-rounding + stored-activation SIMULATION of the 16-bit modes (tools/sim_split.py,
bf16x1s / f16x1s) does to the decisions of that head on the training batch and on a held-out batch.

    python tools/make_trained_like_head.py            # ~2 min on 8 cores
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import svt_oracle as O  # noqa: E402
import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402
from svt_speechbrain_amd.agreement import mode_agreement  # noqa: E402
from svt_speechbrain_amd.decode import FRAME_DTYPE  # noqa: E402
from svt_speechbrain_amd.synth import synth_singing  # noqa: E402
import sim_split  # noqa: E402

MODEL, ENC_SEED, CLIPS, SECONDS, TRAIN_SEED, HELD_SEED = "wav2vec2-base", 1986, 32, 10.0, 2986, 3986
WEIGHT_DECAY = 3e-4


def features(sd, cfg, wav, mode=None):
    keep = (F.linear, F.conv1d, torch.matmul, F.gelu)
    if mode:
        F.linear, F.conv1d, torch.matmul, F.gelu = sim_split.make_ops(mode)
    try:
        with torch.no_grad():
            return O.encoder_forward(sd, cfg, torch.from_numpy(wav))
    finally:
        F.linear, F.conv1d, torch.matmul, F.gelu = keep


def recipe_loss(logits, lab):
    on = F.binary_cross_entropy_with_logits(logits[:, 0], lab[:, 0].float(), pos_weight=torch.tensor(15.0))
    off = F.binary_cross_entropy_with_logits(logits[:, 1], lab[:, 1].float(), pos_weight=torch.tensor(1.0))
    octv = F.nll_loss(F.log_softmax(logits[:, 2:7], -1), lab[:, 2])
    pc = F.nll_loss(F.log_softmax(logits[:, 7:], -1), lab[:, 3])
    return on + off + octv + pc


def frames_of(logits):
    p_on, p_off, octv, pc = O.decode_frames(logits)
    fr = np.zeros(tuple(logits.shape[:2]), dtype=FRAME_DTYPE)
    fr["p_on"], fr["p_off"], fr["octave"], fr["pitch_class"] = p_on.numpy(), p_off.numpy(), octv.numpy(), pc.numpy()
    return fr


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    cfg = S.PRESETS[MODEL]
    sd = W.seeded_encoder_state_dict(cfg, seed=ENC_SEED)
    wav, lab, notes = synth_singing(CLIPS, SECONDS, seed=TRAIN_SEED)
    t0 = time.time()
    feats = features(sd, cfg, wav)
    print(f"oracle features of {CLIPS} x {SECONDS:g} s: {time.time() - t0:.1f} s", flush=True)
    X, Y = feats.reshape(-1, feats.shape[-1]).double(), torch.from_numpy(lab).reshape(-1, 4)
    Wt = torch.zeros(20, X.shape[1], dtype=torch.float64, requires_grad=True)
    b = torch.zeros(20, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.LBFGS([Wt, b], lr=1.0, max_iter=400, history_size=30, line_search_fn="strong_wolfe", tolerance_grad=1e-9, tolerance_change=1e-12)

    def closure():
        opt.zero_grad()
        loss = recipe_loss(X @ Wt.t() + b, Y) + 0.5 * WEIGHT_DECAY * (Wt ** 2).sum()
        loss.backward()
        return loss

    opt.step(closure)
    w32, b32 = Wt.detach().float().contiguous(), b.detach().float().contiguous()
    out = {"model": MODEL, "encoder_seed": ENC_SEED, "clips": CLIPS, "seconds": SECONDS, "train_seed": TRAIN_SEED, "held_out_seed": HELD_SEED,
           "weight_decay": WEIGHT_DECAY, "w.weight": w32, "w.bias": b32, "loss": "BCE(onset, pos_weight 15) + BCE(offset) + NLL(octave) + NLL(pitch class), L-BFGS"}

    def report(tag, wav_, lab_, f_exact):
        logits = O.head_forward(f_exact, w32, b32)
        fr = frames_of(logits)
        lab_t = torch.from_numpy(lab_)
        acc = {"octave": float((torch.from_numpy(fr["octave"].astype(np.int64)) == lab_t[..., 2]).float().mean()),
               "pitch_class": float((torch.from_numpy(fr["pitch_class"].astype(np.int64)) == lab_t[..., 3]).float().mean())}
        srt = logits[..., 7:].sort(-1).values
        margin = (srt[..., -1] - srt[..., -2]).reshape(-1)
        srt_o = logits[..., 2:7].sort(-1).values
        margin_o = (srt_o[..., -1] - srt_o[..., -2]).reshape(-1)
        rec = {"frame_accuracy": acc, "logit_std": float(logits.std()), "loss": float(recipe_loss(logits.reshape(-1, 20).double(), lab_t.reshape(-1, 4))),
               "pitch_class_top2_margin_percentiles_1_5_25_50": [float(np.percentile(margin.numpy(), q)) for q in (1, 5, 25, 50)],
               "octave_top2_margin_percentiles_1_5_25_50": [float(np.percentile(margin_o.numpy(), q)) for q in (1, 5, 25, 50)]}
        for mode in ("bf16x1s", "f16x1s"):
            lg = O.head_forward(features(sd, cfg, wav_, mode), w32, b32)
            ag = mode_agreement(lg, frames_of(lg), logits, fr, 0.4, 0.5, 1 / 49.8)
            rec[f"simulated_{mode}"] = {k: ag[k] for k in ("max_abs_dlogit", "mean_abs_dlogit", "frames", "frames_argmax_mismatch", "clips",
                                                            "clips_with_identical_notes", "reference_notes", "COnPOff_f1", "COnP_f1", "COn_f1")}
        print(tag, rec, flush=True)
        return rec

    out["train_batch"] = report("train", wav, lab, feats)
    wav_h, lab_h, _ = synth_singing(CLIPS, SECONDS, seed=HELD_SEED)
    out["held_out_batch"] = report("held-out", wav_h, lab_h, features(sd, cfg, wav_h))
    # the random head of bench.py on the same clips, for the same table
    hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=2986)
    lg_r = O.head_forward(feats, hd["w.weight"], hd["w.bias"])
    srt = lg_r[..., 7:].sort(-1).values
    out["random_head_pitch_class_top2_margin_percentiles_1_5_25_50"] = [float(np.percentile((srt[..., -1] - srt[..., -2]).numpy(), q)) for q in (1, 5, 25, 50)]
    out["random_head_logit_std"] = float(lg_r.std())
    print("random head:", out["random_head_pitch_class_top2_margin_percentiles_1_5_25_50"], out["random_head_logit_std"])
    torch.save(out, os.path.join(ROOT, "tests", "golden", "trained_like_head.pt"))


if __name__ == "__main__":
    main()
