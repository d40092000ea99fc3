"""How far one numeric mode's transcription is from another's, at the three levels the north star names:

* logits  — max / mean ``|logit - reference logit|`` (the "frame logits within 1e-3" bar);
* frames  — frames whose decoded octave or pitch class differs (the per-frame argmax of
  ``MIR_ST500/train_audio_ssl.py:93-100``), and the largest onset / offset probability difference;
* notes   — the note lists both sides produce through ``frame2note`` (``MIR_ST500/utils.py:82-149``, the library's host
  routine ``svt_frames_to_notes``): clips whose lists are identical, and note-level precision / recall / F-measure of
  the mode's notes against the reference mode's notes with the recipes' scoring (``scoring.score_song``: COnPOff, COnP,
  COn — ``train_audio_ssl.py:119-134``), micro-averaged over the batch.

``bench.py`` runs it after the timed region (timed dtype against the exact-fp32 mode, same batch, same weights) and the
GPU suite asserts the same numbers as bounds.  Host code: numpy on frames already copied from the device.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .decode import FRAME_DTYPE, frames2note_batch
from . import scoring


def _frames_host(frames) -> np.ndarray:
    """(clips, T, 4) int32 device / host tensor, or a structured array -> structured (clips, T)."""
    if isinstance(frames, torch.Tensor):
        frames = frames.detach().cpu().contiguous().numpy()
    fr = np.ascontiguousarray(frames)
    if fr.dtype != FRAME_DTYPE:
        fr = fr.view(FRAME_DTYPE).reshape(fr.shape[:-1])
    return fr


def note_agreement(est_notes, ref_notes) -> Dict[str, float]:
    """Micro-averaged note metrics over a batch of clips: matched / estimated / reference note counts summed over the clips
    before the ratios are taken (a clip without notes on either side contributes its counts, not a 0 / 0)."""
    keys = {"COnPOff": ("Precision", "Recall"), "COnP": ("Precision_no_offset", "Recall_no_offset"), "COn": ("Onset_Precision", "Onset_Recall")}
    matched = {k: 0.0 for k in keys}
    n_est = n_ref = 0
    identical = 0
    for est, ref in zip(est_notes, ref_notes):
        identical += int(est == ref)
        n_est += len(est)
        n_ref += len(ref)
        if est and ref:
            sc = scoring.score_song(est, ref)
            for k, (pk, _) in keys.items():
                matched[k] += sc[pk] * len(est)     # precision x estimated notes = size of the matching
    out: Dict[str, float] = {"clips": len(ref_notes), "clips_with_identical_notes": identical, "notes": n_est, "reference_notes": n_ref}
    for k in keys:
        m = round(matched[k])
        p = m / n_est if n_est else (1.0 if n_ref == 0 else 0.0)
        r = m / n_ref if n_ref else (1.0 if n_est == 0 else 0.0)
        out[f"{k}_precision"], out[f"{k}_recall"] = round(p, 6), round(r, 6)
        out[f"{k}_f1"] = round(0.0 if p + r == 0 else 2 * p * r / (p + r), 6)
    return out


def mode_agreement(logits: torch.Tensor, frames, ref_logits: torch.Tensor, ref_frames, onset_thres: float = 0.4,
                   offset_thres: float = 0.5, frame_size: float = 1 / 49.8, lengths: Optional[np.ndarray] = None,
                   n_octave: Optional[int] = None, n_class: int = 13) -> Dict[str, object]:
    """Everything above for one batch: ``logits`` (clips, T, 20) and ``frames`` (clips, T, 4 x int32, what the fused tail /
    ``svt_decode_frames`` wrote) of the mode under test against the reference mode's.  ``n_octave`` / ``n_class``: the head's layout
    (2 + n_octave + n_class logits per frame; ``n_octave`` defaults to what is left beside ``n_class``), as ``svt_frames_to_notes`` takes them."""
    d = (logits.detach().float() - ref_logits.detach().float()).abs()
    fr, rf = _frames_host(frames), _frames_host(ref_frames)
    if fr.shape != rf.shape:
        raise ValueError(f"mode_agreement: frame arrays of different shape {fr.shape} vs {rf.shape}")
    differ = (fr["octave"] != rf["octave"]) | (fr["pitch_class"] != rf["pitch_class"])
    # a differing frame is a NEAR TIE when, in the reference mode's own logits, the class the other mode picked is within 2e-3 of the
    # winner: two forwards that both sit within the north star's 1e-3 of the truth cannot be expected to break such a tie the same way.
    # Logit layout (svt_decode_frames / svt_frames_to_notes): onset, offset, n_octave octave logits, n_class pitch-class logits.
    width = int(ref_logits.shape[-1])
    if n_octave is None:
        n_octave = width - 2 - n_class
    if n_octave < 1 or n_class < 1 or 2 + n_octave + n_class != width:
        raise ValueError(f"mode_agreement: {width} logits per frame are not 2 + n_octave ({n_octave}) + n_class ({n_class})")
    near_tie = 0
    if differ.any() and d.numel():
        tol = 2e-3
        idx = np.flatnonzero(differ.reshape(-1))
        rl = ref_logits.detach().float().cpu().reshape(-1, width).numpy()[idx]
        ok = np.ones(idx.size, dtype=bool)
        for lo, key in ((2, "octave"), (2 + n_octave, "pitch_class")):
            a = fr[key].reshape(-1)[idx].astype(np.int64)      # what the mode under test picked
            b = rf[key].reshape(-1)[idx].astype(np.int64)      # what the reference mode picked
            gap = np.take_along_axis(rl, (lo + b)[:, None], 1)[:, 0] - np.take_along_axis(rl, (lo + a)[:, None], 1)[:, 0]
            ok &= (a == b) | (gap <= tol)
        near_tie = int(ok.sum())
    notes = frames2note_batch(fr, onset_thres, offset_thres, frame_size, lengths)
    ref_notes = frames2note_batch(rf, onset_thres, offset_thres, frame_size, lengths)
    out: Dict[str, object] = {
        "max_abs_dlogit": float(d.max().item()) if d.numel() else 0.0,
        "mean_abs_dlogit": float(d.mean().item()) if d.numel() else 0.0,
        "frames": int(differ.size),
        "frames_argmax_mismatch": int(differ.sum()),
        "frames_argmax_mismatch_beyond_near_ties": int(differ.sum()) - near_tie,
        "max_abs_dp_onset": float(np.abs(fr["p_on"] - rf["p_on"]).max()) if differ.size else 0.0,
        "max_abs_dp_offset": float(np.abs(fr["p_off"] - rf["p_off"]).max()) if differ.size else 0.0,
    }
    out.update(note_agreement(notes, ref_notes))
    out["meets_1e-3_and_identical_notes"] = bool(out["max_abs_dlogit"] <= 1e-3 and out["frames_argmax_mismatch"] == 0 and
                                                 out["clips_with_identical_notes"] == out["clips"])
    out["meets_1e-3_and_identical_argmax_up_to_near_ties"] = bool(out["max_abs_dlogit"] <= 1e-3 and
                                                                  out["frames_argmax_mismatch_beyond_near_ties"] == 0)
    return out


def trained_like_study(device, modes=("bf16", "fp16", "fp16x3"), fixture: Optional[str] = None, held_out: bool = False) -> Dict[str, object]:
    """What the numeric modes do to a transcription whose decisions have TRAINED-LIKE margins (VERDICT r05 "next" #3).

    The bench batch is noise through a random head: every decision is a near tie.  Here the batch is seeded synthetic singing
    (``synth.synth_singing``) and the head is the fixture ``tests/golden/trained_like_head.pt`` -- the 20-way head fitted with the
    recipes' loss to those clips' labels on the frozen seeded encoder's exact features (``tools/make_trained_like_head.py``).  Per mode,
    against the exact-fp32 mode on the same clips: the ``mode_agreement`` figures, and -- what a user of the recipes cares about -- the
    note-level F1 of each mode's notes against the GROUND-TRUTH notes of the clips, beside the exact mode's own.  ``held_out`` uses the
    fixture's second seed (clips the head was not fitted on)."""
    import os
    import svt_speechbrain_amd as S
    from .synth import synth_singing
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = torch.load(fixture or os.path.join(root, "tests", "golden", "trained_like_head.pt"), weights_only=False)
    cfg = S.PRESETS[fx["model"]]
    wav, lab, notes = synth_singing(fx["clips"], fx["seconds"], seed=fx["held_out_seed"] if held_out else fx["train_seed"])
    truth = [[[float(a), float(b), int(m)] for a, b, m in n] for n in notes]
    x = torch.from_numpy(wav).to(device)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict({"w.weight": fx["w.weight"], "w.bias": fx["w.bias"]})
    head = head.to(device)
    T = cfg.frames(x.shape[1])

    def run(precision):
        enc = S.HuggingFaceWav2Vec2(fx["model"], None, config=cfg, precision=precision, normalize_wav=True, seed=fx["encoder_seed"]).to(device)
        frames = torch.empty((x.shape[0], T, 4), dtype=torch.int32, device=device)
        logits = enc.forward_head(x, head, frames=frames)
        torch.cuda.synchronize()
        del enc
        return logits, frames

    ref_logits, ref_frames = run("fp32")
    ref_notes = frames2note_batch(_frames_host(ref_frames), 0.4, 0.5, 1 / 49.8)
    lab_t = torch.from_numpy(lab)
    fr = _frames_host(ref_frames)
    srt = ref_logits[..., 7:].float().sort(-1).values
    margin = (srt[..., -1] - srt[..., -2]).reshape(-1).cpu().numpy()
    out: Dict[str, object] = {
        "batch": f"{x.shape[0]} x {fx['seconds']:g} s of seeded synthetic singing ({'held-out' if held_out else 'the clips the head was fitted on'})",
        "head": "tests/golden/trained_like_head.pt: recipe loss on the frozen seeded encoder's exact features (tools/make_trained_like_head.py)",
        "ground_truth_notes": sum(len(n) for n in truth),
        "exact_fp32": {"frame_accuracy_octave": float((torch.from_numpy(fr["octave"].astype(np.int64)) == lab_t[..., 2]).float().mean()),
                       "frame_accuracy_pitch_class": float((torch.from_numpy(fr["pitch_class"].astype(np.int64)) == lab_t[..., 3]).float().mean()),
                       "pitch_class_top2_margin_percentiles_1_5_25_50": [round(float(np.percentile(margin, q)), 4) for q in (1, 5, 25, 50)],
                       "notes_vs_ground_truth": {k: v for k, v in note_agreement(ref_notes, truth).items() if k.endswith("_f1") or k in ("notes", "reference_notes")}},
        "modes": {},
    }
    for m in modes:
        lg, frm = run(m)
        ag = mode_agreement(lg, frm, ref_logits, ref_frames, 0.4, 0.5, 1 / 49.8)
        own_notes = frames2note_batch(_frames_host(frm), 0.4, 0.5, 1 / 49.8)
        vs_truth = note_agreement(own_notes, truth)
        out["modes"][m] = {"max_abs_dlogit": ag["max_abs_dlogit"], "mean_abs_dlogit": ag["mean_abs_dlogit"], "frames": ag["frames"],
                           "frames_argmax_mismatch": ag["frames_argmax_mismatch"],
                           "frames_argmax_mismatch_beyond_near_ties": ag["frames_argmax_mismatch_beyond_near_ties"],
                           "clips_with_identical_notes": ag["clips_with_identical_notes"], "clips": ag["clips"],
                           "COnPOff_f1_vs_exact_mode": ag["COnPOff_f1"], "COn_f1_vs_exact_mode": ag["COn_f1"],
                           "meets_1e-3_and_identical_notes": ag["meets_1e-3_and_identical_notes"],
                           "notes_vs_ground_truth": {k: v for k, v in vs_truth.items() if k.endswith("_f1") or k in ("notes", "reference_notes")}}
    return out
