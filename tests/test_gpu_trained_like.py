"""VERDICT r05 "next" #3: can a 16-bit mode meet "identical notes" when decisions have trained-like margins?  The study of
svt_speechbrain_amd/agreement.py::trained_like_study -- seeded synthetic singing, the 20-way head FITTED to it with the recipes' loss on
the frozen seeded encoder (tests/golden/trained_like_head.pt) -- on the GPU in every fast mode, against the exact-fp32 mode.

What it shows (printed; asserted loosely because the counts are near-tie statistics): a linear head on a RANDOM encoder separates pitch
classes only partly (frame accuracy ~0.7 on its own training clips), so its margins stay small and the plain 16-bit modes still flip
frames -- bf16 ~9 %, fp16 ~1 % -- while the note-level F1 against the clips' GROUND TRUTH is the same within noise in every mode: the
16-bit error sits below the model's own error.  fp16x3 reproduces the exact mode's notes on every clip."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("held_out", [False, True])
def test_trained_like_head_study(held_out):
    from svt_speechbrain_amd.agreement import trained_like_study
    r = trained_like_study(torch.device("cuda:0"), modes=("bf16", "fp16", "fp16x3"), held_out=held_out)
    print(json.dumps(r))
    ex = r["exact_fp32"]
    if not held_out:
        assert ex["frame_accuracy_octave"] > 0.8 and ex["frame_accuracy_pitch_class"] > 0.6      # the head IS fitted (chance: 0.2 / 0.08)
    x3, f16, b16 = r["modes"]["fp16x3"], r["modes"]["fp16"], r["modes"]["bf16"]
    # (the fitted head's logits have 2.4 x the spread of the random head's -- std 4.1 against 1.7 -- and the error scales with it:
    #  1.1e-3 absolute here is 2.8e-4 of the spread, where the goldens' 1.5e-4 is 0.9e-4)
    assert x3["max_abs_dlogit"] < 2e-3 and x3["frames_argmax_mismatch_beyond_near_ties"] == 0
    assert x3["clips_with_identical_notes"] >= x3["clips"] - 1                                   # (a near tie may move one note)
    assert f16["frames_argmax_mismatch"] < b16["frames_argmax_mismatch"] < 0.2 * b16["frames"]
    assert f16["frames_argmax_mismatch"] < 0.03 * f16["frames"]
    # against the ground truth every mode transcribes as well as the exact one, to within what moving a few notes costs
    f_exact = ex["notes_vs_ground_truth"]["COn_f1"]
    for m in (x3, f16, b16):
        assert abs(m["notes_vs_ground_truth"]["COn_f1"] - f_exact) < 0.05, (m, f_exact)
