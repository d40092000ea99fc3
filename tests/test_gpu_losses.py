"""GPU: validation losses, (log-)softmax and the checkpoint reader through the C-ABI, against the golden vectors captured
from the reference's own functions (tests/golden/losses.pt, tests/golden/ckpt_tree/, make_golden.py) — fp32, tolerance
2e-6 relative (the kernels accumulate in double, the reference in fp32 pairwise sums)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402
from svt_speechbrain_amd.config import PRESETS  # noqa: E402

DEV = "cuda:0"
TOL = 2e-6


def _close(a, b, tol=TOL):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return bool(((a - b).abs() <= tol * (1 + b.abs())).all())


def test_bce_loss_vs_reference_golden(golden):
    for c in golden("losses")["bce"]:
        kw = dict(length=None if c["length"] is None else c["length"].to(DEV), reduction=c["reduction"])
        if c["pos_weight"] is not None:
            kw["pos_weight"] = torch.tensor([c["pos_weight"]], device=DEV)
        got = S.bce_loss(c["x"].to(DEV), c["y"].to(DEV), **kw)
        assert got.is_cuda and _close(got, c["expect"]), ("bce", c["reduction"], c["pos_weight"], tuple(c["x"].shape), tuple(c["y"].shape))


def test_nll_loss_vs_reference_golden(golden):
    for c in golden("losses")["nll"]:
        got = S.nll_loss(c["lp"].to(DEV), c["tg"].to(DEV), length=None if c["length"] is None else c["length"].to(DEV),
                         label_smoothing=c["label_smoothing"], reduction=c["reduction"])
        assert _close(got, c["expect"]), ("nll", c["reduction"], c["label_smoothing"], tuple(c["lp"].shape), tuple(c["tg"].shape))


def test_softmax_vs_reference_golden(golden):
    for c in golden("losses")["softmax"]:
        got = S.Softmax(apply_log=c["apply_log"])(c["x"].to(DEV))
        assert got.shape == c["x"].shape
        assert _close(got, c["expect"].reshape(c["x"].shape))


def test_loss_error_behaviour(golden):
    with pytest.raises(ValueError) as e:
        S.bce_loss(torch.zeros(1, 10, device=DEV), torch.zeros(1, 14, device=DEV))
    assert str(e.value) == golden("losses")["truncate_error"]
    with pytest.raises(ValueError):
        S.nll_loss(torch.zeros(1, 10, 3, device=DEV), torch.zeros(1, 20, dtype=torch.long, device=DEV))
    with pytest.raises(ValueError):
        S.bce_loss(torch.zeros(4, device=DEV), torch.zeros(4, device=DEV), length=torch.ones(4, device=DEV))
    with pytest.raises(IndexError):
        S.nll_loss(torch.zeros(2, 4, 3, device=DEV), torch.full((2, 4), 3, dtype=torch.long, device=DEV))
    with pytest.raises(_lib.SvtError):
        S.bce_loss(torch.zeros(2, 3), torch.zeros(2, 3))  # CPU tensors: no fallback


def test_recipe_objective_terms_on_the_forward_outputs():
    # compute_objectives of MIR_ST500/train_audio_ssl.py:50-76 with this package's modules end to end, against the oracle
    from oracle import svt_oracle as O
    from svt_speechbrain_amd import weights as W
    cfg = PRESETS["tiny-group"]
    enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, precision="fp32", seed=5).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=6))
    head = head.to(DEV)
    g = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(3, 8000, generator=g)).clamp_(-1, 1)
    lens = torch.tensor([1.0, 0.6, 0.85])
    logits = head(enc(wav.to(DEV)))
    T = logits.shape[1]
    anno = torch.stack([(torch.rand(3, T + 2, generator=g) < 0.1).float(), (torch.rand(3, T + 2, generator=g) < 0.1).float(),
                        torch.randint(0, 5, (3, T + 2), generator=g).float(), torch.randint(0, 13, (3, T + 2), generator=g).float()], -1)
    sd = {k[len("model."):]: v.detach().cpu() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        ref_logits = O.head_forward(O.encoder_forward(sd, cfg, wav), head.state_dict()["w.weight"].cpu(), head.state_dict()["w.bias"].cpu())
    lsm = S.Softmax(apply_log=True)
    got = [S.bce_loss(logits[:, :, 0], anno[:, :, 0].to(DEV), length=lens.to(DEV), pos_weight=torch.tensor([15.0], device=DEV)),
           S.bce_loss(logits[:, :, 1], anno[:, :, 1].to(DEV), length=lens.to(DEV)),
           S.nll_loss(lsm(logits[:, :, 2:7]), anno[:, :, 2].long().to(DEV), length=lens.to(DEV)),
           S.nll_loss(lsm(logits[:, :, 7:]), anno[:, :, 3].long().to(DEV), length=lens.to(DEV))]
    want = [O.bce_loss(ref_logits[:, :, 0], anno[:, :, 0], length=lens, pos_weight=15.0),
            O.bce_loss(ref_logits[:, :, 1], anno[:, :, 1], length=lens),
            O.nll_loss(O.softmax(ref_logits[:, :, 2:7], True), anno[:, :, 2].long(), length=lens),
            O.nll_loss(O.softmax(ref_logits[:, :, 7:], True), anno[:, :, 3].long(), length=lens)]
    for a, b in zip(got, want):
        assert abs(float(a) - float(b)) < 1e-4 * (1 + abs(float(b)))


def test_checkpoint_reader_loads_reference_files_and_reproduces_reference_logits():
    tree = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_tree")
    want = json.load(open(os.path.join(tree, "expected.json")))
    cfg = PRESETS["tiny-group"]
    enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, precision="fp32", seed=1).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size).to(DEV)
    for sel, key in [(dict(min_key="loss"), "min_loss"), (dict(), "recent"), (dict(max_key="COnPOff_f1"), "max_f1")]:
        chosen = S.Checkpointer(tree, {"wav2vec2": enc, "model": head}).recover_if_possible(device=DEV, **sel)
        assert chosen.path.name == want["picks"][key]
        d = want["digests"][chosen.path.name]
        g = torch.Generator().manual_seed(d["wav_seed"])
        wav = (0.1 * torch.randn(2, 4000, generator=g)).clamp_(-1, 1)
        logits = head(enc(wav.to(DEV))).cpu()
        assert max(abs(float(a) - b) for a, b in zip(logits[0, 0, :4], d["first"])) < 1e-3
        assert abs(float(logits.double().sum()) - d["logits_sum"]) < 1e-3 * logits.numel()
