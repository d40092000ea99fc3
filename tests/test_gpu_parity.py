"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI
(ctypes -> libsvt_mi355.so), against the oracle and against the golden vectors captured from the
reference.  Tolerances: fp32 parity mode — logits within 1e-3 (north star) and identical per-frame
argmax / note sequences; bf16 throughput mode — reported error bound + decode agreement rate."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import svt_speechbrain_amd as S  # noqa: E402
from oracle import svt_oracle as O  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402
from svt_speechbrain_amd.config import PRESETS  # noqa: E402

DEV = "cuda:0"


def synth_wav(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)


def build(cfg_name, weight_seed, head_seed, precision):
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision=precision, seed=weight_seed).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=head_seed))
    return cfg, enc, head.to(DEV)


def golden_wav(fx):
    if "wav" in fx:
        return fx["wav"]
    wav = synth_wav(fx["B"], fx["L"], fx["wav_seed"])
    if fx.get("lens"):
        for b, n in enumerate(fx["lens"]):
            wav[b, n:] = 0
    return wav


def check_decode(logits_gpu, fx, exact=True):
    frames = S.decode_frames(logits_gpu)
    mism = 0
    for b, d in enumerate(fx["decode"]):
        o = torch.from_numpy(frames["octave"][b].astype(np.int64))
        p = torch.from_numpy(frames["pitch_class"][b].astype(np.int64))
        mism += int(((o != d["oct"]) | (p != d["pc"])).sum())
        if exact:
            assert torch.equal(o, d["oct"]) and torch.equal(p, d["pc"])
            notes = S.frame2note(S.frames_to_info(frames[b]), 0.4, 0.5)
            assert notes == d["notes"], f"clip {b}: note sequence differs from the reference"
    return mism


# parity-grade modes, held to the north-star bar (logits within 1e-3 of the reference, identical per-frame argmax and
# identical note lists): exact fp32 MFMA and "fp16x3", the split-operand mode on the 16-bit matrix pipe (fp32 operands in
# memory, cut into fp16 (hi, lo) pieces inside the product kernels, three MFMAs per block: csrc/gemm.hip; ~2^-22 relative
# operand error).  "bf16x3" (bf16 pieces, ~2^-17) also lands within 1e-3 on every golden (measured 2.9e-4 .. 8.1e-4) but
# that is 3-8x closer to the bar than fp16x3 (5e-5 .. 1.5e-4) and one near-tie frame of large_b2 decodes differently, so
# it is held to the logit bar plus an argmax agreement rate, not to bit-identical decode.
PARITY_MODES = ["fp32", "fp16x3", "bf16x3"]
EXACT_DECODE_MODES = ("fp32", "fp16x3")


@pytest.mark.parametrize("prec", PARITY_MODES)
@pytest.mark.parametrize("name", ["tiny_group", "tiny_layer", "tiny_hubert", "tiny_group_ragged", "tiny_data2vec", "tiny_wavlm", "tiny_wavlm_stable", "tiny_hubert_bn"])
def test_tiny_fp32_vs_reference_golden(golden, name, prec):
    fx = golden(name)
    cfg, enc, head = build(fx["cfg"], fx["weight_seed"], fx["head_seed"], prec)
    wav = golden_wav(fx).to(DEV)
    feats = enc(wav)
    logits = head(feats)
    assert feats.shape == fx["feats"].shape
    assert (feats.cpu() - fx["feats"]).abs().max() < 1e-3
    assert (logits.cpu() - fx["logits"]).abs().max() < 1e-3
    mism = check_decode(logits, fx, exact=prec in EXACT_DECODE_MODES)
    assert mism <= 1


@pytest.mark.parametrize("prec", PARITY_MODES)
@pytest.mark.parametrize("name", ["base_c1", "base_b2", "large_c1", "hubert_large_c1", "data2vec_base_c1", "wavlm_base_c1",
                                  "large_b2", "hubert_large_b2"])  # *_b2: 2 x 10 s, the clip length of every BASELINE config
def test_full_size_fp32_vs_reference_golden(golden, name, prec):
    fx = golden(name)
    cfg, enc, head = build(fx["cfg"], fx["weight_seed"], fx["head_seed"], prec)
    wav = golden_wav(fx)
    assert hashlib.sha256(wav.numpy().tobytes()).hexdigest() == fx["wav_sha256"]
    feats = enc(wav.to(DEV))
    logits = head(feats)
    assert feats.shape[1] == fx["T"]
    assert (feats.cpu()[:, ::25, ::16] - fx["feats_strided"]).abs().max() < 1e-3
    err = (logits.cpu() - fx["logits"]).abs().max().item()
    print(f"{prec}[{name}]: max|dlogit| vs the reference golden {err:.3e}")
    assert err < 1e-3, err
    mism = check_decode(logits, fx, exact=prec in EXACT_DECODE_MODES)
    assert mism <= 2, mism   # bf16x3: at most a near-tie frame or two per golden (fp32 / fp16x3: exact, asserted above)


@pytest.mark.parametrize("prec", ["fp16x3", "fp16"])
@pytest.mark.parametrize("gain", [2.0 ** -10, 3e-4, 2.0 ** 17])
def test_quiet_and_loud_audio_with_ieee_half_pieces(golden, prec, gain):
    """Conv layer 0 of the GroupNorm extractor cuts the RAW samples into 16-bit (hi, lo) pieces (csrc/conv0_mfma.hip): with IEEE-half
    pieces audio peaking at 1e-4 would have subnormal lo pieces, samples above 65 504 would overflow and the folded coefficients
    (~ 1 / level) would leave the range the other way, had the kernel not scaled both sides by powers of two.  Held to the oracle ON THE
    SAME quiet / loud clip (a gain does change the reference's output: the eps of the whole-batch waveform norm is absolute): fp16x3 to
    the 1e-3 bar and identical argmax, fp16 to the error it has at gain 1."""
    fx = golden("base_c1")
    cfg, enc, head = build(fx["cfg"], fx["weight_seed"], fx["head_seed"], prec)
    wav = golden_wav(fx) * gain
    sd = W.seeded_encoder_state_dict(cfg, seed=fx["weight_seed"])
    hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=fx["head_seed"])
    with torch.no_grad():
        ref = O.head_forward(O.encoder_forward(sd, cfg, wav), hd["w.weight"], hd["w.bias"])
    out = head(enc(wav.to(DEV))).cpu()
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max().item()
    print(f"{prec} gain {gain:g}: max|dlogit| vs the oracle on the same clip {err:.2e}")
    if prec == "fp16x3":
        assert err < 1e-3
        assert torch.equal(out[..., 2:7].argmax(-1), ref[..., 2:7].argmax(-1)) and torch.equal(out[..., 7:].argmax(-1), ref[..., 7:].argmax(-1))
    else:
        assert err < 1.6 * 0.0534 + 1e-3   # the f16 operand-rounding simulation of this golden (tests/golden/sim_bounds.json)


_SIM_CACHE = {}


def simulated_error(name, fx, mode):
    """What ROUNDING THE OPERANDS of every dense product to 16 bits costs on this golden, measured on the CPU by the oracle
    (tools/sim_split.py: every F.linear / F.conv1d / matmul of oracle/svt_oracle.py replaced by a product of operands rounded to
    bf16 (`bf16x1`) or IEEE half (`f16x1`), fp32 accumulation, everything else fp32) -> (max |dlogit|, mean |dlogit|, frames with a
    different octave / pitch-class argmax, frames).  The throughput modes are held to a multiple of THIS, not to one kernel's
    measured figures: a change of summation order inside a kernel moves near-tie frames without being a regression."""
    key = (name, mode)
    if key not in _SIM_CACHE:
        # tests/golden/sim_bounds.json holds the simulation's figures for every bound case (tests/golden/make_sim_bounds.py writes it, the
        # CPU suite re-derives the cases that take seconds: tests/test_host_cpu.py); a case it does not hold is simulated here
        table = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sim_bounds.json")
        cached = json.load(open(table)).get(name, {}).get(mode) if os.path.exists(table) else None
        if cached is not None:
            _SIM_CACHE[key] = tuple(cached)
        else:
            tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
            if tools not in sys.path:
                sys.path.insert(0, tools)
            import sim_split
            _SIM_CACHE[key] = sim_split.simulate(fx, mode)
    return _SIM_CACHE[key]


def check_16bit_mode_bound(tag, logits, fx, sim, stored=None, measured=None):
    """GPU 16-bit-operand mode against two CPU simulations of the same golden (tools/sim_split.py, committed as
    tests/golden/sim_bounds.json): `sim` rounds the OPERANDS of every dense product to 16 bits and nothing else; `stored` (round 6,
    modes bf16x1s / f16x1s) also rounds what the kernels STORE in 16 bits -- every GEMM result and every GELU result, a fused
    linear -> GELU pair once.  The stored simulation is the yardstick: over nine goldens x two modes the kernels' mean |dlogit| is
    0.94-1.18 x its figure (1.0-1.25 x the operand-only one), max |dlogit| 0.83-1.30 x (a noisy statistic: the largest of ~20 000 values).
    Bounds: mean <= 1.25 x, max <= 1.4 x, frames whose argmax differs <= 1.2 x + 2.5 sigma of a count of that size + 2 (near ties flip
    like a Poisson process).  wav2vec2-large 2 x 10 s is the one case the simulation does not explain to within its noise: operand
    rounding alone predicts 14 frames, + stored activations 19, the kernels measure 31 (2.7 sigma above 19; mean |dlogit| 1.16 x):
    what is left is outside both simulations -- the multi-frame positional convolution's 16-bit partial rows and the polynomial GELU are
    the candidates -- and stays inside the bound only through the sigma term.  `measured` (tests/golden/kernel_16bit_measured_r06.json,
    this round's kernels on MI355X) adds a REGRESSION ceiling beside the physics one: at most the measured count + 2 sigma + 1 and
    1.15 x the measured mean.
    Note level: the notes frame2note makes of the mode's frames against the REFERENCE's notes of the golden, scored like the recipes
    score a transcription (COnPOff / COnP / COn F1, svt_speechbrain_amd/agreement.py): at least the operand simulation's F1 minus what
    moving two notes of this many would cost, minus 0.03."""
    from svt_speechbrain_amd.agreement import note_agreement
    s_max, s_mean, s_mism, total, n_ref, s_f_full, s_f_nooff, s_f_on = sim
    y_max, y_mean, y_mism = (stored[0], stored[1], stored[2]) if stored is not None else (s_max, s_mean, s_mism)
    err = (logits.cpu() - fx["logits"]).abs()
    mism = check_decode(logits, fx, exact=False)
    frames = S.decode_frames(logits)
    notes = [S.frame2note(S.frames_to_info(frames[b]), 0.4, 0.5) for b in range(len(fx["decode"]))]
    na = note_agreement(notes, [d["notes"] for d in fx["decode"]])
    print(f"{tag}: max|dlogit| {err.max():.4f} mean {err.mean():.4f} (simulation {s_max:.4f} / {s_mean:.4f}, + stored activations {y_max:.4f} / {y_mean:.4f}; "
          f"logit std {fx['logits'].std():.2f}); frames with a different octave/pitch-class argmax: {mism}/{total} (simulation {s_mism}, + stored {y_mism}); "
          f"notes {na['notes']} vs {n_ref} of the reference, F1 COnPOff {na['COnPOff_f1']:.3f} COnP {na['COnP_f1']:.3f} COn {na['COn_f1']:.3f} "
          f"(simulation {s_f_full:.3f} / {s_f_nooff:.3f} / {s_f_on:.3f}), clips with identical notes {na['clips_with_identical_notes']}/{na['clips']}")
    assert err.max() < 1.4 * y_max + 1e-3, (float(err.max()), y_max)
    assert err.mean() < 1.25 * y_mean + 1e-4, (float(err.mean()), y_mean)
    assert mism <= 1.2 * y_mism + 2.5 * (y_mism ** 0.5) + 2, (mism, y_mism, total)
    if measured is not None:
        m_max, m_mean, m_mism, m_total = measured
        assert m_total == total
        assert mism <= m_mism + 2.0 * max(1.0, m_mism) ** 0.5 + 1 and err.mean() < 1.15 * m_mean + 1e-4, ("regression against round 6", mism, m_mism, float(err.mean()), m_mean)
    assert na["reference_notes"] == n_ref
    slack = 0.03 + 2.0 / max(1, n_ref)
    assert na["COnPOff_f1"] >= s_f_full - slack and na["COnP_f1"] >= s_f_nooff - slack and na["COn_f1"] >= s_f_on - slack, (na, sim)
    return na


def measured_16bit(name, mode):
    """this round's kernels on MI355X (tests/golden/kernel_16bit_measured_r06.json; written from the printed figures of a GPU run)"""
    table = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kernel_16bit_measured_r06.json")
    if not os.path.exists(table):
        return None
    return json.load(open(table)).get(name, {}).get(mode)


BOUND_CASES = ["tiny_group", "tiny_layer", "base_c1", "base_b2", "large_c1", "data2vec_base_c1", "wavlm_base_c1", "large_b2",
               "hubert_large_b2"]   # *_b2: 2 x 10 s -> multi-frame positional conv


@pytest.mark.parametrize("name", BOUND_CASES)
def test_bf16_mode_error_bound(golden, name):
    """bf16 MFMA operands, fp32 accumulate / residual / norms: error bounded by a multiple of what rounding the operands to bf16 costs
    in a CPU simulation of the same case (tools/sim_split.py, not a frozen kernel measurement: see check_16bit_mode_bound) -- i.e. the price of bf16 operands on
    these random-init weights (base 5 s clip: simulation 0.38 max / 0.080 mean / 15 of 249 frames; kernels 0.44 / 0.081 / 18)."""
    fx = golden(name)
    cfg, enc, head = build(fx["cfg"], fx["weight_seed"], fx["head_seed"], "bf16")
    wav = golden_wav(fx).to(DEV)
    logits = head(enc(wav))
    assert torch.isfinite(logits).all()
    check_16bit_mode_bound(f"bf16[{name}]", logits, fx, simulated_error(name, fx, "bf16x1"), simulated_error(name, fx, "bf16x1s"), measured_16bit(name, "bf16"))


@pytest.mark.parametrize("name", BOUND_CASES)
def test_fp16_mode_error_bound(golden, name):
    """`precision="fp16"`: the 16-bit throughput mode with IEEE-half operands (the second build of the library,
    libsvt_mi355_f16.so: same kernels, same MFMA rate, three more mantissa bits than bf16), held to the f16 operand-rounding
    simulation the same way: about one eighth of the bf16 mode's error on every golden (base_c1: max |dlogit| 0.055 vs 0.44, mean
    0.0099 vs 0.081, 4 vs 18 of 249 frames with a different octave / pitch-class argmax) at 98 % of its clips/s."""
    fx = golden(name)
    cfg, enc, head = build(fx["cfg"], fx["weight_seed"], fx["head_seed"], "fp16")
    wav = golden_wav(fx).to(DEV)
    logits = head(enc(wav))
    assert torch.isfinite(logits).all()
    check_16bit_mode_bound(f"fp16[{name}]", logits, fx, simulated_error(name, fx, "f16x1"), simulated_error(name, fx, "f16x1s"), measured_16bit(name, "fp16"))
    # the fused tail serves this build too, and agrees with encoder -> head
    fused = enc.forward_head(wav, head) if S.HuggingFaceWav2Vec2.can_fuse_head(head) else logits
    assert (fused - logits).abs().max().item() < 2e-4


def test_encoder_flags_and_batch_coupling():
    """normalize_wav / output_norm off, and the whole-batch norms couple clips."""
    cfg = PRESETS["tiny-group"]
    sd = W.seeded_encoder_state_dict(cfg, seed=3)
    wav = synth_wav(3, 5000, 9)
    for nw, on in [(False, False), (True, False), (False, True)]:
        enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, precision="fp32", seed=3, normalize_wav=nw,
                                    output_norm=on).to(DEV)
        with torch.no_grad():
            ref = O.encoder_forward(sd, cfg, wav, normalize_wav=nw, output_norm=on)
        out = enc(wav.to(DEV)).cpu()
        assert (out - ref).abs().max() < 1e-3, (nw, on)
    enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, precision="fp32", seed=3).to(DEV)
    a = enc(wav.to(DEV))[0].cpu()
    b = enc(wav[:1].to(DEV))[0].cpu()
    assert (a - b).abs().max() > 1e-3  # same clip, different batch-mates -> different output, as in the reference


def test_state_dict_roundtrip_and_old_weight_norm_keys():
    cfg = PRESETS["tiny-layer"]
    enc = S.HuggingFaceWav2Vec2("tiny-layer", None, config=cfg, precision="fp32", seed=1).to(DEV)
    wav = synth_wav(1, 3000, 2).to(DEV)
    before = enc(wav).cpu()
    sd_new = {"model." + k: v for k, v in W.seeded_encoder_state_dict(cfg, seed=77, old_weight_norm_keys=True).items()}
    enc.load_state_dict(sd_new, strict=True)
    after = enc(wav).cpu()
    assert (before - after).abs().max() > 1e-2
    with torch.no_grad():
        ref = O.encoder_forward(W.seeded_encoder_state_dict(cfg, seed=77), cfg, wav.cpu())
    assert (after - ref).abs().max() < 1e-3
    keys = list(enc.state_dict().keys())
    assert all(k.startswith("model.") for k in keys) and len(keys) == len(sd_new)


@pytest.mark.parametrize("name", ["fusion_eq", "fusion_pad", "fusion_trunc"])
@pytest.mark.parametrize("prec,tol", [("fp32", 1e-3), ("fp16x3", 1e-3), ("bf16x3", 1e-3), ("bf16", 0.12)])
def test_fusion_vs_reference_golden(golden, name, prec, tol):
    fx = golden(name)
    fus = S.FusionRCA(precision=prec, seed=fx["weight_seed"]).to(DEV)
    g = torch.Generator().manual_seed(fx["in_seed"])
    a = torch.randn(fx["B"], fx["T1"], 1024, generator=g)
    v = torch.randn(fx["B"], fx["T2"], 1024, generator=g)
    out = fus(a.to(DEV), v.to(DEV)).cpu()
    assert out.shape == (fx["B"], fx["T1"], 1024)
    assert (out[:, ::7, ::5] - fx["out_strided"]).abs().max() < tol
    assert (out[:, :4] - fx["out_first"]).abs().max() < tol


def test_fusion_state_dict_keys():
    fus = S.FusionRCA()
    want = set(W.fusion_param_shapes().keys())
    assert set(fus.state_dict().keys()) == want


def test_linear_general_and_head():
    g = torch.Generator().manual_seed(0)
    for n_in, n_out, rows in [(768, 20, 499), (64, 20, 7), (128, 96, 300), (100, 50, 33)]:
        lin = S.Linear(n_out, input_size=n_in)
        x = torch.randn(3, rows, n_in, generator=g)
        ref = torch.nn.functional.linear(x, lin.w.weight, lin.w.bias)
        out = lin.to(DEV)(x.to(DEV)).cpu()
        assert out.shape == ref.shape
        assert (out - ref).abs().max() < 1e-4, (n_in, n_out)
    nb = S.Linear(20, input_shape=[2, 5, 64], bias=False)
    x = torch.randn(2, 5, 64, generator=g)
    assert (nb.to(DEV)(x.to(DEV)).cpu() - x @ nb.w.weight.detach().cpu().t()).abs().max() < 1e-4


@pytest.mark.parametrize("n_in", [512, 768, 1024])
def test_frame_head_kernel_rows_and_widths(n_in):
    # the LDS-resident frame-head kernel (K in {512,768,1024}, N <= 32): row counts around its 4-rows-per-wave /
    # 16-rows-per-workgroup blocking and output widths around the 16-lane result packing
    g = torch.Generator().manual_seed(n_in)
    for n_out in (1, 15, 16, 17, 20, 32):
        lin = S.Linear(n_out, input_size=n_in)
        wt, bs = lin.w.weight.detach().cpu().double(), lin.w.bias.detach().cpu().double()
        dl = lin.to(DEV)
        for rows in (1, 3, 4, 5, 16, 17, 8191 + 16 * 512):
            x = torch.randn(rows, n_in, generator=g)
            ref = torch.nn.functional.linear(x.double(), wt, bs).float()
            out = dl(x.to(DEV)).cpu()
            assert out.shape == ref.shape
            assert (out - ref).abs().max() < 2e-5, (n_in, n_out, rows)


def test_decode_frames_first_max_and_sigmoid():
    lg = torch.zeros(4, 20)
    lg[0, 2:7] = torch.tensor([1.0, 3.0, 3.0, 0.0, -1.0])  # tie -> first max (index 1)
    lg[1, 7:] = 5.0                                          # all equal -> 0
    lg[2, 0], lg[2, 1] = 2.0, -2.0
    lg[3, 6], lg[3, 19] = 9.0, 9.0
    fr = S.decode_frames(lg.to(DEV))
    assert fr["octave"].tolist() == [1, 0, 0, 4]
    assert fr["pitch_class"].tolist() == [0, 0, 0, 12]
    assert abs(fr["p_on"][2] - float(torch.sigmoid(torch.tensor(2.0)))) < 1e-6
    assert abs(fr["p_off"][2] - float(torch.sigmoid(torch.tensor(-2.0)))) < 1e-6


def test_ctc_greedy_vs_reference_golden(golden):
    for k, c in golden("ctc").items():
        got = S.ctc_greedy_decode(c["probs"].to(DEV), c["lens"].to(DEV), c["blank"])
        assert got == c["expect"], k


def test_fbank_vs_reference_golden(golden):
    fb = S.Fbank()
    for k, c in golden("fbank").items():
        out = fb(c["wav"].to(DEV)).cpu()
        assert out.shape == c["feats"].shape
        assert (out - c["feats"]).abs().max() < 5e-3, k


def test_errors_are_exceptions():
    from svt_speechbrain_amd import _lib
    cfg = PRESETS["tiny-group"]
    enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, precision="fp32").to(DEV)
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 100, device=DEV))  # shorter than the receptive field
    with pytest.raises(_lib.SvtError):
        enc(torch.zeros(1, 4000))  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        enc(torch.zeros(4000, device=DEV))


@pytest.mark.parametrize("cfg_name,B,L", [("wav2vec2-base", 4, 160000), ("wav2vec2-base", 3, 52345),
                                          ("wav2vec2-base", 32, 160000)])  # the last one is BASELINE config C2 itself
def test_full_size_properties(cfg_name, B, L):
    """Size-independent properties at BASELINE sizes (oracle too slow to run per test):
    determinism (bitwise), whole-batch output norm (zero mean / unit variance), permutation equivariance
    over clips (the batch statistics are permutation invariant), fp32 vs bf16 agreement."""
    cfg = PRESETS[cfg_name]
    enc32 = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision="fp32", seed=5).to(DEV)
    enc16 = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision="bf16", seed=5).to(DEV)
    wav = synth_wav(B, L, 123).to(DEV)
    a = enc32(wav)
    b = enc32(wav)
    assert a.shape == (B, cfg.frames(L), cfg.hidden_size)
    assert torch.isfinite(a).all()
    # the three cross-workgroup statistics are summed in a fixed order since round 5 (kernels.hip, last_workgroup): bit for bit
    assert torch.equal(a, b)
    assert abs(a.mean().item()) < 1e-4 and abs(a.var(unbiased=False).item() - 1.0) < 1e-3
    perm = torch.arange(B - 1, -1, -1, device=DEV)
    c = enc32(wav[perm])
    assert (c[perm] - a).abs().max() < 2e-4
    d = enc16(wav)
    print(f"bf16 vs fp32 feats ({cfg_name},B={B},L={L}): mean|d| {(d - a).abs().mean():.4f} max {(d - a).abs().max():.4f}")
    assert (d - a).abs().mean() < 0.08 and (d - a).abs().max() < 1.5


def test_song_transcriber_matches_oracle_per_utterance(tmp_path):
    """Song-level path: 12.3 s song -> 2 utterances (5 s + 7.3 s), batch-1 forwards, frames
    concatenated, one frame2note; features concatenated and written like extract_ssl_feats.py."""
    cfg = PRESETS["tiny-group"]
    sd = W.seeded_encoder_state_dict(cfg, seed=31)
    hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=32)
    enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, precision="fp32", seed=31).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(hd)
    head = head.to(DEV)
    song = synth_wav(1, int(12.3 * 16000), 99)[0]
    notes, feats = S.SongTranscriber(enc, head).transcribe(song.to(DEV), return_feats=True)
    bounds = S.utterance_bounds(song.shape[0])
    assert len(bounds) == 2
    info, ref_feats = [], []
    with torch.no_grad():
        for lo, hi in bounds:
            f = O.encoder_forward(sd, cfg, song[lo:hi][None])
            lg = O.head_forward(f, hd["w.weight"], hd["w.bias"])
            p_on, p_off, octv, pc = O.decode_frames(lg)
            info += list(zip(p_on[0].numpy(), p_off[0].numpy(), octv[0].tolist(), pc[0].tolist()))
            ref_feats.append(f[0])
    assert notes == O.frame2note(info, 0.4, 0.5)
    ref_feats = torch.cat(ref_feats)
    assert feats.shape == ref_feats.shape and (feats.cpu() - ref_feats).abs().max() < 1e-3
    path = S.save_song_features(feats, str(tmp_path / "song"))
    assert path.endswith("noise_data/clean_feats.pt") and torch.load(path).shape == ref_feats.shape


def test_evaluation_loop_manifest_to_scores():
    """The recipes' test stage end to end with this package only (MIR_ST500/train_audio_ssl.py:28-137): utterance plan
    and slicing -> PaddedBatch (batch 1) -> encoder + head -> the four validation losses -> frames -> notes at the last
    utterance -> COnPOff / COnP / COn scores.  Reference = the same loop on the oracle; fp32 mode must reproduce its notes,
    hence score 1.0 against them, and its loss terms."""
    from svt_speechbrain_amd import dataio as D, scoring as SC
    cfg = PRESETS["tiny-group"]
    sd = W.seeded_encoder_state_dict(cfg, seed=41)
    hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=42)
    enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, precision="fp32", seed=41).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(hd)
    amt = S.AMTForward({"wav2vec2": enc, "model": head.to(DEV)})
    lsm = S.Softmax(apply_log=True)
    duration = 12.3
    song = synth_wav(1, int(duration * 16000), 7)[0]
    g = torch.Generator().manual_seed(8)
    n_frames = round(duration * 49.8)
    song_anno = torch.stack([(torch.rand(n_frames, generator=g) < 0.05).float(), (torch.rand(n_frames, generator=g) < 0.05).float(),
                             torch.randint(0, 5, (n_frames,), generator=g).float(), torch.randint(0, 13, (n_frames,), generator=g).float()], 1)
    plan = D.plan_utterances(duration)
    ref_info, got_losses, ref_losses, notes = [], [], [], None
    for uid in range(1, len(plan) + 1):
        ex = {"id": f"s_{uid}", "sig": D.slice_audio(song, uid, len(plan)), "anno": D.slice_annotation(song_anno, uid, len(plan)),
              "cur_utter": uid, "all_utter": len(plan)}
        batch = D.PaddedBatch([ex]).to(DEV)
        wavs, wav_lens = batch.sig
        on, off, octl, pcl, lens = amt.compute_forward(wavs, wav_lens)
        anno, _ = batch.anno
        got_losses.append(float(S.bce_loss(on, anno[:, :, 0], length=lens, pos_weight=torch.tensor([15.0], device=DEV))
                                + S.bce_loss(off, anno[:, :, 1], length=lens)
                                + S.nll_loss(lsm(octl), anno[:, :, 2].long(), length=lens)
                                + S.nll_loss(lsm(pcl), anno[:, :, 3].long(), length=lens)))
        notes = amt.decode_utterance(amt.last_logits, last_of_song=batch.cur_utter.item() == batch.all_utter.item())
        with torch.no_grad():
            rl = O.head_forward(O.encoder_forward(sd, cfg, ex["sig"][None]), hd["w.weight"], hd["w.bias"])
        a = ex["anno"][None]
        one = torch.ones(1)
        ref_losses.append(float(O.bce_loss(rl[:, :, 0], a[:, :, 0], length=one, pos_weight=15.0) + O.bce_loss(rl[:, :, 1], a[:, :, 1], length=one)
                                + O.nll_loss(O.softmax(rl[:, :, 2:7], True), a[:, :, 2].long(), length=one)
                                + O.nll_loss(O.softmax(rl[:, :, 7:], True), a[:, :, 3].long(), length=one)))
        p_on, p_off, octv, pc = O.decode_frames(rl)
        ref_info += list(zip(p_on[0].numpy(), p_off[0].numpy(), octv[0].tolist(), pc[0].tolist()))
    ref_notes = O.frame2note(ref_info, 0.4, 0.5)
    assert notes == ref_notes and len(notes) > 0
    for a_, b_ in zip(got_losses, ref_losses):
        assert abs(a_ - b_) < 1e-4 * (1 + abs(b_))
    scores = SC.score_song(notes, ref_notes)
    assert scores["F-measure"] == 1.0 and scores["F-measure_no_offset"] == 1.0 and scores["Onset_F-measure"] == 1.0
    # and a perturbed transcription scores below 1
    worse = [[n[0] + 0.2, n[1] + 0.2, n[2]] for n in notes[: len(notes) // 2]] + notes[len(notes) // 2:]
    assert SC.score_song(worse, ref_notes)["Onset_F-measure"] < 1.0


def test_two_streams_bitwise_identical_to_one():
    """bench.py issues successive steps on two HIP streams (two encoder objects).  Kernels that share the chip run with
    different timing (the LDS-DMA pipelines order their reads by counted waits, not by luck): every overlapped result must
    be bit-identical to the single-stream result."""
    cfg = PRESETS["wav2vec2-base"]
    encs = [S.HuggingFaceWav2Vec2("wav2vec2-base", None, config=cfg, precision="bf16", seed=3).to(DEV) for _ in range(2)]
    head = S.Linear(20, input_size=cfg.hidden_size).to(DEV)
    wavs = [synth_wav(4, 160000, 50 + i).to(DEV) for i in range(2)]
    ref = [head(encs[0](w)).clone() for w in wavs]
    torch.cuda.synchronize()
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    streams[1].wait_stream(streams[0])
    outs = []
    for it in range(12):
        i = it % 2
        with torch.cuda.stream(streams[i]):
            outs.append((i, head(encs[i](wavs[i]))))
    torch.cuda.synchronize()
    for i, o in outs:
        assert torch.equal(o, ref[i]), "overlapped step differs from the single-stream result"


def test_fbank_deltas_and_context_vs_reference_golden(golden):
    """Fbank(deltas=True, context=True) pieces: svt_deltas / svt_context_window against the reference's Deltas / ContextWindow
    forward outputs, and the assembled lobe against the oracle chain."""
    from svt_speechbrain_amd import _lib
    lib = _lib.load()
    fx = golden("fbank_ext")
    st = torch.cuda.current_stream().cuda_stream
    for c in fx["deltas"]:
        x = c["x"].to(DEV).contiguous()
        B, T, Cc = x.shape
        out = torch.empty_like(x)
        _lib.check(lib.svt_deltas(x.data_ptr(), Cc, B, T, Cc, 5, out.data_ptr(), Cc, 0, st), "svt_deltas")
        assert (out.cpu() - c["expect"]).abs().max() < 1e-6
    for c in fx["context"]:
        x = c["x"].to(DEV).contiguous()
        B, T, Cc = x.shape
        ctx = c["left"] + c["right"] + 1
        out = torch.empty(B, T, Cc * ctx, device=DEV)
        _lib.check(lib.svt_context_window(x.data_ptr(), B, T, Cc, c["left"], c["right"], out.data_ptr(), 0, st), "svt_context_window")
        assert torch.equal(out.cpu(), c["expect"])
    wav = synth_wav(2, 16000, 77)
    fb = S.Fbank(deltas=True, context=True, left_frames=3, right_frames=2).to(DEV)(wav.to(DEV)).cpu()
    base = O.fbank(wav)
    d1 = O.deltas(base)
    want = O.context_window(torch.cat([base, d1, O.deltas(d1)], dim=2), 3, 2)
    assert fb.shape == want.shape == (2, 101, 120 * 6)
    assert (fb - want).abs().max() < 5e-3   # dB scale, same bound as the plain Fbank test


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-3), ("bf16", 0.6)])
def test_hubert_base_layout_vs_oracle(prec, tol):
    """HuBERT-base geometry class (GroupNorm conv stack, post-LN encoder, NO feature-projection LayerNorm — the
    `hubert-base-ls960` preset) at tiny width, against the oracle (no reference golden for this combination)."""
    import dataclasses
    cfg = dataclasses.replace(PRESETS["tiny-group"], name="tiny-hubert-base", family="hubert", feat_proj_layer_norm=False)
    sd = W.seeded_encoder_state_dict(cfg, seed=77)
    assert "feature_projection.layer_norm.weight" not in sd
    enc = S.HuggingFaceWav2Vec2("tiny-hubert-base", None, config=cfg, precision=prec, seed=77).to(DEV)
    wav = synth_wav(2, 6000, 78)
    out = enc(wav.to(DEV)).cpu()
    with torch.no_grad():
        ref = O.encoder_forward(sd, cfg, wav)
    assert out.shape == ref.shape
    assert (out - ref).abs().max() < tol


LOCAL_DIRS = ["tiny-wav2vec2-hf", "tiny-hubert-bn-hf", "tiny-wavlm-hf", "tiny-data2vec-hf", "tiny-wav2vec2-sb"]


@pytest.mark.parametrize("name", LOCAL_DIRS)
def test_local_model_directory_like_the_reference_constructor(golden, name, tmp_path):
    """Same constructor call on both sides: HuggingFaceWav2Vec2(source=<directory>, save_path=...).  The expected output is
    the reference's own constructor (from_pretrained on that directory / SpeechBrain *.ckpt transfer) + forward
    (tests/golden/make_golden.py make_local_dirs; reference huggingface_interface.py:89-262)."""
    import os
    fx = golden("local_ckpt")[name]
    d = os.path.join(os.path.dirname(__file__), "golden", "local_ckpt", name)
    enc = S.HuggingFaceWav2Vec2(d, str(tmp_path), precision="fp32").to(DEV)
    assert enc.normalize_wav == fx["normalize_wav"]
    y = enc(fx["wav"].to(DEV)).cpu()
    assert y.shape == fx["out"].shape
    assert (y - fx["out"]).abs().max().item() < 1e-3


@pytest.mark.parametrize("cfg_name,B,L", [
    ("wav2vec2-base", 1, 400),        # the receptive field exactly: ONE frame (every whole-batch statistic over a single row)
    ("wav2vec2-base", 2, 719),        # still one frame, ragged tail samples that no conv window reaches
    ("wav2vec2-base", 5, 16001),      # odd batch, odd length (T = 49)
    ("wav2vec2-base", 1, 480000),     # 30 s clip: T = 1499 (3 key tiles more than the bench shape, odd T)
    ("hubert-large-ll60k", 3, 24000), # pre-LN / layer-norm conv stack at an odd batch
    ("wavlm-base", 2, 33333),         # relative position bias at T = 103
])
def test_odd_shapes_vs_oracle(cfg_name, B, L):
    """Edge shapes against the oracle on the same seeded weights and input: fp32 features within 1e-3; bf16 within the
    error bound of the bf16 mode."""
    cfg = PRESETS[cfg_name]
    sd = W.seeded_encoder_state_dict(cfg, seed=77)
    wav = synth_wav(B, L, 78)
    want = O.encoder_forward(sd, cfg, wav)
    assert want.shape == (B, cfg.frames(L), cfg.hidden_size)
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision="fp32", seed=77).to(DEV)
    got = enc(wav.to(DEV)).cpu()
    err = (got - want).abs().max().item()
    print(f"{cfg_name} B={B} L={L} T={want.shape[1]}: fp32 max|err| {err:.2e}")
    assert err < 1e-3
    enc16 = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision="bf16", seed=77).to(DEV)
    got16 = enc16(wav.to(DEV)).cpu()
    d = (got16 - want).abs()
    print(f"   bf16 mean|err| {d.mean():.4f} max {d.max():.3f}")
    assert torch.isfinite(got16).all() and d.mean() < 0.08


@pytest.mark.parametrize("cfg_name,B,L,prec,tol", [("tiny-group", 3, 4000, "fp32", 1e-5), ("tiny-layer", 4, 4000, "fp32", 1e-5),
                                                  ("wav2vec2-base", 5, 80000, "fp32", 2e-4), ("wav2vec2-base", 5, 80000, "bf16", 0.5),
                                                  ("wav2vec2-large-lv60", 3, 80000, "bf16", 0.5)])
def test_per_clip_norm_groups_equal_batch1_forwards(cfg_name, B, L, prec, tol):
    """clips_per_norm_group = 1: a batch of B equal-length utterances == B batch-1 forwards (the reference's evaluation loop,
    train_audio_ssl.py:90), and != the whole-batch norm of the default call.  In bf16 the batch runs on other GEMM tilings
    than a single utterance, so the two agree to bf16 rounding only."""
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision=prec, seed=21).to(DEV)
    wav = synth_wav(B, L, 22)
    wav[1] *= 3.0  # clips of different loudness: the whole-batch norm and the per-clip norm differ visibly
    wav = wav.to(DEV)
    one_by_one = torch.cat([enc(wav[b:b + 1]) for b in range(B)])
    batched = enc(wav, clips_per_norm_group=1)
    d = (batched - one_by_one).abs()
    print(f"{cfg_name} {prec}: per-clip-norm batch vs batch-1 forwards max|d| {d.max():.2e} mean {d.mean():.2e}")
    assert d.max().item() < tol and d.mean().item() < tol / 8
    if prec == "fp32":
        whole = enc(wav)
        assert (whole - one_by_one).abs().max().item() > 10 * max(d.max().item(), 1e-4)
    if prec == "fp32" and cfg_name.startswith("tiny"):
        sd = W.seeded_encoder_state_dict(cfg, seed=21)
        want = torch.cat([O.encoder_forward(sd, cfg, wav[b:b + 1].cpu()) for b in range(B)])
        assert (batched.cpu() - want).abs().max().item() < 1e-3
    with pytest.raises(Exception):
        enc(wav, clips_per_norm_group=2 if B % 2 else 3)  # batch not a multiple of the group


def test_song_transcriber_batched_utterances_same_notes():
    cfg = PRESETS["wav2vec2-base"]
    enc = S.HuggingFaceWav2Vec2("wav2vec2-base", None, config=cfg, precision="fp32", seed=31).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=32))
    head = head.to(DEV)
    song = synth_wav(1, int(16000 * 27.3), 33)[0].to(DEV)  # 5 utterances: 4 x 5 s + 7.3 s
    a, fa = S.SongTranscriber(enc, head, batch_utterances=True).transcribe(song, return_feats=True)
    b, fb = S.SongTranscriber(enc, head, batch_utterances=False).transcribe(song, return_feats=True)
    assert fa.shape == fb.shape and (fa - fb).abs().max().item() < 2e-4
    assert a == b and len(a) > 0
    # several batched jobs (max_batch smaller than the run of equal-length utterances), round-robin on the two streams
    c, fc = S.SongTranscriber(enc, head, batch_utterances=True, max_batch=3).transcribe(song, return_feats=True)
    assert (fc - fb).abs().max().item() < 2e-4 and c == b
    d = S.SongTranscriber(enc, head, streams=1, max_batch=2).transcribe(song)
    assert d == b


def test_reload_reaches_replicas_and_parent_module_loads():
    """Advisor round 1: a replica() kept by a caller (SongTranscriber lanes, bench.py streams) must serve the NEW weights after
    load_state_dict on the original, and a state dict loaded through a parent nn.ModuleDict (Brain.modules / Checkpointer)
    must both normalise the old weight-norm keys and re-upload, with freeze=True (the reference default)."""
    cfg = PRESETS["tiny-layer"]
    enc = S.HuggingFaceWav2Vec2("tiny-layer", None, config=cfg, precision="fp32", seed=1).to(DEV)
    rep = enc.replica()
    wav = synth_wav(2, 3000, 2).to(DEV)
    a0, r0 = enc(wav).cpu(), rep(wav).cpu()
    assert torch.equal(a0, r0)
    sd77 = W.seeded_encoder_state_dict(cfg, seed=77)
    enc.load_state_dict({"model." + k: v for k, v in sd77.items()})
    with torch.no_grad():
        ref77 = O.encoder_forward(sd77, cfg, wav.cpu())
    a1, r1 = enc(wav).cpu(), rep(wav).cpu()
    assert (a1 - ref77).abs().max() < 1e-3
    assert torch.equal(a1, r1), "the replica kept serving the previous checkpoint"
    # through a parent module, old key spelling + HF-only key, strict
    parent = torch.nn.ModuleDict({"wav2vec2": enc})
    sd5 = W.seeded_encoder_state_dict(cfg, seed=5)
    new = {"wav2vec2.model." + k: v for k, v in W.seeded_encoder_state_dict(cfg, seed=5, old_weight_norm_keys=True).items()}
    new["wav2vec2.model.masked_spec_embed"] = torch.zeros(cfg.hidden_size)
    parent.load_state_dict(new, strict=True)
    with torch.no_grad():
        ref5 = O.encoder_forward(sd5, cfg, wav.cpu())
    assert (enc(wav).cpu() - ref5).abs().max() < 1e-3
    assert (rep(wav).cpu() - ref5).abs().max() < 1e-3
    # an in-place edit without any hook: the sentinel tensors catch the first / middle / last parameter
    with torch.no_grad():
        next(iter(enc.model.parameters())).mul_(1.5)
    assert (enc(wav).cpu() - ref5).abs().max() > 1e-3


@pytest.mark.parametrize("cfg_name", ["hubert-large-ll60k", "wav2vec2-large-lv60"])
def test_c3_c5_full_size_64x10s(cfg_name):
    """BASELINE configs C3 (HuBERT-large, 64 x 10 s) and the per-GPU shard of C5 (wav2vec2-large, 64 x 10 s) at FULL size.
    The oracle needs minutes for such a batch, so: size-independent properties of the bf16 throughput mode (finite,
    reproducible, whole-batch norm, permutation equivariance over clips), and -- because per-clip norm groups make every
    clip independent of its batch-mates -- a real parity check: four clips of the 64-clip batch against an exact-fp32
    forward of just those four clips, to the 1e-3 bar for the split-operand parity mode and to the bf16 bound for bf16."""
    cfg = PRESETS[cfg_name]
    B, L = 64, 160000
    wav = synth_wav(B, L, 321).to(DEV)
    enc16 = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision="bf16", seed=9).to(DEV)
    a = enc16(wav)
    assert a.shape == (B, 499, 1024) and torch.isfinite(a).all()
    assert abs(a.mean().item()) < 1e-4 and abs(a.var(unbiased=False).item() - 1.0) < 1e-3
    b = enc16(wav)
    assert torch.equal(a, b)                                # ordered sums since round 5: a forward is reproducible bit for bit
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    c = enc16(wav[perm])
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(B, device=DEV)
    assert (c[inv] - a).abs().mean() < 2e-3
    del b, c
    sel = torch.tensor([0, 21, 42, 63], device=DEV)
    enc32 = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision="fp32", seed=9).to(DEV)
    ref = enc32(wav[sel], clips_per_norm_group=1)
    g16 = enc16(wav, clips_per_norm_group=1)[sel]
    e16 = (g16 - ref).abs()
    print(f"{cfg_name} 64 x 10 s, bf16 vs fp32 on clips {sel.tolist()}: mean|d| {e16.mean():.4f} max {e16.max():.4f}")
    assert e16.mean() < 0.05 and e16.max() < 1.0
    del enc16, g16
    enc3 = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision="fp16x3", seed=9).to(DEV)
    g3 = enc3(wav, clips_per_norm_group=1)[sel]
    e3 = (g3 - ref).abs().max().item()
    print(f"{cfg_name} 64 x 10 s, fp16x3 vs fp32 on clips {sel.tolist()}: max|d| {e3:.3e}")
    assert e3 < 1e-3


def test_c4_audio_visual_16_clips(golden):
    """BASELINE config C4 at its batch size: FusionRCA on 16 x (499 audio + 500 video frames) in every precision -- the first
    two clips are the reference's own golden (``fusion_trunc``: the fusion has no cross-clip coupling), the rest are checked
    by permutation equivariance -- and the AV-HuBERT video branch on 16 x 500 lip frames of 88 x 88 (finite, reproducible)."""
    fx = golden("fusion_trunc")
    assert (fx["B"], fx["T1"], fx["T2"]) == (2, 499, 500)
    g = torch.Generator().manual_seed(fx["in_seed"])
    a2 = torch.randn(2, 499, 1024, generator=g)
    v2 = torch.randn(2, 500, 1024, generator=g)
    g2 = torch.Generator().manual_seed(77)
    a = torch.cat([a2, torch.randn(14, 499, 1024, generator=g2)]).to(DEV)
    v = torch.cat([v2, torch.randn(14, 500, 1024, generator=g2)]).to(DEV)
    perm = torch.randperm(16, generator=g2).to(DEV)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(16, device=DEV)
    for prec, tol in (("fp32", 1e-3), ("fp16x3", 1e-3), ("bf16x3", 1e-3), ("bf16", 0.12)):
        fus = S.FusionRCA(precision=prec, seed=fx["weight_seed"]).to(DEV)
        out = fus(a, v)
        assert out.shape == (16, 499, 1024) and torch.isfinite(out).all()
        o2 = out[:2].cpu()
        assert (o2[:, ::7, ::5] - fx["out_strided"]).abs().max() < tol, prec
        assert (o2[:, :4] - fx["out_first"]).abs().max() < tol, prec
        outp = fus(a[perm], v[perm])
        assert (outp[inv] - out).abs().max() < 1e-5, prec   # clips are independent: same rows, same arithmetic
    from svt_speechbrain_amd.video import FairseqAVHubertPretrain
    m = FairseqAVHubertPretrain(config="avhubert-large-video", precision="bf16", seed=78, output_norm=True).to(DEV)
    x = torch.randn(16, 1, 500, 88, 88, generator=g2).to(DEV)
    y = m({"video": x, "audio": None})
    assert y.shape == (16, 500, 1024) and torch.isfinite(y).all()
    assert abs(y.mean().item()) < 1e-4 and abs(y.var(unbiased=False).item() - 1.0) < 1e-3
    y2 = m({"video": x, "audio": None})
    assert (y - y2).abs().max() < 1e-4


@pytest.mark.parametrize("cfg_name,B,L,cpg", [("wav2vec2-base", 3, 52345, 0), ("wav2vec2-base", 4, 80000, 1), ("wav2vec2-base", 4, 80000, 2),
                                              ("wav2vec2-large-lv60", 2, 48000, 0)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fused_tail_equals_encoder_then_head_then_decode(cfg_name, B, L, cpg, prec):
    """``svt_encoder_forward_head`` (whole-batch output norm + frame head + per-frame decode in one pass over the
    un-normalised encoder output, the features never written) against the three separate calls it replaces."""
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision=prec, seed=13).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=14))
    head = head.to(DEV)
    wav = synth_wav(B, L, 15).to(DEV)
    T = cfg.frames(L)
    ref_logits = head(enc(wav, clips_per_norm_group=cpg))
    ref_frames = S.decode_frames(ref_logits)
    frames = torch.full((B * T, 4), -1, dtype=torch.int32, device=DEV)
    logits = enc.forward_head(wav, head, clips_per_norm_group=cpg, frames=frames)
    assert logits.shape == ref_logits.shape
    err = (logits - ref_logits).abs().max().item()
    assert err < 2e-4, err          # same fp32 arithmetic re-associated: (x.w - mu sum(w)) rstd + b
    got = frames.cpu().numpy().view(S.decode.FRAME_DTYPE).reshape(B, T)
    own = S.decode_frames(logits)   # the fused decode must agree exactly with the decode kernel on the fused logits
    for k in ("octave", "pitch_class"):
        assert (got[k] == own[k]).all()
    assert np.abs(got["p_on"] - own["p_on"]).max() < 1e-6 and np.abs(got["p_off"] - own["p_off"]).max() < 1e-6
    # ... and with the separate path wherever the top-2 logits are further apart than the re-association noise
    mism = (got["octave"] != ref_frames["octave"]) | (got["pitch_class"] != ref_frames["pitch_class"])
    assert mism.mean() < 0.002
    # AMTForward takes the fused path by itself in the throughput precisions; the parity-grade ones keep the reference's order of
    # operations (normalise, then the head) unless fuse_tail is set
    amt = S.AMTForward({"wav2vec2": enc, "model": head})
    amt.compute_forward(wav)
    if cpg == 0:
        assert torch.equal(amt.last_logits, logits if prec in ("bf16", "fp16") else ref_logits)
        amt.fuse_tail = True
        amt.compute_forward(wav)
        assert torch.equal(amt.last_logits, logits)
    # no output norm: the head reads the encoder output as is
    enc2 = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, output_norm=False, precision=prec, seed=13).to(DEV)
    assert (enc2.forward_head(wav, head) - head(enc2(wav))).abs().max() < 2e-4


def test_data_parallel_single_device_and_real_replicas():
    """Round-2 verdict / advisor: `nn.DataParallel` as the recipes' `--data_parallel_backend` wraps the modules
    (speechbrain/core.py:1150-1169).  (i) `DataParallel(module, device_ids=[0])`: forward equals the bare module's, and a reload through
    the wrapper's state dict is served; (ii) REAL replicas -- `torch.nn.parallel.replicate(module, [0, 0])`, the call DataParallel makes
    per forward: their `_parameters` are empty, the upload must read the original's -- give the original's outputs for the encoder, the
    head and the fused tail, before and after a reload of the original."""
    cfg = PRESETS["tiny-layer"]
    enc = S.HuggingFaceWav2Vec2("tiny-layer", None, config=cfg, precision="fp32", seed=4).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=6))
    head = head.to(DEV)
    wav = synth_wav(4, 3000, 8).to(DEV)
    base = head(enc(wav)).cpu()
    dp_enc, dp_head = torch.nn.DataParallel(enc, device_ids=[0]), torch.nn.DataParallel(head, device_ids=[0])
    assert torch.equal(dp_head(dp_enc(wav)).cpu(), base)
    # real replicas on one device
    reps = torch.nn.parallel.replicate(enc, [0, 0])
    hreps = torch.nn.parallel.replicate(head, [0, 0])
    assert all(len(list(r.parameters())) == 0 for r in reps), "replicate() is expected to leave replicas without registered parameters"
    for r, h in zip(reps, hreps):
        assert torch.equal(h(r(wav)).cpu(), base)
        if S.HuggingFaceWav2Vec2.can_fuse_head(h):
            assert (r.forward_head(wav, h).cpu() - base).abs().max() < 2e-4
    # reload through the DataParallel wrapper's key space ("module." prefix), then fresh replicas serve the new weights
    sd9 = W.seeded_encoder_state_dict(cfg, seed=9)
    dp_enc.load_state_dict({"module.model." + k: v for k, v in sd9.items()})
    with torch.no_grad():
        ref9 = O.encoder_forward(sd9, cfg, wav.cpu())
    assert (enc(wav).cpu() - ref9).abs().max() < 1e-3
    for r in torch.nn.parallel.replicate(enc, [0, 0]):
        assert (r(wav).cpu() - ref9).abs().max() < 1e-3


def test_audio_visual_compute_forward_with_the_recipe_module_names(golden):
    """Round-2 verdict: `AMT.compute_forward(wavs, wav_lens, videos)` as the audio-visual recipe calls it
    (N20EMv2/audio_visual/train_rca_av.py:28-51: `self.modules.fusion(audio_feats, video_feats)` then `self.modules.head`).
    The fusion output of the first two clips is the reference's golden (`fusion_trunc`); the logits are the head applied to it; the four
    returned views are the recipe's slices of the 20-way logits; a DataParallel-wrapped pair of modules gives the same."""
    fx = golden("fusion_trunc")
    g = torch.Generator().manual_seed(fx["in_seed"])
    a = torch.randn(2, 499, 1024, generator=g).to(DEV)
    v = torch.randn(2, 500, 1024, generator=g).to(DEV)
    fus = S.FusionRCA(precision="fp32", seed=fx["weight_seed"]).to(DEV)
    head = S.Linear(20, input_size=1024)
    head.load_state_dict(W.seeded_head_state_dict(1024, 20, seed=3))
    head = head.to(DEV)
    amt = S.AMTForward({"fusion": fus, "head": head})
    lens = torch.ones(2)
    onset, offset, octave, pitch, lens_out = amt.compute_forward(a, lens, videos=v)
    feats = fus(a, v)
    assert (feats[:, :4].cpu() - fx["out_first"]).abs().max() < 1e-3
    logits = head(feats)
    assert torch.equal(amt.last_logits, logits)
    assert lens_out is lens and onset.shape == (2, 499) and offset.shape == (2, 499)
    assert torch.equal(onset, logits[:, :, 0]) and torch.equal(offset, logits[:, :, 1])
    o = amt.pitch_octave_num
    assert torch.equal(octave, logits[:, :, 2:3 + o]) and torch.equal(pitch, logits[:, :, 3 + o:])
    assert octave.shape[-1] + pitch.shape[-1] == 18
    wrapped = S.AMTForward({"fusion": torch.nn.DataParallel(fus, device_ids=[0]), "head": torch.nn.DataParallel(head, device_ids=[0])})
    assert torch.equal(wrapped.compute_forward(a, lens, videos=v)[0], onset)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("B,L", [(1, 80000), (3, 48000), (8, 80000)])
def test_small_batch_ffn2_k_split_against_the_unsplit_product(precision, B, L):
    """Round 6: for a few utterances (<= 2048 frames) FFN-2 runs as a K-split launch of the one-utterance GEMM -- four workgroups per
    tile, raw fp32 partial tiles -- and the LayerNorm behind it adds the parts and the bias and rounds the sum to the operand type
    (csrc/gemm_skinny.hip ksplit, kernels.hip layernorm_hilo2_kernel<D, true>; svt_debug_set key 36 = 0: the un-split product).  Same
    products, same rounding points, another fp32 summation order: the two forms agree to a few operand ulps of the branch output,
    each is reproducible bit for bit, and the goldens hold both (test_bf16_mode_error_bound runs the default)."""
    cfg = PRESETS["wav2vec2-base"]
    enc = S.HuggingFaceWav2Vec2("wav2vec2-base", None, config=cfg, precision=precision, normalize_wav=True, seed=41).to(DEV)
    lib = enc._lib()
    wav = synth_wav(B, L, 77).to(DEV)
    try:
        lib.svt_debug_set(36, 1)
        a = enc(wav).clone()
        a2 = enc(wav).clone()
        lib.svt_debug_set(36, 0)
        b = enc(wav).clone()
    finally:
        lib.svt_debug_set(36, 1)
    assert torch.isfinite(a).all() and torch.equal(a, a2)
    d = (a - b).abs()
    print(f"FFN-2 K-split vs un-split ({precision}, B={B}, L={L}): max |d| {d.max().item():.4f} mean {d.mean().item():.5f}")
    bound = 0.25 if precision == "bf16" else 0.04      # the normalised features have unit variance; bf16 mode error vs fp32 is ~0.08 mean
    assert d.max().item() < bound and d.mean().item() < bound / 12


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg_name,B,L", [("wav2vec2-base", 4, 160000), ("wav2vec2-base", 3, 52345), ("hubert-large-ll60k", 2, 80000)])
def test_conv_tap_minor_k_order_against_tap_major(precision, cfg_name, B, L):
    """Round 6: the kernel-3 / stride-2 convolutions (layers 1-4) run on gemm_p1w_kernel with their K slabs TAP-MINOR (slab g = tap g % 3
    of channel block g / 3, against a second copy of the weights stored in that order) so that the input frame two neighbouring output
    rows share is re-read two slabs later -- out of L2 -- instead of sixteen (csrc/gemm_p1w.hip, GemmArgs::k_taps; svt_debug_set key 35 = 0:
    tap-major).  The same products in another fp32 summation order: the features agree to a few operand ulps, each form is reproducible
    bit for bit, group-norm (base) and layer-norm (large) extractors, M tails included (52 345 samples)."""
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, precision=precision, normalize_wav=True, seed=43).to(DEV)
    lib = enc._lib()
    wav = synth_wav(B, L, 78).to(DEV)
    try:
        lib.svt_debug_set(35, 1)
        a = enc(wav).clone()
        a2 = enc(wav).clone()
        lib.svt_debug_set(35, 0)
        b = enc(wav).clone()
    finally:
        lib.svt_debug_set(35, 1)
    assert torch.isfinite(a).all() and torch.equal(a, a2)
    d = (a - b).abs()
    print(f"conv K order tap-minor vs tap-major ({cfg_name}, {precision}, B={B}, L={L}): max |d| {d.max().item():.4f} mean {d.mean().item():.5f}")
    # two valid summation orders of a 16-bit forward differ like two draws of its rounding noise (measured: bf16 max 0.18-0.25 / mean 0.025-0.028 on
    # unit-variance features, where the mode's own error against fp32 is ~0.08 mean; fp16 an eighth of that)
    bound = 0.5 if precision == "bf16" else 0.07
    assert d.max().item() < bound and d.mean().item() < bound / 8
