"""Generate the golden fixtures in this directory by importing the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and HF transformers); the fixtures it writes
(*.pt, data only: seeds, inputs, expected outputs) are committed, the reference never is.

    python tests/golden/make_golden.py [--only NAME]

What runs is the reference's own code: ``HuggingFaceWav2Vec2.forward`` (MIR_ST500/huggingface_interface.py:263-297)
over HF ``Wav2Vec2Model`` / ``HubertModel`` (eager attention, fp32), ``speechbrain.nnet.linear.Linear``,
``fusion.FusionRCA``, ``utils.frame2note``, ``speechbrain.decoders.ctc``, and the STFT ->
spectral_magnitude -> Filterbank chain.  Weights come from ``svt_speechbrain_amd.weights`` (seeded) and are
loaded into the reference modules with ``load_state_dict``.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from svt_speechbrain_amd.config import PRESETS  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402

REF = "/root/reference"


def import_reference():
    import transformers  # noqa: F401  (must precede the stubs: the wrapper probes torchaudio via find_spec)
    from transformers import Wav2Vec2Model, HubertModel, Wav2Vec2Config, HubertConfig  # noqa: F401

    class Stub(types.ModuleType):
        __path__ = []

        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return Stub(self.__name__ + "." + n)

        def __call__(self, *a, **k):
            return None

    for m in ["hyperpyyaml", "torchaudio", "ruamel", "ruamel.yaml"]:
        sys.modules.setdefault(m, Stub(m))
    for p in [REF, REF + "/MIR_ST500", REF + "/N20EMv2/audio_visual"]:
        if p not in sys.path:
            sys.path.append(p)
    import speechbrain  # noqa: F401
    import huggingface_interface
    import fusion
    import utils
    return huggingface_interface, fusion, utils


def hf_model(cfg):
    from transformers import Wav2Vec2Model, HubertModel, Wav2Vec2Config, HubertConfig
    kw = dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
              num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
              conv_dim=list(cfg.conv_dim), conv_kernel=list(cfg.conv_kernel), conv_stride=list(cfg.conv_stride),
              feat_extract_norm=cfg.feat_extract_norm, conv_bias=cfg.conv_bias,
              do_stable_layer_norm=cfg.do_stable_layer_norm,
              num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
              num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups,
              layer_norm_eps=cfg.layer_norm_eps, hidden_dropout=0.0, attention_dropout=0.0,
              activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, apply_spec_augment=False,
              attn_implementation="eager")
    if cfg.family == "wavlm":
        from transformers import WavLMModel, WavLMConfig
        m = WavLMModel(WavLMConfig(num_buckets=cfg.rel_pos_buckets, max_bucket_distance=cfg.rel_pos_max_distance, **kw))
    elif cfg.family == "data2vec":
        from transformers import Data2VecAudioModel, Data2VecAudioConfig
        kw.pop("do_stable_layer_norm"); kw.pop("feat_extract_norm"); kw.pop("num_conv_pos_embeddings")
        m = Data2VecAudioModel(Data2VecAudioConfig(conv_pos_kernel_size=cfg.num_conv_pos_embeddings,
                                                   num_conv_pos_embeddings=cfg.pos_conv_depth, **kw))
    elif cfg.family == "hubert":
        hc = HubertConfig(feat_proj_layer_norm=cfg.feat_proj_layer_norm, conv_pos_batch_norm=cfg.conv_pos_batch_norm, **kw)
        m = HubertModel(hc)
    else:
        m = Wav2Vec2Model(Wav2Vec2Config(**kw))
    return m.eval()


def reference_encoder(hi, cfg, sd, normalize_wav=True, output_norm=True):
    """The reference wrapper object with our seeded weights inside (no network: bypass __init__)."""
    model = hf_model(cfg)
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    assert all(k == "masked_spec_embed" for k in missing), missing
    full = dict(sd)
    for k in missing:
        full[k] = own[k]
    model.load_state_dict(full, strict=True)
    w = hi.HuggingFaceWav2Vec2.__new__(hi.HuggingFaceWav2Vec2)
    torch.nn.Module.__init__(w)
    w.model = model
    w.normalize_wav = normalize_wav
    w.output_norm = output_norm
    w.freeze = True
    w.freeze_feature_extractor = False
    return w.eval()


def synth_wav(B, L, seed=1986):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)


def sd_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def decode(utils, logits):
    """The reference's per-frame loop (MIR_ST500/train_audio_ssl.py:93-100) + frame2note, per clip."""
    out = []
    for b in range(logits.shape[0]):
        lg = logits[b]
        on, off = torch.sigmoid(lg[:, 0]), torch.sigmoid(lg[:, 1])
        po, pc = lg[:, 2:7], lg[:, 7:]
        pred = []
        for f in range(lg.shape[0]):
            pred.append((on[f], off[f], torch.argmax(po[f]).item(), torch.argmax(pc[f]).item()))
        notes = utils.frame2note(pred, onset_thres=0.4, offset_thres=0.5, frame_size=1 / 49.8)
        out.append(dict(p_on=on.clone(), p_off=off.clone(),
                        oct=torch.tensor([p[2] for p in pred]), pc=torch.tensor([p[3] for p in pred]),
                        notes=[[float(n[0]), float(n[1]), int(n[2])] for n in notes]))
    return out


def make_encoder_case(hi, utils, name, cfg_name, B, L, seed, pad_to=None, full=True, lens=None):
    import speechbrain as sb
    cfg = PRESETS[cfg_name]
    sd = W.seeded_encoder_state_dict(cfg, seed=seed)
    hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=seed + 1000)
    enc = reference_encoder(hi, cfg, sd)
    head = sb.nnet.linear.Linear(n_neurons=20, input_size=cfg.hidden_size)
    head.load_state_dict(hd, strict=True)
    wav = synth_wav(B, L, seed=seed + 7)
    if lens is not None:  # ragged, right-zero-padded batch
        for b, n in enumerate(lens):
            wav[b, n:] = 0.0
    with torch.no_grad():
        feats = enc(wav)
        logits = head(feats)
    dec = decode(utils, logits)
    fx = dict(name=name, cfg=cfg_name, B=B, L=L, weight_seed=seed, head_seed=seed + 1000, wav_seed=seed + 7,
              lens=lens, sd_sha256=sd_digest(sd), T=feats.shape[1],
              logits=logits.clone(), decode=dec)
    if full:
        fx["wav"] = wav.clone()
        fx["feats"] = feats.clone()
    else:
        fx["feats_strided"] = feats[:, ::25, ::16].clone()
        fx["wav_sha256"] = hashlib.sha256(wav.numpy().tobytes()).hexdigest()
        fx["feats_absmean"] = float(feats.abs().mean())
    torch.save(fx, os.path.join(HERE, name + ".pt"))
    print(name, tuple(feats.shape), "notes/clip", [len(d["notes"]) for d in dec])


def make_fusion_cases(fusion_mod):
    sd = W.seeded_fusion_state_dict(1024, 3072, seed=3986)
    m = fusion_mod.FusionRCA().eval()
    # the positional-encoding table the REFERENCE builds for itself (speechbrain PositionalEncoding's registered buffer), taken before
    # any load: the fixture pins it (sha256 + a strided sample), and the reference runs on ITS OWN table, not on the one
    # svt_speechbrain_amd.weights computes -- a strict load of our state dict would overwrite the buffer and pin nothing
    pe_key = "fusion.positional_encoding.pe"
    ref_pe = m.state_dict()[pe_key].detach().clone()
    sd_ref = {k: (ref_pe if k == pe_key else v) for k, v in sd.items()}
    m.load_state_dict(sd_ref, strict=True)
    pe_pin = dict(pe_shape=tuple(ref_pe.shape), pe_sha256=hashlib.sha256(ref_pe.contiguous().numpy().tobytes()).hexdigest(),
                  pe_strided=ref_pe[0, ::97, ::61].clone())
    cases = [("fusion_trunc", 2, 499, 500), ("fusion_pad", 1, 250, 240), ("fusion_eq", 1, 64, 64)]
    for name, B, T1, T2 in cases:
        g = torch.Generator().manual_seed(77 + T1)
        a = torch.randn(B, T1, 1024, generator=g)
        v = torch.randn(B, T2, 1024, generator=g)
        with torch.no_grad():
            out = m(a, v)
        fx = dict(name=name, B=B, T1=T1, T2=T2, in_seed=77 + T1, weight_seed=3986, sd_sha256=sd_digest(sd),
                  out_strided=out[:, ::7, ::5].clone(), out_absmean=float(out.abs().mean()),
                  out_first=out[:, :4].clone(), **pe_pin)
        torch.save(fx, os.path.join(HERE, name + ".pt"))
        print(name, tuple(out.shape))


def make_frame2note_cases(utils):
    f32 = np.float32
    cases = {}
    rng = np.random.RandomState(5)

    def mk(on, off, octv, pc):
        return [(torch.tensor(f32(a)), torch.tensor(f32(b)), int(c), int(d)) for a, b, c, d in zip(on, off, octv, pc)]

    n = 60
    base_on = np.full(n, 0.1, f32)
    base_off = np.full(n, 0.1, f32)
    octv = np.full(n, 2)
    pc = np.full(n, 5)
    # plateau of equal onset probabilities (every frame equals the window max)
    on = base_on.copy(); on[10:14] = 0.8; off = base_off.copy(); off[30] = 0.9
    cases["plateau"] = (on, off, octv, pc)
    # exact-threshold values (float32(0.4) vs python 0.4)
    on = base_on.copy(); on[5] = f32(0.4); on[20] = np.nextafter(f32(0.4), f32(0)); off = base_off.copy(); off[15] = f32(0.5); off[40] = 0.7
    cases["thresholds"] = (on, off, octv, pc)
    # pitch-mode ties
    on = base_on.copy(); on[3] = 0.9; off = base_off.copy(); off[25] = 0.9
    o2 = octv.copy(); p2 = pc.copy()
    o2[3:14] = 1; p2[3:14] = 7; o2[14:25] = 3; p2[14:25] = 2
    cases["mode_tie"] = (on, off, o2, p2)
    # onset at the last frame, open note at the end
    on = base_on.copy(); on[8] = 0.7; on[n - 1] = 0.95
    cases["last_frame"] = (on, base_off.copy(), octv, pc)
    # silence classes only -> empty result
    on = base_on.copy(); on[8] = 0.7; off = base_off.copy(); off[20] = 0.8
    cases["silence"] = (on, off, np.full(n, 4), np.full(n, 12))
    # consecutive onsets without offsets
    on = base_on.copy(); on[[5, 15, 25, 35]] = [0.5, 0.6, 0.7, 0.8]
    cases["re_onset"] = (on, base_off.copy(), octv, rng.randint(0, 12, n))
    # random
    for s in range(4):
        r = np.random.RandomState(100 + s)
        m = 200 + 37 * s
        cases[f"random{s}"] = (r.rand(m).astype(f32), r.rand(m).astype(f32), r.randint(0, 5, m), r.randint(0, 13, m))
    # tiny
    cases["two_frames"] = (np.array([0.9, 0.9], f32), np.array([0.1, 0.9], f32), np.array([1, 1]), np.array([2, 2]))
    out = {}
    for k, (on, off, o, p) in cases.items():
        notes = utils.frame2note(mk(on, off, o, p), onset_thres=0.4, offset_thres=0.5, frame_size=1 / 49.8)
        out[k] = dict(p_on=torch.from_numpy(np.asarray(on, f32)), p_off=torch.from_numpy(np.asarray(off, f32)),
                      oct=torch.from_numpy(np.asarray(o)), pc=torch.from_numpy(np.asarray(p)),
                      notes=[[float(a), float(b), int(c)] for a, b, c in notes])
        print("frame2note", k, len(notes))
    torch.save(out, os.path.join(HERE, "frame2note.pt"))


def make_ctc_fbank_cases():
    from speechbrain.decoders.ctc import ctc_greedy_decode, filter_ctc_output
    from speechbrain.processing.features import STFT, spectral_magnitude, Filterbank
    out = {}
    # doctest known answers (speechbrain/decoders/ctc.py:317-320, 366-372)
    probs = torch.tensor([[[0.3, 0.7], [0.0, 0.0]], [[0.2, 0.8], [0.9, 0.1]]])
    lens = torch.tensor([0.51, 1.0])
    out["doctest"] = dict(probs=probs, lens=lens, blank=0, expect=ctc_greedy_decode(probs, lens, 0))
    assert out["doctest"]["expect"] == [[1], [1]]
    assert filter_ctc_output(['a', 'a', 'blank', 'b', 'b', 'blank', 'c'], blank_id='blank') == ['a', 'b', 'c']
    g = torch.Generator().manual_seed(11)
    for i, (B, T, V) in enumerate([(3, 50, 7), (2, 120, 31), (4, 9, 3)]):
        p = torch.log_softmax(torch.randn(B, T, V, generator=g) * 2, dim=-1)
        ln = torch.rand(B, generator=g) * 0.6 + 0.4
        ln[0] = 1.0
        out[f"rand{i}"] = dict(probs=p, lens=ln, blank=-1, expect=ctc_greedy_decode(p, ln, -1))
    torch.save(out, os.path.join(HERE, "ctc.pt"))
    # Fbank default chain (lobes/features.py:134-136); the Fbank class itself needs a GPU
    stft = STFT(sample_rate=16000, n_fft=400, win_length=25, hop_length=10)
    fb = Filterbank(sample_rate=16000, n_fft=400, n_mels=40, f_min=0, f_max=8000)
    fo = {}
    for name, B, L in [("a", 2, 16000), ("b", 3, 4321)]:
        wav = synth_wav(B, L, seed=500 + L)
        feats = fb(spectral_magnitude(stft(wav)))
        fo[name] = dict(wav=wav, feats=feats)
        print("fbank", name, tuple(feats.shape))
    assert float(spectral_magnitude(torch.Tensor([[3, 4]]), power=0.5)) == 5.0
    torch.save(fo, os.path.join(HERE, "fbank.pt"))


def make_loss_cases():
    """speechbrain.nnet.losses.bce_loss / nll_loss and activations.Softmax run as the recipes call them
    (MIR_ST500/train_audio_ssl.py:66-76): (B,T) logits / (B,T,C) log-probs, relative lengths, +-3-frame truncate."""
    from speechbrain.nnet.losses import bce_loss, nll_loss
    from speechbrain.nnet.activations import Softmax
    g = torch.Generator().manual_seed(77)
    out = {"bce": [], "nll": [], "softmax": []}
    # doctest known answers (losses.py:437-439, 498-501)
    assert abs(float(nll_loss(torch.log(torch.tensor([[0.9, 0.1], [0.1, 0.9]])), torch.tensor([1, 1]))) - 1.2040) < 1e-4
    assert abs(float(bce_loss(torch.tensor([10.0, -6.0]), torch.tensor([1, 0]))) - 0.0013) < 1e-4
    shapes = [(3, 249, 249), (2, 499, 498), (4, 60, 63), (1, 250, 249), (5, 17, 17)]
    for i, (B, tp, tt) in enumerate(shapes):
        x = torch.randn(B, tp, generator=g) * 3
        y = (torch.rand(B, tt, generator=g) < 0.2).float()
        ln = torch.rand(B, generator=g) * 0.7 + 0.3
        ln[0] = 1.0
        for length in (None, ln):
            for pw in (None, 15.0):
                for red in ("mean", "batch", "batchmean", "none"):
                    kw = dict(length=length, reduction=red)
                    if pw is not None:
                        kw["pos_weight"] = torch.tensor([pw])
                    out["bce"].append(dict(x=x, y=y, length=length, pos_weight=pw, reduction=red,
                                           expect=bce_loss(x, y, **kw)))
        for C in (5, 13):
            lp = Softmax(apply_log=True)(torch.randn(B, tp, C, generator=g) * 2)
            tg = torch.randint(0, C, (B, tt), generator=g)
            for length in (None, ln):
                for ls in (0.0, 0.1):
                    for red in ("mean", "batch", "batchmean") + (("none",) if ls == 0.0 else ()):
                        out["nll"].append(dict(lp=lp, tg=tg, length=length, label_smoothing=ls, reduction=red,
                                               expect=nll_loss(lp, tg, length=length, label_smoothing=ls, reduction=red)))
    # 1-D / 2-D forms of the doctests and an ignored target
    out["bce"].append(dict(x=torch.tensor([10.0, -6.0]), y=torch.tensor([1.0, 0.0]), length=None, pos_weight=None,
                           reduction="mean", expect=bce_loss(torch.tensor([10.0, -6.0]), torch.tensor([1, 0]))))
    lp2 = torch.log(torch.tensor([[0.9, 0.1], [0.1, 0.9]]))
    out["nll"].append(dict(lp=lp2, tg=torch.tensor([1, 1]), length=None, label_smoothing=0.0, reduction="mean",
                           expect=nll_loss(lp2, torch.tensor([1, 1]))))
    lp3 = Softmax(apply_log=True)(torch.randn(2, 9, 5, generator=g))
    tg3 = torch.randint(0, 5, (2, 9), generator=g)
    tg3[0, 3] = -100
    out["nll"].append(dict(lp=lp3, tg=tg3, length=torch.tensor([1.0, 0.5]), label_smoothing=0.0, reduction="batch",
                           expect=nll_loss(lp3, tg3, length=torch.tensor([1.0, 0.5]), reduction="batch")))
    for shape in [(3, 7, 5), (2, 13), (2, 3, 4, 6)]:
        x = torch.randn(*shape, generator=g) * 4
        for apply_log in (False, True):
            out["softmax"].append(dict(x=x, apply_log=apply_log, expect=Softmax(apply_log=apply_log)(x)))
    # error behaviour: 4 frames apart -> ValueError with this text
    try:
        bce_loss(torch.zeros(1, 10), torch.zeros(1, 14))
    except ValueError as e:
        out["truncate_error"] = str(e)
    torch.save(out, os.path.join(HERE, "losses.pt"))
    print("losses", len(out["bce"]), len(out["nll"]), len(out["softmax"]), out["truncate_error"])


def make_ckpt_tree(hi):
    """A save folder written by the reference's own Checkpointer (speechbrain/utils/checkpoints.py:505-568): three
    CKPT+* directories with the recipes' recoverables (wav2vec2 = the reference wrapper around a tiny encoder, model =
    speechbrain Linear) and metas carrying `loss` / `COnPOff_f1` like train_audio_ssl.py:178-186.  The expected picks are
    the reference's own find_checkpoint answers."""
    import json
    import shutil
    import time
    from speechbrain.utils.checkpoints import Checkpointer
    from speechbrain.nnet.linear import Linear
    cfg = PRESETS["tiny-group"]
    root = os.path.join(HERE, "ckpt_tree")
    shutil.rmtree(root, ignore_errors=True)
    expect = {"picks": {}, "digests": {}}
    metas = [("a", dict(loss=0.52, COnPOff_f1=0.31)), ("b", dict(loss=0.47, COnPOff_f1=0.36)), ("c", dict(loss=0.49))]
    for i, (name, meta) in enumerate(metas):
        sd = W.seeded_encoder_state_dict(cfg, seed=300 + i)
        enc = reference_encoder(hi, cfg, sd)
        head = Linear(n_neurons=20, input_size=cfg.hidden_size)
        hd = W.seeded_head_state_dict(cfg.hidden_size, 20, seed=400 + i)
        head.load_state_dict(hd)
        ck = Checkpointer(root, {"wav2vec2": enc, "model": head})
        ck.save_checkpoint(meta=meta, name=name)
        wav = synth_wav(2, 4000, seed=900 + i)
        with torch.no_grad():
            feats = enc(wav)
            logits = head(feats)
        expect["digests"]["CKPT+" + name] = dict(logits_sum=float(logits.double().sum()), logits_abs=float(logits.double().abs().sum()),
                                                 first=[float(v) for v in logits[0, 0, :4]], wav_seed=900 + i)
        time.sleep(0.05)
    ck = Checkpointer(root, {})
    expect["picks"]["recent"] = ck.find_checkpoint().path.name
    expect["picks"]["min_loss"] = ck.find_checkpoint(min_key="loss").path.name
    expect["picks"]["max_f1"] = ck.find_checkpoint(max_key="COnPOff_f1").path.name
    expect["picks"]["ranked_min_loss"] = [c.path.name for c in ck.find_checkpoints(min_key="loss")]
    expect["picks"]["ranked_max_f1"] = [c.path.name for c in ck.find_checkpoints(max_key="COnPOff_f1")]
    json.dump(expect, open(os.path.join(root, "expected.json"), "w"), indent=1)
    sz = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(root) for f in fs)
    print("ckpt_tree", expect["picks"], f"{sz/1e3:.0f} KB")


def make_video_front_cases():
    """The reference's own lip front-end (N20EMv2/video_only/resnet.py, loaded by path: it imports only torch):
    SubModel(input_dim=512, embed_dim, relu_type='prelu') = ResEncoder (3-D stem + ResNet-18 trunk) + proj, eval mode."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_video_resnet", REF + "/N20EMv2/video_only/resnet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, B, T, HW, E, seed in [("roi88", 2, 5, 88, 1024, 4986), ("roi88_t1", 1, 1, 88, 1024, 4987), ("roi32", 1, 7, 32, 256, 4988),
                                   ("roi50", 2, 3, 50, 128, 4989), ("roi60", 1, 3, 60, 64, 4990)]:
        sd = W.seeded_video_frontend_state_dict(E, seed=seed)
        m = mod.SubModel(512, E, "prelu").eval()
        m.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        video = torch.randn(B, 1, T, HW, HW, generator=g)
        with torch.no_grad():
            y = m(video).transpose(1, 2).contiguous()   # (B, T, E)
        out[name] = dict(B=B, T=T, HW=HW, E=E, weight_seed=seed, video_seed=seed + 1, feats=y, sd_sha256=sd_digest(sd))
        print("video_front", name, tuple(y.shape), float(y.std()))
    torch.save(out, os.path.join(HERE, "video_front.pt"))


def make_note2frame_cases(utils):
    """The reference's own note2frame (MIR_ST500/utils.py:10-69) on the note lists of the first synthetic-singing clips of the two
    seeds the trained-like head uses (svt_speechbrain_amd/synth.py): the labels the head is fitted to are the reference's labels."""
    from svt_speechbrain_amd.synth import synth_singing
    out = {}
    for seed in (2986, 3986):
        _, _, notes = synth_singing(6, 10.0, seed=seed)
        for i, n in enumerate(notes):
            gt = [[float(a), float(b), int(m)] for a, b, m in n]
            out[f"seed{seed}_clip{i}"] = dict(notes=gt, frames=torch.from_numpy(np.array(utils.note2frame(gt, 499, 1 / 49.8), dtype=np.int64)))
    torch.save(out, os.path.join(HERE, "note2frame.pt"))
    print("note2frame", len(out), "clips")


def make_video_u8_cases():
    """The video recipes' input side, by the reference's own classes: ``Compose([Normalize(0.0, 255.0), CenterCrop((88, 88)),
    Normalize(0.421, 0.165)])`` (N20EMv2/video_only/train_video_ssl.py:445-457, classes of N20EMv2/video_only/utils.py:22-84; cv2 --
    imported by that file for its video reader only -- is stubbed) applied to uint8 ROIs the way utterance_eval_pipeline does
    (:528-533: transform, expand_dims, astype(np.float32)), then the reference's SubModel on the result.  Cases: the recipes' 96 x 96
    ROI, an odd-sized one (97 x 99: CenterCrop's truncating offsets), a ROI that is already 88 x 88, and every byte value at once."""
    import importlib.util
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_video_utils", REF + "/N20EMv2/video_only/utils.py")
    ut = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ut)
    spec = importlib.util.spec_from_file_location("ref_video_resnet", REF + "/N20EMv2/video_only/resnet.py")
    res = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(res)
    transform_eval = ut.Compose([ut.Normalize(0.0, 255.0), ut.CenterCrop((88, 88)), ut.Normalize(0.421, 0.165)])
    E, seed = 64, 5186
    sd = W.seeded_video_frontend_state_dict(E, seed=seed)
    m = res.SubModel(512, E, "prelu").eval()
    m.load_state_dict(sd, strict=True)
    out = {}
    rng = np.random.RandomState(86)
    for name, T, H, Wd in [("roi96", 6, 96, 96), ("roi97x99", 3, 97, 99), ("roi88", 2, 88, 88), ("ramp", 2, 96, 96)]:
        roi = rng.randint(0, 256, size=(T, H, Wd)).astype(np.uint8)
        if name == "ramp":   # every byte value, many times, inside the crop
            roi = (np.arange(T * H * Wd, dtype=np.int64) % 256).astype(np.uint8).reshape(T, H, Wd)
        sig = transform_eval(roi)
        sig = np.expand_dims(sig, axis=-1)
        sig = torch.from_numpy(sig.astype(np.float32))                     # (T, 88, 88, 1)
        video = sig.unsqueeze(0).permute(0, 4, 1, 2, 3).contiguous()       # extract_ssl_feats.py:34 -> (1, 1, T, 88, 88)
        with torch.no_grad():
            y = m(video).transpose(1, 2).contiguous()                      # (1, T, E)
        out[name] = dict(roi=torch.from_numpy(roi.copy()), sig=sig[..., 0].clone(), feats=y, E=E, weight_seed=seed, sd_sha256=sd_digest(sd))
        print("video_u8", name, tuple(sig.shape), float(sig.mean()), tuple(y.shape))
    torch.save(out, os.path.join(HERE, "video_u8.pt"))


def import_reference_video():
    """``N20EMv2/video_only/hubert.py`` and ``fairseq_interface.py`` THEMSELVES, with the third-party names they import at module
    level (fairseq.*, omegaconf -- absent from the build container) stubbed in ``sys.modules`` for the duration of the import.
    What then runs is the reference's own ``AVHubertModel.extract_finetune`` (hubert.py:688-739), ``forward_features`` (:532-541),
    ``SubModel.forward`` (:318-326) over the real ``resnet.ResEncoder``, and ``FairseqAVHubertPretrain.forward / extract_features``
    (fairseq_interface.py:454-485).  The one module that cannot be the reference's is fairseq's ``TransformerEncoder``: the stub
    class below wraps HF's ``Wav2Vec2Encoder[StableLayerNorm]`` (the module HF ported from it; pinned for the audio path)."""
    import dataclasses  # noqa: F401
    import torch.nn as nn

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    class HFTransformerEncoder(nn.Module):
        """stands where fairseq.models.wav2vec.wav2vec2.TransformerEncoder stands: forward(x, padding_mask, layer) -> (x, None)"""

        def __init__(self, hf_encoder):
            super().__init__()
            self.hf = hf_encoder

        def forward(self, x, padding_mask=None, layer=None):
            assert padding_mask is None and layer is None
            return self.hf(x).last_hidden_state, None

    class GradMultiply:
        @staticmethod
        def apply(x, scale):
            return x

    def layer_norm(normalized_shape, eps=1e-5, elementwise_affine=True):   # fairseq.modules.LayerNorm without apex = torch's
        return nn.LayerNorm(normalized_shape, eps, elementwise_affine)

    def register_model(name, dataclass=None):
        return lambda cls: cls

    names = {
        "fairseq": module("fairseq", utils=module("fairseq.utils", get_available_activation_fns=lambda: ["relu", "gelu"])),
        "fairseq.utils": None,
        "fairseq.data": module("fairseq.data"),
        "fairseq.data.data_utils": module("fairseq.data.data_utils", compute_mask_indices=None),
        "fairseq.data.dictionary": module("fairseq.data.dictionary", Dictionary=object),
        "fairseq.dataclass": module("fairseq.dataclass", ChoiceEnum=lambda choices: str, FairseqDataclass=object),
        "fairseq.models": module("fairseq.models", BaseFairseqModel=nn.Module, register_model=register_model),
        "fairseq.models.wav2vec": module("fairseq.models.wav2vec"),
        "fairseq.models.wav2vec.wav2vec2": module("fairseq.models.wav2vec.wav2vec2", ConvFeatureExtractionModel=None,
                                                  TransformerEncoder=HFTransformerEncoder),
        "fairseq.modules": module("fairseq.modules", GradMultiply=GradMultiply, LayerNorm=layer_norm),
        "omegaconf": module("omegaconf", II=lambda key: None),
        "hubert_pretraining": module("hubert_pretraining", AVHubertPretrainingConfig=object, AVHubertPretrainingTask=object),
        "hubert_asr": module("hubert_asr"),
        "decoder": module("decoder", TransformerDecoder=None),
        "utils": module("utils", compute_mask_indices=None),    # hubert.py's `from utils import ...` (video_only/utils.py imports fairseq)
    }
    names["fairseq.utils"] = names["fairseq"].utils
    vdir = REF + "/N20EMv2/video_only"
    saved = {k: sys.modules.get(k) for k in list(names) + ["resnet", "hubert", "fairseq_interface"]}
    sys.modules.update(names)
    for k in ("resnet", "hubert", "fairseq_interface"):
        sys.modules.pop(k, None)
    sys.path.insert(0, vdir)
    try:
        import speechbrain  # noqa: F401  (fairseq_interface imports speechbrain.utils.data_utils)
        import hubert as ref_hubert
        import fairseq_interface as ref_iface
        import resnet as ref_resnet
    finally:
        sys.path.remove(vdir)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ref_hubert, ref_iface, ref_resnet, HFTransformerEncoder


def make_video_glue_cases():
    """a15 / f2: the AV-HuBERT video branch through the reference's OWN glue -- ``FairseqAVHubertPretrain.forward`` ->
    ``extract_features`` -> ``AVHubertModel.extract_finetune({"video": v, "audio": None})``: real lip front-end, zeros for the audio
    half, cat([audio, video]) order, LayerNorm(2E), post_extract_proj, encoder, the wrapper's whole-tensor output norm.  The
    objects are built without their __init__ (those need fairseq configs / a checkpoint file): ``__new__`` + the attributes the
    methods read, exactly the submodules ``AVHubertModel.__init__`` creates (hubert.py:344-394)."""
    import torch.nn as nn
    from transformers.models.wav2vec2.modeling_wav2vec2 import Wav2Vec2Encoder, Wav2Vec2EncoderStableLayerNorm, Wav2Vec2Config
    ref_hubert, ref_iface, ref_resnet, HFEnc = import_reference_video()
    out = {}
    for name, cfg_name, B, T, HW, seed, output_norm in [("tiny_stable", "tiny-avhubert-video", 2, 9, 40, 6986, True),
                                                         ("tiny_stable_t1", "tiny-avhubert-video", 1, 1, 32, 6987, False),
                                                         ("tiny_postln", "tiny-avhubert-video-postln", 2, 7, 36, 6988, True)]:
        cfg = PRESETS[cfg_name]
        E = cfg.hidden_size
        sd = W.seeded_avhubert_video_state_dict(cfg, seed=seed)
        hc = Wav2Vec2Config(hidden_size=E, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                            intermediate_size=cfg.intermediate_size, do_stable_layer_norm=cfg.do_stable_layer_norm,
                            num_conv_pos_embeddings=cfg.num_conv_pos_embeddings,
                            num_conv_pos_embedding_groups=cfg.num_conv_pos_embedding_groups, layer_norm_eps=cfg.layer_norm_eps,
                            hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, layerdrop=0.0,
                            attn_implementation="eager")
        hf_enc = (Wav2Vec2EncoderStableLayerNorm if cfg.do_stable_layer_norm else Wav2Vec2Encoder)(hc).eval()
        model = ref_hubert.AVHubertModel.__new__(ref_hubert.AVHubertModel)
        nn.Module.__init__(model)
        sub_cfg = types.SimpleNamespace(encoder_embed_dim=E, encoder_layers=0)
        model.feature_extractor_video = ref_hubert.SubModel(resnet=ref_resnet.ResEncoder(relu_type="prelu", weights=None),
                                                            input_dim=512, cfg=sub_cfg)
        model.encoder_embed_dim, model.modality_fuse, model.embed = E, "concat", 2 * E
        model.layer_norm = nn.LayerNorm(2 * E)
        model.post_extract_proj = nn.Linear(2 * E, E)
        model.dropout_input, model.dropout_features = nn.Dropout(0.0), nn.Dropout(0.0)
        model.feature_grad_mult, model.masking_type = 1.0, "input"
        model.encoder = HFEnc(hf_enc)
        # the fairseq-named state dict goes in through the modules' own load_state_dict: front-end / layer_norm / post_extract_proj
        # by their fairseq names, the encoder through the HF spelling of the same tensors (oracle.fairseq_to_hf_key is NOT used here)
        model.feature_extractor_video.load_state_dict({k[len("feature_extractor_video."):]: v for k, v in sd.items()
                                                       if k.startswith("feature_extractor_video.")}, strict=True)
        model.layer_norm.load_state_dict({"weight": sd["layer_norm.weight"], "bias": sd["layer_norm.bias"]})
        model.post_extract_proj.load_state_dict({"weight": sd["post_extract_proj.weight"], "bias": sd["post_extract_proj.bias"]})
        hf_sd = W.seeded_encoder_state_dict(cfg, seed=seed + 1)    # the tensors seeded_avhubert_video_state_dict renamed
        enc_sd = {k[len("encoder."):]: v for k, v in hf_sd.items() if k.startswith("encoder.")}
        own = hf_enc.state_dict()
        assert set(own) == set(enc_sd), (sorted(set(own) ^ set(enc_sd)))
        for k in own:   # same values as the fairseq-named entries of `sd` (weight-norm g / v spelling aside)
            fk = W.hf_to_fairseq_key("encoder." + k)
            if fk in sd:
                assert torch.equal(sd[fk], enc_sd[k]), k
        hf_enc.load_state_dict(enc_sd, strict=True)
        model.eval()
        wrap = ref_iface.FairseqAVHubertPretrain.__new__(ref_iface.FairseqAVHubertPretrain)
        nn.Module.__init__(wrap)
        wrap.model, wrap.freeze, wrap.normalize, wrap.output_norm = model, True, False, output_norm
        g = torch.Generator().manual_seed(seed + 7)
        video = torch.randn(B, 1, T, HW, HW, generator=g)
        with torch.no_grad():
            y = wrap({"video": video, "audio": None})
            front = model.forward_features(video, modality="video").transpose(1, 2).contiguous()   # (B, T, E): the video half
        assert y.shape == (B, T, E)
        out[name] = dict(cfg=cfg_name, B=B, T=T, HW=HW, weight_seed=seed, video_seed=seed + 7, output_norm=output_norm, out=y,
                         front=front, sd_sha256=sd_digest(sd))
        print("video_glue", name, tuple(y.shape), float(y.std()), float(front.std()))
    torch.save(out, os.path.join(HERE, "video_glue.pt"))


def make_dataio_cases():
    """speechbrain.utils.data_utils.batch_pad_right and speechbrain.dataio.batch.PaddedBatch on ragged 1-D signals and
    (frames, 4) annotations, as the recipes' DataLoader collates them (speechbrain/dataio/batch.py:101-137)."""
    from speechbrain.utils.data_utils import batch_pad_right
    from speechbrain.dataio.batch import PaddedBatch
    g = torch.Generator().manual_seed(31)
    out = {"pad": [], "batch": None}
    for lens in [(80000, 61234, 99999), (5,), (7, 7), (1, 9, 4, 9)]:
        ts = [torch.randn(n, generator=g) for n in lens]
        data, valid = batch_pad_right(ts)
        out["pad"].append(dict(tensors=ts if max(lens) < 100 else None, lens=lens, seed_note="randn from Generator(31) in order",
                               valid=valid, shape=tuple(data.shape), checksum=float(data.double().sum()),
                               tail_zero=bool((data[1, lens[1]:] == 0).all()) if len(lens) > 1 else True))
    ex = [{"id": f"song_{i}", "sig": torch.randn(n, generator=g), "anno": torch.randint(0, 5, (n // 3, 4), generator=g).float(),
           "cur_utter": i + 1} for i, n in enumerate((30, 18, 24))]
    pb = PaddedBatch(ex)
    out["batch"] = dict(examples=ex, sig=pb.sig.data, sig_lens=pb.sig.lengths, anno=pb.anno.data, anno_lens=pb.anno.lengths,
                        ids=pb.id, cur=pb.cur_utter)
    torch.save(out, os.path.join(HERE, "dataio.pt"))
    print("dataio", [c["valid"].tolist() for c in out["pad"]])


def make_fbank_ext_cases():
    """speechbrain.processing.features.Deltas / ContextWindow forward methods.  Deltas.__init__ moves its kernel to CUDA
    (features.py:812-816), so the object is built without __init__ and given the same attributes on the CPU."""
    from speechbrain.processing.features import Deltas, ContextWindow
    g = torch.Generator().manual_seed(61)
    out = {"deltas": [], "context": []}
    for B, T, C in [(2, 101, 40), (1, 3, 5), (3, 17, 120)]:
        x = torch.randn(B, T, C, generator=g)
        d = Deltas.__new__(Deltas)
        torch.nn.Module.__init__(d)
        d.n = 2
        d.denom = d.n * (d.n + 1) * (2 * d.n + 1) / 3
        d.register_buffer("kernel", torch.arange(-d.n, d.n + 1, dtype=torch.float32).repeat(C, 1, 1))
        out["deltas"].append(dict(x=x, expect=d(x)))
    for (B, T, C), (l, r) in [((2, 50, 8), (5, 5)), ((1, 9, 3), (2, 4)), ((2, 20, 6), (3, 0)), ((1, 4, 2), (5, 5))]:
        x = torch.randn(B, T, C, generator=g)
        out["context"].append(dict(x=x, left=l, right=r, expect=ContextWindow(left_frames=l, right_frames=r)(x)))
    torch.save(out, os.path.join(HERE, "fbank_ext.pt"))
    print("fbank_ext", [tuple(c["expect"].shape) for c in out["context"]])


def make_local_dirs(hi):
    """Local model directories as the reference's constructor accepts them (huggingface_interface.py:89-262): a HuggingFace
    directory (config.json + preprocessor_config.json + pytorch_model.bin) per model class, and a SpeechBrain-pretrained one
    (config.json + a *.ckpt whose keys carry the "model.wav2vec2." prefix).  The expected outputs come from the reference's
    REAL constructor (from_pretrained on the directory) and forward."""
    import shutil
    import tempfile
    from transformers import Wav2Vec2FeatureExtractor
    root = os.path.join(HERE, "local_ckpt")
    shutil.rmtree(root, ignore_errors=True)
    cases = [("tiny-wav2vec2-hf", "tiny-group", "bin", True, 31), ("tiny-hubert-bn-hf", "tiny-hubert-bn", "bin", False, 32),
             ("tiny-wavlm-hf", "tiny-wavlm", "bin", True, 33), ("tiny-data2vec-hf", "tiny-data2vec", "bin", True, 34),
             ("tiny-wav2vec2-sb", "tiny-layer", "ckpt", True, 35)]
    out = {}
    for name, cfg_name, kind, do_norm, seed in cases:
        cfg = PRESETS[cfg_name]
        d = os.path.join(root, name)
        os.makedirs(d)
        sd = W.seeded_encoder_state_dict(cfg, seed=seed)
        model = hf_model(cfg)
        full = dict(model.state_dict())
        full.update(sd)
        model.load_state_dict(full, strict=True)
        model.config.to_json_file(os.path.join(d, "config.json"))
        Wav2Vec2FeatureExtractor(do_normalize=do_norm).save_pretrained(d)
        if kind == "bin":
            torch.save(model.state_dict(), os.path.join(d, "pytorch_model.bin"))
        else:  # what HuggingFaceWav2Vec2Pretrain's checkpoint looks like: base model under "model.wav2vec2.", plus heads
            ck = {"model.wav2vec2." + k: v for k, v in model.state_dict().items()}
            ck["model.project_hid.weight"] = torch.zeros(4, cfg.hidden_size)
            ck["model.quantizer.codevectors"] = torch.zeros(1, 8, 4)
            torch.save(ck, os.path.join(d, "wav2vec2.ckpt"))
        with tempfile.TemporaryDirectory() as tmp:
            ref = hi.HuggingFaceWav2Vec2(source=d, save_path=tmp)
        wav = synth_wav(2, 4000, seed=seed + 7)
        with torch.no_grad():
            y = ref(wav)
        out[name] = dict(cfg=cfg_name, wav=wav, out=y.clone(), normalize_wav=bool(ref.normalize_wav),
                         keys=sorted(ref.model.state_dict().keys()))
        print(name, tuple(y.shape), "normalize_wav", ref.normalize_wav, type(ref.model).__name__)
    torch.save(out, os.path.join(HERE, "local_ckpt.pt"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    hi, fusion_mod, utils = import_reference()
    jobs = {
        "tiny_group": lambda: make_encoder_case(hi, utils, "tiny_group", "tiny-group", 2, 4000, 11),
        "tiny_layer": lambda: make_encoder_case(hi, utils, "tiny_layer", "tiny-layer", 2, 4000, 12),
        "tiny_hubert": lambda: make_encoder_case(hi, utils, "tiny_hubert", "tiny-hubert", 2, 4000, 13),
        "tiny_group_ragged": lambda: make_encoder_case(hi, utils, "tiny_group_ragged", "tiny-group", 3, 6000, 14,
                                                       lens=[6000, 3300, 4711]),
        "base_c1": lambda: make_encoder_case(hi, utils, "base_c1", "wav2vec2-base", 1, 80000, 21, full=False),
        "base_b2": lambda: make_encoder_case(hi, utils, "base_b2", "wav2vec2-base", 2, 160000, 22, full=False),
        "large_c1": lambda: make_encoder_case(hi, utils, "large_c1", "wav2vec2-large-lv60", 1, 80000, 23, full=False),
        "hubert_large_c1": lambda: make_encoder_case(hi, utils, "hubert_large_c1", "hubert-large-ll60k", 1, 48000, 24,
                                                      full=False),
        # full-size 10 s goldens of the two LARGE BASELINE models (C3 / C5 geometry: 24 pre-LN layers, layer-norm conv stack)
        "large_b2": lambda: make_encoder_case(hi, utils, "large_b2", "wav2vec2-large-lv60", 2, 160000, 27, full=False),
        "hubert_large_b2": lambda: make_encoder_case(hi, utils, "hubert_large_b2", "hubert-large-ll60k", 2, 160000, 28, full=False),
        "tiny_hubert_bn": lambda: make_encoder_case(hi, utils, "tiny_hubert_bn", "tiny-hubert-bn", 2, 4000, 18),
        "tiny_wavlm": lambda: make_encoder_case(hi, utils, "tiny_wavlm", "tiny-wavlm", 2, 4000, 16),
        "tiny_wavlm_stable": lambda: make_encoder_case(hi, utils, "tiny_wavlm_stable", "tiny-wavlm-stable", 2, 4000, 17),
        "wavlm_base_c1": lambda: make_encoder_case(hi, utils, "wavlm_base_c1", "wavlm-base", 1, 48000, 26, full=False),
        "tiny_data2vec": lambda: make_encoder_case(hi, utils, "tiny_data2vec", "tiny-data2vec", 2, 4000, 15),
        "data2vec_base_c1": lambda: make_encoder_case(hi, utils, "data2vec_base_c1", "data2vec-audio-base", 1, 48000, 25, full=False),
        "fusion": lambda: make_fusion_cases(fusion_mod),
        "frame2note": lambda: make_frame2note_cases(utils),
        "note2frame": lambda: make_note2frame_cases(utils),
        "ctc_fbank": make_ctc_fbank_cases,
        "losses": make_loss_cases,
        "video_front": make_video_front_cases,
        "video_glue": make_video_glue_cases,
        "video_u8": make_video_u8_cases,
        "dataio": make_dataio_cases,
        "fbank_ext": make_fbank_ext_cases,
        "ckpt_tree": lambda: make_ckpt_tree(hi),
        "local_ckpt": lambda: make_local_dirs(hi),
    }
    for k, fn in jobs.items():
        if args.only and args.only != k:
            continue
        fn()


if __name__ == "__main__":
    main()
