"""ORACLE — test infrastructure, NOT product code.

CPU fp32 restatement of the reference's singing-transcription forward path, written from the
numerical recipe in SURVEY.md §9 with plain ``torch`` functional ops (the same ATen op classes the
reference reaches through HuggingFace ``transformers``: conv1d, group_norm / layer_norm, erf-GELU,
linear, softmax attention).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module; the product path (``svt_speechbrain_amd``) never does
and fails loudly when the HIP library is missing.

Parity pin: ``tests/golden/*.pt`` were generated in the build container by
``tests/golden/make_golden.py`` from an *import of the reference itself*
(``/root/reference/MIR_ST500/huggingface_interface.py`` + HF transformers 5.15.0 eager attention,
``speechbrain.nnet.linear.Linear``, ``N20EMv2/audio_visual/fusion.py``, ``MIR_ST500/utils.py``,
``speechbrain/decoders/ctc.py``, ``speechbrain/processing/features.py``) and this oracle is checked
against them in ``tests/test_oracle_golden.py`` (CPU, ``-m "not gpu"``).  The reference holds no golden
vectors of its own for the encoder / fusion / frame2note; its doctest known-answers for
``ctc_greedy_decode``/``filter_ctc_output``/``spectral_magnitude`` are included in the same test file.

Reference anchors (file:line relative to /root/reference, ``HF:`` = transformers
``models/wav2vec2/modeling_wav2vec2.py``):
  * wrapper forward            MIR_ST500/huggingface_interface.py:279-297
  * conv feature extractor     HF:254-323, 382-419
  * feature projection         HF:422-434  (HuBERT: modeling_hubert.py:216-232)
  * positional conv            HF:326-379
  * encoder (post-/pre-LN)     HF:575-654, 657-802 ; attention HF:438-548 ; FFN HF:551-572
  * frame head                 speechbrain/nnet/linear.py:63-76 ; slicing MIR_ST500/train_audio_ssl.py:41-46
  * per-frame decode           MIR_ST500/train_audio_ssl.py:93-100
  * frame2note                 MIR_ST500/utils.py:82-149
  * RCA fusion                 N20EMv2/audio_visual/fusion.py:54-79,137-183,192-210
  * ctc greedy                 speechbrain/decoders/ctc.py:297-383
  * Fbank chain                speechbrain/lobes/features.py:126-143, processing/features.py:133-188,327-356,490-712
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# audio encoder
# ------------------------------------------------------------------------------------------------
def _posconv_weight(sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """weight-norm(dim=2): W[:,:,j] = g[j] * v[:,:,j] / ||v[:,:,j]||_F  (HF:326-352; both key spellings)."""
    pc = prefix + "encoder.pos_conv_embed.conv."
    if pc + "parametrizations.weight.original0" in sd:
        g, v = sd[pc + "parametrizations.weight.original0"], sd[pc + "parametrizations.weight.original1"]
    elif pc + "weight_g" in sd:
        g, v = sd[pc + "weight_g"], sd[pc + "weight_v"]
    else:
        return sd[pc + "weight"]
    nrm = v.float().pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return g.float() * v.float() / nrm


def conv_feature_extractor(sd, cfg, x: torch.Tensor, prefix: str = "", taps: Optional[dict] = None) -> torch.Tensor:
    """(B, L) -> (B, C, T).  HF:382-419."""
    h = x[:, None, :]
    for i, (k, s) in enumerate(zip(cfg.conv_kernel, cfg.conv_stride)):
        p = f"{prefix}feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, sd[p + "conv.weight"], sd.get(p + "conv.bias"), stride=s)
        if cfg.feat_extract_norm == "group" and i == 0:
            c = h.shape[1]
            h = F.group_norm(h, c, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps=1e-5)
        elif cfg.feat_extract_norm == "layer":
            c = h.shape[1]
            h = F.layer_norm(h.transpose(1, 2), (c,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"],
                             eps=1e-5).transpose(1, 2)
        h = F.gelu(h)
        if taps is not None:
            taps[f"conv{i}"] = h.transpose(1, 2).contiguous()
    return h


def wavlm_position_bias(table: torch.Tensor, T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """HF modeling_wavlm.py WavLMAttention.compute_bias / _relative_positions_bucket: (H, T, T) bias from the (buckets, H)
    embedding of the bucketed key - query distance (half of the buckets per sign, exact below max_exact, log-spaced above)."""
    ctx = torch.arange(T, dtype=torch.long)[:, None]
    mem = torch.arange(T, dtype=torch.long)[None, :]
    rel = mem - ctx
    nb = num_buckets // 2
    bucket = (rel > 0).to(torch.long) * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
    large = torch.min((max_exact + large).to(torch.long), torch.full_like(rel, nb - 1))
    bucket = bucket + torch.where(rel < max_exact, rel, large)
    return table[bucket].permute(2, 0, 1)


def attention(sd, p: str, u: torch.Tensor, nheads: int, pos_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HF:466-548 eager path; no mask.  pos_bias (H,T,T): WavLM's relative position bias, gated per
    (clip, head, query) by a projection of the attention input (modeling_wavlm.py WavLMAttention.forward)."""
    B, T, D = u.shape
    dh = D // nheads
    q = F.linear(u, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(B, T, nheads, dh).transpose(1, 2)
    k = F.linear(u, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, T, nheads, dh).transpose(1, 2)
    v = F.linear(u, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, T, nheads, dh).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    if pos_bias is not None:
        gh = u.view(B, T, nheads, dh).permute(0, 2, 1, 3)
        proj = F.linear(gh, sd[p + "gru_rel_pos_linear.weight"], sd[p + "gru_rel_pos_linear.bias"])
        proj = proj.view(B, nheads, T, 2, 4).sum(-1)
        gate_a, gate_b = torch.sigmoid(proj).chunk(2, dim=-1)
        gate = gate_a * (gate_b * sd[p + "gru_rel_pos_const"] - 1.0) + 2.0          # (B, H, T, 1)
        scores = scores + gate * pos_bias.unsqueeze(0)
    a = torch.softmax(scores, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, T, D)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def encoder_forward(sd: Dict[str, torch.Tensor], cfg, wav: torch.Tensor, normalize_wav: bool = True,
                    output_norm: bool = True, prefix: str = "", taps: Optional[dict] = None) -> torch.Tensor:
    """``HuggingFaceWav2Vec2.extract_features``: f32 (B, L) -> f32 (B, T, D)."""
    x = wav.float()
    if normalize_wav:
        x = F.layer_norm(x, x.shape)  # whole-batch LN, eps 1e-5
    h = conv_feature_extractor(sd, cfg, x, prefix, taps).transpose(1, 2)  # (B, T, C)
    return encoder_tail(sd, cfg, h, output_norm, prefix, taps)


def encoder_tail(sd: Dict[str, torch.Tensor], cfg, h: torch.Tensor, output_norm: bool = True, prefix: str = "",
                 taps: Optional[dict] = None) -> torch.Tensor:
    """Everything after the conv feature extractor: feature projection, positional conv, transformer layers, the
    wrapper's whole-batch output norm.  h: (B, T, C) features."""
    C = h.shape[-1]
    eps = cfg.layer_norm_eps
    if cfg.feat_proj_layer_norm:
        h = F.layer_norm(h, (C,), sd[prefix + "feature_projection.layer_norm.weight"],
                         sd[prefix + "feature_projection.layer_norm.bias"], eps=eps)
    h = F.linear(h, sd[prefix + "feature_projection.projection.weight"],
                 sd[prefix + "feature_projection.projection.bias"])
    if taps is not None:
        taps["proj"] = h.clone()
    # positional conv embedding (HF:326-379)
    D = cfg.hidden_size
    kp, g = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    if getattr(cfg, "pos_conv_depth", 1) > 1:
        # data2vec-audio (HF modeling_data2vec_audio.py, Data2VecAudioPositionalConvEmbedding): a stack of plain grouped convs,
        # each followed by LayerNorm over channels without affine parameters (eps 1e-5) and GELU
        pos = h.transpose(1, 2)
        for i in range(cfg.pos_conv_depth):
            pl = prefix + f"encoder.pos_conv_embed.layers.{i}.conv."
            pos = F.conv1d(pos, sd[pl + "weight"], sd[pl + "bias"], padding=kp // 2, groups=g)
            if kp % 2 == 0:
                pos = pos[:, :, :-1]
            pos = F.gelu(F.layer_norm(pos.transpose(1, 2), (D,), None, None, 1e-5)).transpose(1, 2)
        pos = pos.transpose(1, 2)
    else:
        w = _posconv_weight(sd, prefix)
        xin = h.transpose(1, 2)
        bn = prefix + "encoder.pos_conv_embed.batch_norm."
        if getattr(cfg, "conv_pos_batch_norm", False):
            # HF modeling_hubert.py HubertPositionalConvEmbedding with conv_pos_batch_norm: eval-mode BatchNorm1d (eps 1e-5) in
            # front of a plain conv; the conv's zero padding is applied AFTER the norm
            xin = F.batch_norm(xin, sd[bn + "running_mean"], sd[bn + "running_var"], sd[bn + "weight"], sd[bn + "bias"], False, 0.0, 1e-5)
        pos = F.conv1d(xin, w, sd[prefix + "encoder.pos_conv_embed.conv.bias"], padding=kp // 2, groups=g)
        if kp % 2 == 0:
            pos = pos[:, :, :-1]
        pos = F.gelu(pos).transpose(1, 2)
    h = h + pos
    if taps is not None:
        taps["pos"] = h.clone()
    nh = cfg.num_attention_heads

    def ln(t, key):
        return F.layer_norm(t, (D,), sd[prefix + key + ".weight"], sd[prefix + key + ".bias"], eps=eps)

    def ffn(t, p):
        t = F.gelu(F.linear(t, sd[p + "intermediate_dense.weight"], sd[p + "intermediate_dense.bias"]))
        return F.linear(t, sd[p + "output_dense.weight"], sd[p + "output_dense.bias"])

    pb = None
    if getattr(cfg, "rel_pos_buckets", 0):
        pb = wavlm_position_bias(sd[prefix + "encoder.layers.0.attention.rel_attn_embed.weight"], h.shape[1], cfg.rel_pos_buckets,
                                 cfg.rel_pos_max_distance)
    if not cfg.do_stable_layer_norm:
        h = ln(h, "encoder.layer_norm")
        for l in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{l}"
            h = ln(h + attention(sd, prefix + p + ".attention.", h, nh, pb), p + ".layer_norm")
            h = ln(h + ffn(h, prefix + p + ".feed_forward."), p + ".final_layer_norm")
            if taps is not None:
                taps[f"layer{l}"] = h.clone()
    else:
        for l in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{l}"
            h = h + attention(sd, prefix + p + ".attention.", ln(h, p + ".layer_norm"), nh, pb)
            h = h + ffn(ln(h, p + ".final_layer_norm"), prefix + p + ".feed_forward.")
            if taps is not None:
                taps[f"layer{l}"] = h.clone()
        h = ln(h, "encoder.layer_norm")
    if output_norm:
        h = F.layer_norm(h, h.shape)
    return h


def head_forward(feats: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    return F.linear(feats, w, b)


def decode_frames(logits: torch.Tensor, n_octave: int = 4, n_class: int = 12):
    """Per-frame decode of ``MIR_ST500/train_audio_ssl.py:41-46,93-100``.
    logits (..., 2 + (n_octave+1) + (n_class+1)) -> p_on f32, p_off f32, oct i64, pc i64."""
    p_on = torch.sigmoid(logits[..., 0])
    p_off = torch.sigmoid(logits[..., 1])
    octv = torch.argmax(logits[..., 2:2 + n_octave + 1], dim=-1)
    pc = torch.argmax(logits[..., 2 + n_octave + 1:], dim=-1)
    return p_on, p_off, octv, pc


def frame2note(frame_info: Sequence, onset_thres: float, offset_thres: float, frame_size: float = 1 / 49.8):
    """Greedy frame -> note scan (``MIR_ST500/utils.py:82-149``).  ``frame_info[i]`` =
    (p_on, p_off, octave, pitch_class); probabilities are float32 (numpy scalars / 0-dim tensors)."""
    n = len(frame_info)
    on = np.array([np.float32(f[0]) for f in frame_info], dtype=np.float32)
    notes: List[list] = []
    start = None
    votes: List[int] = []
    t_now = 0.0
    thr_on, thr_off = onset_thres, offset_thres

    def flush(t_end):
        if votes:
            notes.append([start, t_end, max(set(votes), key=votes.count) + 36])

    for i in range(n):
        t_now = frame_size * i
        p_on, p_off, octv, pc = frame_info[i]
        lo = max(i - 3, 0)
        hi = min(i + 4, n - 1)
        is_onset = bool(np.float32(p_on) >= thr_on) and bool(on[i] == np.amax(on[lo:hi]))
        if is_onset:
            if start is not None:
                flush(t_now)
            start = t_now
            votes = []
        elif bool(np.float32(p_off) >= thr_off):
            if start is not None:
                flush(t_now)
                start = None
                votes = []
        if start is not None:
            if int(octv) != 4 and int(pc) != 12:
                votes.append(int(int(octv) * 12 + int(pc)))
    if start is not None:
        flush(t_now)
    return notes


# ------------------------------------------------------------------------------------------------
# RCA fusion
# ------------------------------------------------------------------------------------------------
def _mha_packed(x_q, x_kv, w_in, b_in, w_out, b_out, nhead):
    B, Tq, D = x_q.shape
    Tk = x_kv.shape[1]
    dh = D // nhead
    q = F.linear(x_q, w_in[:D], b_in[:D]).view(B, Tq, nhead, dh).transpose(1, 2)
    k = F.linear(x_kv, w_in[D:2 * D], b_in[D:2 * D]).view(B, Tk, nhead, dh).transpose(1, 2)
    v = F.linear(x_kv, w_in[2 * D:], b_in[2 * D:]).view(B, Tk, nhead, dh).transpose(1, 2)
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, Tq, D)
    return F.linear(o, w_out, b_out)


def rca_layer(sd, p, src_kv, src_q, alpha, nhead, eps=1e-6):
    """``RCALayer.forward`` (fusion.py:137-183): post-norm, shared attention weights, ReLU FFN."""
    D = src_kv.shape[-1]
    w_in, b_in = sd[p + "self_att.att.in_proj_weight"], sd[p + "self_att.att.in_proj_bias"]
    w_o, b_o = sd[p + "self_att.att.out_proj.weight"], sd[p + "self_att.att.out_proj.bias"]
    sa = _mha_packed(src_kv, src_kv, w_in, b_in, w_o, b_o, nhead)
    ca = _mha_packed(src_q, src_kv, w_in, b_in, w_o, b_o, nhead)
    x = src_kv + sa * alpha + ca * (1 - alpha)
    x = F.layer_norm(x, (D,), sd[p + "norm1.norm.weight"], sd[p + "norm1.norm.bias"], eps=eps)
    y = F.linear(F.relu(F.linear(x, sd[p + "pos_ffn.ffn.0.weight"], sd[p + "pos_ffn.ffn.0.bias"])),
                 sd[p + "pos_ffn.ffn.3.weight"], sd[p + "pos_ffn.ffn.3.bias"])
    return F.layer_norm(x + y, (D,), sd[p + "norm2.norm.weight"], sd[p + "norm2.norm.bias"], eps=eps)


def fusion_forward(sd, audio: torch.Tensor, video: torch.Tensor, alpha: float = 0.5, nhead: int = 8,
                   prefix: str = "") -> torch.Tensor:
    """``FusionRCA.forward`` (fusion.py:192-210)."""
    B, T1, D = audio.shape
    T2 = video.shape[1]
    diff = T1 - T2
    if diff < 0:
        video = video[:, :diff]
    elif diff > 0:
        video = torch.cat([video, torch.zeros(video.shape[0], diff, D, dtype=video.dtype)], dim=1)
    pe = sd[prefix + "fusion.positional_encoding.pe"][:, :T1]
    s1 = audio + pe
    s2 = video + pe
    o1 = rca_layer(sd, prefix + "fusion.layer1.", s1, s2, alpha, nhead)
    o2 = rca_layer(sd, prefix + "fusion.layer2.", s2, s1, alpha, nhead)
    return o1 + o2


# ------------------------------------------------------------------------------------------------
# CTC greedy + Fbank (named by north_star; not on a recipe path — SURVEY.md F3/F4)
# ------------------------------------------------------------------------------------------------
def filter_ctc_output(seq: Sequence, blank_id=-1) -> list:
    out = []
    prev = object()
    for s in seq:
        if s != prev:
            out.append(s)
        prev = s
    return [s for s in out if s != blank_id]


def ctc_greedy_decode(probs: torch.Tensor, seq_lens: torch.Tensor, blank_id: int = -1) -> List[List[int]]:
    """``speechbrain/decoders/ctc.py:341-383``: probs (B, T, V), relative lens (B,)."""
    if isinstance(blank_id, int) and blank_id < 0:
        blank_id = probs.shape[-1] + blank_id
    T = probs.shape[1]
    out = []
    for seq, rel in zip(probs, seq_lens):
        n = int(torch.round(rel * T))
        ids = torch.argmax(seq[:n], dim=-1).tolist() if n > 0 else []
        out.append(filter_ctc_output(ids, blank_id))
    return out


def mel_filterbank(n_mels=40, n_fft=400, sr=16000, f_min=0.0, f_max=8000.0) -> torch.Tensor:
    """Triangular mel matrix (n_stft, n_mels): ``processing/features.py:452-470,586-610``."""
    def to_mel(hz):
        return 2595.0 * math.log10(1.0 + hz / 700.0)
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = hz[1:] - hz[:-1]
    band = band[:-1]
    f_central = hz[1:-1]
    n_stft = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sr // 2, n_stft)
    fc = f_central.repeat(n_stft, 1).transpose(0, 1)
    bd = band.repeat(n_stft, 1).transpose(0, 1)
    slope = (all_freqs.repeat(n_mels, 1) - fc) / bd
    left = slope + 1.0
    right = -slope + 1.0
    fb = torch.max(torch.zeros(1), torch.min(left, right)).transpose(0, 1)
    return fb  # (n_stft, n_mels)


def fbank(wav: torch.Tensor, n_mels=40, n_fft=400, win=400, hop=160, sr=16000, top_db=80.0) -> torch.Tensor:
    """Default ``Fbank`` chain (deltas=False, context=False): STFT -> power -> mel -> dB -> top_db clip."""
    window = torch.hamming_window(win)
    st = torch.stft(wav.float(), n_fft, hop, win, window, center=True, pad_mode="constant", normalized=False,
                    onesided=True, return_complex=True)
    power = (st.real ** 2 + st.imag ** 2).transpose(1, 2)  # (B, frames, n_stft)
    fb = torch.matmul(power, mel_filterbank(n_mels, n_fft, sr))
    db = 10.0 * torch.log10(torch.clamp(fb, min=1e-10))
    db = db - 10.0 * math.log10(max(1e-10, 1.0))  # db_multiplier with ref_value 1.0 -> 0
    mx = db.amax(dim=(-2, -1)) - top_db
    return torch.max(db, mx.view(-1, 1, 1))


# =================================================================================================
# Validation losses — restated from speechbrain/nnet/losses.py, pinned by
# tests/golden/losses.pt (generated by importing the reference's own functions)
# =================================================================================================
def truncate(predictions: torch.Tensor, targets: torch.Tensor, allowed_len_diff: int = 3):
    """losses.py:594-621: equalise dim 1 when the difference is within the tolerance, else ValueError."""
    d = predictions.shape[1] - targets.shape[1]
    if d == 0:
        return predictions, targets
    if abs(d) > allowed_len_diff:
        raise ValueError("Predictions and targets should be same length, but got %s and %s respectively."
                         % (predictions.shape[1], targets.shape[1]))
    if d < 0:
        return predictions, targets[:, : predictions.shape[1]]
    return predictions[:, : targets.shape[1]], targets


def length_mask(rel_len: Optional[torch.Tensor], batch: int, frames: int) -> torch.Tensor:
    """dataio/dataio.py:661-706 as compute_masked_loss calls it (losses.py:655-657): fp32 `t < rel_len * frames`."""
    if rel_len is None:
        return torch.ones(batch, frames)
    lim = rel_len.float() * frames
    return (torch.arange(frames, dtype=torch.float32).expand(batch, frames) < lim.unsqueeze(1)).float()


def _reduce(loss, mask, reg, reduction, smoothing):
    """losses.py:664-684 (loss, reg: (B,T) already masked)."""
    B = loss.shape[0]
    if reduction == "mean":
        l, r = loss.sum() / mask.sum(), reg.sum() / mask.sum()
    elif reduction == "batchmean":
        l, r = loss.sum() / B, reg.sum() / B
    elif reduction == "batch":
        l, r = loss.sum(1) / mask.sum(1), reg.sum(1) / mask.sum(1)
    else:
        l, r = loss, reg
    return l if smoothing == 0 else -smoothing * r + (1 - smoothing) * l


def bce_loss(inputs, targets, length=None, pos_weight=None, reduction="mean", allowed_len_diff=3):
    """losses.py:458-519: BCE-with-logits x length mask, reduced as compute_masked_loss."""
    if inputs.dim() == targets.dim() + 1:
        inputs = inputs.squeeze(-1)
    one_d = inputs.dim() == 1
    if one_d:
        inputs, targets = inputs.unsqueeze(1), targets.unsqueeze(1)
    inputs, targets = truncate(inputs, targets, allowed_len_diff)
    x, y = inputs.float(), targets.float()
    sp = torch.log1p(torch.exp(-x.abs())) + torch.clamp(-x, min=0)
    lw = 1.0 if pos_weight is None else 1 + (float(pos_weight) - 1) * y
    mask = length_mask(length, x.shape[0], x.shape[1])
    loss = ((1 - y) * x + lw * sp) * mask
    out = _reduce(loss, mask, torch.zeros_like(loss), reduction, 0.0)
    return out.squeeze(1) if (one_d and reduction == "none") else out


def nll_loss(log_probabilities, targets, length=None, label_smoothing=0.0, allowed_len_diff=3, reduction="mean"):
    """losses.py:402-455."""
    two_d = log_probabilities.dim() == 2
    if two_d:
        log_probabilities, targets = log_probabilities.unsqueeze(1), targets.unsqueeze(1)
    else:
        log_probabilities, targets = truncate(log_probabilities, targets, allowed_len_diff)
    lp, tg = log_probabilities.float(), targets.long()
    mask = length_mask(length, lp.shape[0], lp.shape[1])
    picked = -torch.gather(lp, 2, tg.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    picked = torch.where(tg == -100, torch.zeros_like(picked), picked)
    loss = picked * mask
    reg = lp.mean(dim=2) * mask
    out = _reduce(loss, mask, reg, reduction, label_smoothing)
    return out.squeeze(1) if (two_d and reduction == "none") else out


def softmax(x: torch.Tensor, apply_log: bool = False) -> torch.Tensor:
    """nnet/activations.py:14-75 over the last axis."""
    return torch.log_softmax(x.float(), -1) if apply_log else torch.softmax(x.float(), -1)


# =================================================================================================
# AV-HuBERT lip front-end — restated from N20EMv2/video_only/resnet.py,
# pinned by tests/golden/video_front.pt (generated by importing that file itself)
# =================================================================================================
def _bn_eval(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def video_transform_eval(roi, crop: int = 88, image_mean: float = 0.421, image_std: float = 0.165):
    """The video recipes' evaluation transform on one song's ``(T, H, W)`` uint8 lip ROI -> ``(T, crop, crop)`` float32:
    ``Normalize(0.0, 255.0)`` -> ``CenterCrop((88, 88))`` -> ``Normalize(0.421, 0.165)`` (N20EMv2/video_only/train_video_ssl.py:445-457;
    utils.py:53-61 -- ``(frames - mean) / std`` on the numpy array, i.e. float64 -- and :79-83 -- offsets ``int(round(w - tw) / 2.)``),
    then ``.astype(np.float32)`` (train_video_ssl.py:530-533).  Plain numpy, same operation order."""
    import numpy as np
    f = (np.asarray(roi) - 0.0) / 255.0
    _, h, w = f.shape
    dw = int(round((w - crop)) / 2.)
    dh = int(round((h - crop)) / 2.)
    f = f[:, dh:dh + crop, dw:dw + crop]
    return ((f - image_mean) / image_std).astype(np.float32)


def video_frontend_forward(sd: Dict[str, torch.Tensor], video: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """video (B,1,T,H,W) -> (B,T,embed).  resnet.py:150-158 (ResEncoder.forward), :36-72 (BasicBlock), :125-132
    (trunk), :183-187 (SubModel: proj on the transposed features; returned here already as (B,T,E))."""
    g = lambda k: sd[prefix + k]  # noqa: E731
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    B, _, T, _, _ = video.shape
    x = F.conv3d(video.float(), g("resnet.frontend3D.0.weight"), None, stride=(1, 2, 2), padding=(2, 3, 3))
    x = _bn_eval(x, sub, "resnet.frontend3D.1")
    x = F.prelu(x, g("resnet.frontend3D.2.weight"))
    x = F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
    x = x.transpose(1, 2).reshape(B * T, 64, x.shape[3], x.shape[4])
    for li in range(1, 5):
        for b in range(2):
            p = f"resnet.trunk.layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            out = F.conv2d(x, sub[p + ".conv1.weight"], None, stride=stride, padding=1)
            out = F.prelu(_bn_eval(out, sub, p + ".bn1"), sub[p + ".relu1.weight"])
            out = _bn_eval(F.conv2d(out, sub[p + ".conv2.weight"], None, stride=1, padding=1), sub, p + ".bn2")
            res = x
            if (p + ".downsample.0.weight") in sub:
                res = _bn_eval(F.conv2d(x, sub[p + ".downsample.0.weight"], None, stride=stride), sub, p + ".downsample.1")
            x = F.prelu(out + res, sub[p + ".relu2.weight"])
    x = x.mean(dim=(2, 3)).view(B, T, 512)
    return F.linear(x, g("proj.weight"), g("proj.bias"))


# fairseq (AV-HuBERT / HuBERT) parameter names -> the HF names the restatement above uses.  The same table as HF's public
# convert_hubert_original_pytorch_checkpoint_to_pytorch.py MAPPING; fairseq's TransformerEncoder
# (pos_conv -> [layer_norm] -> layers -> [layer_norm], layer_norm_first = HF do_stable_layer_norm) is the module HF ported.
FAIRSEQ_TO_HF = [
    ("layer_norm.", "feature_projection.layer_norm."),
    ("post_extract_proj.", "feature_projection.projection."),
    ("encoder.pos_conv.0.", "encoder.pos_conv_embed.conv."),
    ("encoder.layer_norm.", "encoder.layer_norm."),
]
FAIRSEQ_LAYER_TO_HF = [
    ("self_attn.k_proj.", "attention.k_proj."), ("self_attn.v_proj.", "attention.v_proj."),
    ("self_attn.q_proj.", "attention.q_proj."), ("self_attn.out_proj.", "attention.out_proj."),
    ("self_attn_layer_norm.", "layer_norm."), ("fc1.", "feed_forward.intermediate_dense."),
    ("fc2.", "feed_forward.output_dense."), ("final_layer_norm.", "final_layer_norm."),
]


def fairseq_to_hf_key(k: str) -> Optional[str]:
    if k.startswith("encoder.layers."):
        _, _, idx, rest = k.split(".", 3)
        for a, b in FAIRSEQ_LAYER_TO_HF:
            if rest.startswith(a):
                return f"encoder.layers.{idx}.{b}{rest[len(a):]}"
        return None
    for a, b in FAIRSEQ_TO_HF:
        if k.startswith(a):
            return b + k[len(a):]
    return None


def avhubert_video_forward(sd: Dict[str, torch.Tensor], cfg, video: torch.Tensor, output_norm: bool = True,
                           prefix: str = "") -> torch.Tensor:
    """``FairseqAVHubertPretrain.extract_features`` for {"video": video, "audio": None}
    (N20EMv2/video_only/fairseq_interface.py:461-476 over hubert.py:688-739): lip front-end -> cat([zeros, video], dim=C)
    -> LayerNorm(2E) -> post_extract_proj -> TransformerEncoder -> optional whole-tensor layer norm.
    PINNED (round 5) by tests/golden/video_glue.pt for everything the reference itself holds: make_golden.py runs the reference's
    own ``FairseqAVHubertPretrain.forward`` -> ``AVHubertModel.extract_finetune`` / ``forward_features`` / ``SubModel.forward``
    over the real ``resnet.ResEncoder`` (fairseq.* stubbed at import).  What cannot be pinned here: fairseq's
    ``TransformerEncoder`` class (fairseq is absent; an HF ``Wav2Vec2Encoder[StableLayerNorm]`` -- the module HF ported from it,
    pinned for the audio path -- stands in its place in the fixture)."""
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    fv = video_frontend_forward(sub, video, prefix="feature_extractor_video.")  # (B, T, E)
    feats = torch.cat([torch.zeros_like(fv), fv], dim=-1)                        # audio half first (hubert.py:706-707)
    hf = {}
    for k, v in sub.items():
        nk = fairseq_to_hf_key(k)
        if nk is not None:
            hf[nk] = v
    return encoder_tail(hf, cfg, feats, output_norm)


# ---- Fbank add-ons (speechbrain/processing/features.py:788-850, 853-940), pinned by tests/golden/fbank_ext.pt ----
def deltas(x: torch.Tensor, window_length: int = 5) -> torch.Tensor:
    """(B,T,C) -> time derivative with replicate padding: sum_k k x[t+k] / (n(n+1)(2n+1)/3)."""
    n = (window_length - 1) // 2
    denom = n * (n + 1) * (2 * n + 1) / 3
    xt = F.pad(x.float().transpose(1, 2), (n, n), mode="replicate")
    k = torch.arange(-n, n + 1, dtype=torch.float32).repeat(x.shape[-1], 1, 1)
    return (F.conv1d(xt, k, groups=x.shape[-1]) / denom).transpose(1, 2)


def context_window(x: torch.Tensor, left_frames: int = 5, right_frames: int = 5) -> torch.Tensor:
    """(B,T,C) -> (B,T,C*(left+right+1)), column c*ctx + j = x[t + j + lag - pad, c], zeros outside."""
    B, T, C = x.shape
    ctx, pad = left_frames + right_frames + 1, max(left_frames, right_frames)
    lag = max(right_frames - left_frames, 0)
    out = torch.zeros(B, T, C, ctx)
    for j in range(ctx):
        off = j + lag - pad
        lo, hi = max(0, -off), min(T, T - off)
        if hi > lo:
            out[:, lo:hi, :, j] = x[:, lo + off:hi + off, :]
    return out.reshape(B, T, C * ctx)
