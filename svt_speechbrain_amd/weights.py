"""Seeded synthetic weights with the HuggingFace / SpeechBrain state-dict key layout.

There is no network and no checkpoint on disk, so benchmarks, golden fixtures and tests all use
weights from this generator.  The *key names and shapes* are the contract the
drop-in must honour (SURVEY.md §5 "Checkpoint / resume"): ``wav2vec2.ckpt`` holds ``model.<hf_key>``,
the head holds ``w.weight``/``w.bias``, the fusion holds ``fusion.layer{1,2}...``.

The draw order is: keys in the order produced by ``encoder_param_shapes`` (a fixed, documented order),
one ``torch.randn`` per tensor from a single ``torch.Generator`` seeded with ``seed``.  Scales are
fan-in normalised so activations stay O(1) through 12-24 layers (a 0.02-std init would make the
softmax almost uniform and hide attention bugs).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import EncoderConfig


def encoder_param_shapes(cfg: EncoderConfig, old_weight_norm_keys: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """HF ``Wav2Vec2Model`` / ``HubertModel`` parameter names -> shapes (HF:254-434, 551-802)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}"
        sh[f"{p}.conv.weight"] = (c, cin, k)
        if cfg.conv_bias:
            sh[f"{p}.conv.bias"] = (c,)
        if cfg.feat_extract_norm == "layer" or (cfg.feat_extract_norm == "group" and i == 0):
            sh[f"{p}.layer_norm.weight"] = (c,)
            sh[f"{p}.layer_norm.bias"] = (c,)
        cin = c
    if not cfg.conv_kernel:  # features-in configuration (AV-HuBERT video branch): the projection reads conv_dim[0] features
        cin = cfg.conv_dim[0]
    D, F = cfg.hidden_size, cfg.intermediate_size
    if cfg.feat_proj_layer_norm:
        sh["feature_projection.layer_norm.weight"] = (cin,)
        sh["feature_projection.layer_norm.bias"] = (cin,)
    sh["feature_projection.projection.weight"] = (D, cin)
    sh["feature_projection.projection.bias"] = (D,)
    kp, g = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    if cfg.pos_conv_depth > 1:  # data2vec-audio: plain (no weight norm) grouped convs, one per stacked layer
        for i in range(cfg.pos_conv_depth):
            sh[f"encoder.pos_conv_embed.layers.{i}.conv.weight"] = (D, D // g, kp)
            sh[f"encoder.pos_conv_embed.layers.{i}.conv.bias"] = (D,)
    pc = "encoder.pos_conv_embed.conv"
    if cfg.pos_conv_depth > 1:
        pass
    elif cfg.conv_pos_batch_norm:  # HF HubertPositionalConvEmbedding with conv_pos_batch_norm: plain conv, BatchNorm1d in front
        sh[f"{pc}.weight"] = (D, D // g, kp)
        sh[f"{pc}.bias"] = (D,)
        bn = "encoder.pos_conv_embed.batch_norm"
        sh[f"{bn}.weight"] = (D,)
        sh[f"{bn}.bias"] = (D,)
        sh[f"{bn}.running_mean"] = (D,)
        sh[f"{bn}.running_var"] = (D,)
        sh[f"{bn}.num_batches_tracked"] = ()
    elif old_weight_norm_keys:
        sh[f"{pc}.bias"] = (D,)
        sh[f"{pc}.weight_g"] = (1, 1, kp)
        sh[f"{pc}.weight_v"] = (D, D // g, kp)
    else:
        sh[f"{pc}.bias"] = (D,)
        sh[f"{pc}.parametrizations.weight.original0"] = (1, 1, kp)
        sh[f"{pc}.parametrizations.weight.original1"] = (D, D // g, kp)
    sh["encoder.layer_norm.weight"] = (D,)
    sh["encoder.layer_norm.bias"] = (D,)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}"
        if cfg.rel_pos_buckets:  # WavLM (HF modeling_wavlm.py WavLMAttention.__init__): gate parameters in every layer, the
            # bucket embedding in layer 0 only; the bare Parameter precedes the sub-modules in state_dict order
            sh[f"{p}.attention.gru_rel_pos_const"] = (1, cfg.num_attention_heads, 1, 1)
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            sh[f"{p}.attention.{n}.weight"] = (D, D)
            sh[f"{p}.attention.{n}.bias"] = (D,)
        if cfg.rel_pos_buckets:
            sh[f"{p}.attention.gru_rel_pos_linear.weight"] = (8, D // cfg.num_attention_heads)
            sh[f"{p}.attention.gru_rel_pos_linear.bias"] = (8,)
            if l == 0:
                sh[f"{p}.attention.rel_attn_embed.weight"] = (cfg.rel_pos_buckets, cfg.num_attention_heads)
        sh[f"{p}.layer_norm.weight"] = (D,)
        sh[f"{p}.layer_norm.bias"] = (D,)
        sh[f"{p}.feed_forward.intermediate_dense.weight"] = (F, D)
        sh[f"{p}.feed_forward.intermediate_dense.bias"] = (F,)
        sh[f"{p}.feed_forward.output_dense.weight"] = (D, F)
        sh[f"{p}.feed_forward.output_dense.bias"] = (D,)
        sh[f"{p}.final_layer_norm.weight"] = (D,)
        sh[f"{p}.final_layer_norm.bias"] = (D,)
    return sh


def _draw(name: str, shape, gen: torch.Generator) -> torch.Tensor:
    x = torch.randn(shape, generator=gen, dtype=torch.float32)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.tensor(7, dtype=torch.int64)
    if leaf == "running_var":
        return 0.5 + x.abs()
    if leaf == "running_mean":
        return 0.3 * x
    if "layer_norm" in name or "batch_norm" in name or "norm" in name.split(".")[-2:-1]:
        return 1.0 + 0.1 * x if leaf == "weight" else 0.1 * x
    if leaf == "bias":
        return 0.05 * x
    if leaf == "gru_rel_pos_const":
        return 1.0 + 0.3 * x
    if "rel_attn_embed" in name:
        return 1.5 * x          # bias values O(1): a wrong bucket or gate shows in the logits
    if "gru_rel_pos_linear" in name and leaf == "weight":
        return x * (1.0 / math.sqrt(shape[1]))
    if leaf in ("original0", "weight_g"):
        return 0.6 + 0.1 * x.abs()  # per-tap gain g[j]; norm of v per tap is O(sqrt(D*D/g)) -> W ~ small
    if leaf in ("original1", "weight_v"):
        return x
    if len(shape) == 3:  # conv (Cout, Cin, k): kaiming-like so GELU outputs stay O(1)
        fan_in = shape[1] * shape[2]
        return x * math.sqrt(2.0 / fan_in)
    if len(shape) == 2:  # linear (out, in)
        # Sharper attention (q,k x3) and a damped attention branch (out_proj x0.3) keep the 12 post-LN
        # layers from averaging every frame into the same vector (measured: per-frame logit std 0.03 -> 1.0),
        # so per-frame argmax / note parity checks are not trivially satisfied.
        gain = 3.0 if (".q_proj." in name or ".k_proj." in name) else (0.3 if ".attention.out_proj." in name else 1.0)
        return x * (gain / math.sqrt(shape[1]))
    return x


def seeded_encoder_state_dict(cfg: EncoderConfig, seed: int = 1986, prefix: str = "",
                              old_weight_norm_keys: bool = False) -> "OrderedDict[str, torch.Tensor]":
    gen = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in encoder_param_shapes(cfg, old_weight_norm_keys).items():
        t = _draw(k, shp, gen)
        if k.endswith("original0") or k.endswith("weight_g"):
            # make the effective pos-conv weight O(1/sqrt(fan_in)): g = gain * sqrt(#elements per tap)/sqrt(fan_in)
            D = cfg.hidden_size
            g = cfg.num_conv_pos_embedding_groups
            per_tap = D * (D // g)
            fan_in = (D // g) * cfg.num_conv_pos_embeddings
            t = t * math.sqrt(per_tap) / math.sqrt(fan_in)
        sd[prefix + k] = t.contiguous()
    return sd


def seeded_head_state_dict(d_in: int, n_out: int = 20, seed: int = 2986) -> "OrderedDict[str, torch.Tensor]":
    """``speechbrain.nnet.linear.Linear`` keys (``speechbrain/nnet/linear.py:41-61``)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    sd["w.weight"] = (torch.randn((n_out, d_in), generator=gen) * (2.0 / math.sqrt(d_in))).contiguous()
    sd["w.bias"] = (0.3 * torch.randn((n_out,), generator=gen)).contiguous()
    return sd


def positional_encoding_table(d_model: int, max_len: int = 2500) -> torch.Tensor:
    """Sinusoidal table (1, max_len, d_model); formula of
    ``speechbrain/lobes/models/transformer/Transformer.py:200-213``."""
    pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    den = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * den)
    pe[:, 1::2] = torch.cos(pos * den)
    return pe.unsqueeze(0)


def fusion_param_shapes(d_model: int = 1024, d_ffn: int = 3072, max_len: int = 2500):
    """``FusionRCA`` state-dict keys (``N20EMv2/audio_visual/fusion.py:9-209`` via sb wrappers)."""
    sh = OrderedDict()
    sh["fusion.positional_encoding.pe"] = (1, max_len, d_model)
    for l in ("layer1", "layer2"):
        p = f"fusion.{l}"
        sh[f"{p}.self_att.att.in_proj_weight"] = (3 * d_model, d_model)
        sh[f"{p}.self_att.att.in_proj_bias"] = (3 * d_model,)
        sh[f"{p}.self_att.att.out_proj.weight"] = (d_model, d_model)
        sh[f"{p}.self_att.att.out_proj.bias"] = (d_model,)
        sh[f"{p}.pos_ffn.ffn.0.weight"] = (d_ffn, d_model)
        sh[f"{p}.pos_ffn.ffn.0.bias"] = (d_ffn,)
        sh[f"{p}.pos_ffn.ffn.3.weight"] = (d_model, d_ffn)
        sh[f"{p}.pos_ffn.ffn.3.bias"] = (d_model,)
        sh[f"{p}.norm1.norm.weight"] = (d_model,)
        sh[f"{p}.norm1.norm.bias"] = (d_model,)
        sh[f"{p}.norm2.norm.weight"] = (d_model,)
        sh[f"{p}.norm2.norm.bias"] = (d_model,)
    return sh


def seeded_fusion_state_dict(d_model: int = 1024, d_ffn: int = 3072, seed: int = 3986,
                             max_len: int = 2500) -> "OrderedDict[str, torch.Tensor]":
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in fusion_param_shapes(d_model, d_ffn, max_len).items():
        if k.endswith(".pe"):
            sd[k] = positional_encoding_table(d_model, max_len)
            continue
        sd[k] = _draw(k, shp, gen).contiguous()
    return sd


def count_params(sd: Dict[str, torch.Tensor]) -> int:
    return sum(int(v.numel()) for v in sd.values())


# ---- AV-HuBERT lip front-end (ResNet-18 + projection), SURVEY.md §8 a15 ----
def video_frontend_param_shapes(embed_dim: int = 1024) -> "OrderedDict[str, Tuple[int, ...]]":
    """``SubModel.state_dict()`` of ``N20EMv2/video_only/resnet.py:174-187`` (= the ``feature_extractor_video.*`` keys of an
    AV-HuBERT checkpoint): 3-D stem (:141-145), four stages of two PReLU BasicBlocks (:36-72, 84-88), 1x1 stride-2
    downsample convs on stages 2-4 (:21-25), Linear(512 -> embed_dim)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def bn(p, c):
        sh[f"{p}.weight"] = (c,)
        sh[f"{p}.bias"] = (c,)
        sh[f"{p}.running_mean"] = (c,)
        sh[f"{p}.running_var"] = (c,)
        sh[f"{p}.num_batches_tracked"] = ()

    sh["resnet.frontend3D.0.weight"] = (64, 1, 5, 7, 7)
    bn("resnet.frontend3D.1", 64)
    sh["resnet.frontend3D.2.weight"] = (64,)
    cin = 64
    for li, c in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            p = f"resnet.trunk.layer{li}.{b}"
            sh[f"{p}.conv1.weight"] = (c, cin if b == 0 else c, 3, 3)
            bn(f"{p}.bn1", c)
            sh[f"{p}.relu1.weight"] = (c,)
            sh[f"{p}.relu2.weight"] = (c,)
            sh[f"{p}.conv2.weight"] = (c, c, 3, 3)
            bn(f"{p}.bn2", c)
            if b == 0 and li > 1:
                sh[f"{p}.downsample.0.weight"] = (c, cin, 1, 1)
                bn(f"{p}.downsample.1", c)
        cin = c
    sh["proj.weight"] = (embed_dim, 512)
    sh["proj.bias"] = (embed_dim,)
    return sh


def seeded_video_frontend_state_dict(embed_dim: int = 1024, seed: int = 4986, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    """Seeded values in the key order above: He-normal convs (resnet.py:94-97), batch-norm statistics away from the
    identity (so a wrong fold shows), PReLU slopes around 0.25."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shape in video_frontend_param_shapes(embed_dim).items():
        if k.endswith("num_batches_tracked"):
            v = torch.tensor(100, dtype=torch.long)
        elif k.endswith("running_var"):
            v = torch.rand(shape, generator=g) + 0.5
        elif k.endswith("running_mean"):
            v = torch.randn(shape, generator=g) * 0.1
        elif ".bn2." in k and k.endswith("weight"):
            v = torch.rand(shape, generator=g) * 0.3 + 0.3  # keeps the residual stream O(1) through eight blocks
        elif ".bn" in k or "frontend3D.1" in k or "downsample.1" in k:
            v = torch.rand(shape, generator=g) * 0.6 + 0.7 if k.endswith("weight") else torch.randn(shape, generator=g) * 0.1
        elif "relu" in k or k.endswith("frontend3D.2.weight"):
            v = torch.rand(shape, generator=g) * 0.3 + 0.1
        elif k == "proj.bias":
            v = torch.randn(shape, generator=g) * 0.1
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            v = torch.randn(shape, generator=g) * math.sqrt((0.25 if k == "proj.weight" else 2.0) / fan_in)
        sd[prefix + k] = v
    return sd


# ---- AV-HuBERT video encoder: fairseq parameter names (hubert.py:344-394; fairseq TransformerEncoder) ----
HF_TO_FAIRSEQ = [
    ("feature_projection.layer_norm.", "layer_norm."),
    ("feature_projection.projection.", "post_extract_proj."),
    ("encoder.pos_conv_embed.conv.", "encoder.pos_conv.0."),
    ("encoder.layer_norm.", "encoder.layer_norm."),
]
HF_LAYER_TO_FAIRSEQ = [
    ("attention.k_proj.", "self_attn.k_proj."), ("attention.v_proj.", "self_attn.v_proj."),
    ("attention.q_proj.", "self_attn.q_proj."), ("attention.out_proj.", "self_attn.out_proj."),
    ("layer_norm.", "self_attn_layer_norm."), ("feed_forward.intermediate_dense.", "fc1."),
    ("feed_forward.output_dense.", "fc2."), ("final_layer_norm.", "final_layer_norm."),
]


def hf_to_fairseq_key(k: str) -> str:
    if k.startswith("encoder.layers."):
        _, _, idx, rest = k.split(".", 3)
        for a, b in HF_LAYER_TO_FAIRSEQ:
            if rest.startswith(a):
                return f"encoder.layers.{idx}.{b}{rest[len(a):]}"
        raise KeyError(k)
    for a, b in HF_TO_FAIRSEQ:
        if k.startswith(a):
            return b + k[len(a):]
    raise KeyError(k)


def fairseq_to_hf_key(k: str):
    if k.startswith("encoder.layers."):
        _, _, idx, rest = k.split(".", 3)
        for b, a in HF_LAYER_TO_FAIRSEQ:
            if rest.startswith(a):
                return f"encoder.layers.{idx}.{b}{rest[len(a):]}"
        return None
    for b, a in HF_TO_FAIRSEQ:
        if k.startswith(a):
            return b + k[len(a):]
    return None


def seeded_avhubert_video_state_dict(cfg: EncoderConfig, seed: int = 5986, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    """``AVHubertModel.state_dict()`` entries the video-only forward reads: ``feature_extractor_video.*`` (lip front-end),
    ``layer_norm.*`` (over cat([audio, video]) = 2E), ``post_extract_proj.*``, ``encoder.*`` (fairseq names, weight_g/weight_v
    spelling of the positional conv's weight norm)."""
    E = cfg.hidden_size
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in seeded_video_frontend_state_dict(E, seed=seed).items():
        sd[prefix + "feature_extractor_video." + k] = v
    for k, v in seeded_encoder_state_dict(cfg, seed=seed + 1, old_weight_norm_keys=True).items():
        sd[prefix + hf_to_fairseq_key(k)] = v
    return sd
