"""Encoder configuration for the singing-transcription hot path.

The fields mirror the subset of the HuggingFace ``Wav2Vec2Config`` / ``HubertConfig`` that the
reference's encoder forward actually reads (reference call site:
``MIR_ST500/huggingface_interface.py:107-124,169-179``; HF ``modeling_wav2vec2.py:254-434,657-802``).
Everything is a runtime parameter: the C-ABI (``include/svt_mi355.h``, ``svt_config``) carries the
same fields, so large / HuBERT variants need no recompilation.
"""
from __future__ import annotations

import dataclasses
from typing import Tuple


@dataclasses.dataclass(frozen=True)
class EncoderConfig:
    name: str = "wav2vec2-base"
    family: str = "wav2vec2"  # "wav2vec2" | "hubert" | "data2vec" | "avhubert" (selects the key layout only)
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    feat_extract_norm: str = "group"  # "group": GroupNorm on conv0 only; "layer": LN after every conv
    conv_bias: bool = False
    do_stable_layer_norm: bool = False  # False: post-LN encoder; True: pre-LN + final LN
    feat_proj_layer_norm: bool = True  # always True for wav2vec2; optional for HuBERT
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    # data2vec-audio: a STACK of `pos_conv_depth` positional conv layers (kernel num_conv_pos_embeddings, no weight norm), each
    # followed by LayerNorm(no affine) + GELU (HF modeling_data2vec_audio.py Data2VecAudioPositionalConvLayer); 1 = wav2vec2 / HuBERT
    pos_conv_depth: int = 1
    # WavLM: gated relative position bias in every self-attention (HF modeling_wavlm.py WavLMAttention); 0 = none
    rel_pos_buckets: int = 0
    rel_pos_max_distance: int = 800
    # HuBERT ``conv_pos_batch_norm`` (HF modeling_hubert.py HubertPositionalConvEmbedding): BatchNorm1d in front of a plain
    # positional conv instead of the weight-normed conv
    conv_pos_batch_norm: bool = False

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def frames(self, n_samples: int) -> int:
        """Number of encoder frames for a waveform of ``n_samples`` (no padding in any conv)."""
        t = n_samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            t = (t - k) // s + 1
        return t

    def frame_counts(self, n_samples: int):
        out = []
        t = n_samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            t = (t - k) // s + 1
            out.append(t)
        return out

    def flops_per_clip(self, n_samples: int, head_out: int = 20) -> float:
        """Algorithmic FLOPs (2*MAC; norms/activations/softmax excluded) — SURVEY.md §8(d)."""
        ts = self.frame_counts(n_samples)
        fl = 0.0
        cin = 1
        for t, k, c in zip(ts, self.conv_kernel, self.conv_dim):
            fl += 2.0 * t * k * cin * c
            cin = c
        T = ts[-1]
        D, F = self.hidden_size, self.intermediate_size
        fl += 2.0 * T * cin * D  # feature projection
        fl += 2.0 * T * D * (D // self.num_conv_pos_embedding_groups) * self.num_conv_pos_embeddings * self.pos_conv_depth
        per_layer = 4 * 2.0 * T * D * D + 2 * 2.0 * T * T * D + 2 * 2.0 * T * D * F
        fl += self.num_hidden_layers * per_layer
        fl += 2.0 * T * D * head_out
        return fl


PRESETS = {
    "wav2vec2-base": EncoderConfig(),
    "wav2vec2-large-lv60": EncoderConfig(
        name="wav2vec2-large-lv60", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
        intermediate_size=4096, feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True),
    "hubert-large-ll60k": EncoderConfig(
        name="hubert-large-ll60k", family="hubert", hidden_size=1024, num_hidden_layers=24,
        num_attention_heads=16, intermediate_size=4096, feat_extract_norm="layer", conv_bias=True,
        do_stable_layer_norm=True, feat_proj_layer_norm=True),
    # Other public checkpoints the reference's `source` argument commonly names.  Their geometry is restated from the public
    # HF config.json files, which cannot be fetched offline: a local checkpoint directory's own config.json always wins
    # (huggingface_interface._config_from_dir).
    "hubert-base-ls960": EncoderConfig(
        name="hubert-base-ls960", family="hubert", hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
        intermediate_size=3072, feat_extract_norm="group", conv_bias=False, do_stable_layer_norm=False,
        feat_proj_layer_norm=False),
    "hubert-xlarge-ll60k": EncoderConfig(
        name="hubert-xlarge-ll60k", family="hubert", hidden_size=1280, num_hidden_layers=48, num_attention_heads=16,
        intermediate_size=5120, feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True, feat_proj_layer_norm=True),
    "wav2vec2-large": EncoderConfig(
        name="wav2vec2-large", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
        feat_extract_norm="group", conv_bias=False, do_stable_layer_norm=False),
    "wavlm-base": EncoderConfig(
        name="wavlm-base", family="wavlm", hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
        intermediate_size=3072, feat_extract_norm="group", conv_bias=False, do_stable_layer_norm=False, rel_pos_buckets=320),
    "wavlm-large": EncoderConfig(
        name="wavlm-large", family="wavlm", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
        intermediate_size=4096, feat_extract_norm="layer", conv_bias=False, do_stable_layer_norm=True, rel_pos_buckets=320),
    "tiny-wavlm": EncoderConfig(
        name="tiny-wavlm", family="wavlm", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(32,) * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4,
        rel_pos_buckets=32, rel_pos_max_distance=40),
    "tiny-wavlm-stable": EncoderConfig(
        name="tiny-wavlm-stable", family="wavlm", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(32,) * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4,
        feat_extract_norm="layer", do_stable_layer_norm=True, rel_pos_buckets=32, rel_pos_max_distance=40),
    "data2vec-audio-base": EncoderConfig(
        name="data2vec-audio-base", family="data2vec", hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
        intermediate_size=3072, feat_extract_norm="layer", conv_bias=False, do_stable_layer_norm=False,
        num_conv_pos_embeddings=19, pos_conv_depth=5),
    "data2vec-audio-large": EncoderConfig(
        name="data2vec-audio-large", family="data2vec", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
        intermediate_size=4096, feat_extract_norm="layer", conv_bias=False, do_stable_layer_norm=False,
        num_conv_pos_embeddings=19, pos_conv_depth=5),
    "tiny-data2vec": EncoderConfig(
        name="tiny-data2vec", family="data2vec", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(32,) * 7, feat_extract_norm="layer", conv_bias=False, num_conv_pos_embeddings=5,
        num_conv_pos_embedding_groups=4, pos_conv_depth=3),
    # Small configurations for golden fixtures / fast parity tests (SURVEY.md §8c "tiny config").
    "tiny-group": EncoderConfig(
        name="tiny-group", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(32,) * 7, num_conv_pos_embeddings=16,
        num_conv_pos_embedding_groups=4),
    "tiny-layer": EncoderConfig(
        name="tiny-layer", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(32,) * 7, num_conv_pos_embeddings=16,
        num_conv_pos_embedding_groups=4, feat_extract_norm="layer", conv_bias=True,
        do_stable_layer_norm=True),
    "tiny-hubert": EncoderConfig(
        name="tiny-hubert", family="hubert", hidden_size=64, num_hidden_layers=2,
        num_attention_heads=4, intermediate_size=128, conv_dim=(32,) * 7,
        num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, feat_extract_norm="layer",
        conv_bias=True, do_stable_layer_norm=True, feat_proj_layer_norm=False),
    "tiny-hubert-bn": EncoderConfig(
        name="tiny-hubert-bn", family="hubert", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(32,) * 7, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4,
        feat_extract_norm="group", conv_bias=False, do_stable_layer_norm=False, feat_proj_layer_norm=False,
        conv_pos_batch_norm=True),
    # AV-HuBERT video branch (features-in: no waveform conv stack; the transformer reads cat([audio, video]) features of
    # width 2 * hidden_size).  LARGE: 24 layers, layer_norm_first (the public large_vox_iter5 / self_large_vox_433h cfg)
    "avhubert-large-video": EncoderConfig(
        name="avhubert-large-video", family="avhubert", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
        intermediate_size=4096, conv_dim=(2048,), conv_kernel=(), conv_stride=(), do_stable_layer_norm=True,
        feat_proj_layer_norm=True),
    "avhubert-base-video": EncoderConfig(
        name="avhubert-base-video", family="avhubert", hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
        intermediate_size=3072, conv_dim=(1536,), conv_kernel=(), conv_stride=(), do_stable_layer_norm=True,
        feat_proj_layer_norm=True),
    "tiny-avhubert-video": EncoderConfig(
        name="tiny-avhubert-video", family="avhubert", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(128,), conv_kernel=(), conv_stride=(), num_conv_pos_embeddings=16,
        num_conv_pos_embedding_groups=4, do_stable_layer_norm=True, feat_proj_layer_norm=True),
    # the same branch over a post-LN transformer (fairseq layer_norm_first = False): the glue golden covers both encoder forms
    "tiny-avhubert-video-postln": EncoderConfig(
        name="tiny-avhubert-video-postln", family="avhubert", hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=128, conv_dim=(128,), conv_kernel=(), conv_stride=(), num_conv_pos_embeddings=16,
        num_conv_pos_embedding_groups=4, do_stable_layer_norm=False, feat_proj_layer_norm=True),
}


# ``do_normalize`` of the hub models' preprocessor_config.json (the reference sets ``normalize_wav`` from it,
# MIR_ST500/huggingface_interface.py:103,130).  The files cannot be fetched offline; the values are restated from the
# fairseq task configs the checkpoints were trained with (``task.normalize``: False for the LibriSpeech-960 BASE models
# of HuBERT and WavLM, True for wav2vec 2.0, data2vec and every LARGE model).  A local model directory's own
# preprocessor_config.json always wins; a source that is not in the table gets True with a warning.
PRESET_DO_NORMALIZE = {
    "wav2vec2-base": True, "wav2vec2-base-960h": True, "wav2vec2-large-lv60": True, "wav2vec2-large-960h-lv60": True,
    "hubert-base-ls960": False, "hubert-large-ll60k": True, "hubert-large-ls960-ft": True, "hubert-xlarge-ll60k": True,
    "wavlm-base": False, "wavlm-base-plus": False, "wavlm-large": True,
    "data2vec-audio-base": True, "data2vec-audio-large": True,
}


def preset_do_normalize(source: str):
    """``do_normalize`` for a hub id / preset name, or None when it is not known offline."""
    key = str(source).rstrip("/").split("/")[-1]
    if key.startswith("tiny-") or key.startswith("avhubert"):
        return True  # this package's own test geometries
    return PRESET_DO_NORMALIZE.get(key)


def config_from_source(source: str) -> EncoderConfig:
    """Pick a preset from a HF hub id / local name the way the reference picks the model class:
    by substring (``MIR_ST500/huggingface_interface.py:107-119``)."""
    key = source.rstrip("/").split("/")[-1]
    if key in PRESETS:
        return PRESETS[key]
    low = source.lower()
    if "avhubert" in low or "av_hubert" in low:
        return PRESETS["avhubert-base-video" if "base" in low else "avhubert-large-video"]
    if "hubert" in low:
        if "xlarge" in low:
            return PRESETS["hubert-xlarge-ll60k"]
        return PRESETS["hubert-base-ls960"] if "base" in low else PRESETS["hubert-large-ll60k"]
    if "data2vec" in low:
        return PRESETS["data2vec-audio-large" if "large" in low else "data2vec-audio-base"]
    if "wavlm" in low:
        return PRESETS["wavlm-large" if "large" in low else "wavlm-base"]
    if "wav2vec2" in low:
        if "large" not in low and "xls" not in low:
            return PRESETS["wav2vec2-base"]
        # the LibriLight / multilingual large models (lv60, xlsr, xls-r, robust) use the layer-norm conv stack and the pre-LN
        # encoder; the LibriSpeech-only "wav2vec2-large" / "-large-960h" keep the base layout at large width
        if any(t in low for t in ("lv60", "xls", "robust", "voxpopuli")):
            return PRESETS["wav2vec2-large-lv60"]
        return PRESETS["wav2vec2-large"]
    raise ValueError(f"cannot infer an encoder family from source={source!r}")
