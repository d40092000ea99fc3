"""Drop-in for ``MIR_ST500/huggingface_interface.py`` (``HuggingFaceWav2Vec2``, reference :47-297).

Same constructor keywords, same attributes (``.model``, ``.normalize_wav``, ``.output_norm``,
``.freeze``), same ``forward(wav: f32[B, L]) -> f32[B, T, D]`` and the same ``state_dict`` key layout
(``model.<HF key>``), so a recipe yaml selects it by replacing
``!new:huggingface_interface.HuggingFaceWav2Vec2`` with
``!new:svt_speechbrain_amd.huggingface_interface.HuggingFaceWav2Vec2``.

The arithmetic (HF ``Wav2Vec2Model`` / ``HubertModel`` forward + the wrapper's two whole-batch layer
norms) runs in hand-written gfx950 kernels behind the C-ABI; torch is used only as tensor storage.
The path is inference-only (forward target); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import logging
import os
import warnings
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from ._device import DeviceObjects, ReplicaAware
from .config import EncoderConfig, PRESETS, config_from_source
from .weights import encoder_param_shapes, seeded_encoder_state_dict

logger = logging.getLogger(__name__)

PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp16x3": 3, "fp16": 1}
# "fp16": the 16-bit throughput mode with IEEE-half operands instead of bf16 -- precision code 1 of the OTHER build of the library
# (libsvt_mi355_f16.so, the same sources compiled with -DSVT_OPERAND_F16): same MFMA rate, three more mantissa bits
LIB_VARIANT = {"fp16": "f16"}


class ParamTree(nn.Module):
    """A bare module hierarchy that reproduces a dotted state-dict key layout (no forward of its own)."""

    def add(self, dotted: str, tensor: torch.Tensor) -> None:
        head, _, rest = dotted.partition(".")
        if not rest:
            if head in ("running_mean", "running_var", "num_batches_tracked"):  # BatchNorm statistics are buffers in HF too
                self.register_buffer(head, tensor)
            else:
                self.register_parameter(head, nn.Parameter(tensor, requires_grad=True))
            return
        if head not in self._modules:
            self.add_module(head, ParamTree())
        self._modules[head].add(rest, tensor)


def _config_to_c(cfg: EncoderConfig, normalize_wav: bool, output_norm: bool, precision: str) -> _lib.EncoderConfigC:
    c = _lib.EncoderConfigC()
    c.struct_size = C.sizeof(_lib.EncoderConfigC)
    c.hidden_size = cfg.hidden_size
    c.num_layers = cfg.num_hidden_layers
    c.num_heads = cfg.num_attention_heads
    c.intermediate_size = cfg.intermediate_size
    n = len(cfg.conv_kernel)
    if n > _lib.MAX_CONV:
        raise ValueError("too many conv layers")
    c.num_conv_layers = n  # 0 = features-in mode (AV-HuBERT video branch): conv_dim[0] is the input feature width
    c.conv_dim[0] = cfg.conv_dim[0]
    for i in range(n):
        c.conv_dim[i] = cfg.conv_dim[i]
        c.conv_kernel[i] = cfg.conv_kernel[i]
        c.conv_stride[i] = cfg.conv_stride[i]
    c.feat_extract_norm = 0 if cfg.feat_extract_norm == "group" else 1
    c.conv_bias = int(cfg.conv_bias)
    c.stable_layer_norm = int(cfg.do_stable_layer_norm)
    c.feat_proj_layer_norm = int(cfg.feat_proj_layer_norm)
    c.pos_conv_kernel = cfg.num_conv_pos_embeddings
    c.pos_conv_groups = cfg.num_conv_pos_embedding_groups
    c.layer_norm_eps = cfg.layer_norm_eps
    c.normalize_wav = int(normalize_wav)
    c.output_norm = int(output_norm)
    c.precision = PRECISIONS[precision]
    c.pos_conv_depth = cfg.pos_conv_depth
    c.rel_pos_buckets = cfg.rel_pos_buckets
    c.rel_pos_max_distance = cfg.rel_pos_max_distance
    c.pos_conv_batch_norm = int(cfg.conv_pos_batch_norm)
    return c


def _config_from_dir(path: str) -> Optional[EncoderConfig]:
    f = os.path.join(path, "config.json")
    if not os.path.isfile(f):
        return None
    with open(f) as fh:
        j = json.load(fh)
    mt = j.get("model_type", "wav2vec2")
    fam = {"wav2vec2": "wav2vec2", "hubert": "hubert", "data2vec-audio": "data2vec", "wavlm": "wavlm"}.get(mt)
    if fam is None:
        raise NotImplementedError(f"model_type {mt!r}: the reference wrapper only builds wav2vec2 / hubert / data2vec-audio / wavlm "
                                  f"models (MIR_ST500/huggingface_interface.py:107-119)")
    kw = {}
    if fam == "data2vec":  # Data2VecAudioConfig: num_conv_pos_embeddings = number of stacked layers, conv_pos_kernel_size = taps
        kw = dict(num_conv_pos_embeddings=j.get("conv_pos_kernel_size", 19), pos_conv_depth=j.get("num_conv_pos_embeddings", 5))
    else:
        kw = dict(num_conv_pos_embeddings=j.get("num_conv_pos_embeddings", 128))
    if fam == "wavlm":
        kw.update(rel_pos_buckets=j.get("num_buckets", 320), rel_pos_max_distance=j.get("max_bucket_distance", 800))
    # a config.json written with use_diff=True omits the fields that equal the config CLASS defaults: fall back to those
    # (transformers Wav2Vec2Config / HubertConfig / WavLMConfig / Data2VecAudioConfig)
    return EncoderConfig(
        name=os.path.basename(path.rstrip("/")), family=fam, hidden_size=j.get("hidden_size", 768),
        num_hidden_layers=j.get("num_hidden_layers", 12), num_attention_heads=j.get("num_attention_heads", 12),
        intermediate_size=j.get("intermediate_size", 3072), conv_dim=tuple(j.get("conv_dim", (512,) * 7)),
        conv_kernel=tuple(j.get("conv_kernel", (10, 3, 3, 3, 3, 2, 2))), conv_stride=tuple(j.get("conv_stride", (5, 2, 2, 2, 2, 2, 2))),
        feat_extract_norm=j.get("feat_extract_norm", "layer" if fam == "data2vec" else "group"),
        conv_bias=bool(j.get("conv_bias", False)),
        do_stable_layer_norm=bool(j.get("do_stable_layer_norm", False)) and fam != "data2vec",
        feat_proj_layer_norm=bool(j.get("feat_proj_layer_norm", True)) or fam != "hubert",
        num_conv_pos_embedding_groups=j.get("num_conv_pos_embedding_groups", 16),
        layer_norm_eps=j.get("layer_norm_eps", 1e-5),
        conv_pos_batch_norm=bool(j.get("conv_pos_batch_norm", False)) and fam == "hubert", **kw)


class _DevF64:
    """A device pointer to ``n`` doubles as a ``__cuda_array_interface__`` object (``torch.as_tensor`` wraps it without a copy)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class HuggingFaceWav2Vec2(ReplicaAware, nn.Module):
    """wav2vec 2.0 / HuBERT encoder on MI355X with the reference wrapper's surface.

    Arguments (reference ``huggingface_interface.py:89-98``)
    ---------
    source : str
        HF hub id (``facebook/wav2vec2-base``, ``facebook/hubert-large-ll60k`` ...), one of
        ``svt_speechbrain_amd.config.PRESETS`` or a local directory holding ``config.json``
        (+ optionally ``pytorch_model.bin`` / a SpeechBrain ``*.ckpt``).  The model family is picked by
        substring exactly like the reference (:107-119).
    save_path : str
        kept for signature compatibility (the reference caches downloads there); unused offline.
    pretrain, output_norm, freeze, freeze_feature_extractor, apply_spec_augment : as the reference.
        There is no network in this build, so ``pretrain=True`` loads a local checkpoint when ``source``
        is a directory that has one.  With a hub id / preset name and ``pretrain=True`` there is nothing to
        download from: the weights stay seeded-random until ``load_state_dict`` / a ``Checkpointer`` recovers
        the fine-tuned checkpoint (the recipes' evaluation path) -- a warning says so; pass
        ``allow_random_init=False`` to make that case an error instead.

    Extra keyword-only arguments: ``config`` (an ``EncoderConfig``), ``precision`` ("bf16" throughput mode, "fp32"
    exact-fp32 parity mode, "fp16x3" / "bf16x3" split-operand parity modes on the 16-bit matrix pipe; env
    ``SVT_PRECISION`` sets the default), ``normalize_wav`` (the reference reads ``feature_extractor.do_normalize`` of the
    checkpoint's ``preprocessor_config.json``: a local directory is read the same way, presets carry the published value
    of their hub model -- ``config.PRESET_DO_NORMALIZE`` -- and an unknown hub id falls back to True with a warning),
    ``seed``.
    """

    def __init__(self, source, save_path=None, pretrain=True, output_norm=True, freeze=True,
                 freeze_feature_extractor=False, apply_spec_augment=False, *, config: Optional[EncoderConfig] = None,
                 precision: Optional[str] = None, normalize_wav: Optional[bool] = None, seed: int = 1986,
                 allow_random_init: bool = True):
        super().__init__()
        if apply_spec_augment:
            raise NotImplementedError("apply_spec_augment is a training-time mask; the MI355X path is forward-only")
        cfg = config
        local_ckpt = None
        if cfg is None and isinstance(source, str) and os.path.isdir(source):
            cfg = _config_from_dir(source)
            # reference _check_model_source (:215-262): a directory with a *.bin is a HuggingFace model, otherwise the first
            # *.ckpt is a SpeechBrain-pretrained one, otherwise FileNotFoundError.  (*.safetensors -- what current transformers
            # writes -- is accepted as a HuggingFace model too.)
            files = sorted(os.listdir(source))
            # the weights file among the *.bin (optimizer.bin / training_args.bin / scheduler.bin sit beside it in a
            # Trainer output directory): pytorch_model.bin / model.safetensors first, then their shards via the
            # *.index.json, then any other *.bin / *.safetensors that is not a known non-weights file
            not_weights = ("optimizer.bin", "training_args.bin", "scheduler.bin", "rng_state.pth", "scaler.pt")
            hf = [fn for fn in ("pytorch_model.bin", "model.safetensors") if fn in files]
            if not hf:
                hf = [fn for fn in ("pytorch_model.bin.index.json", "model.safetensors.index.json") if fn in files]
            if not hf:
                hf = ([fn for fn in files if fn.endswith(".bin") and fn not in not_weights]
                      or [fn for fn in files if fn.endswith(".safetensors")])
            sb = [fn for fn in files if fn.endswith(".ckpt")]
            if hf:
                local_ckpt = os.path.join(source, hf[0])
            elif sb:
                local_ckpt = os.path.join(source, sb[0])
            else:
                raise FileNotFoundError(f"{source} does not contain a .bin or .ckpt checkpoint !")
            pp = os.path.join(source, "preprocessor_config.json")
            if normalize_wav is None and os.path.isfile(pp):
                with open(pp) as fh:
                    normalize_wav = bool(json.load(fh).get("do_normalize", True))
        if cfg is None:
            cfg = config_from_source(source)
        self.config = cfg
        self.source = source
        precision = precision or os.environ.get("SVT_PRECISION", "bf16")
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}")
        self.precision = precision
        if normalize_wav is None:
            from .config import preset_do_normalize
            normalize_wav = preset_do_normalize(source if isinstance(source, str) else cfg.name)
            if normalize_wav is None:
                logger.warning("HuggingFaceWav2Vec2(%s): do_normalize of this source is not known offline (the reference reads "
                               "it from the hub's preprocessor_config.json); assuming True -- pass normalize_wav= to set it", source)
                normalize_wav = True
        self.normalize_wav = bool(normalize_wav)
        self.output_norm = output_norm
        self.freeze = freeze
        self.freeze_feature_extractor = freeze_feature_extractor
        self.output_size = cfg.hidden_size

        self.model = ParamTree()
        for k, v in seeded_encoder_state_dict(cfg, seed=seed).items():
            self.model.add(k, v)
        # device side: one C object per device (shared with nn.DataParallel replicas, see _device.py) and ONE generation
        # counter of the parameter values, shared BY REFERENCE with every replica(): whoever changes the weights bumps it,
        # every device object compares it before a forward
        self._dev = DeviceObjects("svt_encoder_destroy", LIB_VARIANT.get(self.precision))
        self._gen = [0]
        self._dp_replica = False
        self._warned_grad = False
        # state loaded through a PARENT module (nn.ModuleDict.load_state_dict, Brain.modules, Checkpointer on the whole
        # model) never calls this object's load_state_dict: nn.Module recursion goes through _load_from_state_dict of the
        # children.  The hooks do the key normalisation and the invalidation on that path too.
        self.model._register_load_state_dict_pre_hook(self._pre_load_hook)
        self.model.register_load_state_dict_post_hook(self._post_load_hook)
        if pretrain:
            if local_ckpt is not None:
                self._load_local(local_ckpt)
            elif not allow_random_init:
                raise FileNotFoundError(f"HuggingFaceWav2Vec2({source!r}, pretrain=True): no local checkpoint and no network "
                                        "in this build; point `source` at a model directory or pass allow_random_init=True")
            else:
                logger.warning("HuggingFaceWav2Vec2(%s): pretrain=True but there is no network / local checkpoint in this build; "
                               "the weights are SEEDED RANDOM until load_state_dict() / a Checkpointer loads real ones", source)
        if self.freeze:
            self.model.eval()
            for p in self.model.parameters():
                p.requires_grad = False
        else:
            self.model.train()
            if self.freeze_feature_extractor:
                for n, p in self.model.named_parameters():
                    if n.startswith("feature_extractor."):
                        p.requires_grad = False

    # ------------------------------------------------------------------ checkpoint intake
    @staticmethod
    def _read_checkpoint_file(path: str) -> Dict[str, torch.Tensor]:
        if path.endswith(".index.json"):  # sharded HuggingFace checkpoint: weight_map = {key: shard file}
            with open(path) as fh:
                shards = sorted(set(json.load(fh)["weight_map"].values()))
            sd = {}
            for fn in shards:
                sd.update(HuggingFaceWav2Vec2._read_checkpoint_file(os.path.join(os.path.dirname(path), fn)))
            return sd
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(path)
        return torch.load(path, map_location="cpu")

    def _pre_load_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Runs at the top of ``self.model``'s ``_load_from_state_dict`` (whatever module ``load_state_dict`` was called
        on): both weight-norm spellings of the positional conv are accepted, HF's ``masked_spec_embed`` is dropped."""
        pc = prefix + "encoder.pos_conv_embed.conv."
        for old, new in ((pc + "weight_g", pc + "parametrizations.weight.original0"),
                         (pc + "weight_v", pc + "parametrizations.weight.original1")):
            if old in state_dict:
                state_dict[new] = state_dict.pop(old)
        state_dict.pop(prefix + "masked_spec_embed", None)

    def _post_load_hook(self, module, incompatible_keys):
        self._invalidate()

    def _invalidate(self) -> None:
        self._gen[0] += 1

    def _load_local(self, path: str) -> None:
        sd = self._read_checkpoint_file(path)
        own = set(self.model.state_dict().keys())
        out = {}
        if path.endswith(".ckpt"):
            # SpeechBrain-pretrained (HuggingFaceWav2Vec2Pretrain) checkpoint: the reference keeps the keys that contain
            # "wav2vec2." and strips "model.wav2vec2." (:181-191); everything else is discarded with a warning
            for k, v in sd.items():
                if "wav2vec2." in k:
                    out[k.replace("model.wav2vec2.", "")] = v
                else:
                    logger.warning("The param with the key: %s is discarded as it is useless for wav2vec 2.0 finetuning.", k)
        else:
            # HuggingFace checkpoint: base-model keys, possibly under the task model's prefix (from_pretrained strips it)
            for k, v in sd.items():
                for pre in ("", "wav2vec2.", "hubert.", "data2vec_audio.", "wavlm.", "model."):
                    if k.startswith(pre) and self._normalise_key(k[len(pre):]) in own:
                        out[k[len(pre):]] = v
                        break
        missing = own - set(self._normalise_keys(out).keys())
        for k in sorted(missing):
            logger.warning("During parameter transfer loading from %s, the transferred parameters did not have parameters for "
                           "the key: %s", path, k)
        self._load_model_state(out, strict=False)

    def _load_model_state(self, sd: Dict[str, torch.Tensor], strict: bool) -> None:
        sd = self._normalise_keys(sd)
        self.model.load_state_dict(sd, strict=strict)
        self._invalidate()

    @staticmethod
    def _normalise_key(k: str) -> str:
        pc = "encoder.pos_conv_embed.conv."
        return {pc + "weight_g": pc + "parametrizations.weight.original0",
                pc + "weight_v": pc + "parametrizations.weight.original1"}.get(k, k)

    def _normalise_keys(self, sd):
        """Accept both weight-norm spellings of the positional conv and drop the HF-only
        ``masked_spec_embed`` (SpecAugment embedding, unused: apply_spec_augment=False, reference :127)."""
        pc = "encoder.pos_conv_embed.conv."
        ren = {pc + "weight_g": pc + "parametrizations.weight.original0",
               pc + "weight_v": pc + "parametrizations.weight.original1"}
        out = {}
        for k, v in sd.items():
            for pre in ("model.", ""):
                if k.startswith(pre) and k[len(pre):] in ren:
                    k = pre + ren[k[len(pre):]]
                    break
            if k.endswith("masked_spec_embed"):
                continue
            out[k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        res = super().load_state_dict(self._normalise_keys(dict(state_dict)), strict=strict, **kw)
        self._invalidate()
        return res

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def _replicate_for_data_parallel(self):
        # nn.DataParallel (the reference's multi-GPU mode): the replica shares the registry of device objects and the
        # generation counter by reference; its parameters are per-forward broadcast copies, so it never compares tensor
        # identities -- the slot of its device is (re)filled only when the generation moved
        r = super()._replicate_for_data_parallel()
        r._dp_replica = True
        return r

    # ------------------------------------------------------------------ device-side object
    def _lib(self):
        """The build of the library this object's precision lives in (``libsvt_mi355_f16.so`` for "fp16")."""
        return _lib.load(LIB_VARIANT.get(self.precision))

    def _params_signature(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.model.parameters()) + list(self.model.buffers()))

    def _sync_device(self, device: torch.device):
        """Make the C object of ``device`` current with the parameters; returns its slot (handle + workspace)."""
        lib = self._lib()
        _lib.require_gpu()
        idx = _lib.dev_index(device)
        slot = self._dev.slot(idx, (self.normalize_wav, self.output_norm, self.precision))
        gen = self._gen[0]
        if slot.handle is not None and slot.gen == gen:
            # frozen (the reference default) or a DataParallel replica: the values only change through load_state_dict (on
            # this module, a parent or a replica's original) / .to() / refresh(), all of which move the generation.  A
            # two-tensor sentinel still catches an in-place edit of the weights; walking all ~200 tensors on every
            # forward costs ~0.4 ms of host time, half of a one-clip forward.
            if self._dp_replica:
                return slot
            if self.freeze:
                if slot.sig is not None and self._sentinel() == slot.sig[1]:
                    return slot
            elif slot.sig is not None and self._params_signature() == slot.sig[0]:
                return slot
        if slot.handle is None:
            h = C.c_void_p()
            cc = _config_to_c(self.config, self.normalize_wav, self.output_norm, self.precision)
            _lib.check(lib.svt_encoder_create(C.byref(cc), idx, C.byref(h)), "svt_encoder_create", lib)
            slot.handle = h
        src = self._param_owner()   # a DataParallel replica uploads the ORIGINAL's parameters (its own tree has none)
        for name, p in src._upload_items():
            t = p.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(lib.svt_encoder_load_param(slot.handle, name.encode(), C.c_void_p(t.data_ptr()), 0, shape,
                                                  t.dim()), f"svt_encoder_load_param({name})", lib)
        _lib.check(lib.svt_encoder_finalize(slot.handle), "svt_encoder_finalize", lib)
        slot.sig = (None if self.freeze else src._params_signature(), src._sentinel())
        slot.gen = gen
        return slot

    def _upload_items(self):
        """(name, tensor) of everything svt_encoder_load_param receives, read from THIS module's tree."""
        for name, p in self.model.state_dict().items():
            if p.is_floating_point():   # not BatchNorm's num_batches_tracked
                yield name, p

    def _sentinel(self):
        ts = list(self.model.parameters())
        return tuple((t.data_ptr(), t._version) for t in (ts[0], ts[len(ts) // 2], ts[-1]))

    def refresh(self) -> None:
        """Re-upload the parameters on the next forward (after editing them in place while ``freeze=True``)."""
        self._invalidate()

    def replica(self):
        """A second encoder object over the SAME parameter tensors with its own device handle and workspace: what a caller
        needs to keep two forwards in flight on two HIP streams (bench.py, SongTranscriber).  It shares the generation
        counter with this object: ``load_state_dict`` / ``refresh`` / ``.to()`` on either one re-uploads both."""
        import copy
        c = copy.copy(self)
        c._dev = DeviceObjects("svt_encoder_destroy", LIB_VARIANT.get(self.precision))
        # copy.copy shares _parameters / _modules dicts and the _gen list; the load hooks of self.model fire for both
        return c

    def num_frames(self, n_samples: int) -> int:
        return self.config.frames(n_samples)

    # ------------------------------------------------------------------ "global-batch-equivalent" norms
    def set_norm_reduce(self, fn, global_clips: int = 0) -> None:
        """Make the wrapper's two whole-batch layer norms (reference ``huggingface_interface.py:289-295``) those of a LARGER batch
        this object only sees a shard of.  ``fn(sums)`` receives a 2-element float64 CUDA tensor -- (sum, sum of squares) of the
        local shard, first of the waveform, then of the encoder output -- and must add the other shards' pairs in place
        (``torch.distributed.all_reduce``); it runs inside the forward, in stream order on the current stream.  ``global_clips`` is
        the clip count of the whole batch (equal lengths).  ``fn=None`` restores per-shard norms, which is what the reference's own
        multi-GPU modes compute.  ``set_global_batch_norm`` is the ``torch.distributed`` form."""
        if fn is None:
            self._norm_reduce = None
            return
        if int(global_clips) < 1:
            raise ValueError("set_norm_reduce: global_clips = the number of clips of the whole global batch")

        def cb(ptr, n, stream, user):
            try:
                t = torch.as_tensor(_DevF64(ptr, n), device=torch.device("cuda", torch.cuda.current_device()))
                fn(t)
                return 0
            except Exception:   # the C side turns a non-zero return into an error of the forward call
                import traceback
                traceback.print_exc()
                return 1

        self._norm_reduce = (_lib.NORM_REDUCE_FN(cb), int(global_clips))

    def set_global_batch_norm(self, global_clips: int, group=None) -> None:
        """``set_norm_reduce`` with ``torch.distributed.all_reduce`` (RCCL: two 16-byte all-reduces per forward): N ranks holding
        contiguous shards of one global batch return what a single device returns for the whole batch."""
        import torch.distributed as dist

        def reduce(t):
            if dist.get_backend(group) == "gloo":   # test mode (ranks sharing one GPU): staged through the host
                h = t.cpu()
                dist.all_reduce(h, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=group)
        self.set_norm_reduce(reduce, global_clips)

    def _apply_norm_reduce(self, lib, handle) -> None:
        nr = getattr(self, "_norm_reduce", None)
        if nr is None:
            _lib.check(lib.svt_encoder_set_norm_reduce(handle, None, None, 0), "svt_encoder_set_norm_reduce", lib)
        else:
            _lib.check(lib.svt_encoder_set_norm_reduce(handle, C.cast(nr[0], C.c_void_p), None, nr[1]), "svt_encoder_set_norm_reduce", lib)

    # ------------------------------------------------------------------ forward (reference :263-297)
    def forward(self, wav: torch.Tensor, clips_per_norm_group: int = 0) -> torch.Tensor:
        """``clips_per_norm_group`` (extension; 0 = the reference: both whole-tensor layer norms over the entire batch):
        with 1, a batch of B equal-length utterances returns what B batch-1 forwards return, which is how the reference
        evaluates (``train_audio_ssl.py:90`` asserts batch 1) -- one batched launch sequence instead of B."""
        if not self.freeze and torch.is_grad_enabled() and not self._warned_grad:
            warnings.warn("svt_speechbrain_amd.HuggingFaceWav2Vec2 is forward-only: the output is detached "
                          "(fine-tuning the encoder is out of scope of the MI355X path)")
            self._warned_grad = True
        with torch.no_grad():
            return self.extract_features(wav, clips_per_norm_group).detach()

    def forward_head(self, wav: torch.Tensor, head, clips_per_norm_group: int = 0, frames: Optional[torch.Tensor] = None,
                     pitch_octave_num: int = 4, pitch_class_num: int = 12) -> torch.Tensor:
        """``head(self(wav))`` -- the two module calls of ``AMT.compute_forward`` (train_audio_ssl.py:36-39) -- as ONE C-ABI call
        that never writes the (B, T, D) features: the whole-batch output norm, the frame head (``svt_speechbrain_amd.Linear``
        with <= 32 outputs) and, when ``frames`` (a ``(B*T, 4)`` int32 device tensor) is given, the per-frame sigmoid / argmax
        are fused behind the encoder (``svt_encoder_forward_head``).  Returns the ``(B, T, n_out)`` logits."""
        if wav.dim() != 2:
            raise ValueError(f"expected a (batch, samples) waveform, got shape {tuple(wav.shape)}")
        if not wav.is_cuda:
            raise _lib.SvtError("the MI355X encoder needs its input on the GPU ('cuda:N'); there is no CPU fallback")
        lib = self._lib()
        slot = self._sync_device(wav.device)
        hslot = head._sync(wav.device, LIB_VARIANT.get(self.precision))   # the head's C object lives in the encoder's library build
        x = wav.detach().to(torch.float32).contiguous()
        B, L = x.shape
        T = self.config.frames(L)
        if B < 1 or T < 1:
            raise ValueError(f"waveform of {L} samples is shorter than the encoder's receptive field")
        need = lib.svt_encoder_workspace_bytes(slot.handle, B, L)
        if need < 0:
            raise _lib.SvtError(_lib.last_error(lib))
        ws = slot.workspace(need, x.device)
        n_out = head.w.out_features
        logits = torch.empty((B, T, n_out), dtype=torch.float32, device=x.device)
        if frames is not None and (frames.dtype != torch.int32 or frames.numel() != B * T * 4 or not frames.is_contiguous()):
            raise ValueError("frames must be a contiguous int32 tensor of B*T*4 elements (16 bytes per frame)")
        self._apply_norm_reduce(lib, slot.handle)
        with torch.no_grad():
            _lib.check(lib.svt_encoder_forward_head(slot.handle, hslot.handle, _lib.ptr(x), B, L, _lib.ptr(logits),
                                                    _lib.ptr(frames) if frames is not None else None, int(pitch_octave_num),
                                                    int(pitch_class_num), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device),
                                                    int(clips_per_norm_group)), "svt_encoder_forward_head", lib)
        return logits

    @staticmethod
    def can_fuse_head(head) -> bool:
        """True when ``head`` is this package's ``Linear`` with a geometry the fused tail serves."""
        from .linear import Linear
        return (isinstance(head, Linear) and not head.combine_dims and head.w.in_features in (512, 768, 1024)
                and 1 <= head.w.out_features <= 32)

    def extract_features(self, wav: torch.Tensor, clips_per_norm_group: int = 0) -> torch.Tensor:
        if wav.dim() != 2:
            raise ValueError(f"expected a (batch, samples) waveform, got shape {tuple(wav.shape)}")
        if not wav.is_cuda:
            raise _lib.SvtError("the MI355X encoder needs its input on the GPU ('cuda:N'); there is no CPU fallback")
        lib = self._lib()
        slot = self._sync_device(wav.device)
        x = wav.detach().to(torch.float32).contiguous()
        B, L = x.shape
        T = self.config.frames(L)
        if B < 1 or T < 1:
            raise ValueError(f"waveform of {L} samples is shorter than the encoder's receptive field")
        need = lib.svt_encoder_workspace_bytes(slot.handle, B, L)
        if need < 0:
            raise _lib.SvtError(_lib.last_error(lib))
        ws = slot.workspace(need, x.device)
        out = torch.empty((B, T, self.config.hidden_size), dtype=torch.float32, device=x.device)
        self._apply_norm_reduce(lib, slot.handle)
        _lib.check(lib.svt_encoder_forward_ex(slot.handle, _lib.ptr(x), B, L, _lib.ptr(out), _lib.ptr(ws),
                                              ws.numel(), _lib.stream_ptr(x.device), int(clips_per_norm_group)),
                   "svt_encoder_forward", lib)
        return out
