"""Drop-in for the lip front-end of the AV-HuBERT video branch — ``SubModel`` / ``ResEncoder`` of
``N20EMv2/video_only/resnet.py`` (:134-187; the model's ``feature_extractor_video``, ``hubert.py:344-346``): 3-D stem
(Conv3d 1->64, 5x7x7, stride 1x2x2) + BatchNorm3d + PReLU + 3x3/2 max-pool, ResNet-18 trunk of PReLU BasicBlocks, global
average pool, ``Linear(512, embed_dim)``; eval mode (running statistics), as the frozen recipes run it.

``forward(x)`` takes the reference's ``(B, 1, T, H, W)`` lip ROI tensor and returns ``(B, embed_dim, T)`` like
``SubModel.forward``.  The state-dict keys are ``SubModel.state_dict()``'s (``resnet.frontend3D.0.weight``,
``resnet.trunk.layer1.0.bn1.running_mean``, ..., ``proj.weight``), so the ``feature_extractor_video.*`` slice of an AV-HuBERT
checkpoint loads unchanged.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
from torch import nn

from . import _lib
from ._device import DeviceObjects, ReplicaAware
from .huggingface_interface import ParamTree, PRECISIONS, LIB_VARIANT
from .weights import seeded_video_frontend_state_dict


class EvalTransform:
    """``transform_eval`` of the video recipes (``N20EMv2/video_only/train_video_ssl.py:445-457``): ``Normalize(0.0, 255.0)`` ->
    ``CenterCrop((88, 88))`` -> ``Normalize(0.421, 0.165)`` on the ``np.load``-ed uint8 frames, then ``.astype(np.float32)`` (:530-533).
    Passed to the lip front-end together with the RAW uint8 ROI, it runs inside the padding kernel (``svt_video_forward_u8``): one byte
    per pixel crosses HBM instead of four, and no host pass touches the frames.  ``__call__`` is the host restatement (numpy, float64
    like the reference) for callers that still want the float tensor."""

    def __init__(self, crop=(88, 88), image_mean: float = 0.421, image_std: float = 0.165, scale_sub: float = 0.0, scale_div: float = 255.0):
        self.crop_h, self.crop_w = int(crop[0]), int(crop[1])
        self.sub0, self.div0, self.mean, self.std = float(scale_sub), float(scale_div), float(image_mean), float(image_std)

    def offsets(self, h: int, w: int):
        if h < self.crop_h or w < self.crop_w:
            raise ValueError(f"CenterCrop({self.crop_h}, {self.crop_w}) of a {h} x {w} frame")
        return int(round(h - self.crop_h) / 2.0), int(round(w - self.crop_w) / 2.0)     # utils.py:79-83

    def __call__(self, frames):
        import numpy as np
        f = np.asarray(frames)
        dy, dx = self.offsets(f.shape[-2], f.shape[-1])
        f = (f - self.sub0) / self.div0
        f = f[..., dy:dy + self.crop_h, dx:dx + self.crop_w]
        return ((f - self.mean) / self.std).astype(np.float32)


class SubModel(ReplicaAware, nn.Module):
    def __init__(self, input_dim=512, embed_dim=1024, relu_type="prelu", weights=None, *, precision=None, seed=4986):
        super().__init__()
        if input_dim != 512:
            raise ValueError("the ResNet-18 back end emits 512 features (resnet.py:137)")
        if relu_type != "prelu":
            raise NotImplementedError("only relu_type='prelu' (the AV-HuBERT configuration) is provided")
        self.embed_dim = int(embed_dim)
        self.precision = precision or os.environ.get("SVT_PRECISION", "bf16")
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}")
        tree = ParamTree()
        for k, v in seeded_video_frontend_state_dict(self.embed_dim, seed=seed).items():
            if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
                mod = tree
                parts = k.split(".")
                for part in parts[:-1]:
                    mod = mod._modules.setdefault(part, ParamTree())
                mod.register_buffer(parts[-1], v)
            else:
                tree.add(k, v)
        self.resnet = tree._modules["resnet"]
        self.proj = tree._modules["proj"]
        if weights is not None:  # resnet.py:146-157: a lip-reading checkpoint with 'model_state_dict'
            std = torch.load(weights, map_location="cpu")["model_state_dict"]
            own = self.state_dict()
            for key, val in std.items():
                new_key = "resnet." + ".".join(key.split(".")[1:])
                if ("frontend3D" in key or "trunk" in key) and new_key in own:
                    own[new_key] = val
            self.load_state_dict(own)
        self._dev = DeviceObjects("svt_video_destroy", LIB_VARIANT.get(self.precision))  # one C object per device, shared with DataParallel replicas

    def _tensors(self):
        for n, p in self.named_parameters():
            yield n, p
        for n, b in self.named_buffers():
            yield n, b

    def _sync(self, device):
        lib = _lib.load(LIB_VARIANT.get(self.precision))
        _lib.require_gpu()
        idx = _lib.dev_index(device)
        slot = self._dev.slot(idx, (self.precision,))
        src = self._param_owner()   # a DataParallel replica reads the ORIGINAL's tensors (its own tree holds no parameters)
        sig = tuple((t.data_ptr(), t._version) for _, t in src._tensors())
        if slot.handle is not None and sig == slot.sig:
            return slot
        if slot.handle is None:
            h = C.c_void_p()
            _lib.check(lib.svt_video_create(self.embed_dim, PRECISIONS[self.precision], idx, C.byref(h)), "svt_video_create", lib)
            slot.handle = h
            # the slot's workspace tensor belongs to this handle alone (DeviceSlot.workspace): the stage buffers' zero halos survive
            # from call to call and are written once per geometry instead of on every forward
            _lib.check(lib.svt_video_keep_workspace(slot.handle, 1), "svt_video_keep_workspace", lib)
        for name, t in src._tensors():
            if name.endswith("num_batches_tracked"):
                continue
            c = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * c.dim())(*c.shape)
            _lib.check(lib.svt_video_load_param(slot.handle, name.encode(), C.c_void_p(c.data_ptr()), 0, shape, c.dim()),
                       f"svt_video_load_param({name})", lib)
        _lib.check(lib.svt_video_finalize(slot.handle), "svt_video_finalize", lib)
        slot.sig = sig
        return slot

    def forward(self, x: torch.Tensor, transform: "Optional[EvalTransform]" = None) -> torch.Tensor:
        return self.forward_into(x, None, transform).transpose(1, 2)  # (B, embed_dim, T), the reference's layout

    def forward_into(self, x: torch.Tensor, fused: Optional[torch.Tensor] = None, transform: "Optional[EvalTransform]" = None) -> torch.Tensor:
        """The front-end on ``x`` -- a ``(B, 1, T, H, W)`` float tensor already normalised (what the reference's ``SubModel`` takes), or the
        raw ``(B, T, H, W)`` / ``(B, 1, T, H, W)`` / ``(B, T, H, W, 1)`` **uint8** lip ROI, in which case ``transform`` (default: the recipes'
        ``transform_eval``) is applied inside the padding kernel.  Returns ``(B, T, embed_dim)``; with ``fused`` -- an uninitialised
        ``(B, T, 2 * embed_dim)`` fp32 buffer -- the result is written into its right half and the left half is zeroed by the same C-ABI
        call (AV-HuBERT's ``cat([zeros, video], -1)``, ``hubert.py:700-712``), and ``fused`` is returned."""
        if not x.is_cuda:
            raise _lib.SvtError("the lip front-end needs its input on the GPU; there is no CPU fallback")
        u8 = x.dtype == torch.uint8
        if u8 and x.dim() == 5 and x.shape[-1] == 1 and x.shape[1] != 1:
            x = x[..., 0]                         # (B, T, H, W, 1): the recipe's batch.sig before its permute
        if u8 and x.dim() == 5 and x.shape[1] == 1:
            x = x[:, 0]
        if (u8 and x.dim() != 4) or (not u8 and (x.dim() != 5 or x.shape[1] != 1)):
            raise ValueError(f"expected a (B, 1, T, H, W) float lip ROI or a (B, T, H, W) uint8 one, got {tuple(x.shape)} {x.dtype}")
        if transform is not None and not u8:
            raise ValueError("transform applies to the raw uint8 ROI; a float tensor is taken as already normalised")
        B, T = (x.shape[0], x.shape[1]) if u8 else (x.shape[0], x.shape[2])
        Hin, Win = x.shape[-2], x.shape[-1]
        tf = (transform or EvalTransform()) if u8 else None
        H, W = (tf.crop_h, tf.crop_w) if u8 else (Hin, Win)
        lib = _lib.load(LIB_VARIANT.get(self.precision))
        slot = self._sync(x.device)
        v = x.detach().contiguous() if u8 else x.detach().to(torch.float32).contiguous()
        need = lib.svt_video_workspace_bytes(slot.handle, B, T, H, W)
        if need < 0:
            raise ValueError(f"unsupported geometry {tuple(x.shape)}")
        ws = slot.workspace(need, v.device)
        E = self.embed_dim
        if fused is not None:
            if fused.shape != (B, T, 2 * E) or fused.dtype != torch.float32 or not fused.is_contiguous() or fused.device != v.device:
                raise ValueError(f"fused must be a contiguous fp32 (B, T, 2 * embed_dim) tensor on {v.device}")
            out, out_ptr, ld, zl = fused, fused.data_ptr() + 4 * E, 2 * E, E
        else:
            out = torch.empty((B, T, E), dtype=torch.float32, device=v.device)
            out_ptr, ld, zl = out.data_ptr(), E, 0
        if u8:
            tc = _lib.VideoTransformC(tf.sub0, tf.div0, tf.mean, tf.std, tf.crop_h, tf.crop_w)
            _lib.check(lib.svt_video_forward_u8(slot.handle, _lib.ptr(v), B, T, Hin, Win, C.byref(tc), C.c_void_p(out_ptr), ld, zl,
                                                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(v.device)), "svt_video_forward_u8", lib)
        else:
            _lib.check(lib.svt_video_forward_ex(slot.handle, _lib.ptr(v), B, T, H, W, C.c_void_p(out_ptr), ld, zl, _lib.ptr(ws),
                                                ws.numel(), _lib.stream_ptr(v.device)), "svt_video_forward_ex", lib)
        return out


VideoFrontend = SubModel


# =================================================================================================
# AV-HuBERT video encoder = lip front-end + transformer (features-in mode of the encoder C-ABI)
# =================================================================================================
from .config import EncoderConfig, PRESETS  # noqa: E402
from .huggingface_interface import _config_to_c  # noqa: E402
from .weights import seeded_avhubert_video_state_dict, fairseq_to_hf_key  # noqa: E402

_IGNORED_PREFIXES = ("mask_emb", "label_embs_concat", "final_proj.", "target_glu.", "feature_extractor_audio.")


class FairseqAVHubertPretrain(ReplicaAware, nn.Module):
    """Drop-in for ``N20EMv2/video_only/fairseq_interface.py:350-499`` on the video modality: ``forward({"video": x,
    "audio": None})`` with ``x`` the ``(B, 1, T, H, W)`` lip ROI returns the ``(B, T, D)`` AV-HuBERT encoding
    (``extract_finetune``, ``hubert.py:688-739``: front-end -> cat([zeros, video]) -> LayerNorm -> post_extract_proj ->
    TransformerEncoder), then the wrapper's optional whole-tensor ``F.layer_norm`` (``output_norm``).

    Parameters keep the fairseq names under ``model.`` (``model.feature_extractor_video.resnet...``, ``model.layer_norm.weight``,
    ``model.post_extract_proj.weight``, ``model.encoder.pos_conv.0.weight_g``, ``model.encoder.layers.N.self_attn.q_proj.weight`` ...), so the
    ``model`` entry of a fairseq AV-HuBERT checkpoint loads with ``load_fairseq_model_state``; pre-training-only entries
    (``mask_emb``, ``final_proj``, ``label_embs_concat``, ``feature_extractor_audio``) are ignored.

    The lip front-end is pinned to the reference; the transformer runs the same kernels as the (pinned) wav2vec2 /
    HuBERT encoders but fairseq itself is absent from the build container, so that part is parity-unpinned."""

    def __init__(self, pretrained_path=None, save_path=None, input_norm=None, output_norm=True, freeze=True, pretrain=True,
                 dropout=None, *, config: "EncoderConfig | str" = "avhubert-large-video", precision=None, seed: int = 5986):
        super().__init__()
        if input_norm:
            raise NotImplementedError("input_norm: the reference would layer-norm the modality dict itself; unused by the recipes")
        cfg = PRESETS[config] if isinstance(config, str) else config
        if cfg.conv_kernel:
            raise ValueError("the AV-HuBERT video encoder needs a features-in configuration (empty conv_kernel)")
        if cfg.conv_dim[0] != 2 * cfg.hidden_size:
            raise ValueError("modality_fuse='concat': the transformer input width must be 2 * hidden_size")
        self.config = cfg
        self.output_norm = output_norm
        self.freeze = freeze
        self.normalize = False
        self.precision = precision or os.environ.get("SVT_PRECISION", "bf16")
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}")
        self.model = ParamTree()
        self.model.add_module("feature_extractor_video", SubModel(512, cfg.hidden_size, "prelu", precision=self.precision, seed=seed))
        for k, v in seeded_avhubert_video_state_dict(cfg, seed=seed).items():
            if not k.startswith("feature_extractor_video."):
                self.model.add(k, v)
        if pretrained_path is not None and pretrain:
            # the reference downloads `pretrained_path` (a URL in the recipes' yaml) to `save_path` and loads that file
            # (fairseq_interface.py:392-412); there is no network here, so the file has to be in place already
            path = save_path if (save_path and os.path.exists(save_path)) else pretrained_path
            if str(path).startswith(("http://", "https://")):
                raise _lib.SvtError(f"no network in this build: put the AV-HuBERT checkpoint at save_path ({save_path!r}) "
                                    f"instead of downloading {pretrained_path}")
            self.load_fairseq_model_state(self._read_fairseq_checkpoint(path))
        if self.freeze:
            self.eval()
            for p in self.parameters():
                p.requires_grad = False
        self._dev = DeviceObjects("svt_encoder_destroy", LIB_VARIANT.get(self.precision))  # one C object per device, shared with DataParallel replicas

    @staticmethod
    def _read_fairseq_checkpoint(path):
        try:
            state = torch.load(path, map_location="cpu", weights_only=False)
        except Exception as ex:  # a fairseq checkpoint pickles its cfg with omegaconf / fairseq classes
            raise _lib.SvtError(f"cannot unpickle {path} without fairseq/omegaconf installed ({type(ex).__name__}: {ex}); "
                                "re-save its state['model'] as a plain tensor dict and pass it to load_fairseq_model_state") from ex
        return state["model"] if isinstance(state, dict) and "model" in state else state

    def load_fairseq_model_state(self, model_state):
        own = self.model.state_dict()
        sd = {}
        for k, v in model_state.items():
            if k.startswith(_IGNORED_PREFIXES):
                continue
            sd[k] = v
        missing = [k for k in own if k not in sd and not k.endswith("num_batches_tracked")]
        unexpected = [k for k in sd if k not in own]
        if missing or unexpected:
            raise RuntimeError(f"AV-HuBERT state mismatch: missing {missing[:5]} unexpected {unexpected[:5]}")
        for k in own:
            if k not in sd:
                sd[k] = own[k]
        self.model.load_state_dict(sd, strict=True)

    def _transformer_tensors(self):
        for n, p in self.model.named_parameters():
            if not n.startswith("feature_extractor_video."):
                yield n, p

    def _sync(self, device):
        lib = _lib.load(LIB_VARIANT.get(self.precision))
        _lib.require_gpu()
        idx = _lib.dev_index(device)
        slot = self._dev.slot(idx, (self.precision, bool(self.output_norm)))
        src = self._param_owner()   # a DataParallel replica reads the ORIGINAL's tensors
        sig = tuple((p.data_ptr(), p._version) for _, p in src._transformer_tensors())
        if slot.handle is not None and sig == slot.sig:
            return slot
        if slot.handle is None:
            h = C.c_void_p()
            cc = _config_to_c(self.config, False, bool(self.output_norm), self.precision)
            _lib.check(lib.svt_encoder_create(C.byref(cc), idx, C.byref(h)), "svt_encoder_create", lib)
            slot.handle = h
        for name, p in src._transformer_tensors():
            hf = fairseq_to_hf_key(name)
            if hf is None:
                raise _lib.SvtError(f"no encoder slot for parameter {name}")
            t = p.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(lib.svt_encoder_load_param(slot.handle, hf.encode(), C.c_void_p(t.data_ptr()), 0, shape, t.dim()),
                       f"svt_encoder_load_param({hf})", lib)
        _lib.check(lib.svt_encoder_finalize(slot.handle), "svt_encoder_finalize", lib)
        slot.sig = sig
        return slot

    def forward(self, wav, clips_per_norm_group: int = 0):
        """``clips_per_norm_group`` (extension, 0 = the reference): the wrapper's whole-tensor output norm over groups of that
        many clips; 1 makes a batch of equal-length clips equal to batch-1 forwards (see HuggingFaceWav2Vec2.forward)."""
        with torch.no_grad():
            return self.extract_features(wav, clips_per_norm_group).detach()

    def extract_features(self, wav, clips_per_norm_group: int = 0):
        if not isinstance(wav, dict) or "video" not in wav:
            raise ValueError('expected {"video": (B,1,T,H,W) tensor, "audio": None}')
        if wav.get("audio") is not None:
            raise NotImplementedError("the recipes feed AV-HuBERT the video modality only (audio=None)")
        video = wav["video"]            # (B, 1, T, H, W) float, normalised -- or the raw uint8 ROI (transform_eval runs in the kernel)
        fe = self.model.feature_extractor_video
        E = fe.embed_dim
        u8 = video.dtype == torch.uint8
        B = video.shape[0]
        T = video.shape[1] if (u8 and not (video.dim() == 5 and video.shape[1] == 1)) else video.shape[2]
        # cat([zeros (absent audio), video], -1) (hubert.py:700-712) written in place by the front-end's own call: its projection stores
        # rows with pitch 2E into the right half, one 2-D memset node zeroes the left half -- no torch fill / strided copy
        feats = fe.forward_into(video, torch.empty((B, T, 2 * E), dtype=torch.float32, device=video.device), wav.get("transform"))
        lib = _lib.load(LIB_VARIANT.get(self.precision))
        slot = self._sync(feats.device)
        need = lib.svt_encoder_workspace_bytes(slot.handle, B, T)
        if need < 0:
            raise _lib.SvtError(_lib.last_error(lib))
        ws = slot.workspace(need, feats.device)
        out = torch.empty((B, T, self.config.hidden_size), dtype=torch.float32, device=feats.device)
        _lib.check(lib.svt_encoder_forward_ex(slot.handle, _lib.ptr(feats), B, T, _lib.ptr(out), _lib.ptr(ws),
                                              ws.numel(), _lib.stream_ptr(feats.device), int(clips_per_norm_group)),
                   "svt_encoder_forward", lib)
        return out
