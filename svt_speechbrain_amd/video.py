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

import torch
from torch import nn

from . import _lib
from .huggingface_interface import ParamTree, PRECISIONS
from .weights import seeded_video_frontend_state_dict


class SubModel(nn.Module):
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
        self._handle = None
        self._key = None
        self._sig = None
        self._ws = None

    def _tensors(self):
        for n, p in self.named_parameters():
            yield n, p
        for n, b in self.named_buffers():
            yield n, b

    def _sync(self, device):
        lib = _lib.load()
        _lib.require_gpu()
        idx = _lib.dev_index(device)
        key = (idx, self.precision)
        sig = tuple((t.data_ptr(), t._version) for _, t in self._tensors())
        if self._handle is not None and key == self._key and sig == self._sig:
            return
        if self._handle is not None and key != self._key:
            lib.svt_video_destroy(self._handle)
            self._handle = None
        if self._handle is None:
            h = C.c_void_p()
            _lib.check(lib.svt_video_create(self.embed_dim, PRECISIONS[self.precision], idx, C.byref(h)), "svt_video_create")
            self._handle, self._key = h, key
        for name, t in self._tensors():
            if name.endswith("num_batches_tracked"):
                continue
            c = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * c.dim())(*c.shape)
            _lib.check(lib.svt_video_load_param(self._handle, name.encode(), C.c_void_p(c.data_ptr()), 0, shape, c.dim()),
                       f"svt_video_load_param({name})")
        _lib.check(lib.svt_video_finalize(self._handle), "svt_video_finalize")
        self._sig = sig

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._sig = None
        return r

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.load().svt_video_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise _lib.SvtError("the lip front-end needs its input on the GPU; there is no CPU fallback")
        if x.dim() != 5 or x.shape[1] != 1:
            raise ValueError(f"expected a (B, 1, T, H, W) lip ROI tensor, got {tuple(x.shape)}")
        B, _, T, H, W = x.shape
        lib = _lib.load()
        self._sync(x.device)
        v = x.detach().to(torch.float32).contiguous()
        need = lib.svt_video_workspace_bytes(self._handle, B, T, H, W)
        if need < 0:
            raise ValueError(f"unsupported geometry {tuple(x.shape)}")
        if self._ws is None or self._ws.numel() < need or self._ws.device != v.device:
            self._ws = None
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=v.device)
        out = torch.empty((B, T, self.embed_dim), dtype=torch.float32, device=v.device)
        _lib.check(lib.svt_video_forward(self._handle, _lib.ptr(v), B, T, H, W, _lib.ptr(out), _lib.ptr(self._ws),
                                         self._ws.numel(), _lib.stream_ptr(v.device)), "svt_video_forward")
        return out.transpose(1, 2)  # (B, embed_dim, T), the reference's layout


VideoFrontend = SubModel
