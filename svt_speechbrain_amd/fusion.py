"""Drop-in for ``N20EMv2/audio_visual/fusion.py`` (``FusionRCA``, reference :186-209; ``RCANet`` :9-79,
``RCALayer`` :82-183).  Same constructor and ``forward(audio_feats, video_feats)``; the state-dict keys are
the reference's (``fusion.layer{1,2}.self_att.att.in_proj_weight`` ..., ``fusion.positional_encoding.pe``).

MI355X design notes (vs the reference's two ``nn.MultiheadAttention`` calls per layer): the kv stream is
projected ONCE per layer with the packed (3D, D) matrix and its K/V serve both the self- and the
cross-attention (the reference recomputes them); the two attention outputs are blended
before the single output projection (the projection is linear)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from ._device import DeviceObjects, ReplicaAware
from .huggingface_interface import ParamTree, PRECISIONS, LIB_VARIANT
from .weights import seeded_fusion_state_dict


class FusionRCA(ReplicaAware, nn.Module):
    def __init__(self, alpha=0.5, nhead=8, d_ffn=3072, d_model=1024, *, precision=None, max_length=2500, seed=3986):
        super().__init__()
        import os
        self.alpha, self.nhead, self.d_ffn, self.d_model, self.max_length = alpha, nhead, d_ffn, d_model, max_length
        self.precision = precision or os.environ.get("SVT_PRECISION", "bf16")
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}")
        tree = ParamTree()
        pe = None
        for k, v in seeded_fusion_state_dict(d_model, d_ffn, seed=seed, max_len=max_length).items():
            k = k[len("fusion."):]
            if k.endswith(".pe"):
                pe = v
                continue
            tree.add(k, v)
        # the sinusoidal table is a buffer in the reference (Transformer.py:200-213): part of the state dict
        tree._modules.setdefault("positional_encoding", ParamTree())
        tree._modules["positional_encoding"].register_buffer("pe", pe)
        # keep module order of the reference: positional_encoding, layer1, layer2
        ordered = ParamTree()
        for name in ("positional_encoding", "layer1", "layer2"):
            ordered.add_module(name, tree._modules[name])
        self.fusion = ordered
        self._dev = DeviceObjects("svt_rca_destroy", LIB_VARIANT.get(self.precision))  # one C object per device, shared with DataParallel replicas

    def _tensors(self):
        for n, p in self.fusion.named_parameters():
            yield "fusion." + n, p
        for n, b in self.fusion.named_buffers():
            yield "fusion." + n, b

    def _sync(self, device):
        lib = _lib.load(LIB_VARIANT.get(self.precision))
        _lib.require_gpu()
        idx = _lib.dev_index(device)
        slot = self._dev.slot(idx, (self.precision, float(self.alpha)))
        src = self._param_owner()   # a DataParallel replica reads the ORIGINAL's tensors (its own tree holds no parameters)
        sig = tuple((t.data_ptr(), t._version) for _, t in src._tensors())
        if slot.handle is not None and sig == slot.sig:
            return slot
        if slot.handle is None:
            h = C.c_void_p()
            _lib.check(lib.svt_rca_create(self.d_model, self.nhead, self.d_ffn, float(self.alpha), self.max_length,
                                          PRECISIONS[self.precision], idx, C.byref(h)), "svt_rca_create", lib)
            slot.handle = h
        for name, t in src._tensors():
            c = t.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * c.dim())(*c.shape)
            _lib.check(lib.svt_rca_load_param(slot.handle, name.encode(), C.c_void_p(c.data_ptr()), 0, shape, c.dim()),
                       f"svt_rca_load_param({name})", lib)
        _lib.check(lib.svt_rca_finalize(slot.handle), "svt_rca_finalize", lib)
        slot.sig = sig
        return slot

    def forward(self, audio_feats: torch.Tensor, video_feats: torch.Tensor) -> torch.Tensor:
        if not (audio_feats.is_cuda and video_feats.is_cuda):
            raise _lib.SvtError("FusionRCA needs its inputs on the GPU; there is no CPU fallback")
        B, T1, D = audio_feats.shape
        B2, T2, D2 = video_feats.shape
        if D != self.d_model or D2 != self.d_model or B2 != B:
            raise ValueError(f"expected (B, T, {self.d_model}) inputs with equal batch, got {tuple(audio_feats.shape)} "
                             f"and {tuple(video_feats.shape)}")
        if abs(T1 - T2) > 15:
            print("Alignment is wrong")  # the reference's diagnostic (fusion.py:204-205)
        lib = _lib.load(LIB_VARIANT.get(self.precision))
        slot = self._sync(audio_feats.device)
        a = audio_feats.detach().to(torch.float32).contiguous()
        v = video_feats.detach().to(torch.float32).contiguous()
        need = lib.svt_rca_workspace_bytes(slot.handle, B, T1)
        if need < 0:
            raise _lib.SvtError(_lib.last_error(lib))
        ws = slot.workspace(need, a.device)
        out = torch.empty((B, T1, D), dtype=torch.float32, device=a.device)
        _lib.check(lib.svt_rca_forward(slot.handle, _lib.ptr(a), T1, _lib.ptr(v), T2, B, _lib.ptr(out),
                                       _lib.ptr(ws), ws.numel(), _lib.stream_ptr(a.device)),
                   "svt_rca_forward", lib)
        return out
