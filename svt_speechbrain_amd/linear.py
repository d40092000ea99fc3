"""Drop-in for ``speechbrain.nnet.linear.Linear`` (reference ``speechbrain/nnet/linear.py:15-76``): the
20-way frame head of the AMT recipes (``hparams/train_audio_ssl.yaml:113-115``).  Same constructor, the
``nn.Linear`` lives under attribute ``.w`` so the state-dict keys are ``w.weight`` / ``w.bias``.
fp32 in, fp32 accumulate, fp32 out on the GPU (decode thresholds are compared in fp32)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from ._device import DeviceObjects, ReplicaAware


class Linear(ReplicaAware, nn.Module):
    def __init__(self, n_neurons, input_shape=None, input_size=None, bias=True, combine_dims=False):
        super().__init__()
        self.combine_dims = combine_dims
        if input_shape is None and input_size is None:
            raise ValueError("Expected one of input_shape or input_size")
        if input_size is None:
            input_size = input_shape[-1]
            if len(input_shape) == 4 and self.combine_dims:
                input_size = input_shape[2] * input_shape[3]
        self.w = nn.Linear(input_size, n_neurons, bias=bias)
        # one C object per device AND per build of the library ("" = libsvt_mi355.so, "f16" = the IEEE-half build): a handle is
        # only ever handed to the library that created it (each build has its own allocator state, registries and last_error)
        # Both registries exist from the start: DataParallel replicas share this dict by reference and run `_sync` from threads, so
        # a lazy "create the entry on first use" would let two threads each build an entry and drop one that already owns a handle.
        self._devs = {"": DeviceObjects("svt_linear_destroy"), "f16": DeviceObjects("svt_linear_destroy", "f16")}
        self._dev = self._devs[""]

    def _upload_items(self):
        return self.w.weight, self.w.bias

    def _sync(self, device, lib_variant: str = None):
        """The head's C object on ``device`` inside the library build ``lib_variant`` (the fused tail of an encoder that lives in
        the IEEE-half build asks for "f16"); shared with DataParallel replicas, which upload the original's parameters."""
        lib = _lib.load(lib_variant)
        _lib.require_gpu()
        idx = _lib.dev_index(device)
        key = lib_variant or ""
        if key not in self._devs:
            raise _lib.SvtError(f"Linear: unknown library build {lib_variant!r}")
        slot = self._devs[key].slot(idx, (self.w.in_features, self.w.out_features, self.w.bias is not None))
        weight, bias = self._param_owner()._upload_items()
        sig = (weight.data_ptr(), weight._version, None if bias is None else (bias.data_ptr(), bias._version))
        if slot.handle is not None and sig == slot.sig:
            return slot
        if slot.handle is None:
            h = C.c_void_p()
            _lib.check(lib.svt_linear_create(self.w.in_features, self.w.out_features, int(bias is not None), idx,
                                             C.byref(h)), "svt_linear_create", lib)
            slot.handle = h
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
        _lib.check(lib.svt_linear_load(slot.handle, C.c_void_p(w.data_ptr()),
                                       C.c_void_p(b.data_ptr()) if b is not None else None), "svt_linear_load", lib)
        slot.sig = sig
        return slot

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 4 and self.combine_dims:
            x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
        if not x.is_cuda:
            raise _lib.SvtError("svt_speechbrain_amd.Linear needs its input on the GPU; there is no CPU fallback")
        lib = _lib.load()
        slot = self._sync(x.device)
        xf = x.detach().to(torch.float32).contiguous()
        rows = xf.numel() // xf.shape[-1] if xf.numel() else 0
        y = torch.empty(xf.shape[:-1] + (self.w.out_features,), dtype=torch.float32, device=x.device)
        if rows:
            _lib.check(lib.svt_linear_forward(slot.handle, _lib.ptr(xf), rows, _lib.ptr(y), _lib.stream_ptr(x.device)),
                       "svt_linear_forward")
        return y
