"""Device-side objects of a host module (C handles of ``libsvt_mi355.so`` + their workspaces).

A module of this package (encoder, head, fusion, lip front-end) owns ONE ``DeviceObjects`` registry with one
slot per device index.  Why a registry instead of a bare ``_handle`` attribute:

* ``nn.DataParallel`` -- the reference's multi-GPU mode (``speechbrain/core.py:1164-1169``) -- replicates a module by
  shallow-copying its ``__dict__`` on every forward.  With a bare handle every replica would share (and at collection
  time free) the original's handle, and a replica placed on another device would destroy and re-create it.  With the
  registry shared by reference, replica ``k`` finds (or creates once) the slot of ITS device, nothing is freed while any
  replica is alive, and the C objects die exactly once, with the registry.
* an explicit second object on the SAME device (``HuggingFaceWav2Vec2.replica()``: two forwards in flight on two
  streams) gets a registry of its own.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib


class ReplicaAware:
    """``nn.DataParallel`` support for a module that uploads its parameters into a C object.

    ``torch.nn.parallel.replicate`` gives every replica -- and every child module of it -- ``_parameters = {}`` and hangs the
    per-forward broadcast copies on it as plain attributes: on a replica ``parameters()``, ``named_parameters()`` and
    ``state_dict()`` are EMPTY (buffers survive), and the copies are new tensors on every forward.  A replica therefore never reads
    its own tree: it remembers the module it was replicated from (``_dp_origin``, kept out of ``_modules``) and both the upload
    loop and the "did the weights change" signature walk THAT module's tensors; the values travel through the host, so the slot of
    device k is filled once from the original's parameters wherever they live."""

    def _replicate_for_data_parallel(self):
        r = super()._replicate_for_data_parallel()
        r.__dict__["_dp_origin"] = self.__dict__.get("_dp_origin") or self   # not via setattr: a Module value would be registered as a child
        return r

    def _param_owner(self):
        return self.__dict__.get("_dp_origin") or self


class DeviceSlot:
    """One C object on one device: its handle, what was uploaded into it, and its workspace."""
    __slots__ = ("key", "handle", "sig", "gen", "ws")

    def __init__(self, key):
        self.key = key
        self.handle = None
        self.sig = None   # signature of the uploaded parameters (tuple of (data_ptr, version)), or None
        self.gen = -1     # generation of the owner's parameters at upload time
        self.ws: Optional[torch.Tensor] = None

    def workspace(self, nbytes: int, device: torch.device) -> torch.Tensor:
        if self.ws is None or self.ws.numel() < nbytes or self.ws.device != device:
            self.ws = None  # release before allocating the larger one
            self.ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.ws


class DeviceObjects:
    def __init__(self, destroy_symbol: str, lib_variant: str = None):
        self._destroy = destroy_symbol
        self._variant = lib_variant   # which build of the library created the handles (None: libsvt_mi355.so)
        self._slots: Dict[int, DeviceSlot] = {}

    def slot(self, dev_index: int, key) -> DeviceSlot:
        """The slot of ``dev_index``; a slot built for other flags (``key`` differs) is destroyed and replaced."""
        s = self._slots.get(dev_index)
        if s is not None and s.key != key:
            self._free(s)
            s = None
        if s is None:
            s = DeviceSlot(key)
            self._slots[dev_index] = s
        return s

    def _free(self, s: DeviceSlot) -> None:
        if s.handle is not None:
            try:
                getattr(_lib.load(getattr(self, "_variant", None)), self._destroy)(s.handle)
            except Exception:
                pass
            s.handle = None
        s.ws = None

    def close(self) -> None:
        for s in list(self._slots.values()):
            self._free(s)
        self._slots.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
