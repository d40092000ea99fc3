"""Out-of-bounds hunt (this pool has no GPU sanitizer): the forward passes with every device buffer the kernels see — weights and
packed pieces (library allocations), the workspace, the input waveform — placed as its own mapping between two UNMAPPED granules
of address space, flush against the end (mode 1) or the start (mode 2) of the mapping (svt_debug_set key 13, svt_debug_alloc).
A kernel that reads or writes one element past that edge takes a page fault (the process aborts with the runtime's "Memory
access fault" message) instead of silently touching a neighbour; outputs must also equal the ordinary run bit for bit."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import _device, _lib  # noqa: E402

from test_gpu_fuzz import random_case  # noqa: E402

DEV = torch.device("cuda:0")


class _Guarded:
    """A device buffer from svt_debug_alloc exposed through __cuda_array_interface__ (torch.as_tensor wraps it without a copy)."""

    def __init__(self, nbytes):
        self.lib = _lib.load()
        p = C.c_void_p()
        _lib.check(self.lib.svt_debug_alloc(C.byref(p), max(int(nbytes), 16), 0), "svt_debug_alloc")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (max(int(nbytes), 16),), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        torch.cuda.synchronize()
        self.lib.svt_debug_free(self.ptr, 0)


def guarded_bytes(nbytes):
    owner = _Guarded(nbytes)
    t = torch.as_tensor(owner, device=DEV)
    t._svt_owner = owner   # the mapping lives as long as the tensor object the test holds
    return t


def guarded_like(x):
    t = guarded_bytes(x.numel() * x.element_size())
    v = t[:x.numel() * x.element_size()].view(x.dtype).view(x.shape)
    v.copy_(x)
    v._svt_owner = t
    return v


def _guarded_workspace(self, nbytes, device):   # exactly the bytes the library asked for
    self.ws = None
    self.ws = guarded_bytes(nbytes)
    return self.ws


PRECISIONS = ("fp32", "fp16x3", "bf16x3", "bf16")


@pytest.mark.parametrize("seed", list(range(24)))
def test_encoder_random_geometries_stay_in_bounds(seed, monkeypatch):
    lib = _lib.load()
    cfg, B, L = random_case(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)

    def build(prec):
        return S.HuggingFaceWav2Vec2(cfg.name, None, config=cfg, normalize_wav=True, precision=prec, seed=seed).to(DEV)

    want = {prec: build(prec)(wav.to(DEV)).cpu() for prec in PRECISIONS}
    monkeypatch.setattr(_device.DeviceSlot, "workspace", _guarded_workspace)
    try:
        for mode in (1, 2):
            lib.svt_debug_set(13, mode)
            for prec in PRECISIONS:
                enc = build(prec)
                got = enc(guarded_like(wav.to(DEV))).cpu()
                assert torch.equal(got, want[prec]), (seed, prec, mode)
                del enc
    finally:
        torch.cuda.synchronize()
        lib.svt_debug_set(13, 0)
