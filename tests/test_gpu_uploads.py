"""Parameter uploads never free device memory (round 5).

With eight processes uploading to one GPU at the same time, the round-4 upload -- a temporary fp32 staging buffer per tensor, i.e. ~75
hipMalloc / hipFree pairs per encoder object -- left 12 of 1 152 encoder objects with wrong weights for their whole life
(profiles/r05_determinism_under_gpu_sharing.txt; tools/determinism_stress.py --encoders reproduces the statistics, which a single test
run cannot).  What a test CAN pin is the structural rule that removed it (csrc/api.hip, upload_operand): on a path that launches kernels
nothing is freed and re-allocated -- one staging buffer per device is kept, a re-upload of the same tensors reuses their buffers, the
split-operand registry re-packs in place.  svt_debug_set key 31 returns the number of device frees of the process so far."""
import gc

import pytest
import torch

import svt_speechbrain_amd as S
from svt_speechbrain_amd import _lib, weights as W
from svt_speechbrain_amd.config import PRESETS

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def frees(lib):
    return lib.svt_debug_set(31, 0)


@pytest.mark.parametrize("precision", ["bf16", "fp16x3", "fp32"])
def test_creating_and_reloading_an_encoder_frees_nothing(precision):
    cfg = PRESETS["tiny-group"]
    name = "tiny-group"
    warm = S.HuggingFaceWav2Vec2(name, None, config=cfg, precision=precision, normalize_wav=True, seed=3).to(DEV)
    wav = (0.1 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(0))).to(DEV)
    warm(wav)                                   # the staging buffer of this device now exists (it may have been grown: a free)
    lib = warm._lib()
    gc.collect()                                # objects of earlier tests are destroyed (and their buffers freed) NOW, not in the middle
    base = frees(lib)
    enc = S.HuggingFaceWav2Vec2(name, None, config=cfg, precision=precision, normalize_wav=True, seed=4).to(DEV)
    y0 = enc(wav)
    assert frees(lib) == base, "creating an encoder object freed device memory"
    rep = enc.replica()
    y1 = rep(wav)
    assert frees(lib) == base and torch.equal(y0, y1)
    # the same parameters again (what a training loop with an unfrozen encoder does every step): every buffer is reused
    enc.refresh()
    y2 = enc(wav)
    assert frees(lib) == base, "re-uploading unchanged shapes freed device memory"
    assert torch.equal(y0, y2)
    # new VALUES in the same shapes arrive in the same buffers
    sd = {k: (v + 0.01 * torch.randn_like(v) if v.is_floating_point() else v) for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)
    y3 = enc(wav)
    assert frees(lib) == base
    assert not torch.equal(y0, y3) and torch.isfinite(y3).all()


def test_a_reupload_waits_for_forwards_still_in_flight():
    """a re-upload overwrites the live weight buffers in place (same-size buffers are kept), on the null stream, while torch's
    side streams are non-blocking -- so svt_*_finalize must wait for the device before the first byte changes.  A queue of forwards on a
    side stream (tens of milliseconds of work), then new parameter VALUES and a forward on the main stream right away: every queued
    forward must still return what the OLD parameters give, bit for bit, and the new forward what a fresh object gives."""
    cfg = PRESETS["tiny-layer"]
    enc = S.HuggingFaceWav2Vec2("tiny-layer", None, config=cfg, precision="bf16", normalize_wav=True, seed=21).to(DEV)
    wav = (0.1 * torch.randn(24, 80000, generator=torch.Generator().manual_seed(5))).to(DEV)
    small = wav[:2, :8000].contiguous()
    want_old = enc(wav).clone()
    sd_new = {k: (v + 0.02 * torch.randn_like(v) if v.is_floating_point() else v) for k, v in enc.state_dict().items()}
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    with torch.cuda.stream(side):
        for _ in range(12):
            outs.append(enc(wav).clone())
    enc.load_state_dict(sd_new)            # marks the parameters stale; the upload happens inside the next forward
    got_new = enc(small).clone()           # main stream: re-upload (must drain `side` first), then a short forward
    torch.cuda.synchronize()
    for i, y in enumerate(outs):
        assert torch.equal(y, want_old), f"forward {i} queued before the re-upload read half-updated weights"
    fresh = S.HuggingFaceWav2Vec2("tiny-layer", None, config=cfg, precision="bf16", normalize_wav=True, seed=21).to(DEV)
    fresh.load_state_dict(sd_new)
    assert torch.equal(got_new, fresh(small))
