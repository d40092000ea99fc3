"""Out-of-bounds hunt (this pool has no GPU sanitizer): the forward passes with every device buffer the kernels see — weights and
packed pieces (library allocations), the workspace, the inputs, the outputs — placed as its own mapping between two UNMAPPED
pages of address space, flush against the end (mode 1) or the start (mode 2) of the mapping (svt_debug_set key 13,
svt_debug_alloc).  A kernel that reads or writes one element past that edge takes a page fault (the process aborts with the
runtime's "Memory access fault" message; tools/guard_hunt.py runs the cases one per process and names the kernel) instead of
silently touching a neighbour; outputs must also equal the ordinary run bit for bit.

Round 2 found its first bug this way: the register-staged GEMM read up to 112 bytes past the last operand row when K was
shorter than one K slab (a fuzz case aborted one run in three, depending on what the allocator had put behind the buffer)."""
import ctypes as C
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import _device, _lib  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402
from svt_speechbrain_amd.config import PRESETS  # noqa: E402

from test_gpu_fuzz import random_case  # noqa: E402

DEV = torch.device("cuda:0")
_real_empty = torch.empty


class _Guarded:
    """A device buffer from svt_debug_alloc exposed through __cuda_array_interface__ (torch.as_tensor wraps it without a copy
    and keeps this object alive for as long as the storage lives)."""

    def __init__(self, nbytes):
        self.lib = _lib.load()
        self.n = max(int(nbytes), 16)
        p = C.c_void_p()
        _lib.check(self.lib.svt_debug_alloc(C.byref(p), self.n, 0), "svt_debug_alloc")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (self.n,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        torch.cuda.synchronize()
        self.lib.svt_debug_free(self.ptr, 0)


def guarded_bytes(nbytes):
    return torch.as_tensor(_Guarded(nbytes), device=DEV)


def guarded_empty(*size, dtype=None, device=None, **kw):
    """torch.empty for the package's device-side outputs, page-guarded."""
    if device is None or torch.device(device).type != "cuda":
        return _real_empty(*size, dtype=dtype, device=device, **kw)
    shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(int(s) for s in size)
    dtype = dtype or torch.float32
    n = math.prod(shape) * _real_empty((), dtype=dtype).element_size()
    return guarded_bytes(n)[:n].view(dtype).view(shape)


def guarded_like(x):
    v = guarded_empty(tuple(x.shape), dtype=x.dtype, device=x.device)
    v.copy_(x)
    return v


def _guarded_workspace(self, nbytes, device):   # exactly the bytes the library asked for
    self.ws = None
    self.ws = guarded_bytes(nbytes)
    return self.ws


class Guard:
    """with Guard(mode): library allocations, workspaces and torch.empty outputs are page-guarded."""

    def __init__(self, mode):
        self.mode = mode
        self.mp = pytest.MonkeyPatch()

    def __enter__(self):
        for lib in (_lib.load(), _lib.load("f16")):   # both builds of the library keep their own allocator switch
            lib.svt_debug_set(13, self.mode)
        self.mp.setattr(_device.DeviceSlot, "workspace", _guarded_workspace)
        self.mp.setattr(torch, "empty", guarded_empty)
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.mp.undo()
        for lib in (_lib.load(), _lib.load("f16")):
            lib.svt_debug_set(13, 0)


def same(a, b):
    if isinstance(a, torch.Tensor):
        return torch.equal(a.cpu(), b.cpu())
    if isinstance(a, dict):
        return all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    try:
        return bool((a == b).all())
    except AttributeError:
        return a == b


def check_guarded(run, what):
    """run(to_dev): builds its modules, moves inputs with to_dev, returns outputs.  Ordinary run, then both guard modes."""
    want = run(lambda x: x.to(DEV))
    torch.cuda.synchronize()
    for mode in (1, 2):
        with Guard(mode):
            got = run(lambda x: guarded_like(x.to(DEV)))
            torch.cuda.synchronize()
            assert same(want, got), (what, mode)
            del got


PRECISIONS = ("fp32", "fp16x3", "bf16x3", "bf16", "fp16")


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("SVT_GUARD_CASES", "24")))))
def test_encoder_random_geometries_stay_in_bounds(seed):
    cfg, B, L = random_case(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)
    for prec in PRECISIONS:
        def run(to_dev):
            enc = S.HuggingFaceWav2Vec2(cfg.name, None, config=cfg, normalize_wav=True, precision=prec, seed=seed).to(DEV)
            return enc(to_dev(wav)).cpu()
        check_guarded(run, (seed, prec))


@pytest.mark.parametrize("cfg_name,prec,shapes", [
    ("wav2vec2-base", "bf16", [(1, 16000), (5, 80000), (9, 31000)]),
    ("wav2vec2-base", "fp16x3", [(1, 16000), (3, 47000)]),
    ("wav2vec2-base", "fp16", [(1, 16000), (5, 47000)]),
    ("wav2vec2-base", "bf16x3", [(2, 23000)]),
    ("wav2vec2-base", "fp32", [(2, 23000)]),
    ("wavlm-base", "bf16", [(2, 47000)]),
    ("wavlm-base", "fp16x3", [(1, 30000)]),
    ("wav2vec2-large-lv60", "bf16", [(3, 47000)]),
    ("wav2vec2-large-lv60", "fp16x3", [(2, 20000)]),
    ("hubert-large-ll60k", "bf16", [(2, 33000)]),
    ("data2vec-audio-base", "bf16", [(2, 33000)]),
])
def test_full_width_models_stay_in_bounds(cfg_name, prec, shapes):
    """The kernels only full-width models reach (persistent LDS-DMA GEMMs, fused out-projection + LayerNorm, the small-problem
    kernel, 8-wave fused attention, the LDS-DMA split-operand GEMM), fused tail (encoder + head + decode) included."""
    if cfg_name not in PRESETS:
        pytest.skip(f"no preset {cfg_name}")
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision=prec, seed=3).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=4))
    head = head.to(DEV)
    for B, L in shapes:
        g = torch.Generator().manual_seed(B * 100 + L)
        wav = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)

        def run(to_dev):
            enc._dev.close()    # the device-side objects are rebuilt (weights uploaded again) under the allocator of this run
            head._dev.close()
            x = to_dev(wav)
            feats = enc(x)
            logits = head(feats)
            frames = torch.empty((B * cfg.frames(L), 4), dtype=torch.int32, device=DEV)
            fused = enc.forward_head(x, head, frames=frames)
            return feats.cpu(), logits.cpu(), S.decode_frames(logits), fused.cpu(), frames.cpu()
        check_guarded(run, (cfg_name, prec, B, L))
    enc._dev.close()
    head._dev.close()


@pytest.mark.parametrize("seed", list(range(12)))
def test_other_entry_points_stay_in_bounds(seed):
    r = random.Random(5000 + seed)
    g = torch.Generator().manual_seed(seed)
    d_model = r.choice([64, 128, 256, 1024])
    nhead = r.choice([h for h in (1, 2, 4, 8) if d_model % h == 0 and (d_model // h) % 8 == 0])
    d_ffn = r.choice([64, 128, 256])
    B, T1 = r.choice([1, 2, 3]), r.choice([1, 7, 33, 100, 250])
    T2 = max(1, T1 + r.choice([-5, -1, 0, 1, 4]))
    sd = W.seeded_fusion_state_dict(d_model, d_ffn, seed=seed, max_len=300)
    a = torch.randn(B, T1, d_model, generator=g)
    v = torch.randn(B, T2, d_model, generator=g)
    for prec in PRECISIONS:
        def run(to_dev):
            fus = S.FusionRCA(nhead=nhead, d_ffn=d_ffn, d_model=d_model, precision=prec, max_length=300, seed=seed).to(DEV)
            fus.load_state_dict(sd)
            return fus(to_dev(a), to_dev(v)).cpu()
        check_guarded(run, ("fusion", seed, prec))
    Bc, Tc, V = r.choice([1, 3, 6]), r.choice([1, 5, 40, 200]), r.choice([2, 5, 31])
    probs = torch.rand(Bc, Tc, V, generator=g)
    lens = torch.rand(Bc, generator=g) * 0.9 + 0.1
    blank = r.choice([0, -1, V - 1])
    check_guarded(lambda to_dev: S.ctc_greedy_decode(to_dev(probs), to_dev(lens), blank), ("ctc", seed))
    Lw = r.choice([400, 1600, 4801, 16000, 23457])
    wav = 0.1 * torch.randn(r.choice([1, 2, 5]), Lw, generator=g)
    check_guarded(lambda to_dev: S.Fbank()(to_dev(wav)).cpu(), ("fbank", seed))
    n_in, rows = r.choice([64, 512, 768, 1024]), r.choice([1, 7, 249, 1000])
    hd = W.seeded_head_state_dict(n_in, 20, seed=seed)
    feats = torch.randn(2, rows, n_in, generator=g)

    def run_head(to_dev):
        head = S.Linear(20, input_size=n_in)
        head.load_state_dict(hd)
        logits = head.to(DEV)(to_dev(feats))
        return logits.cpu(), S.decode_frames(logits)
    check_guarded(run_head, ("head", seed))
    # losses over the logits (bce on the onset / offset columns, nll on the class columns), with relative lengths
    lg = torch.randn(3, 40, 20, generator=g)
    tgt = (torch.rand(3, 40, generator=g) > 0.5).float()
    cls = torch.randint(0, 5, (3, 40), generator=g)
    ln = torch.tensor([1.0, 0.55, 0.8])

    def run_losses(to_dev):
        x = to_dev(lg)
        return (S.bce_loss(x[:, :, 0].contiguous(), to_dev(tgt), to_dev(ln)).cpu(),
                S.nll_loss(torch.log_softmax(x[:, :, 2:7], -1).contiguous(), to_dev(cls), to_dev(ln)).cpu())
    check_guarded(run_losses, ("losses", seed))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_video_branch_stays_in_bounds(prec):
    from svt_speechbrain_amd.video import FairseqAVHubertPretrain, SubModel
    cfg = PRESETS["tiny-avhubert-video"]
    sd = W.seeded_avhubert_video_state_dict(cfg, seed=123)
    g = torch.Generator().manual_seed(5)
    video = torch.randn(2, 1, 9, 40, 40, generator=g)

    def run(to_dev):
        m = FairseqAVHubertPretrain(config=cfg, precision=prec, seed=77, output_norm=True)
        m.load_fairseq_model_state(sd)
        return m.to(DEV)({"video": to_dev(video), "audio": None}).cpu()
    check_guarded(run, ("avhubert", prec))
    for hw, T in ((88, 3), (32, 1), (50, 2), (60, 2)):   # stage-1 widths 22 / 8 / 13 / 15: every row layout of the front-end
        clip = torch.randn(1, 1, T, hw, hw, generator=g)

        def run_front(to_dev):
            m = SubModel(512, 64, "prelu", precision=prec, seed=1).to(DEV)
            return m(to_dev(clip)).cpu()
        check_guarded(run_front, ("lip front-end", prec, hw, T))
