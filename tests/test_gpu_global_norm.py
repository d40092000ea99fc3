"""The optional "global-batch-equivalent" norms: a batch that is one shard of a larger one, with the
wrapper's two whole-batch layer norms taken over the WHOLE global batch through a cross-rank reduction of the (sum, sum of
squares) pairs (`svt_encoder_set_norm_reduce`, `HuggingFaceWav2Vec2.set_norm_reduce / set_global_batch_norm`).  One GPU is enough to
check the arithmetic: the two "ranks" run one after the other and the reduction function adds the other shard's recorded pair --
three passes, because the output statistics depend on the input ones.  Uneven shards (3 + 2 clips), both conv-norm layouts, the
separate and the fused (encoder + head + decode) tail."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402
from svt_speechbrain_amd.config import PRESETS  # noqa: E402

DEV = "cuda:0"


def sharded(run, enc, shards, n_clips):
    """run(x) -> output of one shard under the CURRENT reduction function of `enc`; returns the per-shard outputs with both
    norms reduced over all shards (sequential emulation of the ranks)."""
    n = len(shards)
    wav_sums, out_sums = [None] * n, [None] * n

    def reducer(me, stage):
        calls = {"i": 0}

        def fn(t):
            assert t.dtype == torch.float64 and t.numel() == 2 and t.is_cuda
            k = calls["i"]
            calls["i"] += 1
            if k == 0:
                if stage == 0:
                    wav_sums[me] = t.clone()
                else:
                    t += sum(wav_sums[o] for o in range(n) if o != me)
            else:
                if stage == 1:
                    out_sums[me] = t.clone()
                elif stage == 2:
                    t += sum(out_sums[o] for o in range(n) if o != me)
        return fn

    outs = None
    for stage in range(3):
        outs = []
        for me, x in enumerate(shards):
            enc.set_norm_reduce(reducer(me, stage), n_clips)
            outs.append(run(x))
    enc.set_norm_reduce(None)
    return outs


@pytest.mark.parametrize("cfg_name,L", [("tiny-group", 4000), ("tiny-layer", 4000), ("wav2vec2-base", 16000)])
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("fp16x3", 2e-4), ("bf16", None)])
def test_shards_with_reduced_norms_equal_the_whole_batch(cfg_name, L, prec, tol):
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, normalize_wav=True, precision=prec, seed=21).to(DEV)
    g = torch.Generator().manual_seed(3)
    wav = ((0.05 + 0.1 * torch.rand(5, 1, generator=g)) * torch.randn(5, L, generator=g)).clamp_(-1, 1).to(DEV)  # clips of different loudness
    want = enc(wav)
    local = torch.cat([enc(wav[:3]), enc(wav[3:])])          # per-shard norms: what the reference's DataParallel computes
    got = torch.cat(sharded(lambda x: enc(x), enc, [wav[:3], wav[3:]], 5))
    d_local = (local - want).abs().max().item()
    d = (got - want).abs().max().item()
    print(f"{cfg_name} {prec}: reduced norms vs whole batch {d:.2e}; per-shard norms vs whole batch {d_local:.2e}")
    assert d_local > 1e-3, "the case must tell the two apart"
    if tol is not None:
        assert d < tol, d
    else:
        assert (got - want).abs().mean().item() < 0.05   # bf16: different tile heights at different batch sizes round differently
    assert torch.equal(enc(wav), want), "the reduction is off again"


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("fp16x3", 2e-4)])
def test_fused_tail_with_reduced_norms(prec, tol):
    cfg = PRESETS["wav2vec2-base"]
    enc = S.HuggingFaceWav2Vec2("wav2vec2-base", None, config=cfg, normalize_wav=True, precision=prec, seed=22).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=23))
    head = head.to(DEV)
    g = torch.Generator().manual_seed(4)
    wav = ((0.05 + 0.1 * torch.rand(4, 1, generator=g)) * torch.randn(4, 24000, generator=g)).clamp_(-1, 1).to(DEV)
    want = enc.forward_head(wav, head)
    got = torch.cat(sharded(lambda x: enc.forward_head(x, head), enc, [wav[:1], wav[1:]], 4))
    assert (got - want).abs().max().item() < tol


def test_reduction_contract_errors():
    cfg = PRESETS["tiny-group"]
    enc = S.HuggingFaceWav2Vec2("tiny-group", None, config=cfg, normalize_wav=True, precision="fp32", seed=1).to(DEV)
    wav = torch.zeros(2, 4000, device=DEV)
    with pytest.raises(ValueError):
        enc.set_norm_reduce(lambda t: None, 0)
    enc.set_norm_reduce(lambda t: None, 1)       # smaller than the batch
    with pytest.raises(_lib.SvtError, match="global_clips"):
        enc(wav)
    enc.set_norm_reduce(lambda t: None, 2)
    with pytest.raises(_lib.SvtError, match="whole-batch"):
        enc(wav, clips_per_norm_group=1)

    def boom(t):
        raise RuntimeError("reduction failed")
    enc.set_norm_reduce(boom, 2)
    with pytest.raises(_lib.SvtError, match="callback"):
        enc(wav)
    enc.set_norm_reduce(None)
    assert torch.isfinite(enc(wav)).all()
