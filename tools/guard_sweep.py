"""One-off companion of tests/test_gpu_guard.py: the full-width presets over random batch sizes (1-33) and lengths (0.2-6 s) in bf16
and fp16x3 mode with every buffer page-guarded (plain forward, fused tail, per-clip norm groups).  Round 2: 40 cases, no fault, no
mismatch.  python tools/guard_sweep.py"""
import sys, torch, logging, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
logging.disable(logging.WARNING)
import svt_speechbrain_amd as S
from svt_speechbrain_amd import weights as W
from svt_speechbrain_amd.config import PRESETS
import test_gpu_guard as G
DEV = torch.device("cuda:0")
r = random.Random(7)
cases = []
for name in ("wav2vec2-base", "wav2vec2-large-lv60", "wavlm-base", "hubert-base-ls960", "data2vec-audio-base"):
    for prec in ("bf16", "fp16x3"):
        shapes = [(r.choice([1, 2, 3, 4, 6, 8, 12, 16, 24, 33]), r.choice([3200, 8000, 16000, 23456, 40000, 47000, 80000, 100001])) for _ in range(6 if prec == "bf16" else 2)]
        cases.append((name, prec, shapes))
nfail = 0
for name, prec, shapes in cases:
    cfg = PRESETS[name]
    enc = S.HuggingFaceWav2Vec2(name, None, config=cfg, normalize_wav=True, precision=prec, seed=3).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size); head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=4)); head = head.to(DEV)
    for B, L in shapes:
        if B * L > 16 * 100001: B = max(1, 16 * 100001 // L)
        g = torch.Generator().manual_seed(B * 100 + L)
        wav = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)
        def run(to_dev):
            enc._dev.close(); head._dev.close()
            x = to_dev(wav)
            feats = enc(x); fused = enc.forward_head(x, head)
            per = enc(x, clips_per_norm_group=1) if L % 4 == 0 else feats
            return feats.cpu(), fused.cpu(), per.cpu()
        try:
            G.check_guarded(run, (name, prec, B, L))
            print("ok", name, prec, B, L, flush=True)
        except AssertionError as e:
            nfail += 1; print("MISMATCH", name, prec, B, L, e, flush=True)
    enc._dev.close(); head._dev.close()
print("sweep done, mismatches:", nfail)
