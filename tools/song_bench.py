"""Song-level throughput: a 3-minute song (36 utterances of 5 s, batch 1 each, as the reference's eval) -> notes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svt_speechbrain_amd as S
dev = "cuda:0"
model = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2-base"
cfg = S.PRESETS[model]
enc = S.HuggingFaceWav2Vec2(model, None, config=cfg, precision="bf16").to(dev)
torch.manual_seed(0)
head = S.Linear(20, input_size=cfg.hidden_size).to(dev)
g = torch.Generator().manual_seed(0)
song = (0.1 * torch.randn(180 * 16000, generator=g)).clamp_(-1, 1).to(dev)
for ns, batched in ((1, False), (2, False), (2, True)):
    tr = S.SongTranscriber(enc, head, streams=ns, batch_utterances=batched)
    for _ in range(2):
        notes = tr.transcribe(song)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 5
    for _ in range(n):
        notes = tr.transcribe(song)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"{model}: 3-minute song, {'utterances as one batch (per-clip norms)' if batched else f'utterance by utterance on {ns} stream(s)'}: "
          f"{dt*1e3:.1f} ms per song = {180/dt:.0f}x real time ({len(notes)} notes)")
