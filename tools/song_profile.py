"""Where the time of SongTranscriber.transcribe goes (3-minute song, wav2vec2-base, bf16): batched forward, last utterance,
head, frame decode + D2H, note assembly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svt_speechbrain_amd as S
from svt_speechbrain_amd.decode import decode_frames, frames2note
dev="cuda:0"
cfg=S.PRESETS["wav2vec2-base"]
enc=S.HuggingFaceWav2Vec2("wav2vec2-base", None, config=cfg, precision="bf16").to(dev)
torch.manual_seed(0)
head=S.Linear(20, input_size=cfg.hidden_size).to(dev)
g=torch.Generator().manual_seed(0)
song=(0.1*torch.randn(180*16000, generator=g)).clamp_(-1,1).to(dev)
def T(f, n=20):
    for _ in range(3): r=f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3, r
w35=song[:35*80000].view(35,-1)
ms, feats = T(lambda: enc(w35, clips_per_norm_group=1)); print("enc B=35 per-clip norm", ms)
ms, _ = T(lambda: enc(w35)); print("enc B=35 whole-batch norm", ms)
ms, f1 = T(lambda: enc(song[35*80000:].unsqueeze(0))); print("enc last utterance", ms)
ms, lg = T(lambda: head(feats)); print("head", ms)
lg2 = lg.reshape(-1, 20)
ms, fr = T(lambda: decode_frames(lg2, 4, 12)); print("decode_frames (+D2H)", ms)
t=time.perf_counter()
for _ in range(20): notes = frames2note(fr, 0.4, 0.5, 1/49.8)
print("frames2note", (time.perf_counter()-t)/20*1e3, len(notes))
tr=S.SongTranscriber(enc, head)
ms,_=T(lambda: tr.transcribe(song)); print("transcribe", ms)
