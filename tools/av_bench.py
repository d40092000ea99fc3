"""BASELINE config C4 end to end on one GPU: 16 x (10 s audio @16 kHz + 500 lip-ROI frames of 88x88) ->
HuBERT-large audio features + AV-HuBERT-large video features -> RCA fusion -> 20-way head -> frame decode.
Prints per-stage and total times (synthetic inputs, seeded weights)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
B = a.batch
g = torch.Generator().manual_seed(1986)
wav = (0.1 * torch.randn(B, 160000, generator=g)).clamp_(-1, 1).to(dev)
video = torch.randn(B, 1, 500, 88, 88, generator=g).to(dev)
audio_enc = S.HuggingFaceWav2Vec2("hubert-large-ll60k", None, config=S.PRESETS["hubert-large-ll60k"], precision=a.precision).to(dev)
video_enc = S.FairseqAVHubertPretrain(config="avhubert-large-video", precision=a.precision).to(dev)
fusion = S.FusionRCA(precision=a.precision).to(dev)
head = S.Linear(20, input_size=1024).to(dev)
lib = _lib.load()


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n, r


def step():
    fa = audio_enc(wav)
    fv = video_enc({"video": video, "audio": None})
    fused = fusion(fa, fv)
    logits = head(fused)
    return S.decode_frames(logits)


for _ in range(2):
    step()
ta, fa = timed(lambda: audio_enc(wav), a.iters)
tf_, ff = timed(lambda: video_enc.model.feature_extractor_video(video), a.iters)
tv, fv = timed(lambda: video_enc({"video": video, "audio": None}), a.iters)
tr, fused = timed(lambda: fusion(fa, fv), a.iters)
th, _ = timed(lambda: S.decode_frames(head(fused)), a.iters)
tt, _ = timed(step, a.iters)
side = torch.cuda.Stream()


def step_overlapped():
    # the two encoders are independent: run them on two streams, join before the fusion
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        fv_ = video_enc({"video": video, "audio": None})
    fa_ = audio_enc(wav)
    cur.wait_stream(side)
    return S.decode_frames(head(fusion(fa_, fv_)))


for _ in range(2):
    step_overlapped()
to, _ = timed(step_overlapped, a.iters)
print(f"C4 ({a.precision}, B={B}): audio encoder (HuBERT-large) {ta*1e3:.2f} ms | video encoder {tv*1e3:.2f} ms "
      f"(lip front-end {tf_*1e3:.2f} ms) | RCA fusion {tr*1e3:.2f} ms | head+decode {th*1e3:.2f} ms | "
      f"end to end {tt*1e3:.2f} ms = {B/tt:.0f} clips/s; audio and video encoders on two streams {to*1e3:.2f} ms = {B/to:.0f} clips/s")
gf = B * (383.86e9 + 500 * 632e6 + 500 * (2 * 2048 * 1024 + 0) + 383.86e9 - 49.078e9 - 0.523e9 + 33.38e9)
print(f"  algorithmic work ~{gf/1e12:.1f} TFLOP per batch -> {gf/tt/1e12:.0f} TFLOP/s end to end")
