"""Race / stability soak: the same forward N times (alternating two streams over two replicas), every output compared with the
first one.  The whole-batch norms accumulate with fp64 atomics, so the last bits may move (<= ~1e-5); anything larger is a
race (a missing barrier or wait in a kernel shows up as a rare large difference)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svt_speechbrain_amd as S

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="wav2vec2-base")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--video", action="store_true", help="soak the lip front-end (16 x --frames lip ROIs of 88 x 88) instead of an audio encoder: two "
                "front-end objects on two streams, every output compared bit for bit (no atomics in that path)")
ap.add_argument("--frames", type=int, default=500)
ap.add_argument("--inputs", type=int, default=1, help="--video: this many different inputs in turn (the last one 100 frames shorter)")
a = ap.parse_args()
if a.video:
    from svt_speechbrain_amd.video import SubModel
    dev = "cuda:0"
    B = min(a.batch, 16)
    ms = [SubModel(512, 1024, "prelu", precision=a.precision, seed=4).to(dev) for _ in range(2)]
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    g = torch.Generator().manual_seed(10)
    # --inputs N: N different inputs in turn (the last one 100 frames shorter: the kept zero halos of the workspace are re-made when the
    # geometry changes) -- a forward then finds ANOTHER input's tensors in the workspace, which one repeated input can never show
    xs = [torch.randn(B, 1, a.frames - (100 if (j == a.inputs - 1 and a.inputs > 1 and a.frames > 150) else 0), 88, 88, generator=g).to(dev)
          for j in range(a.inputs)]
    refs = [ms[0](x).clone() for x in xs]
    torch.cuda.synchronize()
    streams[1].wait_stream(streams[0])
    bad, outs, t0 = 0, [], time.time()
    for i in range(a.iters):
        k = i % 2
        j = (i // 2 + i) % a.inputs
        with torch.cuda.stream(streams[k]):
            outs.append((ms[k](xs[j]) != refs[j]).any())
        if len(outs) == 20 or i + 1 == a.iters:
            torch.cuda.synchronize()
            bad += sum(int(o.item()) for o in outs)
            outs = []
    print(f"lip front-end B={B} x {a.frames} frames {a.precision}: {a.iters} forwards of {a.inputs} input(s) in {time.time() - t0:.1f} s on two streams, "
          f"{bad} forwards differ from the first")
    sys.exit(1 if bad else 0)
dev = "cuda:0"
cfg = S.PRESETS[a.model]
enc = S.HuggingFaceWav2Vec2(a.model, None, config=cfg, precision=a.precision, seed=3).to(dev)
encs = [enc, enc.replica()]
streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
g = torch.Generator().manual_seed(9)
wav = (0.1 * torch.randn(a.batch, int(16000 * a.seconds), generator=g)).clamp_(-1, 1).to(dev)
ref = enc(wav).clone()
torch.cuda.synchronize()
worst, bad = 0.0, 0
t0 = time.time()
outs = []
for i in range(a.iters):
    k = i % 2
    with torch.cuda.stream(streams[k]):
        y = encs[k](wav)
        d = (y - ref).abs().max()
    outs.append(d)
    if len(outs) == 20 or i + 1 == a.iters:
        torch.cuda.synchronize()
        for d in outs:
            v = d.item()
            worst = max(worst, v)
            bad += v > 1e-4
        outs = []
print(f"{a.model} B={a.batch} {a.seconds:g}s {a.precision}: {a.iters} forwards in {time.time() - t0:.1f} s, max |y - y0| = {worst:.3e}, "
      f"{bad} forwards above 1e-4")
sys.exit(1 if bad else 0)
