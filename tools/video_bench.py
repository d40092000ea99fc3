"""Timing of the lip front-end (BASELINE config C4 shape: B=16 clips x 500 frames of 88x88)."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svt_speechbrain_amd.video import SubModel
from svt_speechbrain_amd import _lib
import ctypes as C

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--frames", type=int, default=500)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--debug", default="", help="svt_debug_set pairs key=value,key=value")
a = ap.parse_args()
dev = "cuda:0"
for kv in filter(None, a.debug.split(",")):
    _lib.load("f16" if a.precision == "fp16" else "").svt_debug_set(*map(int, kv.split("=")))
m = SubModel(512, 1024, "prelu", precision=a.precision).to(dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(a.batch, 1, a.frames, 88, 88, generator=g).to(dev)
for _ in range(2):
    y = m(x)
torch.cuda.synchronize()
n = 5
t = time.perf_counter()
for _ in range(n):
    y = m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
gf = 632e6 * a.batch * a.frames
print(f"lip front-end {a.precision}: {dt*1e3:.2f} ms per batch of {a.batch} x {a.frames} frames -> {a.batch/dt:.0f} clips/s, "
      f"{a.batch*a.frames/dt:.0f} frames/s, {gf/dt/1e12:.0f} TFLOP/s (632 MFLOP per frame)")
# round 6: the recipe's RAW input -- (B, T, 96, 96) uint8 ROI, transform_eval inside the padding kernel -- against the float path above,
# and against the float path WITH the host-side work it replaces left out (the float tensor is already on the device here)
roi = torch.randint(0, 256, (a.batch, a.frames, 96, 96), generator=g, dtype=torch.uint8).to(dev)
for _ in range(2):
    y8 = m.forward_into(roi)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    y8 = m.forward_into(roi)
torch.cuda.synchronize()
dt8 = (time.perf_counter() - t) / n
print(f"lip front-end {a.precision}, uint8 96 x 96 ROI in (crop 88 + normalisation in the padding kernel, {roi.numel() / 1e6:.0f} MB instead of "
      f"{x.numel() * 4 / 1e6:.0f} MB of input): {dt8*1e3:.2f} ms per batch -> {a.batch/dt8:.0f} clips/s")
