"""Where a wave of flash_attn_pipe_kernel spends its cycles (svt_debug_set key 21 = 99: the stamp build of the kernel writes 16 s_memtime
stamps per wave over the head of O).  Usage: python tools/attn_pipe_stamps.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B, T, H, dh = 32, 499, 12, 64
D = H * dh
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(B, T, 3 * D, generator=g) * 1.5).to(dev, torch.bfloat16)
out = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream


def call():
    _lib.check(lib.svt_debug_attention(1, qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, out.data_ptr(), B, T, H, dh, 3 * D, 3 * D, D,
                                       dh ** -0.5, 0, st), "svt_debug_attention")


lib.svt_debug_set(21, 99)
for _ in range(5):
    call()
torch.cuda.synchronize()
nw = B * H * ((T + 127) // 128) * 4
rec = out.view(-1).view(torch.int32)[: nw * 16].view(nw, 16).cpu().to(torch.int64) & 0xFFFFFFFF
rec = rec[(rec[:, 15] >> 16) == 0x5A5A]
names = ["entry .. head requests issued (Q loads, addresses)", "wait for K_0 + barrier", "S_0 (8 MFMAs alone)", "S_0 .. top of iteration 3",
         "iteration 3: counted vmcnt wait", "iteration 3: barrier", "iteration 3: (bookkeeping)", "iteration 3: row max if not taken in the region",
         "iteration 3: rescale test + interleaved region (requests, 16 MFMAs, softmax, next max)", "end of iteration 3 .. end of the loop", "vmcnt(0) + barrier + last P V"]
idx = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 11)]
print(f"{rec.shape[0]} waves, median core cycles per phase:")
for n, (a, b) in zip(names, idx):
    d = ((rec[:, b] - rec[:, a]) & 0xFFFFFFFF).double()
    print(f"  {n:60s} {d.median().item():8.0f}   (p10 {d.quantile(0.1).item():.0f}, p90 {d.quantile(0.9).item():.0f})")
tot = ((rec[:, 12] - rec[:, 0]) & 0xFFFFFFFF).double()
print(f"  whole wave {tot.median().item():.0f} cycles")
