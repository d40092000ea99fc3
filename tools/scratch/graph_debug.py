import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import svt_speechbrain_amd as S
from svt_speechbrain_amd import weights as W
from svt_speechbrain_amd.config import PRESETS
DEV = "cuda:0"
for cfg_name, prec, B, L, nw in [("tiny-group", "fp32", 3, 16000, True), ("tiny-group", "fp32", 3, 16000, False), ("tiny-group", "bf16", 3, 16000, True),
                                 ("tiny-layer", "fp32", 2, 24000, True), ("wav2vec2-base", "bf16", 1, 80000, True)]:
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, precision=prec, normalize_wav=nw, seed=31).to(DEV)
    g = torch.Generator().manual_seed(9)
    a = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(DEV)
    b = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(DEV)
    wa, wb = enc(a).clone(), enc(b).clone()
    st = a.clone()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enc(st)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        out = enc(st)
    res = []
    for name, src, want in [("a", a, wa), ("b", b, wb), ("b", b, wb), ("a", a, wa)]:
        st.copy_(src); out.zero_(); graph.replay(); torch.cuda.synchronize()
        res.append((name, float((out - want).abs().max()), float(out.abs().max())))
    print(cfg_name, prec, "normalize_wav", nw, res, flush=True)
