"""include/svt_mi355.h promises that a forward call neither synchronises the device nor allocates, "so calls may be captured into a
hipGraph" (no test captured one).  Here `svt_encoder_forward_head` -- encoder, whole-batch norms, frame head and
per-frame decode, ~130 launches and one memset for the one-utterance case -- is captured on a stream (torch.cuda.CUDAGraph drives
hipStreamBeginCapture / hipGraphLaunch) and replayed: logits and decoded frames bit-identical to the eager call, on new input written
into the captured buffer too."""
import pytest
import torch

import svt_speechbrain_amd as S
from svt_speechbrain_amd import weights as W
from svt_speechbrain_amd.config import PRESETS

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("cfg_name,precision,B,L", [("tiny-group", "fp32", 3, 16000), ("tiny-layer", "bf16", 2, 24000),
                                                   ("wav2vec2-base", "bf16", 1, 80000),      # BASELINE C1: one 5 s utterance
                                                   ("wav2vec2-base", "fp16x3", 2, 48000)])
def test_forward_head_captured_in_a_hip_graph_replays_bit_identically(cfg_name, precision, B, L):
    cfg = PRESETS[cfg_name]
    enc = S.HuggingFaceWav2Vec2(cfg_name, None, config=cfg, precision=precision, normalize_wav=True, seed=31).to(DEV)
    head = S.Linear(20, input_size=cfg.hidden_size)
    head.load_state_dict(W.seeded_head_state_dict(cfg.hidden_size, 20, seed=1031))
    head = head.to(DEV)
    T = cfg.frames(L)
    g = torch.Generator().manual_seed(9)
    wav_a = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(DEV)
    wav_b = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(DEV)
    frames_e = torch.empty((B, T, 4), dtype=torch.int32, device=DEV)
    fused = S.HuggingFaceWav2Vec2.can_fuse_head(head)        # hidden sizes 512 / 768 / 1024; the tiny presets take the three-call path

    def call(x, frames):
        if fused:
            return enc.forward_head(x, head, frames=frames)
        logits = head(enc(x))                                # svt_encoder_forward_ex + svt_linear_forward + svt_decode_frames
        frames.copy_(_decode(logits))
        return logits

    def _decode(logits):
        from svt_speechbrain_amd import _lib
        lib = enc._lib()
        out = torch.empty((B, T, 4), dtype=torch.int32, device=DEV)
        _lib.check(lib.svt_decode_frames(_lib.ptr(logits), B * T, 20, 4, 12, _lib.ptr(out), 0, _lib.stream_ptr(torch.device(DEV))), "svt_decode_frames")
        return out

    want_a = call(wav_a, frames_e).clone()      # eager (also: uploads, workspace, kernel attributes exist now)
    want_fa = frames_e.clone()
    want_b = call(wav_b, frames_e).clone()
    want_fb = frames_e.clone()
    torch.cuda.synchronize()

    static_in = wav_a.clone()
    frames_g = torch.empty((B, T, 4), dtype=torch.int32, device=DEV)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call(static_in, frames_g)                                        # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        logits_g = call(static_in, frames_g)
    logits_g.zero_(); frames_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(logits_g, want_a) and torch.equal(frames_g, want_fa)
    static_in.copy_(wav_b)
    graph.replay()
    graph.replay()                                                        # back to back: the workspace's tickets are re-armed by the graph itself
    torch.cuda.synchronize()
    assert torch.equal(logits_g, want_b) and torch.equal(frames_g, want_fb)
    # and the eager path still works on the same object afterwards
    assert torch.equal(call(wav_a, frames_e), want_a)
