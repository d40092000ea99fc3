"""Drop-in for the default chain of ``speechbrain.lobes.features.Fbank`` (reference
``speechbrain/lobes/features.py:18-143``: STFT -> power -> 40 triangular mel filters -> dB -> top_db clip).
Named by the north star although no AMT recipe calls it.  On MI355X the STFT is a
dense fp32 contraction with a (2*208, 400) DFT basis (``v_mfma_f32_16x16x4_f32``), the mel projection a
second one; ``deltas=True`` appends first and second time derivatives (``Deltas``, features.py:788-850) and
``context=True`` gathers ``left_frames`` / ``right_frames`` neighbours per frame (``ContextWindow``, :853-940)."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib


class Fbank(nn.Module):
    def __init__(self, deltas=False, context=False, requires_grad=False, sample_rate=16000, f_min=0, f_max=None,
                 n_fft=400, n_mels=40, filter_shape="triangular", param_change_factor=1.0, param_rand_factor=0.0,
                 left_frames=5, right_frames=5, win_length=25, hop_length=10):
        super().__init__()
        self.deltas, self.context = bool(deltas), bool(context)
        self.left_frames, self.right_frames = int(left_frames), int(right_frames)
        if filter_shape != "triangular" or requires_grad:
            raise NotImplementedError("only frozen triangular filters are built")
        self.sample_rate = sample_rate
        self.f_min = float(f_min)
        self.f_max = float(sample_rate / 2 if f_max is None else f_max)
        self.n_fft, self.n_mels = n_fft, n_mels
        self.win = int(round(sample_rate / 1000.0 * win_length))
        self.hop = int(round(sample_rate / 1000.0 * hop_length))
        self.top_db = 80.0
        self._ws = None

    def forward(self, wav: torch.Tensor) -> torch.Tensor:
        if not wav.is_cuda:
            raise _lib.SvtError("Fbank needs its input on the GPU; there is no CPU fallback")
        lib = _lib.load()
        x = wav.detach().to(torch.float32).contiguous()
        B, L = x.shape
        nf = 1 + L // self.hop
        need = lib.svt_fbank_workspace_bytes(B, L, self.n_fft, self.hop, self.n_mels)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        out = torch.empty((B, nf, self.n_mels), dtype=torch.float32, device=x.device)
        _lib.check(lib.svt_fbank(_lib.ptr(x), B, L, self.sample_rate, self.n_fft, self.win, self.hop, self.n_mels,
                                 self.f_min, self.f_max, self.top_db, _lib.ptr(out), _lib.ptr(self._ws),
                                 self._ws.numel(), _lib.dev_index(x.device), _lib.stream_ptr(x.device)), "svt_fbank")
        if self.deltas:   # [fbank, delta, delta-delta] side by side (lobes/features.py:137-140)
            C0 = self.n_mels
            cat = torch.empty((B, nf, 3 * C0), dtype=torch.float32, device=x.device)
            cat[:, :, :C0] = out
            for k in (1, 2):
                _lib.check(lib.svt_deltas(_lib.ptr(cat) + 4 * (k - 1) * C0, 3 * C0, B, nf, C0, 5, _lib.ptr(cat) + 4 * k * C0, 3 * C0,
                                          _lib.dev_index(x.device), _lib.stream_ptr(x.device)), "svt_deltas")
            out = cat
        if self.context:
            C1 = out.shape[-1]
            ctx = self.left_frames + self.right_frames + 1
            cw = torch.empty((B, nf, C1 * ctx), dtype=torch.float32, device=x.device)
            _lib.check(lib.svt_context_window(_lib.ptr(out), B, nf, C1, self.left_frames, self.right_frames, _lib.ptr(cw),
                                              _lib.dev_index(x.device), _lib.stream_ptr(x.device)), "svt_context_window")
            out = cw
        return out
