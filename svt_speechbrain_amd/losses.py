"""Validation losses of the recipes on the GPU — same names, signatures and error behaviour as
``speechbrain.nnet.losses`` (``bce_loss`` :458-519, ``nll_loss`` :402-455, ``truncate`` :594-621,
``compute_masked_loss`` :624-684) and ``speechbrain.nnet.activations.Softmax`` (:14-75), forward only (the
accelerated path is inference / validation: ``compute_objectives`` of ``MIR_ST500/train_audio_ssl.py:50-84`` calls
them under ``torch.no_grad`` for ``Stage.VALID`` / ``Stage.TEST``).

Each call is one HIP kernel per batch (per-frame loss x length mask, fixed-order block reduction into double sums)
plus a one-thread reduction kernel; nothing is computed by torch.  There is no CPU fallback.
"""
from __future__ import annotations

import torch

from . import _lib

_REDUCTIONS = {"mean": 0, "batchmean": 1, "batch": 2, "none": 3}


def truncate(predictions: torch.Tensor, targets: torch.Tensor, allowed_len_diff: int = 3):
    """Same contract as ``speechbrain.nnet.losses.truncate``: equalise dim 1, ValueError beyond the tolerance."""
    len_diff = predictions.shape[1] - targets.shape[1]
    if len_diff == 0:
        return predictions, targets
    if abs(len_diff) > allowed_len_diff:
        raise ValueError("Predictions and targets should be same length, but got %s and %s respectively."
                         % (predictions.shape[1], targets.shape[1]))
    if len_diff < 0:
        return predictions, targets[:, : predictions.shape[1]]
    return predictions[:, : targets.shape[1]], targets


def _need_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise _lib.SvtError(f"{who} needs GPU tensors; there is no CPU fallback")


def _reduction(reduction: str) -> int:
    if reduction not in _REDUCTIONS:
        raise ValueError(f"reduction must be one of {sorted(_REDUCTIONS)}, got {reduction!r}")
    return _REDUCTIONS[reduction]


def _length(length, batch, device):
    if length is None:
        return None
    length = torch.as_tensor(length, dtype=torch.float32, device=device).contiguous()
    assert len(length.shape) == 1  # same assertion as length_to_mask
    if length.shape[0] != batch:
        raise ValueError(f"length has {length.shape[0]} entries for a batch of {batch}")
    return length


def _out(red: int, batch: int, frames: int, device):
    if red == 3:
        return torch.empty((batch, frames), dtype=torch.float32, device=device)
    if red == 2:
        return torch.empty((batch,), dtype=torch.float32, device=device)
    return torch.empty((), dtype=torch.float32, device=device)


def bce_loss(inputs, targets, length=None, weight=None, pos_weight=None, reduction="mean", allowed_len_diff=3,
             label_smoothing=0.0):
    """Binary cross-entropy with logits over (batch[, frames]) with the relative-length mask."""
    _need_gpu(inputs, "bce_loss")
    if weight is not None:
        raise NotImplementedError("bce_loss: the elementwise `weight` is not used by the recipes and is not provided")
    if label_smoothing != 0.0:
        raise ValueError("bce_loss: label_smoothing should only be used for the NLL loss")
    if len(inputs.shape) == len(targets.shape) + 1:
        inputs = inputs.squeeze(-1)
    if len(inputs.shape) >= 2:
        if abs(inputs.shape[1] - targets.shape[1]) > allowed_len_diff:
            truncate(inputs, targets, allowed_len_diff)  # raises the reference's ValueError
    elif length is not None:
        raise ValueError("length can be passed only for >= 2D inputs.")
    if len(inputs.shape) > 2:
        raise NotImplementedError("bce_loss: inputs beyond (batch, frames) are not used by the recipes")
    red = _reduction(reduction)
    lib = _lib.load()
    dev = inputs.device
    x = inputs.detach().to(torch.float32)
    y = targets.detach().to(device=dev, dtype=torch.float32)
    if x.dim() == 1:
        x, y = x.unsqueeze(1), y.unsqueeze(1)
    x, y = x.contiguous(), y.contiguous()
    B, tp, tt = x.shape[0], x.shape[1], y.shape[1]
    T = min(tp, tt)
    ln = _length(length, B, dev)
    pw = None
    if pos_weight is not None:
        pw = torch.as_tensor(pos_weight, dtype=torch.float32, device=dev).reshape(-1).contiguous()
        if pw.numel() != 1:
            raise NotImplementedError("bce_loss: pos_weight must hold one value (one class per call in the recipes)")
    out = _out(red, B, T, dev)
    ws = torch.empty(B * 24 + 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.svt_bce_loss(_lib.ptr(x), B, tp, _lib.ptr(y), tt, _lib.ptr(ln) if ln is not None else None,
                                _lib.ptr(pw) if pw is not None else None, int(allowed_len_diff), red, _lib.ptr(out),
                                _lib.ptr(ws), ws.numel(), _lib.dev_index(dev), _lib.stream_ptr(dev)), "svt_bce_loss")
    if red == 3 and inputs.dim() == 1:
        out = out.squeeze(1)
    return out


def nll_loss(log_probabilities, targets, length=None, label_smoothing=0.0, allowed_len_diff=3, reduction="mean"):
    """Negative log-likelihood over (batch, frames, classes) log-probabilities (or (batch, classes))."""
    _need_gpu(log_probabilities, "nll_loss")
    red = _reduction(reduction)
    lib = _lib.load()
    dev = log_probabilities.device
    lp = log_probabilities.detach().to(torch.float32)
    tg = targets.detach().to(device=dev, dtype=torch.int64)
    squeeze = False
    if lp.dim() == 2:
        lp, tg = lp.unsqueeze(1), tg.unsqueeze(1)
        squeeze = True
    elif lp.dim() == 3:
        if abs(lp.shape[1] - tg.shape[1]) > allowed_len_diff:
            truncate(lp, tg, allowed_len_diff)  # raises the reference's ValueError
    else:
        raise NotImplementedError("nll_loss: log_probabilities must be (batch, classes) or (batch, frames, classes)")
    lp, tg = lp.contiguous(), tg.contiguous()
    B, tp, C, tt = lp.shape[0], lp.shape[1], lp.shape[2], tg.shape[1]
    T = min(tp, tt)
    ln = _length(length, B, dev)
    out = _out(red, B, T, dev)
    ws = torch.empty(B * 24 + 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.svt_nll_loss(_lib.ptr(lp), B, tp, C, _lib.ptr(tg), tt, _lib.ptr(ln) if ln is not None else None,
                                float(label_smoothing), int(allowed_len_diff), red, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                _lib.dev_index(dev), _lib.stream_ptr(dev)), "svt_nll_loss")
    if int(ws[B * 24: B * 24 + 4].view(torch.int32).item()) != 0:
        raise IndexError(f"nll_loss: target out of bounds for {C} classes")  # torch raises IndexError on CPU
    if red == 3 and squeeze:
        out = out.squeeze(1)
    return out


class Softmax(torch.nn.Module):
    """``speechbrain.nnet.activations.Softmax``: (log-)softmax over the last axis of a 2-d / 3-d / 4-d tensor."""

    def __init__(self, apply_log: bool = False, dim: int = -1):
        super().__init__()
        self.apply_log = bool(apply_log)
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _need_gpu(x, "Softmax")
        if self.dim not in (-1, x.dim() - 1):
            raise NotImplementedError("Softmax: only the last axis (the recipes' use) is provided")
        lib = _lib.load()
        xx = x.detach().to(torch.float32).contiguous()
        n = xx.shape[-1]
        y = torch.empty_like(xx)
        rows = xx.numel() // max(1, n)
        _lib.check(lib.svt_softmax(_lib.ptr(xx), rows, n, int(self.apply_log), _lib.ptr(y), _lib.dev_index(xx.device),
                                   _lib.stream_ptr(xx.device)), "svt_softmax")
        return y
