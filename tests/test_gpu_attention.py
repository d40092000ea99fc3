"""GPU unit tests of the fused attention kernel through the C-ABI test hook (svt_debug_attention) against a torch fp32
softmax(q k^T) v on the same bf16 inputs: the encoder shapes (T = 499 / 249, head_dim 64), the RCA shape (head_dim
128) and the edge lengths around the 64-key / 128-query tiles (1, 63, 64, 65, 127, 128, 129)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from svt_speechbrain_amd import _lib  # noqa: E402

DEV = "cuda:0"


def run_attention(B, T, H, dh, seed=0, gain=1.5):
    lib = _lib.load()
    D = H * dh
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B, T, 3 * D, generator=g) * gain).to(DEV, torch.bfloat16)
    out = torch.full((B, T, D), float("nan"), device=DEV, dtype=torch.bfloat16)
    scale = dh ** -0.5
    _lib.check(lib.svt_debug_attention(1, qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, out.data_ptr(),
                                       B, T, H, dh, 3 * D, 3 * D, D, scale, 0, torch.cuda.current_stream().cuda_stream),
               "svt_debug_attention")
    torch.cuda.synchronize()
    q, k, v = [x.float().cpu().view(B, T, H, dh).transpose(1, 2) for x in qkv.split(D, dim=-1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B, T, D)
    return out.float().cpu(), ref


@pytest.mark.parametrize("B,T,H,dh", [(2, 499, 12, 64), (1, 249, 12, 64), (2, 499, 16, 64), (2, 499, 8, 128), (1, 500, 8, 128),
                                      (3, 1, 2, 64), (2, 63, 2, 64), (2, 64, 2, 64), (2, 65, 3, 64), (1, 127, 2, 64),
                                      (1, 128, 2, 128), (1, 129, 2, 64), (1, 1000, 2, 64)])
def test_attention_vs_torch(B, T, H, dh):
    got, ref = run_attention(B, T, H, dh)
    assert torch.isfinite(got).all(), "unwritten (NaN-poisoned) outputs"
    err = (got - ref).abs().max().item()
    # P and the output are rounded to bf16 (2^-9 relative); |o| <= max|v| ~ 6
    assert err < 4e-2, (B, T, H, dh, err)
    assert (got - ref).abs().mean().item() < 2e-3


def test_attention_peaked_rows_do_not_overflow():
    # one dominant key per row (scores ~ +-60 after scaling): exercises the deferred-rescale path of the running max
    got, ref = run_attention(1, 499, 4, 64, seed=5, gain=8.0)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 0.25  # |v| ~ 30 here; bf16 output rounding alone is 0.12


def test_attention_rejects_other_head_dims():
    lib = _lib.load()
    x = torch.zeros(1, 8, 3 * 96, device=DEV, dtype=torch.bfloat16)
    o = torch.zeros(1, 8, 96, device=DEV, dtype=torch.bfloat16)
    rc = lib.svt_debug_attention(1, x.data_ptr(), x.data_ptr(), x.data_ptr(), o.data_ptr(), 1, 8, 2, 48, 288, 288, 96, 1.0, 0, None)
    assert rc != 0 and b"head_dim" in lib.svt_last_error()


@pytest.mark.parametrize("prec,tol", [(3, 2e-5), (2, 4e-4)])
@pytest.mark.parametrize("B,T,H,dh", [(2, 499, 12, 64), (1, 249, 3, 64), (2, 499, 8, 128), (2, 65, 2, 64), (1, 1, 2, 64), (3, 130, 2, 128)])
def test_split_operand_attention_vs_torch(prec, tol, B, T, H, dh):
    """The fused attention of the split-operand modes (fp32 q|k|v cut into 16-bit (hi, lo) planes, three MFMAs per product,
    fp32 output) against fp64 softmax(q k^T) v on the same fp32 inputs: fp16 pieces to ~1e-6, bf16 pieces to ~1e-4."""
    lib = _lib.load()
    D = H * dh
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B, T, 3 * D, generator=g) * 1.5).to(DEV)
    out = torch.full((B, T, D), float("nan"), device=DEV)
    scale = dh ** -0.5
    _lib.check(lib.svt_debug_attention(prec, qkv.data_ptr(), qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, out.data_ptr(), B, T, H, dh,
                                       3 * D, 3 * D, D, scale, 0, torch.cuda.current_stream().cuda_stream), "svt_debug_attention")
    torch.cuda.synchronize()
    q, k, v = [x.double().cpu().view(B, T, H, dh).transpose(1, 2) for x in qkv.split(D, dim=-1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B, T, D).float()
    got = out.cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    print(f"split attention prec={prec} B={B} T={T} H={H} dh={dh}: max|err| {err:.3e}")
    assert err < tol, err
