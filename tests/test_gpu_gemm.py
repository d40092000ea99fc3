"""GPU unit tests of the dense contraction kernels through the C-ABI test hook (svt_debug_gemm): every dispatch path
(persistent LDS-DMA, one-tile LDS-DMA, register-staged bf16 / exact-fp32) against a torch fp32 reference of the same
op, including implicit-conv row addressing, M / N / K tails, bias, GELU / ReLU and the fp32 residual epilogue."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from svt_speechbrain_amd import _lib  # noqa: E402

DEV = "cuda:0"


def run_gemm(prec, M, N, K, conv=None, act=0, out_f32=0, resid=False, bias=True, seed=0):
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    dt = torch.bfloat16 if prec == 1 else torch.float32  # prec 2 / 3 (bf16x3 / fp16x3): fp32 operands in memory
    if conv:
        T_in, T_out, st, cin = conv
        B = M // T_out
        A = (torch.rand(B, T_in, cin, generator=g) * 2 - 1).to(DEV, dt)
        rpb, bstr, rstr = T_out, T_in * cin, st * cin
        k = K // cin
        idx = (torch.arange(T_out) * st)[:, None] + torch.arange(k)[None, :]
        A_rows = A.cpu().float()[:, idx].reshape(M, K)
    else:
        A = (torch.rand(M, K, generator=g) * 2 - 1).to(DEV, dt)
        rpb, bstr, rstr = M, 0, K
        A_rows = A.cpu().float()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(DEV, dt)
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    R = torch.randn(M, N, generator=g).to(DEV) if resid else None
    C = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32 if (out_f32 or prec != 1) else dt)
    _lib.check(lib.svt_debug_gemm(prec, A.data_ptr(), W.data_ptr(), C.data_ptr(), b.data_ptr() if bias else None,
                                  R.data_ptr() if resid else None, M, N, K, rpb, bstr, rstr, K, act, out_f32, 0,
                                  torch.cuda.current_stream().cuda_stream), "svt_debug_gemm")
    torch.cuda.synchronize()
    ref = (A_rows.double() @ W.cpu().double().t()).float() if prec >= 2 else A_rows @ W.cpu().float().t()
    if bias:
        ref = ref + b.cpu()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.relu(ref)
    if resid:
        ref = ref + R.cpu()
    return C.cpu().float(), ref


CASES = [
    # name, prec, M, N, K, conv, act, out_f32, resid
    ("pers_bf16_gelu", 1, 70000, 512, 1536, None, 1, 0, False),        # >= 512 tiles -> persistent kernel, M tail
    ("pers_conv", 1, 8 * 15999, 512, 1536, (31999, 15999, 2, 512), 1, 0, False),
    ("pers_f32out", 1, 66000, 768, 768, None, 0, 1, False),
    ("uring_f32out_resid", 1, 15968, 768, 768, None, 0, 1, True),      # single round -> one-tile kernel
    ("uring_bf16", 1, 4999, 2304, 768, None, 0, 0, False),
    ("uring_relu", 1, 998, 3072, 1024, None, 2, 0, False),
    ("v1_bf16_small_k", 1, 300, 256, 96, None, 1, 0, False),           # K % 64 != 0 -> register-staged kernel
    ("v1_bf16_narrow", 1, 499, 48, 6144, None, 1, 1, True),            # grouped pos-conv shape (N = 48)
    ("v1_bf16_ntail", 1, 777, 200, 128, None, 0, 0, False),
    ("fp32_exact", 0, 1000, 768, 512, None, 1, 0, True),
    ("fp32_conv", 0, 2 * 999, 512, 1024, (1999, 999, 2, 512), 1, 0, False),
    ("fp32_tiny", 0, 24, 64, 32, None, 0, 0, False),
    ("fp32_head", 0, 499, 20, 768, None, 0, 0, False),
    # split-operand engine (fp32 operands cut into 16-bit (hi, lo) pieces inside the kernel, three MFMAs per block)
    ("bf16x3", 2, 1000, 768, 512, None, 1, 0, True),
    ("bf16x3_conv", 2, 2 * 999, 512, 1024, (1999, 999, 2, 512), 1, 0, False),
    ("bf16x3_tiny", 2, 24, 64, 32, None, 0, 0, False),
    ("bf16x3_narrow_ktail", 2, 499, 20, 772, None, 0, 0, False),
    ("fp16x3", 3, 1000, 768, 512, None, 1, 0, True),
    ("fp16x3_conv", 3, 2 * 999, 512, 1024, (1999, 999, 2, 512), 1, 0, False),
    ("fp16x3_ntail", 3, 777, 200, 3072, None, 2, 0, False),
    ("fp16x3_narrow_ktail", 3, 499, 20, 772, None, 0, 0, False),
    # ... on the LDS-DMA split kernel (256 x 256 tiles, weight pieces cut once): M / N tails, conv rows, every epilogue
    ("x3dma_bf16_conv", 2, 8 * 1999, 512, 1536, (3999, 1999, 2, 512), 1, 0, False),
    ("x3dma_fp16_qkv", 3, 15968, 2304, 768, None, 0, 0, False),
    ("x3dma_fp16_mtail_ntail", 3, 777, 200, 3072, None, 2, 0, True),
    ("x3dma_bf16_resid_k32", 2, 1000, 768, 32, None, 0, 0, True),
    ("x3dma_fp16_one_tile", 3, 130, 136, 64, None, 1, 0, False),
]


@pytest.mark.parametrize("name,prec,M,N,K,conv,act,out_f32,resid", CASES, ids=[c[0] for c in CASES])
def test_gemm_vs_torch(name, prec, M, N, K, conv, act, out_f32, resid):
    got, ref = run_gemm(prec, M, N, K, conv, act, out_f32, resid)
    assert torch.isfinite(got).all(), "unwritten (NaN-poisoned) outputs"
    err = (got - ref).abs().max().item()
    # bf16 output rounding dominates in bf16 mode (values O(1..3)); fp32 path is an exact fp32 fma chain
    # split-operand products: ~2^-17 (bf16 pieces) / ~2^-22 (fp16 pieces) relative operand error, against an fp64 reference
    tol = {0: 1e-4, 1: 3e-2 if not out_f32 else 2e-4, 2: 3e-5, 3: 4e-6}[prec]
    print(f"{name}: max|err| {err:.3e} (tol {tol:g})")
    assert err < tol, (name, err)


PPS_CASES = [
    # name, M, N, K, conv, act, bias   -- gemm_pps_kernel: bf16 output, no residual, N % 256 == 0, K % 64 == 0, K >= 128
    ("ffn1_gelu_3_tiles_per_cu", 15968, 3072, 768, None, 1, True),
    ("qkv_no_act", 15968, 2304, 768, None, 0, True),
    ("conv_gelu_mtail", 8 * 1999, 512, 1536, (3999, 1999, 2, 512), 1, True),
    ("conv_k1024", 4 * 999, 512, 1024, (1999, 999, 2, 512), 1, True),
    ("no_bias_two_slabs", 70000, 512, 128, None, 0, False),
    ("one_tile_ragged", 845, 512, 256, None, 0, True),
    ("many_tiles_per_cu", 70000, 512, 256, None, 1, True),
    ("long_k_single_round", 15968, 768, 3072, None, 0, True),
]


@pytest.mark.parametrize("bm", [0, 256, 192, 128])
@pytest.mark.parametrize("name,M,N,K,conv,act,bias", PPS_CASES, ids=[c[0] for c in PPS_CASES])
def test_persistent_staggered_kernel(bm, name, M, N, K, conv, act, bias):
    """gemm_pps_kernel forced for every eligible launch (svt_debug_set key 3 = 70: write-through stores, the form the dispatch uses;
    key 1 = tile height, 0 = the dispatch's choice) against the torch reference: several tiles per workgroup (the ring and the
    source offsets carry over tile boundaries), M tails (rows >= M dropped by the buffer range check), conv rows, bias fetched
    inside the stream, GELU."""
    lib = _lib.load()
    lib.svt_debug_set(3, 70)
    lib.svt_debug_set(1, bm)
    try:
        got, ref = run_gemm(1, M, N, K, conv, act, 0, False, bias=bias)
    finally:
        lib.svt_debug_set(3, 0)
        lib.svt_debug_set(1, 0)
    assert torch.isfinite(got).all(), "unwritten (NaN-poisoned) outputs"
    err = ((got - ref).abs() / (1.0 + ref.abs())).max().item()
    assert err < 8e-3, (name, bm, err)   # bf16 rounding of the stored value: 2^-9 relative


@pytest.mark.parametrize("variant", [50, 70])
def test_persistent_kernel_exact_integers_under_load(variant):
    """Regression for a store-data hazard met on gfx950: with an SGPR in the soffset field of buffer_store_dwordx4 hipcc adds no
    wait state before the next VALU write of the data registers, and while the persistent stream's LDS-DMA kept the memory
    pipeline busy the first dword of a store was torn in lanes 12-15 of every 16.  Exact small-integer data, several tiles per
    workgroup, every output element compared."""
    lib = _lib.load()
    M, N, K = 256 * 96, 2048, 768
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-2, 3, (M, K), generator=g).to(DEV, torch.bfloat16)
    W = torch.randint(-2, 3, (N, K), generator=g).to(DEV, torch.bfloat16)
    C = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    lib.svt_debug_set(3, variant)
    try:
        _lib.check(lib.svt_debug_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, M, N, K, M, 0, K, K, 0, 0, 0,
                                      torch.cuda.current_stream().cuda_stream), "svt_debug_gemm")
        torch.cuda.synchronize()
    finally:
        lib.svt_debug_set(3, 0)
    ref = (A.float() @ W.float().t()).to(torch.bfloat16)
    assert torch.equal(C, ref)


X3P_CASES = [
    # name, prec, M, N, K, conv, act, bias  -- gemm_x3p_kernel: N % 256 == 0, K % 32 == 0, K >= 64, no residual, act none / GELU
    ("fp16_conv_gelu_mtail", 3, 8 * 1999, 512, 1536, (3999, 1999, 2, 512), 1, True),
    ("bf16_conv_k1024", 2, 4 * 999, 512, 1024, (1999, 999, 2, 512), 1, True),
    ("fp16_qkv_3_tiles_per_cu", 3, 15968, 2304, 768, None, 0, True),
    ("fp16_ffn1_gelu", 3, 15968, 3072, 768, None, 1, True),
    ("bf16_two_slabs_no_bias", 2, 70000, 512, 64, None, 0, False),
    ("fp16_one_tile_ragged", 3, 845, 512, 256, None, 0, True),
    ("fp16_n768_single_round", 3, 15968, 768, 3072, None, 0, True),
]


@pytest.mark.parametrize("variant", [0, 34])
@pytest.mark.parametrize("name,prec,M,N,K,conv,act,bias", X3P_CASES, ids=[c[0] for c in X3P_CASES])
def test_split_operand_persistent_kernel(variant, name, prec, M, N, K, conv, act, bias):
    """gemm_x3p_kernel (split-operand products as a persistent staggered stream; svt_debug_set key 3 = 34 forces it wherever it is
    eligible, 0 = the dispatch's choice) against an fp64 reference: tile boundaries inside a workgroup's stream, M tails, conv rows,
    GELU, bias fetched in the epilogue."""
    lib = _lib.load()
    lib.svt_debug_set(3, variant)
    try:
        got, ref = run_gemm(prec, M, N, K, conv, act, 0, False, bias=bias)
    finally:
        lib.svt_debug_set(3, 0)
    assert torch.isfinite(got).all(), "unwritten (NaN-poisoned) outputs"
    err = (got - ref).abs().max().item()
    assert err < {2: 3e-5, 3: 4e-6}[prec] * max(1.0, (K / 768) ** 0.5), (name, variant, err)


def run_gemm_pairs(prec, M, N, K, conv, act, out_kind, bias=True, seed=0, want_ref=True):
    """gemm_x3q_kernel through svt_debug_gemm_pairs: the fp32 A is converted to pair rows inside the hook, the result comes back as fp32."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    if conv:
        T_in, T_out, st, cin = conv
        B = M // T_out
        A = (torch.rand(B, T_in, cin, generator=g) * 2 - 1).to(DEV)
        rpb, bstr, rstr = T_out, T_in * cin, st * cin
        k = K // cin
        idx = (torch.arange(T_out) * st)[:, None] + torch.arange(k)[None, :]
        A_rows = A.cpu()[:, idx].reshape(M, K)
    else:
        A = (torch.rand(M, K, generator=g) * 2 - 1).to(DEV)
        rpb, bstr, rstr = M, 0, K
        A_rows = A.cpu()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    C = torch.full((M, N), float("nan"), device=DEV)
    _lib.check(lib.svt_debug_gemm_pairs(prec, A.data_ptr(), A.numel(), W.data_ptr(), C.data_ptr(), b.data_ptr() if bias else None, M, N, K,
                                        rpb, bstr, rstr, act, out_kind, 0, torch.cuda.current_stream().cuda_stream, 0, None), "svt_debug_gemm_pairs")
    torch.cuda.synchronize()
    if not want_ref:
        return C.cpu(), None
    ref = A_rows.double() @ W.cpu().double().t()
    if bias:
        ref = ref + b.cpu().double()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    return C.cpu(), ref.float()


X3Q_CASES = [
    # name, prec, M, N, K, conv, act, bias -- gemm_x3q_kernel: pair-row operands, N % 256 == 0, K % 32 == 0, K >= 64, M >= 128
    ("fp16_conv_gelu_mtail", 3, 8 * 1999, 512, 1536, (3999, 1999, 2, 512), 1, True),
    ("bf16_conv_k1024", 2, 4 * 999, 512, 1024, (1999, 999, 2, 512), 1, True),
    ("fp16_qkv_3_tiles_per_cu", 3, 15968, 2304, 768, None, 0, True),
    ("fp16_ffn1_gelu", 3, 15968, 3072, 768, None, 1, True),
    ("bf16_two_slabs_no_bias", 2, 70000, 512, 64, None, 0, False),
    ("fp16_one_tile_ragged", 3, 864, 512, 256, None, 0, True),
    ("fp16_n768_single_round", 3, 15968, 768, 3072, None, 0, True),
    ("fp16_three_slabs", 3, 4096, 256, 96, None, 0, True),
]


@pytest.mark.parametrize("bm", [0, 256, 192, 128])
@pytest.mark.parametrize("out_kind", [0, 1, 2])
@pytest.mark.parametrize("name,prec,M,N,K,conv,act,bias", X3Q_CASES, ids=[c[0] for c in X3Q_CASES])
def test_pair_row_split_kernel(bm, out_kind, name, prec, M, N, K, conv, act, bias):
    """gemm_x3q_kernel (both operands pre-cut into pair rows; the schedule of gemm_pps_kernel with hi*hi + hi*lo + lo*hi per block)
    against an fp64 reference, for its three outputs (fp32 rows / pair rows / separate planes: the last two read back as hi + lo,
    which carries 22 of fp32's 24 mantissa bits), every tile height, tile boundaries inside a workgroup's stream, M tails, conv
    rows, GELU, bias as the accumulators' initial value."""
    if out_kind == 2 and act:
        pytest.skip("the plane output (QKV projection) has no activation")
    lib = _lib.load()
    lib.svt_debug_set(1, bm)
    try:
        got, ref = run_gemm_pairs(prec, M, N, K, conv, act, out_kind, bias=bias)
    finally:
        lib.svt_debug_set(1, 0)
    assert torch.isfinite(got).all(), "unwritten (NaN-poisoned) outputs"
    err = (got - ref).abs().max().item()
    # (the bias is the accumulators' initial value: the fp32 partial sums round at the magnitude of bias + sum, not of the sum alone)
    tol = {2: 3e-5, 3: 1e-5}[prec] * max(1.0, (K / 768) ** 0.5)
    if out_kind:   # the stored value itself is cut: 2^-16 (bf16 pieces) / 2^-22 (fp16 pieces) relative
        tol += {2: 2.0 ** -15, 3: 2.0 ** -21}[prec] * ref.abs().max().item()
    assert err < tol, (name, bm, out_kind, err, tol)


@pytest.mark.parametrize("bm", [256, 192, 128])
@pytest.mark.parametrize("out_kind", [0, 1, 2])
@pytest.mark.parametrize("name,prec,M,N,K,conv,act,bias", [c for c in X3Q_CASES if c[4] >= 96], ids=[c[0] for c in X3Q_CASES if c[4] >= 96])
def test_pair_row_one_wave_kernel_equals_two_wave_kernel(bm, out_kind, name, prec, M, N, K, conv, act, bias):
    """gemm_p1x_kernel (round 5: one wave per SIMD, three phases per slab, K >= 96; svt_debug_set key 30 = 1) against gemm_x3q_kernel (the
    dispatched kernel, which the test above holds to the fp64 reference): the same MFMAs in the same order per accumulator -- lo x hi,
    hi x hi, hi x lo from the bias -- so the outputs are bit-identical, for every tile height, output form, M tail, conv rows and GELU."""
    if out_kind == 2 and act:
        pytest.skip("the plane output (QKV projection) has no activation")
    lib = _lib.load()
    if lib.svt_debug_set(30, 1) != 0:
        pytest.skip("gemm_p1x_kernel is an A/B arm: built by `make DIAG=1` only (the shipped library holds what it dispatches)")
    lib.svt_debug_set(1, bm)
    try:
        lib.svt_debug_set(30, 1)
        one, _ = run_gemm_pairs(prec, M, N, K, conv, act, out_kind, bias=bias, want_ref=False)
        lib.svt_debug_set(30, 0)
        two, _ = run_gemm_pairs(prec, M, N, K, conv, act, out_kind, bias=bias, want_ref=False)
    finally:
        lib.svt_debug_set(30, 0)
        lib.svt_debug_set(1, 0)
    assert torch.isfinite(one).all() and torch.equal(one, two), (name, bm, out_kind, (one - two).abs().max().item())


def test_gemm_rejects_unaligned():
    lib = _lib.load()
    A = torch.zeros(64, 36, device=DEV, dtype=torch.bfloat16)
    W = torch.zeros(64, 36, device=DEV, dtype=torch.bfloat16)
    C = torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16)
    rc = lib.svt_debug_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, 64, 64, 36, 64, 0, 36, 36, 0, 0, 0, None)
    assert rc != 0 and b"multiple" in lib.svt_last_error()




@pytest.mark.parametrize("name,M,N,K,act", [("ffn1_base", 15968, 3072, 768, 1),        # gemm_pps_kernel (GELU, K < 1024), 63 x 12 tiles of 256 rows
                                            ("ragged_panels", 5200, 2304, 768, 0),      # gemm_p1w_kernel, 28 tile rows: the last panel is short for pm = 3 / 5 / 8 / 16
                                            ("large_ffn1", 12000, 4096, 1024, 1),       # gemm_p1w_kernel<256, GELU>
                                            ("narrow", 40000, 512, 1536, 0)])           # 2 columns of tiles: fewer than a panel is wide
def test_persistent_gemm_tile_walk_is_a_permutation(name, M, N, K, act):
    """Round 6 (csrc/common.h, tile_walk; svt_debug_set key 34): the persistent 16-bit GEMM kernels map logical tile indices to (tile_m, tile_n)
    either n fastest (0) or in panels of pm tile rows walked column by column.  The map must be a permutation of the tiles for EVERY pm --
    a tile computed twice or never would leave NaN-poisoned or stale rows -- and a tile's arithmetic does not depend on who computes it:
    the outputs are bit-identical to the n-fastest walk for panel heights that divide the tile rows, that do not, and that exceed them."""
    lib = _lib.load()
    outs = {}
    try:
        for pm in (0, 3, 5, 8, 16, 64):
            lib.svt_debug_set(34, pm)
            C, ref = run_gemm(1, M, N, K, None, act, 0, False, seed=5)
            assert torch.isfinite(C).all(), (name, pm)
            outs[pm] = C
    finally:
        lib.svt_debug_set(34, -1)
    err = (outs[0] - ref).abs().max().item()
    assert err < 0.05, (name, err)
    for pm, C in outs.items():
        assert torch.equal(C, outs[0]), (name, pm, (C - outs[0]).abs().max().item())
