"""GPU: the AV-HuBERT lip front-end through the C-ABI against the golden vectors captured from the
reference's own resnet.py (tests/golden/video_front.pt).  fp32 mode: within 1e-3 (relative to the output scale);
bf16 mode: bounded error, printed."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402
from svt_speechbrain_amd.video import SubModel  # noqa: E402

DEV = "cuda:0"
CASES = ["roi88", "roi88_t1", "roi32", "roi50", "roi60"]  # stage-1 widths 22 (two-pixel rows), 8, 13 (plain), 15 (four-pixel rows)


def _run(fx, precision):
    m = SubModel(512, fx["E"], "prelu", precision=precision, seed=1)
    m.load_state_dict(W.seeded_video_frontend_state_dict(fx["E"], seed=fx["weight_seed"]), strict=True)
    m = m.to(DEV)
    g = torch.Generator().manual_seed(fx["video_seed"])
    video = torch.randn(fx["B"], 1, fx["T"], fx["HW"], fx["HW"], generator=g)
    y = m(video.to(DEV))
    assert y.shape == (fx["B"], fx["E"], fx["T"])
    return y.transpose(1, 2).cpu()


@pytest.mark.parametrize("name", CASES)
def test_video_frontend_fp32_vs_reference_golden(golden, name):
    fx = golden("video_front")[name]
    y = _run(fx, "fp32")
    ref = fx["feats"]
    err = (y - ref).abs().max().item()
    assert err < 1e-3 * max(1.0, ref.abs().max().item()), (name, err)


@pytest.mark.parametrize("name", CASES)
def test_video_frontend_bf16_error_bound(golden, name):
    fx = golden("video_front")[name]
    y = _run(fx, "bf16")
    ref = fx["feats"]
    d = (y - ref).abs()
    scale = ref.abs().max().item()
    print(f"video front-end bf16 {name}: max |d| {d.max().item():.4f}  mean |d| {d.mean().item():.5f}  (|ref| max {scale:.2f}, std {ref.std().item():.3f})")
    assert d.max().item() < 0.06 * scale and d.mean().item() < 0.01 * scale


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("hw,B,T", [(88, 2, 300), (88, 1, 7), (80, 1, 40), (60, 3, 100), (50, 2, 31), (92, 1, 30)])
def test_stage1_frame_resident_conv_against_the_gemm_path(precision, hw, B, T):
    """conv3x3_c64_kernel (two padded frames resident in LDS, the 3x3 weights in registers; ROIs of 50-88 pixels) against the same
    front-end with stage 1 on the GEMM kernels (svt_debug_set(23, 0)).  Same 16-bit storage, and the same fp32 summation order (tap by
    tap, 32 channels per MFMA; the GEMM path's structural zeros add exact zeros) in stage 1, two separately summed input-channel halves
    in stage 2's conv3x3_c128_kernel: a few ulps of the activations, and svt_debug_set(24, 0)'s launch count proves the direct path ran.  600 frames = more than two frames per workgroup (both LDS buffers re-used), 7 = fewer
    frames than CUs; 92 pixels = two 25 x 25 padded frames do not fit the LDS and fall back."""
    lib = _lib.load("f16" if precision == "fp16" else "")
    m = SubModel(512, 256, "prelu", precision=precision, seed=5).to(DEV)
    g = torch.Generator().manual_seed(hw + T)
    video = torch.randn(B, 1, T, hw, hw, generator=g).to(DEV)
    n0 = lib.svt_debug_set(24, 0)
    y = m(video).float()
    n1 = lib.svt_debug_set(24, 0)
    # stage 1: two BasicBlocks x two convolutions, stage 2: its three stride-1 convolutions (92: neither fits the LDS)
    assert n1 - n0 == (0 if hw == 92 else 7)
    assert torch.equal(y, m(video).float())
    lib.svt_debug_set(23, 0)
    try:
        ref = m(video).float()
    finally:
        lib.svt_debug_set(23, 1)
    assert lib.svt_debug_set(24, 0) == 2 * n1 - n0
    assert torch.isfinite(y).all() and y.abs().max().item() > 1.0
    # stage 1 is bit-identical to the GEMM path; stage 2 sums its two input-channel halves separately (another fp32 summation order)
    d = (y - ref).abs()
    scale = ref.abs().max().item()
    print(f"direct convolutions {precision} {hw}: max |d| {d.max().item():.5f} mean {d.mean().item():.6f} scale {scale:.2f}")
    if hw == 92:
        assert torch.equal(y, ref)
    else:
        assert 0 < d.max().item() < 0.03 * scale and d.mean().item() < 0.003 * scale


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("hw,B,T", [(88, 2, 301), (88, 9, 33), (88, 1, 5), (60, 3, 50), (50, 2, 31), (32, 4, 20), (92, 1, 12)])
def test_fused_stem_and_maxpool_against_the_two_kernels(precision, hw, B, T):
    """conv3d_front_pool_kernel (stem + 3x3/2 max-pool, persistent over runs of consecutive frames with one new plane per frame) against
    the stem and pool kernels it replaces (svt_debug_set(26, 0)): it pools the same bf16-rounded stem outputs -> bit-identical.
    602 = 3 frames per workgroup with a run crossing from clip 0 into clip 1; 9 x 33: runs of two frames crossing clips at odd
    boundaries; 5 frames: one frame per workgroup; 60 / 50 / 32: 30, 25 (odd) and 16 stem rows (short last bands); 92: the six plane
    slots and the band do not fit the LDS, the two kernels run."""
    lib = _lib.load("f16" if precision == "fp16" else "")
    m = SubModel(512, 128, "prelu", precision=precision, seed=9).to(DEV)
    g = torch.Generator().manual_seed(hw * 7 + T)
    video = torch.randn(B, 1, T, hw, hw, generator=g).to(DEV)
    y = m(video).float()
    assert torch.equal(y, m(video).float())
    lib.svt_debug_set(26, 0)
    try:
        ref = m(video).float()
    finally:
        lib.svt_debug_set(26, 1)
    assert torch.isfinite(y).all() and y.abs().max().item() > 1.0
    assert torch.equal(y, ref)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("h,w", [(72, 88), (88, 64), (56, 80), (24, 136)])
def test_non_square_roi_on_every_path(precision, h, w):
    """Non-square lip ROIs (the C-ABI takes h and w separately; the reference crops 88 x 88): the round-4 kernels (fused stem + pool,
    frame-resident stage 1-2 convolutions) against the kernels they replace, and the fp32 mode as the yardstick of both."""
    lib = _lib.load("f16" if precision == "fp16" else "")
    m = SubModel(512, 128, "prelu", precision=precision, seed=3).to(DEV)
    m32 = SubModel(512, 128, "prelu", precision="fp32", seed=3).to(DEV)
    g = torch.Generator().manual_seed(h * 100 + w)
    video = torch.randn(2, 1, 37, h, w, generator=g).to(DEV)
    n0 = lib.svt_debug_set(24, 0)
    y = m(video).float()
    assert lib.svt_debug_set(24, 0) - n0 == 7   # (24 x 136: 34 pooled columns exceed the fused stem's item index -> the two kernels)
    lib.svt_debug_set(23, 0)
    lib.svt_debug_set(26, 0)
    try:
        old = m(video).float()
    finally:
        lib.svt_debug_set(23, 1)
        lib.svt_debug_set(26, 1)
    ref = m32(video).float()
    scale = ref.abs().max().item()
    e_new, e_old = (y - ref).abs().max().item(), (old - ref).abs().max().item()
    print(f"non-square {precision} {h}x{w}: new {e_new:.4f} old {e_old:.4f} scale {scale:.2f}")
    lim = 0.06 if precision == "bf16" else 0.01
    assert e_new < lim * scale and e_old < lim * scale and e_new < 1.5 * e_old + 1e-3 * scale


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("hw,B,T", [(88, 2, 40), (60, 1, 9), (92, 1, 6)])
def test_stage2_downsample_inside_conv1_product(precision, hw, B, T):
    """Stage 2, block 0: the 1x1 stride-2 downsample as columns 128..255 of conv1's 3x3 stride-2 product (centre-tap weights, slope 1;
    GemmArgs::c_nsplit sends them to the next buffer) against the two separate products (svt_debug_set(27, 0)): the extra K range
    multiplies exact zeros -> bit-identical."""
    lib = _lib.load("f16" if precision == "fp16" else "")
    m = SubModel(512, 128, "prelu", precision=precision, seed=11).to(DEV)
    g = torch.Generator().manual_seed(hw + 3 * T)
    video = torch.randn(B, 1, T, hw, hw, generator=g).to(DEV)
    y = m(video).float()
    lib.svt_debug_set(27, 0)
    try:
        ref = m(video).float()
    finally:
        lib.svt_debug_set(27, 1)
    assert torch.isfinite(y).all() and y.abs().max().item() > 1.0
    assert torch.equal(y, ref)


def test_video_frontend_errors():
    m = SubModel(512, 64, "prelu", precision="fp32").to(DEV)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 2, 32, 32, device=DEV))
    with pytest.raises(_lib.SvtError):
        m(torch.zeros(1, 1, 2, 32, 32))
    with pytest.raises(RuntimeError):  # strict load, like torch
        m.load_state_dict({"proj.weight": torch.zeros(64, 512)})


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_kept_workspace_writes_the_zero_halos_once_and_again_when_it_must(prec):
    """svt_video_keep_workspace: the module owns its workspace, so the zero halos of the stage buffers are written on the first call of
    a geometry only.  Same input -> bit-identical output on the kept halos; another geometry in between -> rewritten; with keep = 0 (the
    C-ABI default: the workspace is scratch) a workspace full of NaN patterns in front of every call changes nothing."""
    from svt_speechbrain_amd import _lib
    lib = _lib.load()
    m = SubModel(512, 128, "prelu", precision=prec, seed=4991).to(DEV)
    g = torch.Generator().manual_seed(12)
    a = torch.randn(2, 1, 6, 88, 88, generator=g).to(DEV)
    b = torch.randn(1, 1, 3, 60, 60, generator=g).to(DEV)
    ya = m(a).clone()
    assert torch.isfinite(ya).all() and torch.equal(m(a), ya) and torch.equal(m(a), ya)        # second and third call: halos kept
    yb = m(b).clone()                                                                           # another geometry, same (larger) workspace
    assert torch.equal(m(a), ya) and torch.equal(m(b), yb) and torch.equal(m(a), ya)
    slot = m._sync(a.device)
    _lib.check(lib.svt_video_keep_workspace(slot.handle, 0), "svt_video_keep_workspace")
    for _ in range(2):
        slot.ws.fill_(0xFF)                                                                     # bf16 / fp32 NaN patterns everywhere
        assert torch.equal(m(a), ya)
    _lib.check(lib.svt_video_keep_workspace(slot.handle, 1), "svt_video_keep_workspace")       # forgets what it had written
    slot.ws.fill_(0xFF)
    assert torch.equal(m(a), ya) and torch.equal(m(a), ya)
    side = torch.cuda.Stream()                                                                  # another stream: written again, in ITS order
    with torch.cuda.stream(side):
        yc = m(a)
    side.synchronize()
    assert torch.equal(yc, ya)


# ---- AV-HuBERT video encoder end to end (front-end pinned above; transformer checked against the oracle restatement,
# which is parity-unpinned for fairseq: see oracle/svt_oracle.py::avhubert_video_forward) ----
@pytest.mark.parametrize("prec,tol", [("fp32", 1e-3), ("bf16", 0.35), ("fp16", 0.05)])
def test_avhubert_video_encoder_vs_oracle(prec, tol):
    from oracle import svt_oracle as O
    from svt_speechbrain_amd.video import FairseqAVHubertPretrain
    cfg = S.PRESETS["tiny-avhubert-video"]
    m = FairseqAVHubertPretrain(config=cfg, precision=prec, seed=77, output_norm=True)
    sd = W.seeded_avhubert_video_state_dict(cfg, seed=123)
    sd["mask_emb"] = torch.zeros(4)                         # pre-training leftovers in a real checkpoint: ignored
    sd["final_proj.weight"] = torch.zeros(3, 3)
    sd["feature_extractor_audio.proj.weight"] = torch.zeros(cfg.hidden_size, 104)
    m.load_fairseq_model_state(sd)
    m = m.to(DEV)
    g = torch.Generator().manual_seed(5)
    video = torch.randn(2, 1, 9, 40, 40, generator=g)
    out = m({"video": video.to(DEV), "audio": None}).cpu()
    with torch.no_grad():
        ref = O.avhubert_video_forward(sd, cfg, video, output_norm=True)
    assert out.shape == ref.shape == (2, 9, cfg.hidden_size)
    err = (out - ref).abs().max().item()
    print(f"AV-HuBERT video encoder {prec}: max |d| {err:.5f} (ref std {ref.std().item():.3f})")
    assert err < tol
    with pytest.raises(NotImplementedError):
        m({"video": video.to(DEV), "audio": torch.zeros(1)})
    with pytest.raises(RuntimeError):
        m.load_fairseq_model_state({"layer_norm.weight": torch.zeros(128)})


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-3), ("fp16x3", 1e-3), ("bf16", 0.35), ("fp16", 0.05)])
@pytest.mark.parametrize("name", ["tiny_stable", "tiny_stable_t1", "tiny_postln"])
def test_avhubert_video_encoder_vs_reference_glue_golden(golden, name, prec, tol):
    """a15 / f2 pinned: FairseqAVHubertPretrain on the HIP path against what the REFERENCE's own ``FairseqAVHubertPretrain.forward`` ->
    ``AVHubertModel.extract_finetune`` produced (tests/golden/video_glue.pt: real resnet.ResEncoder front-end, the reference's
    zeros-for-audio / concat / LayerNorm(2E) / post_extract_proj glue, an HF encoder module in the place of fairseq's)."""
    from svt_speechbrain_amd.video import FairseqAVHubertPretrain
    fx = golden("video_glue")[name]
    cfg = S.PRESETS[fx["cfg"]]
    m = FairseqAVHubertPretrain(config=cfg, precision=prec, seed=1, output_norm=fx["output_norm"])
    m.load_fairseq_model_state(W.seeded_avhubert_video_state_dict(cfg, seed=fx["weight_seed"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(fx["video_seed"])
    video = torch.randn(fx["B"], 1, fx["T"], fx["HW"], fx["HW"], generator=g)
    out = m({"video": video.to(DEV), "audio": None}).cpu()
    assert out.shape == fx["out"].shape
    err = (out - fx["out"]).abs().max().item()
    print(f"AV-HuBERT video branch {prec}[{name}]: max |d| vs the reference's extract_finetune {err:.2e}")
    assert err < tol, err


def test_avhubert_per_clip_norm_batch_equals_batch1():
    """The video wrapper's output norm per clip: a batch of clips == the clips forwarded one at a time (fp32: to the last bits)."""
    from svt_speechbrain_amd.video import FairseqAVHubertPretrain
    cfg = S.PRESETS["tiny-avhubert-video"]
    m = FairseqAVHubertPretrain(config=cfg, precision="fp32", seed=78, output_norm=True)
    m.load_fairseq_model_state(W.seeded_avhubert_video_state_dict(cfg, seed=124))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(6)
    video = torch.randn(3, 1, 8, 40, 40, generator=g)
    video[1] *= 2.5
    video = video.to(DEV)
    one = torch.cat([m({"video": video[b:b + 1], "audio": None}) for b in range(3)])
    batched = m({"video": video, "audio": None}, clips_per_norm_group=1)
    whole = m({"video": video, "audio": None})
    assert (batched - one).abs().max().item() < 1e-5
    assert (whole - one).abs().max().item() > 1e-3


# ---- the recipe's input side (round 6): uint8 ROI -> transform_eval inside the padding kernel (svt_video_forward_u8) ----
U8_CASES = ["roi96", "roi97x99", "roi88", "ramp"]


def _u8_model(fx, precision):
    m = SubModel(512, fx["E"], "prelu", precision=precision, seed=1)
    m.load_state_dict(W.seeded_video_frontend_state_dict(fx["E"], seed=fx["weight_seed"]), strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("name", U8_CASES)
def test_u8_input_transform_is_bit_identical_to_the_reference(golden, name):
    """The padded operand the stem reads, written by video_pad_u8_kernel from the RAW uint8 ROI (exact-fp32 mode: fp32 pixels at
    (t + 2, y + 3, x + 4) of the workspace's first region, zeros elsewhere), against what the reference's own
    Compose([Normalize(0, 255), CenterCrop((88, 88)), Normalize(0.421, 0.165)]) + astype(float32) made of the same bytes
    (tests/golden/video_u8.pt): every bit.  Then the features against the reference SubModel's on its transformed frames."""
    fx = golden("video_u8")[name]
    m = _u8_model(fx, "fp32")
    roi = fx["roi"].to(DEV)                                    # (T, H, W) uint8, as np.load gives it
    y = m.forward_into(roi.unsqueeze(0))                      # (1, T, E)
    slot = m._sync(torch.device(DEV))
    T = roi.shape[0]
    Hp, Wp = 88 + 6, ((88 + 8 + 7) // 8) * 8
    torch.cuda.synchronize()
    vp = slot.ws[:(T + 4) * Hp * Wp * 4].view(torch.float32).view(T + 4, Hp, Wp).cpu()
    inner = vp[2:T + 2, 3:3 + 88, 4:4 + 88]
    assert torch.equal(inner.view(torch.int32), fx["sig"].view(torch.int32)), "transform_eval in the kernel differs from numpy's bits"
    halo = vp.clone()
    halo[2:T + 2, 3:3 + 88, 4:4 + 88] = 0
    assert not halo.any()
    ref = fx["feats"]
    assert (y.cpu() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16", "fp16x3"])
@pytest.mark.parametrize("name", U8_CASES)
def test_u8_path_equals_float_path_on_the_references_transformed_frames(golden, name, precision):
    """Same model, two inputs: the raw uint8 ROI (transform in the kernel, 1 byte per pixel) and the reference's transformed float32
    frames through the float entry point (4 bytes per pixel).  The padded operand is the same, so the features are bit-identical in
    every precision; accepted layouts: (B, T, H, W), (B, 1, T, H, W) and the recipe's (B, T, H, W, 1)."""
    fx = golden("video_u8")[name]
    m = _u8_model(fx, precision)
    roi = fx["roi"].to(DEV)
    a = m.forward_into(roi.unsqueeze(0)).clone()
    b = m.forward_into(fx["sig"].to(DEV)[None, None]).clone()             # (1, 1, T, 88, 88) float
    assert torch.equal(a, b)
    assert torch.equal(m.forward_into(roi[None, None]), a) and torch.equal(m.forward_into(roi[None, ..., None]), a)
    assert torch.equal(m(roi.unsqueeze(0)), a.transpose(1, 2))              # the reference's (B, E, T) layout
    with pytest.raises(ValueError):
        m.forward_into(fx["sig"].to(DEV)[None, None], transform=S.EvalTransform())   # a float tensor is already normalised
    with pytest.raises(_lib.SvtError):
        m.forward_into(roi[:, :80, :80].contiguous().unsqueeze(0))                   # CenterCrop(88) of an 80 x 80 frame


def test_fused_concat_buffer_is_written_in_place(golden):
    """cat([zeros, video], -1) of AV-HuBERT's concat fusion (hubert.py:700-712) written by the front-end's own call: right half = the
    features (row pitch 2E), left half zeroed by the call's 2-D memset -- from a buffer full of NaNs, and identical to the plain call."""
    fx = golden("video_u8")["roi96"]
    for precision in ("fp32", "bf16"):
        m = _u8_model(fx, precision)
        roi = fx["roi"].to(DEV).unsqueeze(0).repeat(2, 1, 1, 1)
        E, T = fx["E"], roi.shape[1]
        plain = m.forward_into(roi).clone()
        fused = torch.full((2, T, 2 * E), float("nan"), device=DEV)
        out = m.forward_into(roi, fused)
        assert out.data_ptr() == fused.data_ptr()
        assert torch.equal(fused[..., E:], plain) and not fused[..., :E].any()
        with pytest.raises(ValueError):
            m.forward_into(roi, torch.empty((2, T, 2 * E + 1), device=DEV))


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-3), ("bf16", 0.35)])
def test_avhubert_encoder_takes_the_raw_uint8_roi_and_writes_song_features(golden, tmp_path, prec, tol):
    """FairseqAVHubertPretrain on {"video": uint8 ROI}: equal to the float path on the host-transformed frames (bit for bit: same
    operand), and the per-song extraction pass (N20EMv2/video_only/extract_ssl_feats.py:28-36, 99-111): utterances of 5 s at 50
    frames/s as batch-1 forwards, concatenated, saved as <song folder>/noise_data/video_feats.pt."""
    from svt_speechbrain_amd.video import FairseqAVHubertPretrain
    cfg = S.PRESETS["tiny-avhubert-video"]
    m = FairseqAVHubertPretrain(config=cfg, precision=prec, seed=1, output_norm=True)
    m.load_fairseq_model_state(W.seeded_avhubert_video_state_dict(cfg, seed=321))
    m = m.to(DEV)
    rng = torch.Generator().manual_seed(4)
    song = torch.randint(0, 256, (560, 96, 96), generator=rng, dtype=torch.uint8)        # 11.2 s at 50 fps -> 2 utterances (250 + 310)
    tf = S.EvalTransform()
    bounds = S.utterance_bounds(560, 50, 5.0)
    assert bounds == [(0, 250), (250, 560)]
    want = []
    for lo, hi in bounds:
        fl = torch.from_numpy(tf(song[lo:hi].numpy()))[None, None].to(DEV)
        want.append(m({"video": fl, "audio": None})[0])
    want = torch.cat(want)
    got = S.song_video_features(m, song.to(DEV))
    assert got.shape == (560, cfg.hidden_size) and torch.equal(got, want)
    path = S.save_song_video_features(got, str(tmp_path / "song_0001"))
    assert path.endswith("song_0001/noise_data/video_feats.pt")
    back = torch.load(path)
    assert back.device.type == "cpu" and torch.equal(back, got.cpu())
    # the recipe's own batch layout: batch.sig is (B, T, H, W, 1) before its permute (extract_ssl_feats.py:30-35)
    one = m({"video": song[:250].to(DEV)[None, ..., None], "audio": None})
    assert torch.equal(one[0], want[:250])
