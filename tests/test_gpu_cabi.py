"""The C-ABI used by a caller that is not Python: tests/cabi/cabi_driver.c (plain C11, gcc, HIP runtime + dlopen, no torch)
creates an encoder from a config struct, loads the parameters by HF key, runs svt_encoder_forward on its own stream with
its own device buffers and writes the features.  Same blob through the Python binding -> same features, and both within the
fp32 tolerance of the reference golden."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import svt_speechbrain_amd as S  # noqa: E402
from svt_speechbrain_amd import _lib  # noqa: E402
from svt_speechbrain_amd import weights as W  # noqa: E402
from svt_speechbrain_amd.config import PRESETS  # noqa: E402
from svt_speechbrain_amd.huggingface_interface import _config_to_c  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build_driver(tmp_path) -> str:
    exe = str(tmp_path / "cabi_driver")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(HERE, "cabi", "cabi_driver.c"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-ldl",
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def write_blob(path, cfg, sd, wav, precision):
    cc = _config_to_c(cfg, True, True, precision)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", 0x53565431))
        f.write(bytes(cc))
        f.write(struct.pack("<i", len(sd)))
        for k, v in sd.items():
            t = v.detach().to(torch.float32).contiguous()
            kb = k.encode()
            f.write(struct.pack("<i", len(kb)))
            f.write(kb)
            f.write(struct.pack("<i", t.dim()))
            f.write(struct.pack(f"<{t.dim()}q", *t.shape))
            f.write(t.numpy().tobytes())
        f.write(struct.pack("<iq", wav.shape[0], wav.shape[1]))
        f.write(wav.contiguous().numpy().tobytes())


@pytest.mark.parametrize("name,precision", [("tiny_group", "fp32"), ("tiny_layer", "fp32"), ("tiny_wavlm", "fp32"), ("base_c1", "bf16")])
def test_plain_c_caller_matches_python_binding_and_golden(golden, tmp_path, name, precision):
    fx = golden(name)
    cfg = PRESETS[fx["cfg"]]
    sd = W.seeded_encoder_state_dict(cfg, seed=fx["weight_seed"])
    sd = {k: v for k, v in sd.items() if v.is_floating_point()}
    if "wav" in fx:
        wav = fx["wav"]
    else:
        g = torch.Generator().manual_seed(fx["wav_seed"])
        wav = (0.1 * torch.randn(fx["B"], fx["L"], generator=g)).clamp_(-1, 1)
    blob, out = str(tmp_path / "blob.bin"), str(tmp_path / "out.bin")
    write_blob(blob, cfg, sd, wav, precision)
    exe = build_driver(tmp_path)
    r = subprocess.run([exe, _lib.LIB_PATH, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "cabi_driver: B=" in r.stdout
    raw = open(out, "rb").read()
    B, T, D = struct.unpack("<iqi", raw[:16])
    got = torch.from_numpy(np.frombuffer(raw[16:], dtype=np.float32).reshape(B, T, D).copy())
    enc = S.HuggingFaceWav2Vec2(fx["cfg"], None, config=cfg, normalize_wav=True, precision=precision, seed=fx["weight_seed"]).to("cuda:0")
    py = enc(wav.to("cuda:0")).cpu()
    assert got.shape == py.shape
    # same library, same kernels, same inputs: the two callers agree to the last bits (fp64 atomics in the whole-batch norms)
    assert (got - py).abs().max().item() < 1e-5
    if precision == "fp32" and "feats" in fx:
        assert (got - fx["feats"]).abs().max().item() < 1e-3
