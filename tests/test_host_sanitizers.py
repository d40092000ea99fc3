"""CPU: the host-only routines of the C-ABI (svt_speechbrain_amd/csrc/host.cpp: error string, parameter intake, configuration
validation, the frames -> notes scan) built with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C svt_speechbrain_amd/csrc san`)
and driven by tests/cabi/host_san_driver.cpp: the reference's frame2note fixture, random sequences against the Python frame loop,
exact-size output arrays, and every refusal path with hostile arguments.  A sanitizer finding aborts the binary (non-zero exit).
SURVEY.md §5 (sanitizers): CPU build only -- the GPU pool has no sanitizer runs, and this code has no GPU part."""
import os
import subprocess

import numpy as np
import pytest

import svt_speechbrain_amd as S
from svt_speechbrain_amd.decode import FRAME_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svt_speechbrain_amd", "csrc")
BIN = os.path.join(CSRC, "build_san", "host_san_test")


@pytest.fixture(scope="module")
def san_bin():
    r = subprocess.run(["make", "-C", CSRC, "san"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return BIN


def run(binary, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([binary, *map(str, args)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, f"exit {r.returncode}\n{r.stdout}\n{r.stderr}"
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr
    return r.stdout


def pack(p_on, p_off, octv, pc):
    fr = np.zeros(len(p_on), dtype=FRAME_DTYPE)
    fr["p_on"], fr["p_off"], fr["octave"], fr["pitch_class"] = p_on, p_off, octv, pc
    return fr


def notes_via_binary(binary, tmp_path, frames, cap=None):
    """frames: structured (B, T) -> per clip [[t_on, t_off, pitch or -1, lo, hi], ...] as the sanitized routine returned them"""
    B, T = frames.shape
    path = tmp_path / "frames.bin"
    np.ascontiguousarray(frames).tofile(path)
    args = ["notes", path, B, T, repr(float(np.float32(0.4))), repr(float(np.float32(0.5))), repr(1 / 49.8)]
    if cap is not None:
        args.append(cap)
    out = run(binary, *args)
    if out.startswith("error"):
        return out.strip()
    clips = [[] for _ in range(B)]
    for line in out.splitlines():
        b, t_on, t_off, pitch, lo, hi = line.split()
        clips[int(b)].append([float(t_on), float(t_off), int(pitch), int(lo), int(hi)])
    return clips


def resolve(clip_notes, fr_row):
    """tied pitch histograms come back as -1 with their frame range: the reference's own expression on the reference's own list"""
    out = []
    for t_on, t_off, pitch, lo, hi in clip_notes:
        if pitch < 0:
            bag = [int(o) * 12 + int(p) for o, p in zip(fr_row["octave"][lo:hi], fr_row["pitch_class"][lo:hi]) if o != 4 and p != 12]
            pitch = max(set(bag), key=bag.count) + 36
        out.append([t_on, t_off, pitch])
    return out


def test_hostile_arguments_under_sanitizers(san_bin):
    assert run(san_bin, "hostile").strip() == "hostile ok"


def test_reference_frame2note_fixture_under_sanitizers(san_bin, tmp_path, golden):
    for k, c in golden("frame2note").items():
        fr = pack(c["p_on"].numpy(), c["p_off"].numpy(), c["oct"].numpy(), c["pc"].numpy())
        got = notes_via_binary(san_bin, tmp_path, fr[None])
        assert resolve(got[0], fr) == c["notes"], k


def test_random_sequences_under_sanitizers_match_the_frame_loop(san_bin, tmp_path):
    rng = np.random.default_rng(11)
    grid = np.array([0.0, 0.1, 0.4, 0.5, 0.7, 0.7, 0.9, 1.0], dtype=np.float32)
    for trial in range(12):
        B, T = int(rng.integers(1, 5)), int(rng.integers(2, 300))
        fr = np.zeros((B, T), dtype=FRAME_DTYPE)
        fr["p_on"] = grid[rng.integers(0, len(grid), (B, T))] * (rng.random((B, T)) < 0.3)
        fr["p_off"] = grid[rng.integers(0, len(grid), (B, T))] * (rng.random((B, T)) < 0.2)
        fr["octave"] = np.repeat(rng.integers(0, 5, (B, T // 7 + 1)), 7, axis=1)[:, :T]
        fr["pitch_class"] = np.repeat(rng.integers(0, 13, (B, T // 3 + 1)), 3, axis=1)[:, :T]
        want = [S.frame2note(S.frames_to_info(fr[b]), 0.4, 0.5, 1 / 49.8) for b in range(B)]
        most = max(1, max(len(w) for w in want))
        got = notes_via_binary(san_bin, tmp_path, fr, cap=most)       # output arrays of EXACTLY the size needed
        assert [resolve(got[b], fr[b]) for b in range(B)] == want, trial
        if most > 1:                                                  # one slot short: refused, nothing written past the arrays
            assert "more notes than capacity_per_clip" in notes_via_binary(san_bin, tmp_path, fr, cap=most - 1)
