"""ctypes binding of ``libsvt_mi355.so`` (C-ABI: ``include/svt_mi355.h``).

The HIP library is the product; there is NO CPU fallback.  Importing this module never touches the GPU,
but every compute entry point raises ``SvtError`` when the library is missing or no gfx950 device is
visible (SURVEY.md §8b error conventions: Python exceptions, never an abort).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVT_LIB_PATH: load another build of the bf16 library (kernel experiments: tools/*_bench.py, bench.py); default = the in-tree one
LIB_PATH = os.environ.get("SVT_LIB_PATH") or os.path.join(_HERE, "libsvt_mi355.so")
MAX_CONV = 8


class SvtError(RuntimeError):
    pass


class EncoderConfigC(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("hidden_size", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("intermediate_size", C.c_int32),
        ("num_conv_layers", C.c_int32),
        ("conv_dim", C.c_int32 * MAX_CONV),
        ("conv_kernel", C.c_int32 * MAX_CONV),
        ("conv_stride", C.c_int32 * MAX_CONV),
        ("feat_extract_norm", C.c_int32),
        ("conv_bias", C.c_int32),
        ("stable_layer_norm", C.c_int32),
        ("feat_proj_layer_norm", C.c_int32),
        ("pos_conv_kernel", C.c_int32),
        ("pos_conv_groups", C.c_int32),
        ("layer_norm_eps", C.c_float),
        ("normalize_wav", C.c_int32),
        ("output_norm", C.c_int32),
        ("precision", C.c_int32),
        ("pos_conv_depth", C.c_int32),
        ("rel_pos_buckets", C.c_int32),
        ("rel_pos_max_distance", C.c_int32),
        ("pos_conv_batch_norm", C.c_int32),
    ]


class VideoTransformC(C.Structure):
    """svt_video_transform: ((u - sub0) / div0 - mean) / std in float64 on the centre crop_h x crop_w window of a uint8 lip ROI."""
    _fields_ = [("sub0", C.c_double), ("div0", C.c_double), ("mean", C.c_double), ("std", C.c_double), ("crop_h", C.c_int32), ("crop_w", C.c_int32)]


class FrameC(C.Structure):
    _fields_ = [("p_on", C.c_float), ("p_off", C.c_float), ("octave", C.c_int32), ("pitch_class", C.c_int32)]


# every symbol include/svt_mi355.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I64P = C.POINTER(C.c_int64)
# svt_norm_reduce_fn (include/svt_mi355.h): int fn(double* sums_dev, int32_t n_doubles, void* stream, void* user)
NORM_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)

SYMBOLS = {
    "svt_last_error": (C.c_char_p, []),
    "svt_abi_version": (C.c_int, []),
    "svt_device_count": (C.c_int, []),
    "svt_encoder_create": (C.c_int, [C.POINTER(EncoderConfigC), C.c_int, C.POINTER(_P)]),
    "svt_encoder_destroy": (None, [_P]),
    "svt_encoder_load_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int, _I64P, C.c_int]),
    "svt_encoder_finalize": (C.c_int, [_P]),
    "svt_encoder_get_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "svt_encoder_num_frames": (C.c_int64, [_P, C.c_int64]),
    "svt_encoder_workspace_bytes": (C.c_int64, [_P, C.c_int32, C.c_int64]),
    "svt_encoder_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, C.c_size_t, _P]),
    "svt_encoder_forward_ex": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, C.c_size_t, _P, C.c_int32]),
    "svt_encoder_forward_head": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P, _P, C.c_int32, C.c_int32, _P, C.c_size_t, _P, C.c_int32]),
    "svt_linear_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int, C.c_int, C.POINTER(_P)]),
    "svt_linear_destroy": (None, [_P]),
    "svt_linear_load": (C.c_int, [_P, _P, _P]),
    "svt_linear_forward": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "svt_decode_frames": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int, _P]),
    "svt_rca_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int, C.POINTER(_P)]),
    "svt_rca_destroy": (None, [_P]),
    "svt_rca_load_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int, _I64P, C.c_int]),
    "svt_rca_finalize": (C.c_int, [_P]),
    "svt_rca_workspace_bytes": (C.c_int64, [_P, C.c_int32, C.c_int32]),
    "svt_rca_forward": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, C.c_size_t, _P]),
    "svt_ctc_greedy": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, C.c_int, _P]),
    "svt_fbank_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "svt_fbank": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                            C.c_float, C.c_float, C.c_float, _P, _P, C.c_size_t, C.c_int, _P]),
    "svt_debug_gemm": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                 C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int, _P]),
    "svt_debug_gemm_pairs": (C.c_int, [C.c_int32, _P, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                       C.c_int64, C.c_int32, C.c_int32, C.c_int, _P, C.c_int32, C.POINTER(C.c_float)]),
    "svt_video_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int, C.POINTER(C.c_void_p)]),
    "svt_video_destroy": (None, [C.c_void_p]),
    "svt_video_load_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "svt_video_finalize": (C.c_int, [C.c_void_p]),
    "svt_video_keep_workspace": (C.c_int, [C.c_void_p, C.c_int]),
    "svt_video_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "svt_video_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "svt_video_forward_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "svt_video_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(VideoTransformC),
                                       C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "svt_deltas": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "svt_context_window": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_void_p]),
    "svt_bce_loss": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                               C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "svt_nll_loss": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                               C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "svt_softmax": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int, C.c_void_p]),
    "svt_debug_attention": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "svt_debug_set": (C.c_int, [C.c_int, C.c_int]),
    "svt_debug_clock": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "svt_debug_encoder_layout": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.c_int]),
    "svt_operand_type": (C.c_int, []),
    "svt_encoder_set_norm_reduce": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_int64]),
    "svt_frames_to_notes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_double, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "svt_debug_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]),
    "svt_debug_free": (C.c_int, [C.c_void_p, C.c_int]),
    "svt_prof_enable": (C.c_int, [C.c_int]),
    "svt_prof_reset": (C.c_int, []),
    "svt_prof_read": (C.c_int, [C.c_int, _I64P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None
_libs: dict = {}   # one handle per build of the library ("" = bf16 operands, "f16" = IEEE-half operands)


def load(variant: str = None) -> C.CDLL:
    """dlopen the in-tree library and bind every declared symbol; raises SvtError if it is missing.
    ``variant="f16"``: the IEEE-half build of the same sources (``libsvt_mi355_f16.so``, precision="fp16" of the modules)."""
    global _lib
    key = variant or ""
    if key in _libs:
        return _libs[key]
    if not variant and os.environ.get("SVT_LIB_SUFFIX"):   # diagnostics: an experimental build of the bf16 library (csrc/Makefile, XNAME)
        variant_path = LIB_PATH.replace("libsvt_mi355.so", f"libsvt_mi355_{os.environ['SVT_LIB_SUFFIX']}.so")
        if not os.path.exists(variant_path):
            raise SvtError(f"SVT_LIB_SUFFIX: {variant_path} not found")
        globals()["LIB_PATH"] = variant_path
    root, ext = os.path.splitext(LIB_PATH)   # only the file's suffix: the checkout may live under a directory with ".so" in its name
    path = LIB_PATH if not variant else f"{root}_{variant}{ext}"
    if not os.path.exists(path):
        raise SvtError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C svt_speechbrain_amd/csrc`). The MI355X path has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.svt_abi_version() != 1:
        raise SvtError(f"{os.path.basename(path)} ABI version mismatch")
    if lib.svt_operand_type() != (1 if variant == "f16" else 0):
        raise SvtError(f"{os.path.basename(path)} was built for another operand type")
    _libs[key] = lib
    if not variant:
        _lib = lib
    return lib


def last_error(lib: C.CDLL = None) -> str:
    return ((lib or load()).svt_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "", lib: C.CDLL = None) -> None:
    if rc != 0:
        raise SvtError(f"{what}: {last_error(lib)} (status {rc})")


def require_gpu() -> None:
    lib = load()
    if lib.svt_device_count() < 1:
        raise SvtError("no gfx950 (MI355X) device is visible: the HIP path cannot run and there is no CPU fallback")


def stream_ptr(device) -> int:
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)


def dev_index(device) -> int:
    import torch
    d = torch.device(device)
    if d.type != "cuda":
        raise SvtError(f"tensors must live on a ROCm device ('cuda:N'), got {d}")
    return d.index if d.index is not None else torch.cuda.current_device()


def ptr(t) -> int:
    return int(t.data_ptr())
