"""ctypes binding of libcfd_amd.so (declarations: include/cfd_amd.h).

The library is the ONLY compute path.  If it is missing this module raises at first use --
there is deliberately no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcfd_amd.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
CONV_K4S2, CONV_K4S1, CONV_K3S1, CONVT_K4S2, DENSE, CONV_K4S1_BWD = 0, 1, 2, 3, 4, 5
TUNE_IGEMM256, TUNE_IGEMM256_PD, TUNE_WGRAD256, TUNE_CLASS_ORDER, TUNE_STAGED_OUT, TUNE_RESAMPLE_DENSITY, TUNE_NTILE, TUNE_PATCH = 0, 1, 2, 3, 4, 5, 6, 7
TUNE_RESAMPLE_FUSED = 8
TUNE_ONEPHASE = 9
TUNE_SKINNY = 10


class ConvDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("dtype", C.c_int), ("out_dtype", C.c_int),
                ("B", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("ldx", C.c_int), ("ldy", C.c_int), ("act", C.c_int), ("leaky_slope", C.c_float),
                ("split_k", C.c_int)]


class CfdError(RuntimeError):
    pass


_vp, _i, _f, _d, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64

# name -> (restype, argtypes); every symbol include/cfd_amd.h declares
SIGNATURES = {
    "cfd_version": (_i, []),
    "cfd_last_error": (C.c_char_p, []),
    "cfd_set_tuning": (_i, [_i, _i]),
    "cfd_conv_workspace_bytes": (_i64, [C.POINTER(ConvDesc)]),
    "cfd_conv_suggest_split_k": (_i, [C.POINTER(ConvDesc)]),
    "cfd_conv_forward": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfd_conv_stats_fusable": (_i, [C.POINTER(ConvDesc)]),
    "cfd_conv_forward_stats": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cfd_conv_pool2_fusable": (_i, [C.POINTER(ConvDesc)]),
    "cfd_conv_forward_pool2": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp]),
    "cfd_conv_pool2_keep_fusable": (_i, [C.POINTER(ConvDesc)]),
    "cfd_conv_forward_pool2_keep": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cfd_pack_conv_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cfd_pack_convT_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cfd_pack_dense_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "cfd_conv_first": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cfd_convT_out": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cfd_convT_out_relu": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cfd_convT_out_bn": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
                              _i, _vp]),
    "cfd_conv3_pool2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cfd_bn_act": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, C.c_uint64, _i64, _vp, _i, _vp, _i, _i, _i, _i, _f, _i,
                        _i, _f, _i, _vp]),
    "cfd_bn_apply": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, C.c_uint64, _i64, _vp, _i, _vp, _i, _i, _i,
                          _i, _f, _i, _i, _f, _i, _vp]),
    "cfd_dropout_epoch": (_i, [C.c_uint64, _i, _vp]),
    "cfd_bn_stats": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "cfd_dropout_mask": (_i, [_vp, _i64, _i, _i, _i, _f, C.c_uint64, _vp]),
    "cfd_maxpool2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cfd_cast_copy": (_i, [_vp, _i, _i, _vp, _i, _i, _i64, _i, _vp]),
    "cfd_conv_wgrad": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp]),
    "cfd_conv_wgrad_suggest_split": (_i, [C.POINTER(ConvDesc)]),
    "cfd_conv_wgrad_workspace_bytes": (_i64, [C.POINTER(ConvDesc), _i]),
    "cfd_conv_wgrad_ws": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp, _vp]),
    "cfd_transpose_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cfd_colsum": (_i, [_vp, _i, _vp, _i64, _i, _i, _vp]),
    "cfd_bn_act_bwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _f, C.c_uint64, _i64, _vp, _i, _vp,
                            _vp, _i, _i, _i, _f, _i, _i, _f, _i, _vp]),
    "cfd_bn_act_bwd_ws": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _f, C.c_uint64, _i64, _vp, _i, _vp,
                               _vp, _i, _i, _i, _f, _i, _i, _f, _i, _vp, _vp]),
    "cfd_act_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _i, _i, _f, _i, _vp]),
    "cfd_maxpool2_act_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cfd_act_bwd_dbias": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _i, _i, _f, _i, _vp, _vp]),
    "cfd_maxpool2_act_bwd_dbias": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "cfd_dropout_apply": (_i, [_vp, _vp, _vp, _i64, _f, C.c_uint64, _i64, _i, _vp]),
    "cfd_bce_logits": (_i, [_vp, _vp, _f, _i64, _vp, _vp, _f, _vp]),
    "cfd_field_loss": (_i, [_vp, _vp, _i64, _i, _f, _f, _vp, _vp, _vp, _i, _vp]),
    "cfd_mfma_probe": (_i, [_vp, _vp, _i, _vp]),
    "cfd_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _f, _f, _f, _f, _f, _vp]),
    "cfd_bn_update_moving": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "cfd_bn_update_moving_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    "cfd_zero_ranges": (_i, [_vp, _vp, _i, _i64, _vp]),
    "cfd_tap_scatter_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "cfd_tap_dx": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "cfd_tap_dw": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp]),
    "cfd_conv_first_wgrad": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cfd_dataset_workspace_bytes": (_i64, [_i, _i, _i]),
    "cfd_dataset_grids": (_i, [_vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "cfd_eval_metrics": (_i, [_vp, _f, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "cfd_resample_workspace_bytes": (_i64, [_i, _i, _i, _i64]),
    "cfd_resample_noalloc_max_bytes": (_i64, []),
    "cfd_resample_nearest_noalloc": (_i, [_vp, _vp, _i, _i64, _vp, _i, _vp, _i, _vp, _i, _d, _d, _d, _vp, _vp, _vp]),
    "cfd_resample_nearest": (_i, [_vp, _vp, _i, _i64, _vp, _i, _vp, _i, _vp, _i, _d, _d, _d, _vp, _vp, _vp]),
    "cfd_resample_nearest_counted": (_i, [_vp, _vp, _i, _i64, _vp, _i, _vp, _i, _vp, _i, _d, _d, _d, _vp, _vp, _vp, _vp]),
    "cfd_csv_format_rows": (_i64, [_vp, _i64, _i, _vp, _i64]),
    "cfd_csv_parse_columns": (_i64, [C.c_char_p, _i64, C.c_char_p, C.c_char_p, _vp, _vp, _i64]),
}

_lib = None


def load():
    """dlopen the library and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    # CFD_AMD_LIB=<file>: another build of the SAME library (scripts/build_variant.sh: an earlier commit's kernels, for
    # interleaved A/B runs on one box); still the HIP library, still no fallback
    path = os.environ.get("CFD_AMD_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise CfdError(f"{path} is missing: build it with `python -m cfd_amd.build` "
                       "(the HIP library is the only compute path; there is no fallback)")
    # torch ships its own libamdhip64; import it FIRST so that this library binds to the same HIP
    # runtime (same soname) instead of pulling a second copy from /opt/rocm -- two runtimes in one
    # process do not share devices, streams or allocations.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    # A/B switch for whole-program runs (bench.py, the CLIs): CFD_AMD_TUNE="6=256,0=1" -> cfd_set_tuning(6, 256), (0, 1)
    for kv in filter(None, os.environ.get("CFD_AMD_TUNE", "").split(",")):
        k, v = kv.split("=")
        if lib.cfd_set_tuning(int(k), int(v)) != 0:
            raise CfdError(f"CFD_AMD_TUNE: bad entry {kv!r}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().cfd_last_error().decode(errors="replace")
        raise CfdError(f"{what} failed (status {rc}): {msg}")
