"""ctypes binding of libnufft_mi355x.so (the C ABI declared in include/nufft_mi355x.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing or cannot be
loaded, importing this module raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NUFFT_LIB_PATH") or os.path.join(_HERE, "libnufft_mi355x.so")   # override: ablation builds

# error codes (include/nufft_mi355x.h)
OK = 0
ERR_INVALID_ARG = 1
ERR_SIZE_TOO_SMALL = 2
ERR_DIM_MISMATCH = 3
ERR_LDS_TOO_SMALL = 4
ERR_UNSUPPORTED = 5
ERR_NO_POINTS = 6
ERR_ALLOC = 7
ERR_HIP = 8
ERR_ROCFFT = 9
ERR_NO_DEVICE = 10

F32, F64 = 0, 1
EVAL_DIRECT, EVAL_FAST_APPROXIMATION = 0, 1
NUM_STAGES = 8
STAGE_NAMES = ("set_points", "t1_zero", "t1_spread", "t1_fft", "t1_deconv", "t2_deconv_pad", "t2_fft", "t2_interp")


class NufftParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("is_complex", C.c_int32), ("ndim", C.c_int32),
        ("N", C.c_int64 * 3),
        ("half_support", C.c_int32),
        ("sigma", C.c_double),
        ("kernel", C.c_int32), ("evalmode", C.c_int32), ("ntransforms", C.c_int32),
        ("fftshift", C.c_int32), ("point_transform", C.c_int32), ("gpu_method", C.c_int32),
        ("device", C.c_int32),
        ("tile_dims", C.c_int32 * 3),
        ("lds_budget_bytes", C.c_int32), ("spread_threads", C.c_int32), ("interp_threads", C.c_int32),
        ("interp_tile_dims", C.c_int32 * 3), ("bin_log2", C.c_int32),
        ("spread_method", C.c_int32),
        ("kernel_param", C.c_double),
        ("struct_size", C.c_int32), ("reserved", C.c_int32),
        ("kernel_param_dim", C.c_double * 3),
        ("N_over", C.c_int64 * 3),
        ("options", C.c_char_p),
    ]


class NufftCallbacks(C.Structure):
    _fields_ = [("point_weights", C.c_void_p), ("mode_factors", C.c_void_p)]


class NufftInfo(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("is_complex", C.c_int32), ("ndim", C.c_int32), ("half_support", C.c_int32),
        ("ntransforms", C.c_int32), ("evalmode", C.c_int32), ("fftshift", C.c_int32), ("device", C.c_int32),
        ("N", C.c_int64 * 3), ("N_over", C.c_int64 * 3), ("N_out", C.c_int64 * 3),
        ("sigma", C.c_double), ("beta", C.c_double * 3),
        ("bin_dims", C.c_int32 * 3), ("nbins", C.c_int32 * 3),
        ("spread_tile", C.c_int32 * 3), ("spread_ntiles", C.c_int32 * 3),
        ("interp_tile", C.c_int32 * 3), ("interp_ntiles", C.c_int32 * 3),
        ("spread_threads", C.c_int32), ("interp_threads", C.c_int32),
        ("lds_bytes_spread", C.c_int64), ("lds_bytes_interp", C.c_int64),
        ("workspace_bytes", C.c_int64), ("num_points", C.c_int64),
        ("npoly", C.c_int32), ("window_scale_log2", C.c_int32 * 3), ("kernel", C.c_int32),
        ("spread_max_items", C.c_int32), ("interp_max_items", C.c_int32), ("spread_method", C.c_int32),
        ("patch_dims", C.c_int32 * 2), ("patch_f32acc", C.c_int32), ("patch_planar", C.c_int32),
        ("ring_column", C.c_int32 * 2), ("ring_segments", C.c_int32), ("ring_halo", C.c_int32),
        ("sort_column", C.c_int32 * 2),
    ]


#: every symbol include/nufft_mi355x.h declares, with (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
SYMBOLS = {
    "nufft_plan_create_ex": (C.c_int, [C.POINTER(_P), C.POINTER(NufftParams)]),
    "nufft_plan_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int,
                                    C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "nufft_plan_destroy": (C.c_int, [_P]),
    "nufft_plan_info": (C.c_int, [_P, C.POINTER(NufftInfo)]),
    "nufft_plan_get_phi_hat": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.c_int64]),
    "nufft_plan_get_poly_coefs": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.c_int64]),
    "nufft_plan_get_index_map": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.c_int64]),
    "nufft_set_points": (C.c_int, [_P, C.c_int64, _PP, _P]),
    "nufft_exec_type1": (C.c_int, [_P, _PP, _PP, _P]),
    "nufft_exec_type2": (C.c_int, [_P, _PP, _PP, _P]),
    "nufft_exec_type1_cb": (C.c_int, [_P, _PP, _PP, _P, _P]),
    "nufft_exec_type2_cb": (C.c_int, [_P, _PP, _PP, _P, _P]),
    "nufft_set_callbacks": (C.c_int, [_P, _P]),
    "nufft_fill_zeros": (C.c_int, [_P, _P]),
    "nufft_spread": (C.c_int, [_P, _PP, _P]),
    "nufft_spread_deferred": (C.c_int, [_P, _PP, _P]),
    "nufft_fft_forward": (C.c_int, [_P, _P]),
    "nufft_deconvolve_truncate": (C.c_int, [_P, _PP, _P]),
    "nufft_deconvolve_pad": (C.c_int, [_P, _PP, _P]),
    "nufft_fft_backward": (C.c_int, [_P, _P]),
    "nufft_interpolate": (C.c_int, [_P, _PP, _P]),
    "nufft_complete_grid": (C.c_int, [_P, _P]),
    "nufft_grid_ptr": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "nufft_copy_grid": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "nufft_get_sort_result": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_uint32), C.c_int64, _P]),
    "nufft_sort_columns_used": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
    "nufft_set_timing": (C.c_int, [_P, C.c_int]),
    "nufft_get_stage_times": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "nufft_spread_engine_used": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
    "nufft_interp_engine_used": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
    "nufft_plan_options": (C.c_char_p, [_P]),
    "nufft_workspace_breakdown": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "nufft_sizeof_params": (C.c_int64, []),
    "nufft_sizeof_info": (C.c_int64, []),
    "nufft_strerror": (C.c_char_p, [C.c_int]),
    "nufft_last_error_message": (C.c_char_p, []),
    "nufft_version": (C.c_int, []),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C nonuniformffts.jl_amd/csrc`). "
            "There is no CPU fallback.")
    try:
        # torch ships its own ROCm runtime with the same sonames; importing it first makes this
        # library bind to the runtime that owns the caller's tensors.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is only needed for device memory
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    # the two structs are mirrored by hand above: refuse a library whose layout differs
    for name, mirror in (("nufft_sizeof_params", NufftParams), ("nufft_sizeof_info", NufftInfo)):
        if getattr(lib, name)() != C.sizeof(mirror):
            raise ImportError(f"{LIB_PATH}: {name}() = {getattr(lib, name)()} but the ctypes mirror has {C.sizeof(mirror)} bytes "
                              "(include/nufft_mi355x.h and _lib.py disagree)")
    return lib


lib = _load()


def error_message(code: int) -> str:
    base = lib.nufft_strerror(code).decode()
    detail = lib.nufft_last_error_message().decode()
    return f"{base}: {detail}" if detail else base
