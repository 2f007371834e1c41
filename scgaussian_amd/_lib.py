"""ctypes binding of libscg_raster.so (include/scg_raster.h).

The library is the product path: if it is missing or fails to load this module raises — there is no
CPU or PyTorch fallback (the oracle under oracle/ is test infrastructure and is never imported here).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (imported first so that torch's bundled libamdhip64.so.7 is the HIP runtime we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCG_LIB_PATH selects an experiment variant built with `python -m scgaussian_amd.build --tag=...` (profiling only)
LIB_PATH = os.environ.get("SCG_LIB_PATH") or os.path.join(_HERE, "libscg_raster.so")

ABI_VERSION = 10


class ScgFrame(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
        ("width", C.c_int32), ("height", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("bg", C.c_void_p),
        ("tile_cost_in", C.c_void_p), ("tile_cost_out", C.c_void_p), ("long_lists_out", C.c_void_p),
        ("bwd_cost_in", C.c_void_p), ("bwd_cost_out", C.c_void_p), ("num_rendered_out", C.c_void_p),
    ]


class ScgWorkspaceLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("splats", "rects", "depth_keys", "clamped", "point_list", "ranges", "final_T",
                                          "n_contrib", "bin_scratch", "total", "partial_words")]


class ScgStageEvents(C.Structure):
    _fields_ = [("begin", C.c_void_p * 3), ("end", C.c_void_p * 3)]


class ScgModelSet(C.Structure):
    """One set of Gaussians in the reference model's raw parameterisation (include/scg_raster.h ScgModelSet)."""
    _fields_ = [("count", C.c_int32)] + [(n, C.c_void_p) for n in ("zval", "rayo", "rayd", "xyz", "features_dc", "features_rest",
                                                                   "opacity", "scaling", "rotation")]


class ScgModel(C.Structure):
    _fields_ = [("ray", ScgModelSet), ("bg", ScgModelSet)]


class ScgModelGradSet(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("zval", "xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")]


class ScgModelGrads(C.Structure):
    _fields_ = [("ray", ScgModelGradSet), ("bg", ScgModelGradSet)]


# name -> (restype, argtypes); must list every symbol declared in include/scg_raster.h
_P = C.c_void_p
SYMBOLS = {
    "scg_last_error": (C.c_char_p, []),
    "scg_abi_version": (C.c_int32, []),
    "scg_struct_bytes": (C.c_size_t, [C.c_int32]),
    "scg_ranges_words": (C.c_size_t, [C.c_int32, C.c_int32]),
    "scg_geometry_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "scg_geometry_forward": (C.c_int, [C.POINTER(ScgFrame)] + [_P] * 7 + [_P] * 6 + [_P, C.c_size_t, _P]),
    "scg_binning_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "scg_binning_accepts_bound": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "scg_binning": (C.c_int, [C.POINTER(ScgFrame), C.c_int64] + [_P] * 2 + [_P] * 3 + [C.c_int32, _P, C.c_size_t, _P]),
    "scg_sort_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "scg_sort_pairs": (C.c_int, [_P] * 4 + [C.c_int64, C.c_int32, _P, C.c_size_t, _P]),
    "scg_scan_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "scg_inclusive_scan_u32": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_size_t, _P]),
    "scg_blend_forward": (C.c_int, [C.POINTER(ScgFrame)] + [_P] * 3 + [_P] * 5 + [_P, _P]),
    "scg_blend_backward": (C.c_int, [C.POINTER(ScgFrame)] + [_P] * 5 + [_P] * 3 + [_P, C.c_int32, _P]),
    "scg_image_loss_dmaps_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "scg_image_loss_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "scg_image_loss_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_size_t, _P]),
    "scg_image_loss_backward": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "scg_image_loss_forward_combined": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, _P, C.c_size_t, _P]),
    "scg_image_loss_backward_combined": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_float, _P, _P]),
    "scg_match_loss_pair": (C.c_int, [_P, C.c_int32, C.c_int32] + [_P] * 9 + [C.c_int32, C.c_float, C.c_float, _P, _P, _P]),
    "scg_knn3_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "scg_knn3_mean_dist2_ws": (C.c_int, [_P, C.c_int64, _P, _P, C.c_size_t, _P]),
    "scg_geometry_backward": (C.c_int, [C.POINTER(ScgFrame)] + [_P] * 7 + [_P] * 3 + [_P] * 8 + [C.c_int32, _P]),
    "scg_workspace_layout": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(ScgWorkspaceLayout)]),
    "scg_forward": (C.c_int, [C.POINTER(ScgFrame)] + [_P] * 7 + [C.c_int64, _P, C.c_size_t] + [_P] * 4 + [_P, _P, _P, C.c_int32, _P, _P]),
    "scg_forward_sorts_in_blend": (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "scg_wait_num_rendered": (C.c_int64, [_P, _P, C.c_int32]),
    "scg_event_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32]),
    "scg_event_destroy": (C.c_int, [_P]),
    "scg_event_elapsed_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "scg_backward": (C.c_int, [C.POINTER(ScgFrame)] + [_P] * 7 + [_P, C.c_int64, _P] + [_P] * 3 + [_P, C.c_int32] + [_P] * 8 + [C.c_int32, _P, _P]),
    "scg_model_activate": (C.c_int, [C.POINTER(ScgModel)] + [_P] * 4 + [_P]),
    "scg_forward_model": (C.c_int, [C.POINTER(ScgFrame), C.POINTER(ScgModel), C.c_int64, _P, C.c_size_t] + [_P] * 4 + [_P, _P, _P, C.c_int32, _P, _P]),
    "scg_backward_model": (C.c_int, [C.POINTER(ScgFrame), C.POINTER(ScgModel), _P, C.c_int64, _P] + [_P] * 3 + [_P, C.c_int32]
                           + [C.POINTER(ScgModelGrads), _P, C.c_int32, _P, _P]),
}

_lib = None


class ScgError(RuntimeError):
    pass


def open_library(path: str) -> C.CDLL:
    """dlopen + type + version-check one build of the library (the product loads exactly one: load(); tools/ab_inproc.py opens
    several variants side by side for same-process A/B runs)."""
    if not os.path.exists(path):
        raise ScgError(f"{path} not found: the HIP extension is required (python -m scgaussian_amd.build); "
                       "there is no CPU fallback")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise ScgError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ScgError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.scg_abi_version() != ABI_VERSION:
        raise ScgError(f"ABI version mismatch: library {lib.scg_abi_version()} != binding {ABI_VERSION}")
    for which, struct in enumerate((ScgFrame, ScgWorkspaceLayout, ScgStageEvents, ScgModel, ScgModelGrads)):
        if lib.scg_struct_bytes(which) != C.sizeof(struct):
            raise ScgError(f"struct layout mismatch: {struct.__name__} is {lib.scg_struct_bytes(which)} bytes in the library, "
                           f"{C.sizeof(struct)} in the binding")
    return lib


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises ScgError when it is absent: build it with
    ``python -m scgaussian_amd.build`` (or ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        _lib = open_library(LIB_PATH)
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().scg_last_error()
        raise ScgError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
