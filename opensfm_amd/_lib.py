"""ctypes binding of ``libosfm_mi355.so`` (the C ABI declared in ``include/osfm_mi355.h``).

There is NO fallback: if the HIP library is missing or no MI355X is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSFM_MI355_LIB") or os.path.join(_HERE, "csrc", "libosfm_mi355.so")  # env: instrumented builds


class OsfmError(RuntimeError):
    """Raised for any non-zero status from the C ABI (the reference raises RuntimeError too)."""


class MatchParams(C.Structure):
    _fields_ = [
        ("lowes_ratio", C.c_double),
        ("symmetric", C.c_int32),
        ("robust", C.c_int32),
        ("robust_matching_min_match", C.c_int32),
        ("robust_matching_threshold", C.c_double),
        ("ransac_confidence", C.c_double),
        ("ransac_max_iters", C.c_int32),
        ("flags", C.c_int32),
    ]


class MatchTimings(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double),
        ("ms_match_kernel", C.c_double),
        ("ms_ransac_kernel", C.c_double),
        ("match_launches", C.c_int64),
        ("pairs", C.c_int64),
        ("pairs_exact_path", C.c_int64),
        ("pairs_ransac", C.c_int64),
        ("ransac_model_points", C.c_int64),
    ]


class RelposeParams(C.Structure):
    _fields_ = [
        ("threshold", C.c_double),
        ("probability", C.c_double),
        ("iterations", C.c_int32),
        ("use_lo", C.c_int32),
        ("lo_iterations", C.c_int32),
        ("refine_iterations", C.c_int32),
    ]


class RelposeResult(C.Structure):
    _fields_ = [
        ("model", C.c_double * 12),
        ("lo_model", C.c_double * 12),
        ("R", C.c_double * 9),
        ("t", C.c_double * 3),
        ("score", C.c_int32),
        ("iterations", C.c_int32),
        ("n_inliers", C.c_int32),
        ("pad", C.c_int32),
    ]


# name -> (restype, argtypes).  tests/test_abi.py checks every symbol of the header is here and exported.
SIGNATURES = {
    "osfm_last_error": (C.c_char_p, []),
    "osfm_version": (C.c_char_p, []),
    "osfm_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "osfm_ctx_destroy": (None, [C.c_void_p]),
    "osfm_ctx_device": (C.c_int, [C.c_void_p]),
    "osfm_ctx_num_cus": (C.c_int, [C.c_void_p]),
    "osfm_ctx_trim_pool": (C.c_int64, [C.c_void_p]),
    "osfm_tracks_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64),
                                     C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "osfm_tracks_num_tracks": (C.c_int64, [C.c_void_p]),
    "osfm_tracks_num_observations": (C.c_int64, [C.c_void_p]),
    "osfm_tracks_device_ms": (C.c_double, [C.c_void_p]),
    "osfm_tracks_fetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "osfm_tracks_destroy": (None, [C.c_void_p]),
    "osfm_store_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "osfm_store_upload_f32": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "osfm_store_upload_u8": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_double)]),
    "osfm_store_set_segmentation": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "osfm_store_upload_binary": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_double)]),
    "osfm_match_hamming_ratio": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, C.c_int,
                                           C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]),
    "osfm_match_hamming_ratio_ex": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                              C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]),
    "osfm_store_destroy": (None, [C.c_void_p]),
    "osfm_store_bytes": (C.c_int64, [C.c_void_p]),
    "osfm_match_params_default": (None, [C.POINTER(MatchParams)]),
    "osfm_match_pairs": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int64, C.POINTER(MatchParams), C.POINTER(C.c_void_p),
         C.POINTER(MatchTimings)],
    ),
    "osfm_match_pairs_calibrated": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64, C.POINTER(MatchParams),
         C.POINTER(RelposeParams), C.POINTER(C.c_void_p), C.POINTER(MatchTimings)],
    ),
    "osfm_match_pairs_guided": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_double), C.c_double, C.POINTER(MatchParams),
         C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(RelposeParams), C.POINTER(C.c_void_p), C.POINTER(MatchTimings)],
    ),
    "osfm_result_num_pairs": (C.c_int64, [C.c_void_p]),
    "osfm_result_total_matches": (C.c_int64, [C.c_void_p]),
    "osfm_result_fetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "osfm_result_host_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]),
    "osfm_result_dev_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "osfm_result_device": (C.c_int, [C.c_void_p]),
    "osfm_result_destroy": (None, [C.c_void_p]),
    "osfm_match_l2_ratio": (
        C.c_int,
        [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double, C.c_int,
         C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)],
    ),
    "osfm_match_l2_ratio_ex": (
        C.c_int,
        [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
         C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)],
    ),
    "osfm_pixel_bearings": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]),
    "osfm_relpose_pairs": (
        C.c_int,
        [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int, C.POINTER(RelposeParams), C.c_int,
         C.POINTER(RelposeResult), C.POINTER(C.c_uint8), C.POINTER(C.c_double)],
    ),
    "osfm_match_guided": (
        C.c_int,
        [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_float),
         C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int32), C.c_int,
         C.POINTER(C.c_int)],
    ),
    "osfm_words_store_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_void_p)]),
    "osfm_words_store_destroy": (None, [C.c_void_p]),
    "osfm_words_store_max_count": (C.c_int, [C.c_void_p]),
    "osfm_match_words_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int64, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "osfm_vlad_descriptor": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "osfm_vlad_distances": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "osfm_bow_distances": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "osfm_knn_points": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double,
                                  C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "osfm_radius_points": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_uint32)]),
    "osfm_hahog_extract": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]),
    "osfm_hahog_extract_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_float, C.c_float,
                                           C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.c_int]),
    "osfm_ransac_fundamental": (
        C.c_int,
        [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, C.c_int,
         C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    ),
}

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load the in-tree shared library.  Raises OsfmError when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise OsfmError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(opensfm_amd has no CPU fallback)"
                )
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in _signatures().items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def _signatures():
    sig = dict(SIGNATURES)
    try:  # bundle-adjustment entry points are declared in their own module
        from . import _ba_abi

        sig.update(_ba_abi.SIGNATURES)
    except ImportError:
        pass
    return sig


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().osfm_last_error()
        raise OsfmError(f"{what or 'osfm call'} failed ({status}): {msg.decode() if msg else ''}")


_tls = threading.local()

MAX_FEATURES = 16000     # OSFM_MAX_FEATURES
MATCH_EXACT_KERNEL = 1   # OSFM_MATCH_EXACT_KERNEL
MATCH_SQUARED_RATIO = 2  # OSFM_MATCH_SQUARED_RATIO
MATCH_KEEP_DEVICE = 4    # OSFM_MATCH_KEEP_DEVICE


class Context:
    """One per (process, GPU).  Owns the HIP stream used by every call made through it."""

    def __init__(self, device: int = 0):
        lib = load()
        h = C.c_void_p()
        check(lib.osfm_ctx_create(int(device), C.byref(h)), "osfm_ctx_create")
        self.handle = h
        self.device = device
        self.num_cus = lib.osfm_ctx_num_cus(h)

    def close(self) -> None:
        if self.handle:
            load().osfm_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def default_context(device: Optional[int] = None) -> Context:
    """The calling THREAD's context for `device` (default: LOCAL_RANK or 0).

    The reference calls the leaf functions from a joblib thread pool (``opensfm/context.py:47-67``); a context owns one HIP
    stream, its events and the buffers of the call in flight, so every thread gets its own (the C library additionally
    serialises calls that share a context, ``osfm_ctx::mu``)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    by_device = getattr(_tls, "ctx", None)
    if by_device is None:
        by_device = _tls.ctx = {}
    ctx = by_device.get(device)
    if ctx is None:
        ctx = by_device[device] = Context(device)
    return ctx
