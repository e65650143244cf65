"""HAHOG feature extraction on the MI355X: ``hahog`` under the name and signature of ``pyfeatures.hahog``
(``opensfm/src/features/src/hahog.cc:125-206``) and ``extract_features_hahog`` as ``opensfm/features.py:516-534`` calls it.  Thin ctypes
glue over ``csrc/hahog.hip``; there is no CPU fallback."""
import ctypes as C
from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import check, default_context

HAHOG_ROOT = 1
HAHOG_UCHAR = 2


def _extract(image: np.ndarray, peak_threshold: float, edge_threshold: float, target_num_features: int, flags: int, ctx=None):
    im = np.ascontiguousarray(image, np.float32)
    if im.ndim != 2:
        raise ValueError("hahog takes one grey-level image (rows x cols)")
    if im.size == 0:
        return None  # hahog.cc:127-129
    ctx = ctx or default_context()
    lib = _lib.load()
    cap = max(16, 4 * int(target_num_features))
    pts = np.empty((cap, 4), np.float32)
    desc = np.empty((cap, 128), np.float32)
    n = C.c_int(0)
    check(lib.osfm_hahog_extract(ctx.handle, im.ctypes.data_as(C.POINTER(C.c_float)), im.shape[0], im.shape[1], float(peak_threshold),
                                 float(edge_threshold), int(target_num_features), int(flags), pts.ctypes.data_as(C.POINTER(C.c_float)),
                                 desc.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)), "osfm_hahog_extract")
    return pts[: n.value].copy(), desc[: n.value].copy()


def hahog(image: np.ndarray, peak_threshold: float, edge_threshold: float, target_num_features: int, ctx=None) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """``pyfeatures.hahog``: image float32 in [0, 1] -> (points n x 4 [x, y, size, angle in degrees], descriptors n x 128 float32), or None
    for an empty image."""
    return _extract(image, peak_threshold, edge_threshold, target_num_features, 0, ctx)


def extract_features_hahog(image: np.ndarray, config: Dict[str, Any], features_count: int, ctx=None) -> Tuple[np.ndarray, np.ndarray]:
    """``features.extract_features_hahog`` (features.py:516-534): grey image with levels 0..255 -> points, descriptors; the square root
    (``feature_root``) and the scaling to integer values in [0, 255] (``hahog_normalize_to_uchar``) are applied on the device."""
    flags = (HAHOG_ROOT if config["feature_root"] else 0) | (HAHOG_UCHAR if config["hahog_normalize_to_uchar"] else 0)
    out = _extract(np.asarray(image).astype(np.float32) / 255, config["hahog_peak_threshold"], config["hahog_edge_threshold"], features_count, flags, ctx)
    if out is None:
        raise TypeError("cannot unpack non-iterable NoneType object")  # what the reference's tuple unpacking raises for an empty image
    return out
