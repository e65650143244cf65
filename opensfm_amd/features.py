"""HAHOG feature extraction on the MI355X: ``hahog`` under the name and signature of ``pyfeatures.hahog``
(``opensfm/src/features/src/hahog.cc:125-206``) and ``extract_features_hahog`` as ``opensfm/features.py:516-534`` calls it.  Thin ctypes
glue over ``csrc/hahog.hip``; there is no CPU fallback."""
import ctypes as C
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check, default_context

HAHOG_ROOT = 1
HAHOG_UCHAR = 2
HAHOG_IMAGE_ON_DEVICE = 4
HAHOG_IMAGE_U8 = 8  # the image is uint8 grey levels: level / 255 in float32 is formed on the device


def _extract(image: np.ndarray, peak_threshold: float, edge_threshold: float, target_num_features: int, flags: int, ctx=None):
    im = np.ascontiguousarray(image, np.uint8 if flags & HAHOG_IMAGE_U8 else np.float32)
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
    check(lib.osfm_hahog_extract(ctx.handle, C.cast(im.ctypes.data, C.POINTER(C.c_float)), im.shape[0], im.shape[1], float(peak_threshold),
                                 float(edge_threshold), int(target_num_features), int(flags), pts.ctypes.data_as(C.POINTER(C.c_float)),
                                 desc.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)), "osfm_hahog_extract")
    # views, not copies (round 6): the buffers are fresh and lazily committed -- only the n rows the library wrote are resident -- and copying 2.8 MB of
    # descriptors into a second fresh array was 0.1 - 0.3 ms of a 1.3 ms call; the unused tail is address space, not memory
    return pts[: n.value], desc[: n.value]


def hahog_batch(images: Sequence[Any], peak_threshold: float, edge_threshold: float, target_num_features: int, flags: int = 0, concurrency: int = 0,
                shapes: Optional[Sequence[Tuple[int, int]]] = None, ctx=None) -> List[Tuple[np.ndarray, np.ndarray]]:
    """``osfm_hahog_extract_batch``: the images of a data set in one call, up to ``concurrency`` (0: 4) in flight on separate streams.
    ``images``: float32 arrays in [0, 1] or uint8 arrays of grey levels (host), or -- with ``shapes`` = their (rows, cols) -- device addresses (ints) of resident images.
    Returns the (points, descriptors) ``hahog`` returns for each image, in order."""
    ctx = ctx or default_context()
    lib = _lib.load()
    n = len(images)
    if n == 0:
        return []
    on_device = shapes is not None
    if on_device:
        flags |= HAHOG_IMAGE_ON_DEVICE
        ims, rc = None, [(int(r), int(c)) for r, c in shapes]
        ptrs = (C.c_void_p * n)(*[int(a) for a in images])
    else:
        kinds = [np.asarray(im).dtype == np.uint8 for im in images]
        u8 = all(kinds)  # grey levels 0 .. 255: converted on the device (HAHOG_IMAGE_U8)
        if u8:
            flags |= HAHOG_IMAGE_U8
        elif any(kinds):  # one flag for the whole call: a list that mixes grey levels 0 .. 255 with floats in [0, 1] has no single meaning
            raise ValueError("hahog_batch takes either uint8 grey levels for every image or float32 values in [0, 1] for every image, not a mix")
        ims = [np.ascontiguousarray(im, np.uint8 if u8 else np.float32) for im in images]
        if any(im.ndim != 2 or im.size == 0 for im in ims):
            raise ValueError("hahog_batch takes non-empty grey-level images (rows x cols)")
        rc = [im.shape for im in ims]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in ims])
    cap = max(16, 4 * int(target_num_features))
    pts = [np.empty((cap, 4), np.float32) for _ in range(n)]
    desc = [np.empty((cap, 128), np.float32) for _ in range(n)]
    rows, cols = (C.c_int * n)(*[r for r, _ in rc]), (C.c_int * n)(*[c for _, c in rc])
    caps, nf = (C.c_int * n)(*([cap] * n)), (C.c_int * n)()
    pp, dp = (C.c_void_p * n)(*[a.ctypes.data for a in pts]), (C.c_void_p * n)(*[a.ctypes.data for a in desc])
    check(lib.osfm_hahog_extract_batch(ctx.handle, n, ptrs, rows, cols, float(peak_threshold), float(edge_threshold), int(target_num_features), int(flags),
                                       pp, dp, caps, nf, int(concurrency)), "osfm_hahog_extract_batch")
    # copies here, unlike _extract: a batch's eight worker threads writing their results into never-touched pages fault them in under one address-space
    # lock (views were tried in round 6: 1 800 -> ~900 images/s); freed at once, the large buffers are what the next call gets back, already resident
    return [(pts[i][: nf[i]].copy(), desc[i][: nf[i]].copy()) for i in range(n)]


def hahog(image: np.ndarray, peak_threshold: float, edge_threshold: float, target_num_features: int, ctx=None) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """``pyfeatures.hahog``: image float32 in [0, 1] -> (points n x 4 [x, y, size, angle in degrees], descriptors n x 128 float32), or None
    for an empty image."""
    return _extract(image, peak_threshold, edge_threshold, target_num_features, 0, ctx)


def extract_features_hahog(image: np.ndarray, config: Dict[str, Any], features_count: int, ctx=None) -> Tuple[np.ndarray, np.ndarray]:
    """``features.extract_features_hahog`` (features.py:516-534): grey image with levels 0..255 -> points, descriptors; the square root
    (``feature_root``) and the scaling to integer values in [0, 255] (``hahog_normalize_to_uchar``) are applied on the device."""
    flags = (HAHOG_ROOT if config["feature_root"] else 0) | (HAHOG_UCHAR if config["hahog_normalize_to_uchar"] else 0)
    image = np.asarray(image)
    if image.dtype == np.uint8:  # the usual case (a decoded grey image): the library divides by 255 on the device -- same float32 values
        out = _extract(image, config["hahog_peak_threshold"], config["hahog_edge_threshold"], features_count, flags | HAHOG_IMAGE_U8, ctx)
    else:
        out = _extract(image.astype(np.float32) / 255, config["hahog_peak_threshold"], config["hahog_edge_threshold"], features_count, flags, ctx)
    if out is None:
        raise TypeError("cannot unpack non-iterable NoneType object")  # what the reference's tuple unpacking raises for an empty image
    return out
