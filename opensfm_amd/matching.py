"""Host-side mirror of the reference's pair-matching interface, backed by the HIP kernels.

Same names, argument meaning and return types as ``opensfm/matching.py`` for the hot path:

================================  =====================================================
this module                        reference
================================  =====================================================
``match_brute_force``              ``opensfm/matching.py:723-756``
``match_brute_force_symmetric``    ``opensfm/matching.py:759-777``
``build_flann_index``              ``opensfm/features.py:638-674`` (exact index, see below)
``match_flann``                    ``opensfm/matching.py:683-697``
``match_flann_symmetric``          ``opensfm/matching.py:700-720``
``robust_match_fundamental``       ``opensfm/matching.py:780-802``
``robust_match``                   ``opensfm/matching.py:906-929``
``robust_match_calibrated``        ``opensfm/matching.py:871-903``
``pixel_bearing_many``             ``pygeometry.Camera.pixel_bearing_many``
``match`` semantics (per pair)     ``opensfm/matching.py:563-634`` (inside ``match_pairs``)
``match_images_with_pairs``        ``opensfm/matching.py:63-98``
``unfilter_matches``               ``opensfm/matching.py:932-936``
================================  =====================================================

Differences that are deliberate: the unordered ``set`` intersection of the reference is returned
sorted by ``(i, j)``; the batched entry point keeps all descriptors resident in HBM
(``DescriptorStore``) instead of the reference's LRU of npz loads (``feature_loading.py``);
``matcher_type: FLANN`` runs the reference's FLANN *semantics* (ratio test on squared float32 distances,
query direction of ``match_flann``) on an EXACT 2-NN search -- the reference's k-means index is approximate and
randomly initialised, so its own output is not reproducible; the exact search is its ``checks -> infinity`` limit.
There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import MatchParams, MatchTimings, OsfmError, RelposeParams, RelposeResult, check, default_context

DEFAULT_CONFIG: Dict[str, Any] = {
    # opensfm/config.py:97-101,191-195
    "lowes_ratio": 0.8,
    "matcher_type": "BRUTEFORCE",
    "symmetric_matching": True,
    "robust_matching_threshold": 0.004,
    "robust_matching_min_match": 20,
    "robust_matching_calib_threshold": 0.004,
    "guided_matching_threshold": 0.006,
    "five_point_refine_match_iterations": 10,
}


def _cfg(config: Optional[Dict[str, Any]], key: str):
    if config is not None and key in config:
        return config[key]
    return DEFAULT_CONFIG[key]


def _fptr(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


# --------------------------------------------------------------------------------------------
# leaf functions (drop-ins)
# --------------------------------------------------------------------------------------------
def _match_hamming_leaf(f1: np.ndarray, f2: np.ndarray, ratio: float, symmetric: bool, ctx=None, flags: int = 0) -> np.ndarray:
    """uint8 bit strings (AKAZE MLDB, ORB): cv2's BruteForce-Hamming branch of match_brute_force (matching.py:737-740); with
    ``MATCH_SQUARED_RATIO`` the FLANN semantics on bit strings (matching.py:683-720 over the LSH index of features.py:660-667, searched
    exactly): ``d0 < lowes_ratio ** 2 * d1`` in doubles on the int Hamming distances, one-way matching queries with ``f2``"""
    ctx = ctx or default_context()
    a, b = np.ascontiguousarray(f1, np.uint8), np.ascontiguousarray(f2, np.uint8)
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("binary descriptors must be two (n, width) uint8 arrays of the same width")
    cap = max(1, min(len(a), len(b)) if symmetric else max(len(a), len(b)))
    out = np.empty((cap, 2), np.int32)
    n = C.c_int(0)
    check(_lib.load().osfm_match_hamming_ratio_ex(ctx.handle, _fptr(a, C.c_uint8), len(a), _fptr(b, C.c_uint8), len(b), a.shape[1], float(ratio),
                                                  int(symmetric), int(flags), _fptr(out, C.c_int32), cap, C.byref(n)), "osfm_match_hamming_ratio")
    return out[: n.value]


def _match_leaf(f1: np.ndarray, f2: np.ndarray, ratio: float, symmetric: bool, ctx=None, flags: int = 0) -> np.ndarray:
    assert f1.dtype.type == f2.dtype.type  # matching.py:737
    if f1.dtype.type == np.uint8:
        # matching.py:738-739: uint8 descriptors switch cv2 to Hamming; never reached by HAHOG/SIFT
        # (descriptors are float32 after loading, features.py:259-262)
        return _match_hamming_leaf(f1, f2, ratio, symmetric, ctx, flags)
    ctx = ctx or default_context()
    lib = _lib.load()
    a = np.ascontiguousarray(f1, np.float32)
    b = np.ascontiguousarray(f2, np.float32)
    dim = a.shape[1] if a.ndim == 2 and len(a) else (b.shape[1] if b.ndim == 2 and len(b) else 128)
    cap = max(1, min(len(a), len(b)) if symmetric else max(len(a), len(b)))
    out = np.empty((cap, 2), np.int32)
    n = C.c_int(0)
    check(
        lib.osfm_match_l2_ratio_ex(ctx.handle, _fptr(a, C.c_float), len(a), _fptr(b, C.c_float), len(b), dim,
                                   float(ratio), int(symmetric), int(flags), _fptr(out, C.c_int32), cap, C.byref(n)),
        "osfm_match_l2_ratio",
    )
    return out[: n.value]


def _match_guided_leaf(f1: np.ndarray, f2: np.ndarray, ratio: float, symmetric: bool, maskij: Optional[np.ndarray] = None,
                       bearings1: Optional[np.ndarray] = None, bearings2: Optional[np.ndarray] = None, R: Optional[np.ndarray] = None,
                       t: Optional[np.ndarray] = None, threshold: float = 0.0, ctx=None) -> np.ndarray:
    """``osfm_match_guided``: explicit ``maskij`` (n1 x n2) or the epipolar mask of (bearings, R, t, threshold) evaluated on the fly."""
    assert f1.dtype.type == f2.dtype.type
    if f1.dtype.type == np.uint8:
        raise NotImplementedError("uint8 descriptors take the reference's BruteForce-Hamming branch")
    ctx = ctx or default_context()
    a = np.ascontiguousarray(f1, np.float32).reshape(-1, 128)
    b = np.ascontiguousarray(f2, np.float32).reshape(-1, 128)
    cap = max(1, len(a))
    out = np.empty((cap, 2), np.int32)
    n = C.c_int(0)
    null_f, null_d = C.POINTER(C.c_float)(), C.POINTER(C.c_double)()
    if maskij is not None:
        m = np.ascontiguousarray(np.asarray(maskij) != 0, np.uint8)
        assert m.shape == (len(a), len(b)), "maskij must be len(f1) x len(f2)"
        args = (_fptr(m, C.c_uint8), null_f, null_f, null_d, null_d, 0.0)
    else:
        b1 = np.ascontiguousarray(bearings1, np.float32).reshape(-1, 3)  # matching.py:859-860: bearings are cast to float32
        b2 = np.ascontiguousarray(bearings2, np.float32).reshape(-1, 3)
        Rm = np.ascontiguousarray(R, np.float64).reshape(3, 3)
        tv = np.ascontiguousarray(t, np.float64).reshape(3)
        assert len(b1) == len(a) and len(b2) == len(b)
        args = (C.POINTER(C.c_uint8)(), _fptr(b1, C.c_float), _fptr(b2, C.c_float), _fptr(Rm, C.c_double), _fptr(tv, C.c_double), float(threshold))
    check(_lib.load().osfm_match_guided(ctx.handle, _fptr(a, C.c_float), len(a), _fptr(b, C.c_float), len(b), 128, *args, float(ratio),
                                        int(symmetric), _fptr(out, C.c_int32), cap, C.byref(n)), "osfm_match_guided")
    return out[: n.value]


def match_brute_force(f1: np.ndarray, f2: np.ndarray, config: Dict[str, Any], maskij: Optional[np.ndarray] = None,
                      ) -> List[Tuple[int, int]]:
    """Brute force matching and Lowe's ratio filtering (``matching.py:723-756``)."""
    if maskij is not None:
        m = _match_guided_leaf(f1, f2, _cfg(config, "lowes_ratio"), False, maskij)
    else:
        m = _match_leaf(f1, f2, _cfg(config, "lowes_ratio"), False)
    return [(int(a), int(b)) for a, b in m]


def match_brute_force_symmetric(fi: np.ndarray, fj: np.ndarray, config: Dict[str, Any],
                                maskij: Optional[np.ndarray] = None) -> List[Tuple[int, int]]:
    """Match with brute force in both directions and keep consistent matches (``matching.py:759-777``)."""
    if maskij is not None:
        m = _match_guided_leaf(fi, fj, _cfg(config, "lowes_ratio"), True, maskij)
    else:
        m = _match_leaf(fi, fj, _cfg(config, "lowes_ratio"), True)
    return [(int(a), int(b)) for a, b in m]


class ExactIndex:
    """What ``build_flann_index`` returns here: the image's descriptors, searched EXACTLY on the GPU.

    The reference builds ``cv2.flann_Index(features, algorithm=KMEANS, branching=8, iterations=10)`` (``features.py:638-674``) and
    searches it with ``checks=flann_checks`` (20): an approximate nearest-neighbour search whose k-means centres are drawn at
    random, so neither the index nor its results are reproducible run to run.  The drop-in keeps the interface (an opaque object
    handed to ``match_flann``; ``knnSearch(queries, 2)`` -> ``(indices, squared float32 distances)``) and answers every query with
    the true two nearest neighbours, i.e. the result the reference converges to as ``checks`` grows."""

    def __init__(self, features: np.ndarray):
        f = np.asarray(features)
        # uint8 bit strings (features.py:660-667 builds cv2's LSH index for them): kept as they are, searched exactly by Hamming distance
        self.features = np.ascontiguousarray(f, np.uint8) if f.dtype == np.uint8 else np.ascontiguousarray(f, np.float32).reshape(-1, 128)

    def __len__(self) -> int:
        return len(self.features)


def build_flann_index(descriptors: np.ndarray, config: Dict[str, Any]) -> ExactIndex:
    """``features.build_flann_index`` (``opensfm/features.py:638-674``): same arguments; the ``flann_*`` tuning keys of ``config``
    have no meaning for an exact search and are ignored."""
    return ExactIndex(descriptors)


def match_flann(index: ExactIndex, f2: np.ndarray, config: Dict[str, Any]) -> List[Tuple[int, int]]:
    """Match using the index of the first image and apply Lowe's ratio filter on squared distances (``matching.py:683-697``):
    for every descriptor of ``f2`` its two nearest neighbours in the index, kept iff ``d0 < float32(lowes_ratio ** 2) * d1``;
    returns ``(index feature, f2 feature)`` in the order of the ``f2`` features, as the reference."""
    m = _match_leaf(index.features, f2, _cfg(config, "lowes_ratio"), False, flags=_lib.MATCH_SQUARED_RATIO)
    return [(int(a), int(b)) for a, b in m]


def match_flann_symmetric(fi: np.ndarray, indexi: ExactIndex, fj: np.ndarray, indexj: ExactIndex, config: Dict[str, Any],
                          ) -> List[Tuple[int, int]]:
    """Match using FLANN semantics in both directions and keep consistent matches (``matching.py:700-720``)."""
    m = _match_leaf(indexi.features, indexj.features, _cfg(config, "lowes_ratio"), True, flags=_lib.MATCH_SQUARED_RATIO)
    return [(int(a), int(b)) for a, b in m]


def match_guided(d1: np.ndarray, d2: np.ndarray, bearings1: np.ndarray, bearings2: np.ndarray, relative_pose, config: Dict[str, Any],
                 ctx=None) -> np.ndarray:
    """The matching step of ``_match_descriptors_guided_impl`` (``matching.py:313-319``): ``compute_inliers_bearing_epipolar`` fused
    into ``match_brute_force_symmetric`` -- the n1 x n2 mask is evaluated inside the kernel, never stored.
    ``relative_pose``: ``pose2.relative_to(pose1)`` (anything with ``get_R_cam_to_world()`` and ``get_origin()``).  -> (K, 2) int32."""
    return _match_guided_leaf(d1, d2, _cfg(config, "lowes_ratio"), True, None, bearings1, bearings2, relative_pose.get_R_cam_to_world(),
                              relative_pose.get_origin(), _cfg(config, "guided_matching_threshold"), ctx)


def find_fundamental_ransac(p1: np.ndarray, p2: np.ndarray, threshold: float, confidence: float = 0.9999,
                            max_iters: int = 1000, ctx=None) -> Tuple[Optional[np.ndarray], np.ndarray]:
    """``cv2.findFundamentalMat(p1, p2, cv2.FM_RANSAC, threshold, confidence)`` -> (F or None, mask (n,1) uint8)."""
    ctx = ctx or default_context()
    lib = _lib.load()
    a = np.ascontiguousarray(p1, np.float64)
    b = np.ascontiguousarray(p2, np.float64)
    n = len(a)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    found = C.c_int(0)
    iters = C.c_int(0)
    check(
        lib.osfm_ransac_fundamental(ctx.handle, _fptr(a, C.c_double), _fptr(b, C.c_double), n, float(threshold),
                                    float(confidence), int(max_iters), _fptr(F, C.c_double), _fptr(mask, C.c_uint8),
                                    C.byref(found), C.byref(iters)),
        "osfm_ransac_fundamental",
    )
    find_fundamental_ransac.last_iters = iters.value  # type: ignore[attr-defined]
    return (F.reshape(3, 3) if found.value else None), mask[:n].reshape(-1, 1)


def robust_match_fundamental(p1: np.ndarray, p2: np.ndarray, matches: np.ndarray, config: Dict[str, Any],
                             ) -> Tuple[Any, np.ndarray]:
    """Filter matches by estimating the Fundamental matrix via RANSAC (``matching.py:780-802``)."""
    if len(matches) < 8:
        return np.array([]), np.array([])
    matches = np.asarray(matches)
    x1 = p1[matches[:, 0]][:, :2].copy()
    x2 = p2[matches[:, 1]][:, :2].copy()
    threshold = _cfg(config, "robust_matching_threshold")
    F, mask = find_fundamental_ransac(x1, x2, threshold, 0.9999)
    inliers = mask.ravel().nonzero()
    if F is None or F[2, 2] == 0.0:
        return F, np.array([])
    return F, matches[inliers]


# projection_type -> (OSFM_CAMERA_* id, attribute names in the native parameter order [projection][distortion][affine]);
# "cx" / "cy" stand for principal_point[0] / [1] (pygeometry.Camera, opensfm/src/geometry/python/pybind.cc:113-292)
_BEARING_MODELS = {
    "perspective": (0, ("k1", "k2", "focal")),
    "fisheye": (1, ("k1", "k2", "focal")),
    "brown": (2, ("k1", "k2", "k3", "p1", "p2", "focal", "aspect_ratio", "cx", "cy")),
    "fisheye_opencv": (3, ("k1", "k2", "k3", "k4", "focal", "aspect_ratio", "cx", "cy")),
    "fisheye62": (4, ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "focal", "aspect_ratio", "cx", "cy")),
    "fisheye624": (5, ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s0", "s1", "s2", "s3", "focal", "aspect_ratio", "cx", "cy")),
    "dual": (6, ("transition", "k1", "k2", "focal")),
    "radial": (7, ("k1", "k2", "focal", "aspect_ratio", "cx", "cy")),
    "simple_radial": (8, ("k1", "focal", "aspect_ratio", "cx", "cy")),
    "spherical": (9, ()),
    "equirectangular": (9, ()),
}


def camera_parameters(camera) -> Tuple[int, np.ndarray]:
    """(OSFM_CAMERA_* id, parameters in the native order) of a pygeometry.Camera-like object (attributes as in the reference)."""
    if camera.projection_type not in _BEARING_MODELS:
        raise NotImplementedError(f"projection type {camera.projection_type!r} is not on the GPU path")
    model, names = _BEARING_MODELS[camera.projection_type]

    def get(name: str) -> float:
        if name in ("cx", "cy"):
            return float(camera.principal_point[0 if name == "cx" else 1])
        return float(getattr(camera, name))

    return model, np.array([get(n) for n in names] + [0.0] * (16 - len(names)), np.float64)


def pixel_bearing_many(camera, points: np.ndarray, ctx=None) -> np.ndarray:
    """``camera.pixel_bearing_many(points)`` (pygeometry.Camera): (n, 2) normalised image coordinates -> (n, 3) unit bearings,
    every projection type of the reference (``camera_instances.h:154-160``)."""
    model, par = camera_parameters(camera)
    ctx = ctx or default_context()
    px = np.ascontiguousarray(np.asarray(points, np.float64)[:, :2])
    out = np.zeros((len(px), 3), np.float64)
    check(_lib.load().osfm_pixel_bearings(ctx.handle, model, _fptr(par, C.c_double), _fptr(px, C.c_double), len(px),
                                          _fptr(out, C.c_double)), "osfm_pixel_bearings")
    return out


def relpose_pairs(b1: np.ndarray, b2: np.ndarray, offsets: Sequence[int], threshold: float, mode: str = "match",
                  iterations: int = 1000, probability: float = 0.99, use_lo: bool = True, lo_iterations: int = 10,
                  refine_iterations: int = 10, ctx=None) -> Tuple[List[Dict[str, Any]], np.ndarray, float]:
    """Batched ``pyrobust.ransac_relative_pose`` (mode "ransac") or the body of ``robust_match_calibrated`` after the bearings
    (mode "match") for pairs whose correspondences are the slices offsets[p]:offsets[p+1] of b1 / b2.
    -> (per-pair dicts, inlier mask over all correspondences, kernel milliseconds)."""
    ctx = ctx or default_context()
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    off = np.ascontiguousarray(offsets, np.int64)
    n_pairs = len(off) - 1
    assert n_pairs >= 0 and len(b1) == len(b2) and (n_pairs == 0 or off[-1] == len(b1))
    prm = RelposeParams(float(threshold), float(probability), int(iterations), int(bool(use_lo)), int(lo_iterations), int(refine_iterations))
    res = (RelposeResult * max(n_pairs, 1))()
    mask = np.zeros(max(len(b1), 1), np.uint8)
    ms = C.c_double(0.0)
    check(_lib.load().osfm_relpose_pairs(ctx.handle, _fptr(b1, C.c_double), _fptr(b2, C.c_double), _fptr(off, C.c_int64), n_pairs,
                                         C.byref(prm), {"ransac": 0, "match": 1}[mode], res, _fptr(mask, C.c_uint8), C.byref(ms)),
          "osfm_relpose_pairs")
    out = []
    for p in range(n_pairs):
        r = res[p]
        out.append({"model": np.array(r.model).reshape(3, 4), "lo_model": np.array(r.lo_model).reshape(3, 4), "R": np.array(r.R).reshape(3, 3),
                    "t": np.array(r.t), "score": r.score, "iterations": r.iterations, "n_inliers": r.n_inliers})
    return out, mask[: len(b1)].astype(bool), ms.value


def robust_match_calibrated(p1: np.ndarray, p2: np.ndarray, camera1, camera2, matches: np.ndarray, config: Dict[str, Any],
                            ctx=None) -> np.ndarray:
    """Filter matches by estimating the Essential matrix via RANSAC (``matching.py:871-903``): same arguments, same return.
    One wavefront of ``relpose_pairs_kernel`` does the whole pair; use ``match_pairs_calibrated`` for batches."""
    if len(matches) < 8:
        return np.array([])
    matches = np.asarray(matches)
    x1 = np.asarray(p1)[matches[:, 0]][:, :2].copy()
    x2 = np.asarray(p2)[matches[:, 1]][:, :2].copy()
    b1 = pixel_bearing_many(camera1, x1, ctx)
    b2 = pixel_bearing_many(camera2, x2, ctx)
    _, mask, _ = relpose_pairs(b1, b2, [0, len(b1)], _cfg(config, "robust_matching_calib_threshold"), "match", 1000, 0.99, True, 10,
                               _cfg(config, "five_point_refine_match_iterations"), ctx)
    if not mask.any():
        return np.array([])
    return matches[mask]


def _is_pinhole(c) -> bool:
    """matching.py:918-925: cameras that take the fundamental-matrix branch."""
    return c.projection_type in ["perspective", "brown"] and c.k1 == 0.0 and c.k2 == 0.0


def robust_match(p1, p2, camera1, camera2, matches, config) -> np.ndarray:
    """``matching.py:906-929``: F-matrix path for undistorted perspective/brown cameras, E-matrix path otherwise."""
    if _is_pinhole(camera1) and _is_pinhole(camera2):
        return robust_match_fundamental(p1, p2, matches, config)[1]
    return robust_match_calibrated(p1, p2, camera1, camera2, matches, config)


def unfilter_matches(matches: np.ndarray, m1: np.ndarray, m2: np.ndarray) -> np.ndarray:
    """Given matches and masking arrays, get matches with un-masked indexes (``matching.py:932-936``)."""
    i1 = np.flatnonzero(m1)
    i2 = np.flatnonzero(m2)
    return np.array([(i1[match[0]], i2[match[1]]) for match in matches])


# --------------------------------------------------------------------------------------------
# batched path
# --------------------------------------------------------------------------------------------
class DescriptorStore:
    """All images' (masked) descriptors + keypoints resident in HBM.

    Replaces ``FeatureLoader.load_all_data`` + its LRU caches (``feature_loading.py:106-173``).
    ``descriptors``: list of (n_i, 128) arrays, float32 (integer-valued: exact int8 path; otherwise, e.g. root-SIFT: quantised
    candidates + float32 evaluation, same results) or uint8; or (n_i, 129) float32 arrays -- 128 integer-valued columns and the
    segmentation column ``feature_loading.py:123-155`` appends (``matching_use_segmentation``): such a store is matched on the exact
    kernel with the label term added in float32 where cv2 adds a 129th element;
    ``points``: list of (n_i, >=2) arrays (normalized image coordinates, ``features.py:324-331``).
    """

    def __init__(self, descriptors: Sequence[np.ndarray], points: Sequence[np.ndarray], ctx=None, hamming: bool = False):
        """``hamming``: the descriptors are uint8 bit strings (AKAZE MLDB, ORB; 1..64 bytes wide) to be matched by Hamming distance,
        as ``match_brute_force`` does for uint8 arrays (``matching.py:737-740``).  Without it a uint8 array is taken as 128 integer
        VALUES (HAHOG uchar files, which the reference converts to float32 when it loads them, ``features.py:169-170``)."""
        self.ctx = ctx or default_context()
        self.hamming = bool(hamming)
        lib = _lib.load()
        counts = np.asarray([len(d) for d in descriptors], np.int32)
        self.counts = counts
        self.n_images = len(counts)
        h = C.c_void_p()
        check(lib.osfm_store_create(self.ctx.handle, self.n_images, _fptr(counts, C.c_int32), C.byref(h)),
              "osfm_store_create")
        self.handle = h
        total = int(counts.sum())
        pts = np.zeros((max(total, 1), 2), np.float64)
        if total:
            pts[:total] = np.concatenate([np.asarray(p, np.float64)[:, :2].reshape(-1, 2) for p in points])
        widths = {np.asarray(d).shape[1] for d in descriptors if np.asarray(d).ndim == 2 and len(d)}
        seg = None
        if hamming:
            if len(widths) > 1 or any(np.asarray(d).dtype != np.uint8 for d in descriptors if len(d)):
                lib.osfm_store_destroy(h)
                self.handle = None
                raise ValueError("hamming=True takes uint8 descriptors of one width")
            w = widths.pop() if widths else 32
            desc = np.zeros((max(total, 1), w), np.uint8)
            if total:
                desc[:total] = np.concatenate([np.asarray(d, np.uint8).reshape(-1, w) for d in descriptors])
            rc = lib.osfm_store_upload_binary(h, _fptr(desc, C.c_uint8), int(w), _fptr(pts, C.c_double))
            if rc != 0:
                msg = lib.osfm_last_error().decode()
                lib.osfm_store_destroy(h)
                self.handle = None
                raise OsfmError(f"osfm_store_upload_binary failed ({rc}): {msg}")
            return
        if widths == {129}:  # segmentation in the descriptor: the last column goes into the store's label array
            seg = np.zeros(max(total, 1), np.float32)
            if total:
                seg[:total] = np.concatenate([np.asarray(d, np.float32).reshape(-1, 129)[:, 128] for d in descriptors])
            descriptors = [np.asarray(d, np.float32).reshape(-1, 129)[:, :128] for d in descriptors]
        elif widths - {128}:
            lib.osfm_store_destroy(h)
            self.handle = None
            raise ValueError(f"descriptors must all be 128 wide, or all 129 wide (segmentation column): got widths {sorted(widths)}")
        all_u8 = all(np.asarray(d).dtype == np.uint8 for d in descriptors)
        if all_u8:
            desc = np.zeros((max(total, 1), 128), np.uint8)
            if total:
                desc[:total] = np.concatenate([np.asarray(d).reshape(-1, 128) for d in descriptors])
            rc = lib.osfm_store_upload_u8(h, _fptr(desc, C.c_uint8), _fptr(pts, C.c_double))
        else:
            desc = np.zeros((max(total, 1), 128), np.float32)
            if total:
                desc[:total] = np.concatenate([np.asarray(d, np.float32).reshape(-1, 128) for d in descriptors])
            rc = lib.osfm_store_upload_f32(h, _fptr(desc, C.c_float), _fptr(pts, C.c_double))
        if rc == 0 and seg is not None:
            rc = lib.osfm_store_set_segmentation(h, _fptr(seg, C.c_float))
        if rc != 0:
            msg = lib.osfm_last_error().decode()
            lib.osfm_store_destroy(h)
            self.handle = None
            raise OsfmError(f"osfm_store_upload failed ({rc}): {msg}")

    @classmethod
    def from_packed(cls, desc: np.ndarray, pts: np.ndarray, offsets: np.ndarray, ctx=None, hamming: bool = False) -> "DescriptorStore":
        ds = [desc[offsets[i]: offsets[i + 1]] for i in range(len(offsets) - 1)]
        ps = [pts[offsets[i]: offsets[i + 1]] for i in range(len(offsets) - 1)]
        return cls(ds, ps, ctx, hamming)

    @property
    def device_bytes(self) -> int:
        return int(_lib.load().osfm_store_bytes(self.handle))

    def close(self) -> None:
        if getattr(self, "handle", None):
            _lib.load().osfm_store_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _matcher_flags(config: Optional[Dict[str, Any]]) -> int:
    """matcher_type -> OSFM_MATCH_* flags; raises for the matchers that are not on the GPU path."""
    mt = str(_cfg(config, "matcher_type")).upper()
    if mt == "BRUTEFORCE":
        return 0
    if mt == "FLANN":  # FLANN semantics on the exact search (see the module docstring)
        return _lib.MATCH_SQUARED_RATIO
    raise NotImplementedError(f"matcher_type {mt!r} is not on the GPU path (BRUTEFORCE, FLANN)")


def make_params(config: Optional[Dict[str, Any]] = None, robust: bool = True) -> MatchParams:
    p = MatchParams()
    _lib.load().osfm_match_params_default(C.byref(p))
    p.flags = _matcher_flags(config)
    p.lowes_ratio = float(_cfg(config, "lowes_ratio"))
    p.symmetric = int(bool(_cfg(config, "symmetric_matching")))
    p.robust = int(robust)
    p.robust_matching_min_match = int(_cfg(config, "robust_matching_min_match"))
    p.robust_matching_threshold = float(_cfg(config, "robust_matching_threshold"))
    return p


class DeviceMatchGraph:
    """Result of ``match_pairs(..., keep_device=True)``: the per-pair counts on the host AND in HBM, the concatenated match rows in
    HBM only (``OSFM_MATCH_KEEP_DEVICE``).  The multi-GPU exchange step (``dist.all_gather_match_graph_device``) all-gathers from
    these buffers; ``fetch()`` brings the rows to the host with one copy.  Owns the C result until ``close()``."""

    def __init__(self, lib, handle):
        self._lib, self.handle = lib, handle
        self.n_pairs = int(lib.osfm_result_num_pairs(handle))
        self.total = int(lib.osfm_result_total_matches(handle))
        self.device = int(lib.osfm_result_device(handle))
        self.counts = np.empty(max(self.n_pairs, 1), np.int32)
        check(lib.osfm_result_fetch(handle, _fptr(self.counts, C.c_int32), None), "osfm_result_fetch")
        self.counts = self.counts[: self.n_pairs]
        dc, dm = C.c_void_p(), C.c_void_p()
        check(lib.osfm_result_dev_ptrs(handle, C.byref(dc), C.byref(dm)), "osfm_result_dev_ptrs")
        self.d_counts, self.d_matches = int(dc.value or 0), int(dm.value or 0)

    def _tensor(self, ptr: int, n: int):
        """a torch view (no copy) of n int32 at a device address, through __cuda_array_interface__"""
        import torch

        if n == 0 or not ptr:
            return torch.empty(0, dtype=torch.int32, device=torch.device("cuda", self.device))

        class _Buf:
            pass

        b = _Buf()
        b.__cuda_array_interface__ = {"shape": (max(n, 1),), "typestr": "<i4", "data": (ptr, False), "version": 2}
        b._keepalive = self
        return torch.as_tensor(b, device=torch.device("cuda", self.device))[:n]

    def counts_tensor(self):
        return self._tensor(self.d_counts, self.n_pairs)

    def matches_tensor(self):
        """flat int32 tensor of 2 * total entries (row k = entries 2k, 2k + 1)"""
        return self._tensor(self.d_matches, 2 * self.total)

    def fetch(self) -> Tuple[np.ndarray, np.ndarray]:
        matches = np.empty((max(self.total, 1), 2), np.int32)
        check(self._lib.osfm_result_fetch(self.handle, None, _fptr(matches, C.c_int32)), "osfm_result_fetch")
        return self.counts, matches[: self.total]

    def close(self) -> None:
        if self.handle:
            self._lib.osfm_result_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def match_pairs(store: DescriptorStore, pairs: np.ndarray, config: Optional[Dict[str, Any]] = None,
                robust: bool = True, timings: Optional[MatchTimings] = None, keep_device: bool = False):
    """Run ``matching.match`` (``matching.py:563-634``) for every pair, on the GPU.

    Returns ``(counts, matches)``: ``counts[p]`` matches for pair ``p`` (0 where the reference returns
    ``[]``), ``matches`` the concatenated ``(K, 2)`` int32 arrays in pair order.  ``keep_device``: a ``DeviceMatchGraph``
    instead (the match rows stay in HBM for the multi-GPU exchange step).
    """
    lib = _lib.load()
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    prm = make_params(config, robust)
    if keep_device:
        prm.flags |= _lib.MATCH_KEEP_DEVICE
    res = C.c_void_p()
    tm = timings if timings is not None else MatchTimings()
    check(lib.osfm_match_pairs(store.ctx.handle, store.handle, _fptr(pairs, C.c_int32), len(pairs), C.byref(prm),
                               C.byref(res), C.byref(tm)), "osfm_match_pairs")
    if keep_device:
        return DeviceMatchGraph(lib, res)
    return _fetch_result(lib, res)


class _ResultOwner:
    """keeps a match result of the library alive while numpy arrays view its host buffers"""

    def __init__(self, lib, res):
        self._lib, self._res = lib, res

    def __del__(self):
        if self._res is not None:
            self._lib.osfm_result_destroy(self._res)
            self._res = None


def _fetch_result(lib, res) -> Tuple[np.ndarray, np.ndarray]:
    """(counts, matches) as VIEWS of the buffers the call left in host memory (osfm_result_host_ptrs): a second copy of the match rows
    into fresh pages was 1 - 13 ms of a 196 ms step, depending on the host.  The result is destroyed when the last view is collected."""
    owner = _ResultOwner(lib, res)
    n = int(lib.osfm_result_num_pairs(res))
    total = int(lib.osfm_result_total_matches(res))
    pc, pm = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    check(lib.osfm_result_host_ptrs(res, C.byref(pc), C.byref(pm)), "osfm_result_host_ptrs")

    def view(ptr, count):
        if count == 0:
            return np.zeros(0, np.int32)
        buf = (C.c_int32 * count).from_address(C.addressof(ptr.contents))
        buf._osfm_owner = owner  # the array's base is this ctypes object: the result lives as long as any view of it
        return np.frombuffer(buf, np.int32)

    return view(pc, n), view(pm, 2 * total).reshape(-1, 2)


def match_pairs_calibrated(store: DescriptorStore, pairs: np.ndarray, cameras: Sequence[Any], points: Optional[Sequence[np.ndarray]] = None,
                           config: Optional[Dict[str, Any]] = None, timings: Optional[MatchTimings] = None) -> Tuple[np.ndarray, np.ndarray]:
    """``matching.match`` for pairs that take the calibrated branch of ``robust_match`` (``matching.py:563-634,871-929``), one
    ``osfm_match_pairs_calibrated`` call: descriptor matching, the ``robust_matching_min_match`` gate, bearings of the matched
    features, the LO-RANSAC + refinement rounds, the gate again -- device-resident from the descriptors to the inlier lists.
    ``cameras[i]`` is the camera of image i of the store (``points`` is not needed any more: the keypoints live in the store).
    Same return convention as ``match_pairs``."""
    lib = _lib.load()
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    models = np.zeros(max(store.n_images, 1), np.int32)
    params = np.zeros((max(store.n_images, 1), 16), np.float64)
    for i in range(store.n_images):
        models[i], params[i] = camera_parameters(cameras[i])
    prm = make_params(config, robust=True)
    rp = RelposeParams(float(_cfg(config, "robust_matching_calib_threshold")), 0.99, 1000, 1, 10, int(_cfg(config, "five_point_refine_match_iterations")))
    res = C.c_void_p()
    tm = timings if timings is not None else MatchTimings()
    check(lib.osfm_match_pairs_calibrated(store.ctx.handle, store.handle, _fptr(models, C.c_int32), _fptr(params, C.c_double),
                                          _fptr(pairs, C.c_int32), len(pairs), C.byref(prm), C.byref(rp), C.byref(res), C.byref(tm)),
          "osfm_match_pairs_calibrated")
    return _fetch_result(lib, res)


def match_pairs_guided(store: DescriptorStore, pairs: np.ndarray, bearings: Sequence[np.ndarray], relative_poses: Sequence[Any],
                       config: Optional[Dict[str, Any]] = None, robust: bool = True, cameras: Optional[Sequence[Any]] = None,
                       timings: Optional[MatchTimings] = None) -> Tuple[np.ndarray, np.ndarray]:
    """``matching.match`` with guided matching (``matching.py:204-207,260-337,563-634``) for every pair, one ``osfm_match_pairs_guided``
    call: the epipolar mask of the pair's relative pose, ``match_brute_force[_symmetric]`` under it, the gates and the robust stage
    (fundamental-matrix RANSAC, or -- with ``cameras`` -- the calibrated branch), device-resident.
    ``bearings[i]``: (n_i, 3) bearings of image i of the store; ``relative_poses[p]``: ``pose2.relative_to(pose1)`` of pair p
    (anything with ``get_R_cam_to_world()`` / ``get_origin()``, or a flat 12-vector R|t).  Same return convention as ``match_pairs``."""
    lib = _lib.load()
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    b = np.ascontiguousarray(np.concatenate([np.asarray(x, np.float32).reshape(-1, 3) for x in bearings]) if len(bearings) else np.zeros((0, 3)),
                             np.float32)
    if len(b) != int(np.sum(store.counts)):
        raise ValueError("bearings do not match the store's feature counts")
    rt = np.zeros((max(len(pairs), 1), 12), np.float64)
    for p, rel in enumerate(relative_poses):
        if hasattr(rel, "get_R_cam_to_world"):
            rt[p, :9] = np.asarray(rel.get_R_cam_to_world(), np.float64).reshape(9)
            rt[p, 9:] = np.asarray(rel.get_origin(), np.float64).reshape(3)
        else:
            rt[p] = np.asarray(rel, np.float64).reshape(12)
    prm = make_params(config, robust)
    res = C.c_void_p()
    tm = timings if timings is not None else MatchTimings()
    thr = float(_cfg(config, "guided_matching_threshold"))
    if cameras is not None:
        models = np.zeros(max(store.n_images, 1), np.int32)
        params = np.zeros((max(store.n_images, 1), 16), np.float64)
        for i in range(store.n_images):
            models[i], params[i] = camera_parameters(cameras[i])
        rp = RelposeParams(float(_cfg(config, "robust_matching_calib_threshold")), 0.99, 1000, 1, 10, int(_cfg(config, "five_point_refine_match_iterations")))
        check(lib.osfm_match_pairs_guided(store.ctx.handle, store.handle, _fptr(b, C.c_float), _fptr(pairs, C.c_int32), len(pairs), _fptr(rt, C.c_double),
                                          thr, C.byref(prm), _fptr(models, C.c_int32), _fptr(params, C.c_double), C.byref(rp), C.byref(res), C.byref(tm)),
              "osfm_match_pairs_guided")
    else:
        check(lib.osfm_match_pairs_guided(store.ctx.handle, store.handle, _fptr(b, C.c_float), _fptr(pairs, C.c_int32), len(pairs), _fptr(rt, C.c_double),
                                          thr, C.byref(prm), None, None, None, C.byref(res), C.byref(tm)), "osfm_match_pairs_guided")
    return _fetch_result(lib, res)


def split_matches(counts: np.ndarray, matches: np.ndarray) -> List[np.ndarray]:
    off = np.concatenate([[0], np.cumsum(counts)])
    return [matches[off[i]: off[i + 1]] for i in range(len(counts))]


# --------------------------------------------------------------------------------------------
# ad-hoc filters (``matching_use_filters``, matching.py:939-1064): per-match predicates on the keypoints between the descriptor stage and
# the gates / robust stage -- host numpy, as in the reference (a few hundred matches per pair)
# --------------------------------------------------------------------------------------------
SEGMENTATION_IN_DESCRIPTOR_MULT = 35  # feature_loading.py:16-18


def _segmentation_of(features_data):
    """FeaturesData.get_segmentation (features.py:74-80): the per-feature class labels, or None"""
    if hasattr(features_data, "get_segmentation"):
        return features_data.get_segmentation()
    semantic = getattr(features_data, "semantic", None)
    return getattr(semantic, "segmentation", None) if semantic else None


def _is_panorama(projection_type: str) -> bool:
    return projection_type in ("equirectangular", "spherical")  # pygeometry.Camera.is_panorama


def apply_adhoc_filters(data, matches: np.ndarray, im1: str, camera1, p1: np.ndarray, im2: str, camera2, p2: np.ndarray) -> np.ndarray:
    """``apply_adhoc_filters`` (matching.py:939-957): static matches, panorama poles, Vermont and BlackVue watermarks, in that order"""
    m = np.asarray(matches, np.int64).reshape(-1, 2)
    p1, p2 = np.asarray(p1, np.float64), np.asarray(p2, np.float64)
    # _non_static_matches (:960-981): drop matches that do not move, unless that would drop more than 85 % of them
    d = p1[m[:, 0], :2] - p2[m[:, 1], :2]
    keep = d[:, 0] ** 2 + d[:, 1] ** 2 >= 0.001 ** 2
    if not (1 - int(keep.sum()) / max(len(m), 1) > 0.85):
        m = m[keep]
    # _not_on_pano_poles_matches (:983-1007)
    pano1, pano2 = _is_panorama(camera1.projection_type), _is_panorama(camera2.projection_type)
    if pano1 or pano2:
        y1, y2 = p1[m[:, 0], 1], p2[m[:, 1], 1]
        ok = np.ones(len(m), bool)
        if pano1:
            ok &= (-0.125 < y1) & (y1 < 0.125)
        if pano2:
            ok &= (-0.125 < y2) & (y2 < 0.125)
        m = m[ok]
    # _not_on_vermont_watermark (:1010-1035), _not_on_blackvue_watermark (:1038-1064)
    meta1, meta2 = data.load_exif(im1), data.load_exif(im2)
    if meta1["make"] == "VTrans_Camera" and meta1["model"] == "VTrans_Camera":
        m = m[p1[m[:, 0], 1] > -0.255]
    if meta2["make"] == "VTrans_Camera" and meta2["model"] == "VTrans_Camera":
        m = m[p2[m[:, 1], 1] > -0.255]
    if meta1["make"].lower() == "blackvue":
        m = m[p1[m[:, 0], 1] < 0.263]
    if meta2["make"].lower() == "blackvue":
        m = m[p2[m[:, 1], 1] < 0.263]
    return m.astype(np.int32)


def _filter_then_robust(data, config, pairs, ipairs, found, pts, cams) -> List[np.ndarray]:
    """matching.py:576-634 with ``matching_use_filters``: descriptor-stage matches -> ad-hoc filters -> gate -> robust_match -> gate.
    The filters sit between two device stages, so the robust stage runs pair by pair through the leaves here."""
    min_match = int(_cfg(config, "robust_matching_min_match"))
    out: List[np.ndarray] = []
    for (im1, im2), (a, b), m in zip(pairs, ipairs, found):
        res = np.zeros((0, 2), np.int32)
        if len(m):
            m = apply_adhoc_filters(data, m, im1, cams[a], pts[a], im2, cams[b], pts[b])
        if len(m) >= min_match:
            rm = np.asarray(robust_match(pts[a], pts[b], cams[a], cams[b], m, config)).reshape(-1, 2)
            if len(rm) >= min_match and len(rm) > 0:
                res = rm.astype(np.int32)
        out.append(res)
    return out


def match_images_with_pairs(data, config_override: Dict[str, Any], exifs: Dict[str, Any],
                            pairs: List[Tuple[str, str]], poses=None) -> Dict[Tuple[str, str], np.ndarray]:
    """Perform pair matchings given pairs (``matching.py:63-98``), all pairs in one GPU batch.

    ``data`` must offer the ``DataSetBase`` methods the reference uses on this path:
    ``config``, ``load_camera_models()``, ``load_features(image)`` (``.points``, ``.descriptors``) and
    optionally ``load_features_mask(image, points)`` (``feature_loading.py:61-71``).
    Only the configuration the GPU path implements is accepted: ``matcher_type`` BRUTEFORCE, FLANN (exact search with FLANN's
    squared-ratio semantics) or WORDS (``data.load_words(image)`` supplies the closest vocabulary words); with ``poses`` the
    descriptor stage is the guided (epipolar-masked) matcher, batched (``match_pairs_guided``);
    pairs of undistorted perspective/brown cameras take the fused matcher + fundamental-matrix RANSAC launch, every other
    pair the calibrated route (``match_pairs_calibrated``).
    """
    config = dict(data.config)
    config.update(config_override)
    # guided matching always runs the brute-force matcher, whatever matcher_type says (matching.py:270-279 logs a warning and switches)
    use_words = str(_cfg(config, "matcher_type")).upper() == "WORDS" and not poses
    if not use_words and not poses:
        _matcher_flags(config)  # raises for matchers that are not on the GPU path
    # matching.py:281,356 + feature_loading.py:118-155: the segmentation label of every feature as a 129th descriptor column
    use_segmentation = bool(config.get("matching_use_segmentation"))
    if use_segmentation:
        if not config.get("hahog_normalize_to_uchar") or config.get("feature_type") != "HAHOG":
            raise RuntimeError("Semantic segmentation in descriptor only supported for HAHOG UCHAR descriptors")  # feature_loading.py:126-133
        if use_words:
            raise NotImplementedError("matcher_type WORDS with matching_use_segmentation is not on the GPU path (the resident WordsStore holds 128-D rows)")
    use_filters = bool(config.get("matching_use_filters"))
    # cv2.findFundamentalMat switches to LMedS below 15 correspondences; the batched fundamental-matrix launch implements RANSAC only (the
    # leaf find_fundamental_ransac has both: osfm_ransac_fundamental).  With robust_matching_min_match < 15 -- not the default 20 -- a pair on
    # the fundamental-matrix route can reach that branch, so those pairs take the batched descriptor stage and then the robust stage pair
    # by pair through the leaf (round 6; rounds 1-5 raised).  Pairs on the calibrated route never reach cv2.
    lmeds_reachable = int(_cfg(config, "robust_matching_min_match")) < 15

    def _robust_through_leaves(sel_ipairs, found) -> List[np.ndarray]:
        min_match = int(_cfg(config, "robust_matching_min_match"))
        res = []
        for (a, b), m in zip(sel_ipairs, found):
            r = np.zeros((0, 2), np.int32)
            if len(m) >= min_match:
                rm = np.asarray(robust_match(pts[a], pts[b], cams[a], cams[b], m, config)).reshape(-1, 2)
                if len(rm) >= min_match and len(rm) > 0:
                    r = rm.astype(np.int32)
            res.append(r)
        return res
    cameras = data.load_camera_models()
    images = sorted({im for pair in pairs for im in pair})
    index = {im: k for k, im in enumerate(images)}
    descs, pts, masks, cams = [], [], [], []
    wordlists: List[np.ndarray] = []
    for im in images:
        cam = cameras[exifs[im]["camera"]]
        if not _is_pinhole(cam) and cam.projection_type not in _BEARING_MODELS:
            raise NotImplementedError(f"camera of {im}: unknown projection type {cam.projection_type!r}")
        cams.append(cam)
        fd = data.load_features(im)
        # an image without (usable) features is a count-0 image: every pair it takes part in returns np.array([]) like the
        # reference's dummy result (matching.py:359-374: features_data is None / fewer than 2 points / no descriptors)
        points = np.zeros((0, 3)) if fd is None or fd.points is None else np.asarray(fd.points)
        desc = None if fd is None else fd.descriptors
        mask = None
        if points.ndim != 2 or len(points) == 0 or desc is None:
            points, desc = np.zeros((0, 3)), np.zeros((0, 128), np.float32)
        else:
            desc = np.asarray(desc)
            if hasattr(data, "load_features_mask"):
                mask = np.asarray(data.load_features_mask(im, points[:, :2]), bool)
                points, desc = points[mask], desc[mask]
            if len(points) < 2:  # the masked set decides (load_all_data(masked=True), matching.py:354-366)
                points, desc = np.zeros((0, 3)), np.zeros((0, 128), np.float32)
            elif use_segmentation:  # FeatureLoader._add_segmentation_in_descriptor (feature_loading.py:123-155), on the masked set
                seg = _segmentation_of(fd)
                if seg is not None:
                    seg = np.asarray(seg)
                    seg = seg[mask] if mask is not None else seg
                    desc = np.concatenate((np.asarray(desc, np.float32), (np.array([seg]).T).astype(np.float32)), axis=1)
                    desc[:, -1] *= SEGMENTATION_IN_DESCRIPTOR_MULT
        if len(points) > _lib.MAX_FEATURES:
            raise NotImplementedError(f"image {im} has {len(points)} features after masking; the GPU path holds at most "
                                      f"{_lib.MAX_FEATURES} per image (OSFM_MAX_FEATURES)")
        descs.append(desc)
        pts.append(points)
        masks.append(mask)
        if use_words:  # feature_loader.load_words(data, image, masked=True) (feature_loader.py:96-107)
            w = np.zeros((0, 1), np.int32)
            if len(points):
                w = data.load_words(im)
                if w is None:  # matching.py:380-381: no words -> the dummy result for every pair of this image: a count-0 image
                    descs[-1], pts[-1], masks[-1] = np.zeros((0, 128), np.float32), np.zeros((0, 3)), None
                    w = np.zeros((0, 1), np.int32)
                else:
                    w = np.asarray(w)
                    w = (w[mask] if mask is not None else w).astype(np.int32).reshape(len(points), -1)
            wordlists.append(w)
    ipairs = np.asarray([(index[a], index[b]) for a, b in pairs], np.int32).reshape(-1, 2)
    per_pair: List[np.ndarray] = [np.zeros((0, 2), np.int32)] * len(ipairs)
    # uint8 descriptors (AKAZE MLDB, ORB): match_brute_force takes cv2's BruteForce-Hamming branch (matching.py:737-740); the data set hands
    # HAHOG uchar descriptors over as float32 (features.py:169-170), like the reference's loader
    live = [np.asarray(d) for d in descs if len(d)]
    hamming = bool(live) and all(d.dtype == np.uint8 for d in live)
    if hamming and live[0].shape[1] > 64:
        # cv2's BruteForce-Hamming has no width cap, the binary store holds 64 bytes (AKAZE MLDB 61, ORB 32): wider uint8 rows are HAHOG /
        # SIFT uchar levels a caller did not convert to float32 as the reference's loader does (features.py:169-170) -- matched as values
        if config.get("feature_type", "HAHOG").upper() in ("AKAZE", "ORB"):
            raise NotImplementedError("binary descriptors of %d bytes: the GPU Hamming matcher holds 1..64 bytes per descriptor" % live[0].shape[1])
        hamming = False
        descs = [np.asarray(d, np.float32) for d in descs]
    if hamming and (poses or use_words or use_segmentation):
        raise NotImplementedError("binary (uint8) descriptors are on the GPU path for matcher_type BRUTEFORCE / FLANN without poses / segmentation")
    if hamming:
        width = live[0].shape[1]
        descs = [np.asarray(d, np.uint8).reshape(-1, width) if len(d) else np.zeros((0, width), np.uint8) for d in descs]
    if poses:  # guided matching (matching.py:204-207,260-337,576-634): every pair in one osfm_match_pairs_guided call per robust branch
        ctx = default_context()
        bearings = [pixel_bearing_many(cams[k], np.asarray(pts[k], np.float64)[:, :2], ctx) if len(pts[k]) else np.zeros((0, 3))
                    for k in range(len(images))]
        cfg_g = dict(config)
        cfg_g["matcher_type"] = "BRUTEFORCE"  # matching.py:272-279: guided matching always runs the brute-force matcher ...
        cfg_g["symmetric_matching"] = True    # ... and match_brute_force_symmetric (matching.py:319)
        rels = [poses[im2].relative_to(poses[im1]) for im1, im2 in pairs]
        store = DescriptorStore(descs, pts)
        try:
            pin = np.array([_is_pinhole(cams[a]) and _is_pinhole(cams[b]) for a, b in ipairs], bool)
            if use_filters:  # matching.py:323-334: the filters follow the guided descriptor stage
                counts, matches = match_pairs_guided(store, ipairs, bearings, rels, cfg_g, robust=False)
                per_pair = _filter_then_robust(data, config, pairs, ipairs, split_matches(counts, matches), pts, cams)
                pin = np.zeros(0, bool)
            for sel, camarg in ((pin, None), (~pin, cams)):
                if not sel.any():
                    continue
                idx = np.flatnonzero(sel)
                leaves = camarg is None and lmeds_reachable  # fundamental-matrix route with cv2's LMedS branch in reach
                counts, matches = match_pairs_guided(store, ipairs[idx], bearings, [rels[p] for p in idx], cfg_g, robust=not leaves, cameras=camarg)
                found = split_matches(counts, matches)
                for p, m in zip(idx, _robust_through_leaves(ipairs[idx], found) if leaves else found):
                    per_pair[p] = m
        finally:
            store.close()
    elif use_words:  # matching.py:388-398: match_words[_symmetric], then the robust stage pair by pair
        from . import words as _words

        nw = max([w.shape[1] for w in wordlists if len(w)] or [1])
        wstore = _words.WordsStore([np.asarray(d, np.float32).reshape(-1, 128) for d in descs],
                                   [w if len(w) else np.zeros((0, nw), np.int32) for w in wordlists])
        try:
            found, _ = _words.match_words_pairs(wstore, ipairs, config, symmetric=bool(_cfg(config, "symmetric_matching")))
        finally:
            wstore.close()
        min_match = int(_cfg(config, "robust_matching_min_match"))
        if use_filters:
            per_pair = _filter_then_robust(data, config, pairs, ipairs, found, pts, cams)
            found = []
        for p, ((a, b), m) in enumerate(zip(ipairs, found)):
            if len(m) < min_match:
                continue
            rm = np.asarray(robust_match(pts[a], pts[b], cams[a], cams[b], m, config))
            if len(rm) >= min_match and len(rm) > 0:
                per_pair[p] = rm.astype(np.int32)
    else:
        store = DescriptorStore(descs, pts, hamming=True) if hamming else DescriptorStore(descs, pts)  # all descriptors resident in HBM
        try:
            pin = np.array([_is_pinhole(cams[a]) and _is_pinhole(cams[b]) for a, b in ipairs], bool)
            if use_filters:  # matching.py:399-411: descriptor stage for every pair, filters on the host, robust stage through the leaves
                counts, matches = match_pairs(store, ipairs, config, robust=False)
                per_pair = _filter_then_robust(data, config, pairs, ipairs, split_matches(counts, matches), pts, cams)
                pin = np.zeros(0, bool)
            if pin.any():
                counts, matches = match_pairs(store, ipairs[pin], config, robust=not lmeds_reachable)
                found = split_matches(counts, matches)
                for p, m in zip(np.flatnonzero(pin), _robust_through_leaves(ipairs[pin], found) if lmeds_reachable else found):
                    per_pair[p] = m
            if (~pin).any():
                counts, matches = match_pairs_calibrated(store, ipairs[~pin], cams, pts, config)
                for p, m in zip(np.flatnonzero(~pin), split_matches(counts, matches)):
                    per_pair[p] = m
        finally:
            store.close()
    out: Dict[Tuple[str, str], np.ndarray] = {}
    for (im1, im2), m in zip(pairs, per_pair):
        if len(m) == 0:
            out[im1, im2] = np.array([])  # matching.py:598,634
            continue
        m1, m2 = masks[index[im1]], masks[index[im2]]
        if m1 is not None and m2 is not None:
            m = unfilter_matches(m, m1, m2)
        out[im1, im2] = np.array(m, dtype=int)
    return out


def match_images(data, config_override: Dict[str, Any], ref_images: List[str], cand_images: List[str], **preselection_kwargs):
    """``matching.match_images`` (matching.py:27-60): pair preselection from the metadata (``preselection.match_candidates_from_metadata``,
    the union of the enabled strategies), then every selected pair in one batch.  Returns ``(matches, preselection_report)``.
    ``preselection_kwargs``: the histogram dictionaries / callables of the BoW and VLAD strategies (see ``preselection``)."""
    from . import preselection

    all_images = list(set(ref_images + cand_images))
    exifs = {im: data.load_exif(im) for im in all_images}
    pairs, preport = preselection.match_candidates_from_metadata(list(ref_images), list(cand_images), exifs, data, config_override, **preselection_kwargs)
    return match_images_with_pairs(data, config_override, exifs, pairs), preport


# --------------------------------------------------------------------------------------------
# the match graph on disk (``dataset.py:344-392``, ``matching.py:128-157``)
# --------------------------------------------------------------------------------------------
def _owner_of_pair(im1: str, im2: str, reference_images) -> Tuple[str, str]:
    """(image whose matches file stores the pair, the other image): the first image when it is a reference image, else the second
    (``matching.py:139-150``); raises like the reference when neither is."""
    if im1 in reference_images:
        return im1, im2
    if im2 in reference_images:
        return im2, im1
    raise RuntimeError("Couldn't save matches for {}. No image found in images_ref.".format((im1, im2)))


def save_matches(data, images_ref: Sequence[str], matched_pairs: Dict[Tuple[str, str], Any]) -> None:
    """``matching.save_matches`` (``matching.py:128-157``): one ``data.save_matches(image, {other image: matches})`` call per reference
    image, every pair filed under the reference image that owns it (also the reference images that own nothing get their call)."""
    reference_images = set(images_ref)
    per_image: Dict[str, Dict[str, Any]] = {im: {} for im in images_ref}
    for (im1, im2), m in matched_pairs.items():
        owner, other = _owner_of_pair(im1, im2, reference_images)
        per_image[owner][other] = m
    for image, its_matches in per_image.items():
        data.save_matches(image, its_matches)


def write_matches_files(data_path: str, images: Sequence[str], pairs: np.ndarray, counts: np.ndarray, matches: np.ndarray,
                        images_ref: Optional[Sequence[str]] = None) -> List[str]:
    """The batched result of ``match_pairs`` (image indices ``pairs``, ``counts``, concatenated ``matches``) straight into the
    reference's on-disk format, ``<data_path>/matches/<image>_matches.pkl.gz`` = gzip(pickle({other image: (K, 2) int array})),
    what ``DataSet.save_matches`` writes and ``DataSet.load_matches`` / ``find_matches`` read (``dataset.py:344-404``), with the
    grouping of ``matching.save_matches``: the pair is stored under its first image when that is a reference image, else under the
    second.  Pairs without matches store ``np.array([])`` as the reference does (``matching.py:598,634``).  -> written files."""
    import gzip
    import os
    import pickle

    ref = list(images) if images_ref is None else list(images_ref)
    ref_set = set(ref)
    per_image: Dict[str, Dict[str, np.ndarray]] = {im: {} for im in ref}
    for (a, b), m in zip(np.asarray(pairs).reshape(-1, 2), split_matches(np.asarray(counts), np.asarray(matches).reshape(-1, 2))):
        im1, im2 = images[int(a)], images[int(b)]
        owner, other = _owner_of_pair(im1, im2, ref_set)
        per_image[owner][other] = np.array(m, dtype=int) if len(m) else np.array([])
    out_dir = os.path.join(data_path, "matches")
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for im, d in per_image.items():
        path = os.path.join(out_dir, "{}_matches.pkl.gz".format(im))
        with open(path, "wb") as fw, gzip.GzipFile(fileobj=fw, mode="w") as fzip:
            pickle.dump(d, fzip)
        written.append(path)
    return written

