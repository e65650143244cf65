"""Pair preselection for matching (``opensfm/pairs_selection.py:581-687`` ``match_candidates_from_metadata`` and the strategies it
unions), with the neighbour searches and descriptor distances on the MI355X (``csrc/words.hip``):

  * by GPS distance  -- ``match_candidates_by_distance`` (pairs_selection.py:154-212): ``osfm_knn_points`` / ``osfm_radius_points`` in
    place of ``scipy.spatial.cKDTree.query(k, distance_upper_bound)``;
  * by capture time  -- ``match_candidates_by_time`` (:527-558): the same search on a one-dimensional point set;
  * by order         -- ``match_candidates_by_order`` (:561-578): index arithmetic;
  * by VLAD distance -- ``match_candidates_with_vlad`` (:351-440): GPS preemption, then ``osfm_vlad_distances``;
  * by BoW distance  -- ``match_candidates_with_bow`` (:285-349, :690-730): GPS preemption, then ``osfm_bow_distances`` (L1 over the word
    histograms, summed in numpy's order);
  * graph rounds     -- ``match_candidates_by_graph`` (:220-282): Delaunay triangulations of the (jittered) positions; scipy on the
    host as in the reference -- a few thousand points, nothing for the GPU.

Exifs / reference / dataset objects are the reference's (duck-typed: ``exifs[image]["gps"]["latitude"]``,
``reference.to_topocentric(lat, lon, alt)``, ``data.config``).  Ties (equal distances at the k-th place) resolve to the lower candidate
index; scipy's kd-tree leaves them unspecified."""
import ctypes as C
import logging
import math
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np

from . import _lib, words
from ._lib import check, default_context

logger = logging.getLogger(__name__)
_EVERYTHING = 99999999  # the reference's stand-in for "no limit" (pairs_selection.py:173-174)
DEFAULT_Z, MAXIMUM_Z, SAMPLE_Z = 1.0, 8000, 100

CONFIG_DEFAULTS = {  # opensfm/config.py
    "matching_gps_distance": 150, "matching_gps_neighbors": 0, "matching_time_neighbors": 0, "matching_order_neighbors": 0,
    "matching_bow_neighbors": 0, "matching_bow_gps_distance": 0, "matching_bow_gps_neighbors": 0, "matching_bow_other_cameras": False,
    "matching_vlad_neighbors": 0, "matching_vlad_gps_distance": 0, "matching_vlad_gps_neighbors": 0, "matching_vlad_other_cameras": False,
    "matching_graph_rounds": 0,
}


def has_gps_info(exif: Dict[str, Any]) -> bool:
    return bool(exif) and "gps" in exif and "latitude" in exif["gps"] and "longitude" in exif["gps"]


def sorted_pair(im1: str, im2: str) -> Tuple[str, str]:
    return (im1, im2) if im1 < im2 else (im2, im1)


def _fp(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---- device searches ----
def knn_points(candidates: np.ndarray, queries: np.ndarray, k: int, max_distance: float = math.inf, ctx=None) -> Tuple[np.ndarray, np.ndarray]:
    """the k nearest candidates (n x 3) of every query (m x 3) within ``max_distance``, ascending: (distances m x k, indices m x k);
    missing neighbours have index -1 and distance inf (cKDTree reports index n)"""
    ctx = ctx or default_context()
    cand = np.ascontiguousarray(candidates, np.float64).reshape(-1, 3)
    qry = np.ascontiguousarray(queries, np.float64).reshape(-1, 3)
    k = int(k)
    dist = np.full((len(qry), k), np.inf)
    idx = np.full((len(qry), k), -1, np.int32)
    check(_lib.load().osfm_knn_points(ctx.handle, _fp(cand, C.c_double), len(cand), _fp(qry, C.c_double), len(qry), k, float(max_distance),
                                      _fp(dist, C.c_double), _fp(idx, C.c_int32)), "osfm_knn_points")
    return dist, idx


def radius_points(candidates: np.ndarray, queries: np.ndarray, max_distance: float, ctx=None) -> np.ndarray:
    """boolean m x n: candidate within ``max_distance`` of the query"""
    ctx = ctx or default_context()
    cand = np.ascontiguousarray(candidates, np.float64).reshape(-1, 3)
    qry = np.ascontiguousarray(queries, np.float64).reshape(-1, 3)
    nwords = (len(cand) + 31) // 32
    mask = np.zeros((len(qry), max(nwords, 1)), np.uint32)
    check(_lib.load().osfm_radius_points(ctx.handle, _fp(cand, C.c_double), len(cand), _fp(qry, C.c_double), len(qry), float(max_distance),
                                         _fp(mask, C.c_uint32)), "osfm_radius_points")
    bits = np.unpackbits(mask.view(np.uint8), axis=1, bitorder="little")
    return bits[:, : len(cand)].astype(bool)


_RADIUS_CHUNK = 4096  # queries per device call of the radius search: the dense hit table of a chunk is chunk x candidates bytes on the host


def _radius_lists(points: np.ndarray, queries: np.ndarray, max_distance: float) -> List[np.ndarray]:
    """per query the candidates within range, a chunk of queries at a time (host memory stays O(chunk x candidates + hits): the
    reference asks its k-d tree one image at a time)"""
    out: List[np.ndarray] = []
    for q0 in range(0, len(queries), _RADIUS_CHUNK):
        hit = radius_points(points, queries[q0:q0 + _RADIUS_CHUNK], max_distance)
        out.extend(np.flatnonzero(row) for row in hit)
    return out


def _neighbours(points: np.ndarray, queries: np.ndarray, k_of_query: np.ndarray, max_distance: float) -> List[np.ndarray]:
    """per query the indices ``tree.query(point, k, distance_upper_bound)`` would return (without the "missing" entries)"""
    n = len(points)
    if len(queries) == 0 or n == 0:
        return [np.zeros(0, np.int64) for _ in range(len(queries))]
    kmax = int(k_of_query.max()) if len(k_of_query) else 0
    if kmax <= 0:
        return [np.zeros(0, np.int64) for _ in range(len(queries))]
    if int(k_of_query.min()) >= n:  # EVERY query may take every point within range: no ranking needed
        return _radius_lists(points, queries, max_distance)
    # a query whose own k covers every candidate still gets the nearest min(k, n): rank with min(kmax, n), cut each row to its k
    _, idx = knn_points(points, queries, min(kmax, n), max_distance)
    return [row[:k][row[:k] >= 0] for row, k in zip(idx, k_of_query)]


# ---- representative points (pairs_selection.py:37-151) ----
def _rotation_from_opk(omega: float, phi: float, kappa: float) -> np.ndarray:
    """opensfm/geometry.py:72-91 (world-to-camera from omega / phi / kappa, z forward)"""
    def rx(a):
        return np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])

    def ry(a):
        return np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])

    def rz(a):
        return np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])

    flip = np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
    return flip @ rz(-kappa) @ ry(-phi) @ rx(-omega)


def get_gps_point(exif: Dict[str, Any], reference) -> Tuple[np.ndarray, np.ndarray]:
    """pairs_selection.py:35-46: the GPS position at altitude 0 and a vertical viewing direction"""
    gps = exif["gps"]
    return np.array(reference.to_topocentric(gps["latitude"], gps["longitude"], 0)), np.array([0, 0, 1])


def get_gps_opk_point(exif: Dict[str, Any], reference) -> Tuple[np.ndarray, np.ndarray]:
    """pairs_selection.py:58-74: the same position with the camera's viewing axis from omega / phi / kappa, scaled to |z| = DEFAULT_Z"""
    opk = exif["opk"]
    z_axis = _rotation_from_opk(math.radians(opk["omega"]), math.radians(opk["phi"]), math.radians(opk["kappa"]))[2]
    return get_gps_point(exif, reference)[0], z_axis / ((1.0 if z_axis[2] > 0.0 else -1.0) * z_axis[2]) * DEFAULT_Z


def find_best_altitude(origin: Dict[str, np.ndarray], directions: Dict[str, np.ndarray]) -> float:
    """pairs_selection.py:77-105: the altitude in [0, MAXIMUM_Z] that makes the X / Y footprint smallest -- the squared diagonal sampled
    every SAMPLE_Z, the extremum of the fitted parabola; DEFAULT_Z when it is negative (diverging views)"""
    o, d = np.array(list(origin.values())), np.array(list(directions.values()))
    zs = np.arange(1, MAXIMUM_Z, SAMPLE_Z)
    size = []
    for z in zs:
        s = o + d / DEFAULT_Z * z
        size.append((s[:, 0].max() - s[:, 0].min()) ** 2 + (s[:, 1].max() - s[:, 1].min()) ** 2)
    c = np.polyfit(zs, size, 2)
    altitude = -c[1] / (2 * c[0])
    return DEFAULT_Z if altitude < 0 else altitude


def get_representative_points(images: Sequence[str], exifs: Dict[str, Any], reference) -> Dict[str, np.ndarray]:
    """pairs_selection.py:108-151: a topocentric point per image with GPS: its position at altitude 0, pushed along the viewing direction to
    the altitude that makes the footprint smallest when any image carries an orientation (opk)"""
    origin: Dict[str, np.ndarray] = {}
    direction: Dict[str, np.ndarray] = {}
    oriented = False
    for image in images:
        exif = exifs[image]
        if not ("gps" in exif and "latitude" in exif["gps"] and "longitude" in exif["gps"]):
            continue
        if "ypr" in exif:
            raise RuntimeError(f"GPS / OPK / YPR {(True, 'opk' in exif, True)} tag combination unsupported")
        if "opk" in exif:
            oriented = True
            origin[image], direction[image] = get_gps_opk_point(exif, reference)
        else:
            o, d = get_gps_point(exif, reference)
            origin[image], direction[image] = np.asarray(o, float), np.asarray(d, float)
    if not oriented:
        return origin
    altitude = find_best_altitude(origin, direction)
    return {k: origin[k] + direction[k] / DEFAULT_Z * altitude for k in images}


def match_candidates_by_distance(images_ref: List[str], images_cand: List[str], exifs: Dict[str, Any], reference, max_neighbors: int,
                                 max_distance: float) -> Set[Tuple[str, str]]:
    """pairs_selection.py:154-212.  Faithful to the detail that the point array has one row per distinct image of cand + ref but only
    the candidate rows are filled: the zero rows can take neighbour slots and are then dropped."""
    if len(images_cand) == 0 or (max_neighbors <= 0 and max_distance <= 0):
        return set()
    max_neighbors = max_neighbors or _EVERYTHING
    max_distance = max_distance or float(_EVERYTHING)
    k = min(len(images_cand), max_neighbors)
    rep = get_representative_points(list(images_cand) + list(images_ref), exifs, reference)
    missing = abs(len(rep) - len(set(images_cand) | set(images_ref)))
    if missing > 0:
        logger.warning("Couldn't fetch %d images. Returning NO pairs.", missing)
        return set()
    points = np.zeros((len(rep), 3))
    for i, image in enumerate(images_cand):
        points[i] = rep[image]
    in_cand = set(images_cand)
    queries = np.array([rep[im] for im in images_ref]).reshape(-1, 3)
    want = np.array([k + 1 if im in in_cand else k for im in images_ref], np.int64)
    pairs: Set[Tuple[str, str]] = set()
    for image_ref, js in zip(images_ref, _neighbours(points, queries, want, max_distance)):
        for j in js:
            if j < len(images_cand) and images_cand[j] != image_ref:
                pairs.add(sorted_pair(image_ref, images_cand[j]))
    return pairs


def match_candidates_by_time(images_ref: List[str], images_cand: List[str], exifs: Dict[str, Any], max_neighbors: int) -> Set[Tuple[str, str]]:
    """pairs_selection.py:527-558: nearest capture times (a kd-tree on one coordinate there, the same point search here)"""
    if max_neighbors <= 0 or len(images_cand) == 0:
        return set()
    k = min(len(images_cand), max_neighbors)
    t0 = float(exifs[images_cand[0]]["capture_time"])  # times are ~1e9 s: search on offsets, the differences stay exact in doubles
    points = np.zeros((len(images_cand), 3))
    points[:, 0] = [float(exifs[im]["capture_time"]) - t0 for im in images_cand]
    queries = np.zeros((len(images_ref), 3))
    queries[:, 0] = [float(exifs[im]["capture_time"]) - t0 for im in images_ref]
    in_cand = set(images_cand)
    want = np.array([k + 1 if im in in_cand else k for im in images_ref], np.int64)
    want = np.minimum(want, len(images_cand))  # cKDTree pads with "missing" beyond the number of points
    pairs: Set[Tuple[str, str]] = set()
    for image_ref, js in zip(images_ref, _neighbours(points, queries, want, math.inf)):
        for j in js:
            if images_cand[j] != image_ref:
                pairs.add(sorted_pair(image_ref, images_cand[j]))
    return pairs


def match_candidates_by_order(images_ref: List[str], images_cand: List[str], max_neighbors: int) -> Set[Tuple[str, str]]:
    """pairs_selection.py:561-578: the (max_neighbors + 1) // 2 images before and after, by position"""
    if max_neighbors <= 0:
        return set()
    n = (max_neighbors + 1) // 2
    pairs = set()
    for i, image_ref in enumerate(images_ref):
        for j in range(max(0, i - n), min(len(images_cand), i + n)):
            if images_cand[j] != image_ref:
                pairs.add(sorted_pair(image_ref, images_cand[j]))
    return pairs


def preempt_candidates(images_ref, images_cand, exifs, reference, max_gps_neighbors: int, max_gps_distance: float):
    """pairs_selection.py:444-471: the candidates each reference image keeps after the GPS filter, and the images whose data is needed"""
    kept: Dict[str, List[str]] = {im: images_cand for im in images_ref}
    if max_gps_distance > 0 or max_gps_neighbors > 0:
        refs = set(images_ref)
        kept = {}
        for a, b in match_candidates_by_distance(images_ref, images_cand, exifs, reference, max_gps_neighbors, max_gps_distance):
            if a in refs:
                kept.setdefault(a, []).append(b)
            if b in refs:
                kept.setdefault(b, []).append(a)
    need = set(kept)
    for v in kept.values():
        need.update(v)
    return kept, need


def pairs_from_neighbors(image: str, exifs, distances, order, other, max_neighbors: int) -> Dict[Tuple[str, str], float]:
    """pairs_selection.py:764-795: the closest max_neighbors of the same camera AND the closest max_neighbors of other cameras"""
    same: List[Tuple[str, float]] = []
    rest: List[Tuple[str, float]] = []
    for i in order:
        bucket = same if exifs[other[i]]["camera"] == exifs[image]["camera"] else rest
        if len(bucket) < max_neighbors:
            bucket.append((other[i], distances[i]))
        if len(same) + len(rest) >= 2 * max_neighbors:
            break
    return {tuple(sorted((image, im2))): d for im2, d in same + rest}


def construct_pairs(results, max_neighbors: int, exifs, enforce_other_cameras: bool) -> Dict[Tuple[str, str], float]:
    """pairs_selection.py:474-489"""
    pairs: Dict[Tuple[str, str], float] = {}
    for im, distances, other in results:
        order = np.argsort(distances)
        if enforce_other_cameras:
            pairs.update(pairs_from_neighbors(im, exifs, distances, order, other, max_neighbors))
        else:
            for i in order[:max_neighbors]:
                pairs[sorted_pair(im, other[i])] = distances[i]
    return pairs


def match_candidates_with_vlad(images_ref, images_cand, exifs, reference, max_neighbors: int, max_gps_distance: float, max_gps_neighbors: int,
                               enforce_other_cameras: bool, histograms: Dict[str, np.ndarray],
                               compute_histograms: Optional[Callable[[Set[str]], Dict[str, np.ndarray]]] = None) -> Dict[Tuple[str, str], float]:
    """pairs_selection.py:351-440.  ``histograms`` holds the (normalised) VLAD descriptors already known; ``compute_histograms(images)``
    supplies the missing ones (the reference loads features through the dataset and calls ``vlad.vlad_histograms``; see
    ``vlad_histogram`` below for the device version of that computation)."""
    if max_neighbors <= 0:
        return {}
    kept, need = preempt_candidates(images_ref, images_cand, exifs, reference, max_gps_neighbors, max_gps_distance)
    if len(kept) == 0:
        logger.warning("Couldn't preempt any candidate with GPS, using ALL %d as candidates", len(images_cand))
        kept = {image: images_cand for image in images_ref}
        need = set(images_ref) | set(images_cand)
    need = {im for im in need if im not in histograms}
    if need:
        if compute_histograms is None:
            raise ValueError("VLAD descriptors of %d images are missing and no compute_histograms callable was given" % len(need))
        histograms.update(compute_histograms(need))
    results = [words.vlad_distances(im, cands, histograms) for im, cands in kept.items()]
    return construct_pairs(results, max_neighbors, exifs, enforce_other_cameras)


def match_candidates_with_bow(images_ref, images_cand, exifs, reference, max_neighbors: int, max_gps_distance: float, max_gps_neighbors: int,
                              enforce_other_cameras: bool, histograms: Dict[str, np.ndarray],
                              compute_histograms: Optional[Callable[[Set[str]], Dict[str, np.ndarray]]] = None) -> Dict[Tuple[str, str], float]:
    """pairs_selection.py:285-349.  ``histograms`` holds the BoW histograms already known; ``compute_histograms(images)`` supplies the
    missing ones (the reference's ``load_histograms``: ``bows.histogram(words[:, 0])`` for images with more than 8 masked words, see
    ``words.bow_histogram``); images without a histogram drop out, as in ``bow_distances``."""
    if max_neighbors <= 0:
        return {}
    kept, need = preempt_candidates(images_ref, images_cand, exifs, reference, max_gps_neighbors, max_gps_distance)
    need = {im for im in need if im not in histograms}
    if need and compute_histograms is not None:
        histograms.update(compute_histograms(need))
    results = [words.bow_distances(im, cands, histograms) for im, cands in kept.items()]
    return construct_pairs(results, max_neighbors, exifs, enforce_other_cameras)


def match_candidates_by_graph(images_ref: List[str], images_cand: List[str], exifs: Dict[str, Any], reference, rounds: int,
                              rng: Optional[np.random.Generator] = None) -> Set[Tuple[str, str]]:
    """pairs_selection.py:220-282: the edges of the Delaunay triangulation of the images' X/Y positions, plus those of ``rounds``
    triangulations of positions jittered by up to the median edge length.  Host code over scipy, like the reference (which draws the
    jitter from numpy's global generator; pass ``rng`` for a reproducible one)."""
    from scipy import spatial

    if len(images_cand) < 4 or rounds < 1:
        return set()
    cand, ref = set(images_cand), set(images_ref)
    images = list(cand | ref)
    rep = get_representative_points(images, exifs, reference)
    xy = np.array([rep[im][0:2] for im in images], float).reshape(-1, 2)

    def edges(simplices):
        for tri in simplices:
            for u, v in ((tri[0], tri[1]), (tri[0], tri[2]), (tri[1], tri[2])):
                a, b = images[u], images[v]
                if a != b and ((a in cand and b in ref) or (b in cand and a in ref)):
                    yield sorted_pair(a, b), (u, v)

    try:
        first = spatial.Delaunay(xy).simplices
    except spatial.QhullError:  # flat initial simplex: let qhull rescale the input
        first = spatial.Delaunay(xy, qhull_options="Qbb Qc Qz Q12 QbB").simplices
    pairs: Set[Tuple[str, str]] = set()
    lengths = []
    for pair, (u, v) in edges(first):
        pairs.add(pair)
        lengths.append(float(np.hypot(*(xy[u] - xy[v]))))
    scale = np.median(lengths)
    for _ in range(rounds):
        jitter = (rng.random(xy.shape) if rng is not None else np.random.rand(*xy.shape)) * scale
        for pair, _uv in edges(spatial.Delaunay(xy + jitter).simplices):
            pairs.add(pair)
    return pairs


def vlad_histogram(features: np.ndarray, vlad_words: np.ndarray) -> Optional[np.ndarray]:
    """``VladCache.vlad_histogram`` without the dataset plumbing (vlad.py:66-78): unnormalised VLAD on the device, SSR normalisation"""
    if vlad_words.shape[1] != features.shape[1] or vlad_words.dtype != features.dtype:
        return None
    return words.signed_square_root_normalize(words.unnormalized_vlad(features, vlad_words))


def ordered_pairs(pairs: Iterable[Tuple[str, str]], images_ref: List[str]) -> List[Tuple[str, str]]:
    """pairs_selection.py:798-826: every pair once, as (im1, im2) with im1 in images_ref, walking from image to image so that
    consecutive pairs share an image.  The reference starts from ``set.pop()`` (arbitrary); here from the smallest name."""
    per_image: Dict[str, List[str]] = {}
    for a, b in sorted(pairs):
        per_image.setdefault(a, []).append(b)
        per_image.setdefault(b, []).append(a)
    done: Set[Tuple[str, str]] = set()
    out: List[Tuple[str, str]] = []
    remaining = sorted(set(images_ref), reverse=True)
    left = set(remaining)
    current = remaining.pop() if remaining else None
    if current is not None:
        left.discard(current)
    while current is not None:
        following = None
        for other in per_image.get(current, []):
            if (other, current) not in done and (current, other) not in done:
                done.add((current, other))
                out.append((current, other))
                if following is None and other in left:
                    following = other
                    left.discard(other)
        if following is None:
            while remaining and remaining[-1] not in left:
                remaining.pop()
            if remaining:
                following = remaining.pop()
                left.discard(following)
        current = following
    return out


def match_candidates_from_metadata(images_ref: List[str], images_cand: List[str], exifs: Dict[str, Any], data, config_override: Dict[str, Any],
                                   vlad_histograms: Optional[Dict[str, np.ndarray]] = None,
                                   compute_vlad_histograms: Optional[Callable[[Set[str]], Dict[str, np.ndarray]]] = None,
                                   bow_histograms: Optional[Dict[str, np.ndarray]] = None,
                                   compute_bow_histograms: Optional[Callable[[Set[str]], Dict[str, np.ndarray]]] = None):
    """``pairs_selection.match_candidates_from_metadata`` (pairs_selection.py:581-687): the union of the enabled strategies, as a list
    of pairs (im1, im2) with im1 in images_ref, and the per-strategy report."""
    cfg = dict(CONFIG_DEFAULTS)
    cfg.update(getattr(data, "config", {}) or {})
    cfg.update(config_override or {})
    max_distance, gps_neighbors = cfg["matching_gps_distance"], cfg["matching_gps_neighbors"]
    graph_rounds, time_neighbors, order_neighbors = cfg["matching_graph_rounds"], cfg["matching_time_neighbors"], cfg["matching_order_neighbors"]
    bow_neighbors, vlad_neighbors = cfg["matching_bow_neighbors"], cfg["matching_vlad_neighbors"]
    if hasattr(data, "init_reference"):
        data.init_reference()
    reference = data.load_reference()
    if not all(map(has_gps_info, exifs.values())):
        if gps_neighbors != 0:
            logger.warning("Not all images have GPS info. Disabling matching_gps_neighbors.")
        gps_neighbors = max_distance = graph_rounds = 0
    images_ref.sort()
    d: Set[Tuple[str, str]] = set()
    t: Set[Tuple[str, str]] = set()
    o: Set[Tuple[str, str]] = set()
    v: Dict[Tuple[str, str], float] = {}
    g: Set[Tuple[str, str]] = set()
    b: Dict[Tuple[str, str], float] = {}
    if max_distance == gps_neighbors == time_neighbors == order_neighbors == bow_neighbors == vlad_neighbors == graph_rounds == 0:
        pairs = {sorted_pair(i, j) for i in images_ref for j in images_cand if i != j}  # nothing enabled: match everything
    else:
        d = match_candidates_by_distance(images_ref, images_cand, exifs, reference, gps_neighbors, max_distance)
        t = match_candidates_by_time(images_ref, images_cand, exifs, time_neighbors)
        o = match_candidates_by_order(images_ref, images_cand, order_neighbors)
        v = match_candidates_with_vlad(images_ref, images_cand, exifs, reference, vlad_neighbors, cfg["matching_vlad_gps_distance"],
                                       cfg["matching_vlad_gps_neighbors"], cfg["matching_vlad_other_cameras"],
                                       vlad_histograms if vlad_histograms is not None else {}, compute_vlad_histograms)
        g = match_candidates_by_graph(images_ref, images_cand, exifs, reference, graph_rounds)
        b = match_candidates_with_bow(images_ref, images_cand, exifs, reference, bow_neighbors, cfg["matching_bow_gps_distance"],
                                      cfg["matching_bow_gps_neighbors"], cfg["matching_bow_other_cameras"],
                                      bow_histograms if bow_histograms is not None else {}, compute_bow_histograms)
        pairs = d | g | t | o | set(b) | set(v)
    report = {"num_pairs_distance": len(d), "num_pairs_graph": len(g), "num_pairs_time": len(t), "num_pairs_order": len(o),
              "num_pairs_bow": len(b), "num_pairs_vlad": len(v)}
    return ordered_pairs(pairs, images_ref), report
