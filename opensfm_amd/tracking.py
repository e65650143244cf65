"""Tracks creation on the MI355X -- host-side mirror of ``opensfm/tracking.py``.

``create_tracks_manager`` (tracking.py:68-140) links matches into tracks with a pure-Python
union-find over ``(image, feature)`` tuples; at 10 000 images it is the next serial bottleneck after
matching.  Here the match graph (already resident as arrays after
``matching.match_pairs`` + all-gather) goes to ``osfm_tracks_create``: GPU connected components +
two radix sorts, and comes back with exactly the reference's track numbering.

No CPU fallback: a missing ``libosfm_mi355.so`` or GPU raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import Context, OsfmError, check, default_context, load


def _ip(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def edges_from_match_graph(pairs: np.ndarray, counts: np.ndarray, matches: np.ndarray, node_offsets: np.ndarray):
    """Flatten the match graph (pair list, per-pair counts, concatenated (i, j) arrays -- the layout
    ``matching.match_pairs`` / ``dist.all_gather_match_graph`` return) into global node ids, in the
    reference's union order (pair by pair, match by match; tracking.py:84-86)."""
    pairs = np.asarray(pairs, np.int64).reshape(-1, 2)
    counts = np.asarray(counts, np.int64)
    matches = np.asarray(matches, np.int64).reshape(-1, 2)
    rep = np.repeat(np.arange(len(pairs)), counts)
    node_offsets = np.asarray(node_offsets, np.int64)
    if len(node_offsets) and node_offsets[-1] >= 2 ** 31:  # node ids are int32 at the C ABI: refuse instead of wrapping
        raise OsfmError("more than 2^31 features: node ids do not fit the int32 of osfm_tracks_create")
    ea = node_offsets[pairs[rep, 0]] + matches[:, 0]
    eb = node_offsets[pairs[rep, 1]] + matches[:, 1]
    return np.ascontiguousarray(ea, np.int32), np.ascontiguousarray(eb, np.int32)


def create_tracks_arrays(edge_a: np.ndarray, edge_b: np.ndarray, node_offsets: np.ndarray, min_length: int = 2,
                         ctx: Optional[Context] = None, timings: Optional[dict] = None):
    """-> (n_tracks, obs_track, obs_image, obs_feature); observations grouped by track in the
    reference's track order, members in the reference's insertion order."""
    ctx = ctx or default_context()
    lib = load()
    off = np.ascontiguousarray(node_offsets, np.int64)
    if len(off) and off[-1] >= 2 ** 31:
        raise OsfmError("more than 2^31 features: node ids do not fit the int32 of osfm_tracks_create")
    for e in (edge_a, edge_b):  # the cast below must not wrap an out-of-range id into a valid one
        e = np.asarray(e)
        if e.size and (e.min() < 0 or e.max() >= 2 ** 31):
            raise OsfmError("match references a node id outside [0, 2^31)")
    ea = np.ascontiguousarray(edge_a, np.int32)
    eb = np.ascontiguousarray(edge_b, np.int32)
    if len(off) == 0:
        off = np.zeros(1, np.int64)
    h = C.c_void_p()
    check(lib.osfm_tracks_create(ctx.handle, _ip(ea, C.c_int32), _ip(eb, C.c_int32), len(ea), _ip(off, C.c_int64), len(off) - 1,
                                 int(min_length), C.byref(h)), "osfm_tracks_create")
    try:
        n_tracks = int(lib.osfm_tracks_num_tracks(h))
        n_obs = int(lib.osfm_tracks_num_observations(h))
        ot, oi, of = (np.empty(max(n_obs, 1), np.int32) for _ in range(3))
        check(lib.osfm_tracks_fetch(h, _ip(ot, C.c_int32), _ip(oi, C.c_int32), _ip(of, C.c_int32)), "osfm_tracks_fetch")
        if timings is not None:
            timings["ms_device"] = float(lib.osfm_tracks_device_ms(h))
    finally:
        lib.osfm_tracks_destroy(h)
    return n_tracks, ot[:n_obs], oi[:n_obs], of[:n_obs]


class TracksTable:
    """The part of ``pymap.TracksManager`` that ``create_tracks_manager`` fills: observations
    ``(image, track_id, feature_id)`` with the reference's string track ids; per-image lookups."""

    def __init__(self, images: Sequence[str], obs_track, obs_image, obs_feature, n_tracks: int):
        self.images = list(images)
        self.obs_track, self.obs_image, self.obs_feature = obs_track, obs_image, obs_feature
        self.n_tracks = n_tracks

    def num_tracks(self) -> int:
        return self.n_tracks

    def num_observations(self) -> int:
        return len(self.obs_track)

    def get_track_ids(self) -> List[str]:
        return [str(t) for t in range(self.n_tracks)]

    def get_shot_ids(self) -> List[str]:
        return [self.images[i] for i in np.unique(self.obs_image)]

    def get_track_observations(self, track_id: str) -> Dict[str, int]:
        sel = self.obs_track == int(track_id)
        return {self.images[i]: int(f) for i, f in zip(self.obs_image[sel], self.obs_feature[sel])}

    def get_shot_observations(self, image: str) -> Dict[str, int]:
        sel = self.obs_image == self.images.index(image)
        return {str(t): int(f) for t, f in zip(self.obs_track[sel], self.obs_feature[sel])}


def create_tracks_manager(features: Dict[str, np.ndarray], colors, segmentations, instances,
                          matches: Dict[Tuple[str, str], Sequence[Tuple[int, int]]], min_length: int, depths=None,
                          depth_is_radial: bool = True, depth_std_deviation: float = 1.0, ctx: Optional[Context] = None) -> TracksTable:
    """Same arguments as the reference (tracking.py:68-78).  Images are numbered in the order they
    first appear in ``matches``; feature counts come from ``features`` (or from the matches when an
    image has no features entry, which the reference skips when emitting observations,
    tracking.py:105-107)."""
    images: List[str] = []
    index: Dict[str, int] = {}
    for im1, im2 in matches:
        for im in (im1, im2):
            if im not in index:
                index[im] = len(images)
                images.append(im)
    counts = np.zeros(len(images) + 1, np.int64)
    for (im1, im2), m in matches.items():
        m = np.asarray(m, np.int64).reshape(-1, 2)
        if len(m):
            counts[index[im1] + 1] = max(counts[index[im1] + 1], m[:, 0].max() + 1)
            counts[index[im2] + 1] = max(counts[index[im2] + 1], m[:, 1].max() + 1)
    for im, f in features.items():
        if im in index:
            counts[index[im] + 1] = max(counts[index[im] + 1], len(f))
    off = np.cumsum(counts)
    ea, eb = [], []
    for (im1, im2), m in matches.items():
        m = np.asarray(m, np.int64).reshape(-1, 2)
        ea.append(off[index[im1]] + m[:, 0])
        eb.append(off[index[im2]] + m[:, 1])
    ea = np.concatenate(ea) if ea else np.zeros(0, np.int64)
    eb = np.concatenate(eb) if eb else np.zeros(0, np.int64)
    n_tracks, ot, oi, of = create_tracks_arrays(ea, eb, off, min_length, ctx)
    keep = np.array([images[i] in features for i in oi], bool) if len(oi) else np.zeros(0, bool)
    return TracksTable(images, ot[keep], oi[keep], of[keep], n_tracks)
