"""Seeded synthetic inputs for the matching + bundle-adjustment hot path.

Recipes follow SURVEY.md 8(d), which in turn mirrors the reference's own synthetic fixtures:

* descriptors: HAHOG-style integer-valued float32 in [0, 255]
  (``opensfm/features.py:526-534``: ``sqrt`` -> ``x362`` -> ``clip(0,255)`` -> ``round``; stored
  uint8, loaded back as float32 ``features.py:259-262``);
* keypoints: normalized image coordinates ``(px + 0.5 - w/2) / max(w, h)``
  (``features.py:324-331``), focal 0.85, 1 px noise at 2000 px;
* BA scenes: one shared perspective camera ``[k1, k2, focal]``, sigma 0.004
  (``opensfm/synthetic_data/synthetic_generator.py:404``), 1 px noise, outliers.

Everything here is input generation; none of it is on the timed path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np


def _hahog_like(rng: np.random.Generator, n: int, dim: int = 128) -> np.ndarray:
    """Non-negative sparse-ish vectors -> L1 normalise -> sqrt -> x362 -> clip -> round (uint8)."""
    v = np.abs(rng.standard_normal((n, dim), dtype=np.float32))
    keep = rng.random((n, dim), dtype=np.float32) < 0.6
    v *= keep
    v[:, 0] += 1e-3  # never all-zero
    v /= v.sum(axis=1, keepdims=True)
    d = np.sqrt(v) * 362.0
    return np.clip(np.rint(d), 0, 255).astype(np.uint8)


@dataclass
class MatchingScene:
    """A set of images: ragged descriptors/keypoints packed back to back."""

    desc: np.ndarray  # (sum_n, 128) uint8, integer-valued descriptors
    pts: np.ndarray  # (sum_n, 2) float64 normalized image coordinates
    offsets: np.ndarray  # (n_images + 1,) int64 row offsets
    point_ids: np.ndarray  # (sum_n,) int64 scene point id, -1 for distractors
    cam_R: Optional[np.ndarray] = None  # (n_images, 3, 3) world-to-camera rotations of the views
    cam_o: Optional[np.ndarray] = None  # (n_images, 3) camera origins

    @property
    def n_images(self) -> int:
        return len(self.offsets) - 1

    def image(self, i: int) -> Tuple[np.ndarray, np.ndarray]:
        a, b = self.offsets[i], self.offsets[i + 1]
        return self.desc[a:b], self.pts[a:b]

    def desc_f32(self, i: int) -> np.ndarray:
        a, b = self.offsets[i], self.offsets[i + 1]
        return self.desc[a:b].astype(np.float32)


def all_pairs(n_images: int) -> np.ndarray:
    """All i<j pairs in lexicographic order: what ``pairs_selection.py:624-644`` emits when every
    selector is 0 (exhaustive matching)."""
    i, j = np.triu_indices(n_images, k=1)
    return np.stack([i, j], axis=1).astype(np.int32)


def make_matching_scene(
    n_images: int,
    n_features: int = 2000,
    seed: int = 42,
    distractor_frac: float = 0.3,
    desc_noise: float = 4.0,
    px_noise: float = 1.0 / 2000.0,
    ragged: bool = False,
    dim: int = 128,
) -> MatchingScene:
    """Street scene: cameras advance along +x looking at a cloud of points at depth 4..12.

    Neighbouring cameras share scene points (true matches with a consistent epipolar geometry);
    distant cameras share none, so most pairs of an exhaustive run fail the min-match gate, as in
    a real exhaustive matching job.
    """
    rng = np.random.default_rng(seed)
    focal = 0.85
    step = 1.0
    n_real_target = int(round(n_features * (1.0 - distractor_frac)))
    # density so that ~2x n_real_target points are visible per camera
    depth_lo, depth_hi = 4.0, 12.0
    x_extent = n_images * step + 2 * depth_hi
    # visible volume per camera: integrate over depth of (z/f) * (0.75 z/f) -> area in (x, y)
    vol = (1.0 / focal) * (0.75 / focal) * (depth_hi**3 - depth_lo**3) / 3.0
    total_vol = x_extent * (2 * 0.375 * depth_hi / focal) * (depth_hi - depth_lo)
    n_points = int(2.0 * n_real_target * total_vol / vol)
    X = np.empty((n_points, 3))
    X[:, 0] = rng.uniform(-depth_hi, n_images * step + depth_hi, n_points)
    X[:, 2] = rng.uniform(depth_lo, depth_hi, n_points)
    X[:, 1] = rng.uniform(-0.375 * depth_hi / focal, 0.375 * depth_hi / focal, n_points)
    order = np.argsort(X[:, 0])
    X = X[order]
    base_desc = _hahog_like(rng, n_points, dim)

    descs: List[np.ndarray] = []
    ptss: List[np.ndarray] = []
    ids: List[np.ndarray] = []
    offsets = [0]
    cam_Rs, cam_os = [], []
    for i in range(n_images):
        n_i = n_features
        if ragged:
            n_i = int(rng.integers(max(2, n_features // 2), n_features + 1))
        cam = np.array([i * step, rng.normal(0, 0.05), rng.normal(0, 0.05)])
        ang = rng.normal(0, 0.03, 3)
        Rm = _rodrigues(ang)
        cam_Rs.append(Rm)
        cam_os.append(cam)
        lo = np.searchsorted(X[:, 0], cam[0] - depth_hi * 0.6 / focal)
        hi = np.searchsorted(X[:, 0], cam[0] + depth_hi * 0.6 / focal)
        Xc = (X[lo:hi] - cam) @ Rm.T
        z = Xc[:, 2]
        u = focal * Xc[:, 0] / z
        v = focal * Xc[:, 1] / z
        vis = np.flatnonzero((z > 0.1) & (np.abs(u) < 0.5) & (np.abs(v) < 0.375))
        n_real = min(int(round(n_i * (1.0 - distractor_frac))), len(vis))
        pick = rng.choice(vis, n_real, replace=False)
        pid = pick + lo
        d = base_desc[pid].astype(np.int16) + np.rint(rng.normal(0, desc_noise, (n_real, dim))).astype(np.int16)
        d = np.clip(d, 0, 255).astype(np.uint8)
        p = np.stack([u[pick], v[pick]], axis=1) + rng.normal(0, px_noise, (n_real, 2))
        n_dis = n_i - n_real
        dd = _hahog_like(rng, n_dis, dim)
        pd = np.stack([rng.uniform(-0.5, 0.5, n_dis), rng.uniform(-0.375, 0.375, n_dis)], axis=1)
        perm = rng.permutation(n_i)
        descs.append(np.concatenate([d, dd])[perm])
        ptss.append(np.concatenate([p, pd])[perm])
        ids.append(np.concatenate([pid, -np.ones(n_dis, dtype=np.int64)])[perm])
        offsets.append(offsets[-1] + n_i)
    return MatchingScene(
        desc=np.ascontiguousarray(np.concatenate(descs)),
        pts=np.ascontiguousarray(np.concatenate(ptss)),
        offsets=np.asarray(offsets, dtype=np.int64),
        point_ids=np.concatenate(ids),
        cam_R=np.asarray(cam_Rs),
        cam_o=np.asarray(cam_os),
    )


def _rodrigues(r: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(r))
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * (K @ K)


def make_two_view(
    n: int, inlier_frac: float = 0.6, seed: int = 42, px_noise: float = 1.0 / 2000.0
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Correspondences (p1, p2, is_inlier) between two pinhole views with outliers, normalized
    image coordinates (what ``robust_match_fundamental`` receives, ``matching.py:790-791``)."""
    rng = np.random.default_rng(seed)
    focal = 0.85
    X = np.stack(
        [rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(4, 12, n)], axis=1
    )
    R2 = _rodrigues(np.array([0.02, -0.1, 0.03]))
    t2 = np.array([1.0, 0.1, 0.2])
    p1 = focal * X[:, :2] / X[:, 2:3]
    Xc = (X - t2) @ R2.T
    p2 = focal * Xc[:, :2] / Xc[:, 2:3]
    p1 = p1 + rng.normal(0, px_noise, p1.shape)
    p2 = p2 + rng.normal(0, px_noise, p2.shape)
    inl = rng.random(n) < inlier_frac
    n_out = int((~inl).sum())
    p2[~inl] = np.stack([rng.uniform(-0.5, 0.5, n_out), rng.uniform(-0.375, 0.375, n_out)], axis=1)
    return np.ascontiguousarray(p1), np.ascontiguousarray(p2), inl


# ------------------------------------------------------------------------------------------------
# bundle adjustment scenes (SURVEY.md 8d "Synthetic BA inputs")
# ------------------------------------------------------------------------------------------------
# model -> (projection, distortion, #distortion parameters, #affine parameters) -- camera_instances.h:183-192
GENERIC_MODELS = {
    "brown": ("perspective", "brown", 5, 4), "fisheye_opencv": ("fisheye", "disto2468", 4, 4),
    "fisheye62": ("fisheye", "disto62", 8, 4), "fisheye624": ("fisheye", "disto624", 12, 4),
    "dual": ("dual", "disto24", 2, 1), "radial": ("perspective", "disto24", 2, 4),
    "simple_radial": ("perspective", "disto2", 1, 4),
}
CAMERA_MODEL_IDS = {"perspective": 0, "fisheye": 1, "brown": 2, "fisheye_opencv": 3, "fisheye62": 4, "fisheye624": 5, "dual": 6,
                    "radial": 7, "simple_radial": 8}


def project_generic(X: np.ndarray, pose: np.ndarray, par: np.ndarray, model: str) -> np.ndarray:
    """ProjectGeneric<PROJ, DISTO, AFF>::Forward for the other 2-D models, parameters in the native
    order [projection][distortion][affine] (camera_instances.h:127-160)."""
    proj, disto, nd, na = GENERIC_MODELS[model]
    R = _rodrigues(-np.asarray(pose[:3], float))
    Xc = (X - pose[3:6]) @ R.T
    k = list(np.asarray(par, float))
    pu, pv = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    r = np.hypot(Xc[:, 0], Xc[:, 1])
    s = np.arctan2(r, Xc[:, 2]) / np.maximum(r, 1e-300)
    fu, fv = s * Xc[:, 0], s * Xc[:, 1]
    if proj == "dual":
        t = k.pop(0)
        u, v = t * pu + (1 - t) * fu, t * pv + (1 - t) * fv
    elif proj == "fisheye":
        u, v = fu, fv
    else:
        u, v = pu, pv
    kd, ka = k[:nd], k[nd:nd + na]
    r2 = u * u + v * v
    tx = ty = 0.0
    if disto == "disto2":
        rad = 1 + r2 * kd[0]
    elif disto == "disto24":
        rad = 1 + r2 * (kd[0] + kd[1] * r2)
    elif disto == "disto2468":
        rad = 1 + r2 * (kd[0] + r2 * (kd[1] + r2 * (kd[2] + r2 * kd[3])))
    elif disto == "brown":
        rad = 1 + r2 * (kd[0] + r2 * (kd[1] + r2 * kd[2]))
        tx, ty = 2 * kd[3] * u * v + kd[4] * (r2 + 2 * u * u), 2 * kd[4] * u * v + kd[3] * (r2 + 2 * v * v)
    else:
        rad = 1 + r2 * (kd[0] + r2 * (kd[1] + r2 * (kd[2] + r2 * (kd[3] + r2 * (kd[4] + r2 * kd[5])))))
        tx, ty = 2 * kd[6] * u * v + kd[7] * (r2 + 2 * u * u), 2 * kd[7] * u * v + kd[6] * (r2 + 2 * v * v)
        if disto == "disto624":
            tx = tx + kd[8] * r2 + kd[9] * r2 * r2
            ty = ty + kd[10] * r2 + kd[11] * r2 * r2
    du, dv = u * rad + tx, v * rad + ty
    if na == 4:
        return np.stack([ka[0] * du + ka[2], ka[0] * ka[1] * dv + ka[3]], axis=1)
    return np.stack([ka[0] * du, ka[0] * dv], axis=1)


def project_perspective(X: np.ndarray, pose: np.ndarray, cam: np.ndarray, model: str = "perspective") -> np.ndarray:
    """PoseFunctor + camera [k1, k2, focal] (transformations_functions.h:112-144,
    camera_projections_functions.h:88-93 perspective / :11-22 fisheye, camera_distortions_functions.h:106-113)."""
    out = np.empty((len(X), 2))
    R = _rodrigues(-np.asarray(pose[:3], float))
    Xc = (X - pose[3:6]) @ R.T
    if model == "fisheye":
        r = np.hypot(Xc[:, 0], Xc[:, 1])
        s = np.arctan2(r, Xc[:, 2]) / np.maximum(r, 1e-300)
        u, v = s * Xc[:, 0], s * Xc[:, 1]
    else:
        u, v = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    r2 = u * u + v * v
    d = 1 + r2 * (cam[0] + cam[1] * r2)
    out[:, 0] = cam[2] * d * u
    out[:, 1] = cam[2] * d * v
    return out


def make_ba_scene(n_shots: int, n_points: int, track_len: int = 10, seed: int = 42, outlier_frac: float = 0.05,
                  px_noise: float = 1.0 / 2000.0, pose_noise_t: float = 0.05, pose_noise_r: float = 0.01,
                  point_noise: float = 0.05, gps_sigma: float = 5.0, use_gps: bool = True, model: str = "perspective",
                  generic_params=None, ragged: bool = False) -> dict:
    """Street scene with `n_points` tracks of length `track_len` over `n_shots` cameras.

    ``ragged``: tracks as a feature tracker leaves them instead of identical windows -- lengths 2 + Poisson(track_len - 2) (mean
    `track_len`, up to ~2.5 x that), 15 % of the sightings inside a window missing (never below two per point).  No two points need
    share a shot set any more, and the co-visibility half-width is the longest track, not `track_len` - 1.

    Returns the flat problem dict consumed by ``bundle_arrays`` / ``oracle.ba_solve``; ground truth
    under the ``gt_*`` keys.  Camera = shared perspective [k1, k2, focal] = [-0.1, 0.01, 0.7]
    (synthetic_examples.py:56,81), observation sigma 0.004 (synthetic_generator.py:404)."""
    rng = np.random.default_rng(seed)
    cam = np.array([-0.1, 0.01, 0.7])
    L = min(track_len, n_shots)
    step = min(0.5, 5.0 / max(L, 1))
    gt_pose = np.zeros((n_shots, 6))
    gt_pose[:, 0:3] = rng.normal(0, 0.03, (n_shots, 3))
    gt_pose[:, 3] = np.arange(n_shots) * step
    gt_pose[:, 4:6] = rng.normal(0, 0.05, (n_shots, 2))
    first = rng.integers(0, max(1, n_shots - L + 1), n_points)
    centre = (first + (L - 1) / 2.0) * step
    gt_pts = np.stack([centre + rng.uniform(-0.4, 0.4, n_points), rng.uniform(-1.5, 1.5, n_points),
                       rng.uniform(4.0, 12.0, n_points)], axis=1)
    obs_point = np.repeat(np.arange(n_points, dtype=np.int32), L)
    obs_shot = (first[:, None] + np.arange(L)[None, :]).reshape(-1).astype(np.int32)
    if ragged:
        rr = np.random.default_rng(seed + 7919)  # its own stream: the plain scene of the same seed stays what it was
        length = np.clip(2 + rr.poisson(max(L - 2, 0), n_points), 2, n_shots)
        first = rr.integers(0, n_shots - length + 1)
        centre = (first + (length - 1) / 2.0) * step
        gt_pts[:, 0] = centre + rr.uniform(-0.4, 0.4, n_points)
        obs_point = np.repeat(np.arange(n_points, dtype=np.int32), length)
        within = np.arange(int(length.sum())) - np.repeat(np.cumsum(length) - length, length)
        obs_shot = (np.repeat(first, length) + within).astype(np.int32)
        keep = rr.random(len(obs_shot)) >= 0.15
        keep[np.cumsum(length) - length] = True  # the first two sightings of every track stay
        keep[np.cumsum(length) - length + 1] = True
        obs_point, obs_shot = obs_point[keep], obs_shot[keep]
    order = np.lexsort((obs_point, obs_shot))  # shot-major like BAHelpers::Bundle's loops (ba_helpers.cc:685-699)
    obs_shot, obs_point = obs_shot[order], obs_point[order]
    xy = np.empty((len(obs_shot), 2))
    bounds = np.searchsorted(obs_shot, np.arange(n_shots + 1))
    for s in range(n_shots):
        a, b = bounds[s], bounds[s + 1]
        if model in GENERIC_MODELS:
            xy[a:b] = project_generic(gt_pts[obs_point[a:b]], gt_pose[s], generic_params, model)
        else:
            xy[a:b] = project_perspective(gt_pts[obs_point[a:b]], gt_pose[s], cam, model)
    xy += rng.normal(0, px_noise, xy.shape)
    out = rng.random(len(xy)) < outlier_frac
    xy[out] += rng.uniform(-0.03, 0.03, (int(out.sum()), 2))  # gross mismatches: up to +-60 px at 2000 px
    pose0 = gt_pose.copy()
    pose0[:, 0:3] += rng.normal(0, pose_noise_r, (n_shots, 3))
    pose0[:, 3:6] += rng.normal(0, pose_noise_t, (n_shots, 3))
    pts0 = gt_pts + rng.normal(0, point_noise, gt_pts.shape)
    prob = {
        "cam_params": cam[None, :].copy(),
        "cam_prior": cam[None, :].copy(),
        "cam_sigma": np.full((1, 3), 0.01),  # config.py:247-263
        "cam_fixed": np.zeros(1, np.uint8),
        "shot_pose": pose0,
        "shot_camera": np.zeros(n_shots, np.int32),
        "points": pts0,
        "obs_shot": obs_shot,
        "obs_point": obs_point,
        "obs_xy": xy,
        "obs_sigma": np.full(len(xy), 0.004),
        "gt_pose": gt_pose,
        "gt_points": gt_pts,
        "gt_cam": cam,
        "is_outlier": out,
    }
    if use_gps:
        prob["shot_gps"] = gt_pose[:, 3:6] + rng.normal(0, gps_sigma / 10.0, (n_shots, 3))
        prob["shot_gps_sigma"] = np.full(n_shots, gps_sigma)
    if model != "perspective":
        prob["cam_model"] = np.full(1, CAMERA_MODEL_IDS[model], np.int32)
    if model in GENERIC_MODELS:  # constant camera: its native parameters, intrinsics not optimised
        ext = np.zeros((1, 16))
        ext[0, : len(generic_params)] = generic_params
        prob["cam_ext"] = ext
        prob["cam_fixed"] = np.ones(1, np.uint8)
    return prob


def make_ba_scene_grid(rows: int, cols: int, n_points: int, track_len: int = 10, seed: int = 42, outlier_frac: float = 0.05,
                       px_noise: float = 1.0 / 2000.0, pose_noise_t: float = 0.05, pose_noise_r: float = 0.01, point_noise: float = 0.05,
                       gps_sigma: float = 5.0) -> dict:
    """A block survey instead of a sequence: rows x cols cameras on a 2-D grid (flight lines, numbered line after line), every point seen
    from a window of ~3 lines x ~track_len / 3 cameras.  Shots of neighbouring lines share points, so the co-visibility half-width in
    shot order is ~2 x cols -- far above what the banded preconditioner of the streaming solver holds.  Same camera,
    noise model and dict layout as ``make_ba_scene``."""
    rng = np.random.default_rng(seed)
    cam = np.array([-0.1, 0.01, 0.7])
    n_shots = rows * cols
    wr = min(3, rows)
    wc = max(1, min(cols, int(round(track_len / wr))))
    L = wr * wc
    step = 0.5
    gt_pose = np.zeros((n_shots, 6))
    gt_pose[:, 0:3] = rng.normal(0, 0.03, (n_shots, 3))
    rr, cc = np.divmod(np.arange(n_shots), cols)
    gt_pose[:, 3] = cc * step
    gt_pose[:, 4] = rr * step
    gt_pose[:, 5] = rng.normal(0, 0.05, n_shots)
    r0 = rng.integers(0, rows - wr + 1, n_points)
    c0 = rng.integers(0, cols - wc + 1, n_points)
    gt_pts = np.stack([(c0 + (wc - 1) / 2.0) * step + rng.uniform(-0.4, 0.4, n_points), (r0 + (wr - 1) / 2.0) * step + rng.uniform(-0.4, 0.4, n_points),
                       rng.uniform(4.0, 12.0, n_points)], axis=1)
    dr, dc = np.divmod(np.arange(L), wc)
    obs_point = np.repeat(np.arange(n_points, dtype=np.int32), L)
    obs_shot = ((r0[:, None] + dr[None, :]) * cols + c0[:, None] + dc[None, :]).reshape(-1).astype(np.int32)
    order = np.lexsort((obs_point, obs_shot))
    obs_shot, obs_point = obs_shot[order], obs_point[order]
    xy = np.empty((len(obs_shot), 2))
    bounds = np.searchsorted(obs_shot, np.arange(n_shots + 1))
    for s in range(n_shots):
        a, b = bounds[s], bounds[s + 1]
        xy[a:b] = project_perspective(gt_pts[obs_point[a:b]], gt_pose[s], cam, "perspective")
    xy += rng.normal(0, px_noise, xy.shape)
    out = rng.random(len(xy)) < outlier_frac
    xy[out] += rng.uniform(-0.03, 0.03, (int(out.sum()), 2))
    pose0 = gt_pose.copy()
    pose0[:, 0:3] += rng.normal(0, pose_noise_r, (n_shots, 3))
    pose0[:, 3:6] += rng.normal(0, pose_noise_t, (n_shots, 3))
    return {
        "cam_params": cam[None, :].copy(), "cam_prior": cam[None, :].copy(), "cam_sigma": np.full((1, 3), 0.01), "cam_fixed": np.zeros(1, np.uint8),
        "shot_pose": pose0, "shot_camera": np.zeros(n_shots, np.int32), "points": gt_pts + rng.normal(0, point_noise, gt_pts.shape),
        "obs_shot": obs_shot, "obs_point": obs_point, "obs_xy": xy, "obs_sigma": np.full(len(xy), 0.004),
        "gt_pose": gt_pose, "gt_points": gt_pts, "gt_cam": cam, "is_outlier": out,
        "shot_gps": gt_pose[:, 3:6] + rng.normal(0, gps_sigma / 10.0, (n_shots, 3)), "shot_gps_sigma": np.full(n_shots, gps_sigma),
    }


# ------------------------------------------------------------------------------------------------
# scenes for the general bundle adjustment (osfm_bundle_solve): rigs, every camera model, biases, control points
# ------------------------------------------------------------------------------------------------
BUNDLE_TEST_CAMERAS = {  # native parameters [projection][distortion][affine]
    "perspective": [-0.1, 0.01, 0.7],
    "fisheye": [-0.05, 0.01, 0.6],
    "brown": [-0.08, 0.01, 0.002, 0.001, -0.001, 0.75, 1.02, 0.01, -0.015],
    "fisheye_opencv": [-0.03, 0.006, -0.001, 0.0002, 0.55, 0.99, 0.005, 0.01],
    "fisheye62": [-0.03, 0.005, -0.001, 0.0003, -0.0001, 0.00002, 0.0008, -0.0006, 0.56, 1.01, 0.004, -0.006],
    "fisheye624": [-0.03, 0.005, -0.001, 0.0003, -0.0001, 0.00002, 0.0008, -0.0006, 0.0005, -0.0002, 0.0004, 0.0001, 0.56, 1.01, 0.004, -0.006],
    "dual": [0.4, -0.06, 0.008, 0.65],
    "radial": [-0.07, 0.009, 0.72, 1.01, 0.006, -0.004],
    "simple_radial": [-0.05, 0.7, 0.99, -0.005, 0.008],
    "spherical": [],
}
MODEL_IDS = dict(CAMERA_MODEL_IDS, spherical=9)


def _compose_world_to_cam(inst: np.ndarray, rc: np.ndarray, X: np.ndarray) -> np.ndarray:
    """WorldToCameraCoordinatesRig (error_utils.h:68-85): x_cam = R_rc^T (R_i^T (x - t_i) - t_rc), poses CAM_TO_WORLD"""
    Xi = (X - inst[3:6]) @ _rodrigues(-inst[:3]).T
    return (Xi - rc[3:6]) @ _rodrigues(-rc[:3]).T


def project_model(model: str, par, Xc: np.ndarray) -> np.ndarray:
    """projection of camera-frame points by any model; spherical: (lon, lat) / 2 pi (geometry/camera_projections_functions.h)"""
    if model == "spherical":
        lon = np.arctan2(Xc[:, 0], Xc[:, 2])
        lat = np.arctan2(-Xc[:, 1], np.hypot(Xc[:, 0], Xc[:, 2]))
        return np.stack([lon / (2 * np.pi), -lat / (2 * np.pi)], axis=1)
    ident = np.zeros(6)
    if model in GENERIC_MODELS:
        return project_generic(Xc, ident, np.asarray(par, float), model)
    return project_perspective(Xc, ident, np.asarray(par, float), model)


def make_bundle_scene(models=("perspective", "brown"), n_instances: int = 12, n_points: int = 150, rig: bool = True, seed: int = 3,
                      free_cameras: bool = True, free_rig_camera: bool = True, gps: bool = True, free_bias: bool = True,
                      n_gcp: int = 4, up_vectors: bool = True, px_noise: float = 2e-4, outlier_frac: float = 0.03) -> dict:
    """A small scene exercising every parameter block and residual family of ``osfm_bundle_solve``: one camera per entry of
    ``models`` (shots alternate between them), a rig of two rig cameras (the first a constant identity, the second with an offset
    and a prior), instances along a street, GPS priors generated THROUGH a non-identity bias per camera, a few control points
    with position priors, gravity-aligned up vectors.  Returns the dict form of an ``osfm_bundle_problem`` plus ``gt_*``."""
    rng = np.random.default_rng(seed)
    NC = len(models)
    cam_gt = np.zeros((NC, 16))
    for c, m in enumerate(models):
        cam_gt[c, : len(BUNDLE_TEST_CAMERAS[m])] = BUNDLE_TEST_CAMERAS[m]
    NR = 2 if rig else 1
    rc_gt = np.zeros((NR, 6))
    if rig:
        rc_gt[1] = [0.02, -0.3, 0.01, 0.4, 0.02, -0.05]
    NI = n_instances
    inst_gt = np.zeros((NI, 6))
    inst_gt[:, 0:3] = rng.normal(0, 0.03, (NI, 3))
    inst_gt[:, 3] = np.arange(NI) * 0.5
    inst_gt[:, 4:6] = rng.normal(0, 0.05, (NI, 2))
    shot_inst = np.repeat(np.arange(NI), NR).astype(np.int32)
    shot_rc = np.tile(np.arange(NR), NI).astype(np.int32)
    shot_cam = ((np.arange(NI * NR) // NR + np.arange(NI * NR) % NR) % NC).astype(np.int32)
    S = NI * NR
    pts_gt = np.stack([rng.uniform(-1.0, NI * 0.5 + 1.0, n_points), rng.uniform(-1.5, 1.5, n_points), rng.uniform(4.0, 10.0, n_points)], axis=1)
    obs_shot, obs_point, obs_xy = [], [], []
    for s in range(S):
        m = models[shot_cam[s]]
        Xc = _compose_world_to_cam(inst_gt[shot_inst[s]], rc_gt[shot_rc[s]], pts_gt)
        uv = project_model(m, cam_gt[shot_cam[s]], Xc)
        ang = np.arctan2(np.hypot(Xc[:, 0], Xc[:, 1]), Xc[:, 2])
        vis = np.flatnonzero(((Xc[:, 2] > 0.5) & (ang < 0.6)) if m != "spherical" else np.ones(len(Xc), bool))
        vis = vis[np.abs(pts_gt[vis, 0] - inst_gt[shot_inst[s], 3]) < 3.0]
        obs_shot += [s] * len(vis)
        obs_point += list(vis)
        obs_xy.append(uv[vis])
    obs_shot, obs_point = np.asarray(obs_shot, np.int32), np.asarray(obs_point, np.int32)
    obs_xy = np.concatenate(obs_xy) + rng.normal(0, px_noise, (len(obs_shot), 2))
    outl = rng.random(len(obs_xy)) < outlier_frac
    obs_xy[outl] += rng.uniform(-0.02, 0.02, (int(outl.sum()), 2))
    seen = np.bincount(obs_point, minlength=n_points) >= 2
    remap = -np.ones(n_points, np.int64)
    remap[seen] = np.arange(int(seen.sum()))
    keep = seen[obs_point]
    obs_shot, obs_point, obs_xy, outl = obs_shot[keep], remap[obs_point[keep]].astype(np.int32), obs_xy[keep], outl[keep]
    pts_gt = pts_gt[seen]
    NP = len(pts_gt)
    cam0 = cam_gt.copy()
    for c, m in enumerate(models):
        nk = len(BUNDLE_TEST_CAMERAS[m])
        cam0[c, :nk] *= 1.0 + rng.normal(0, 0.01, nk) * (1 if free_cameras else 0)
    sig = np.full((NC, 16), 0.01)
    prob = {
        "cam_model": np.asarray([MODEL_IDS[m] for m in models], np.int32), "cam_params": cam0, "cam_prior": cam_gt.copy(), "cam_sigma": sig,
        "cam_fixed": np.full(NC, 0 if free_cameras else 1, np.uint8),
        "rig_camera_pose": rc_gt + (np.r_[np.zeros((1, 6)), rng.normal(0, 0.01, (NR - 1, 6))] if free_rig_camera else 0.0),
        "rig_camera_prior": rc_gt.copy(), "rig_camera_sigma": np.tile([1.0, 1.0, 1.0, 0.1, 0.1, 0.1], (NR, 1)),
        "rig_camera_fixed": np.asarray([1] + [0 if free_rig_camera else 1] * (NR - 1), np.uint8),
        "rig_instance_pose": inst_gt + np.c_[rng.normal(0, 0.01, (NI, 3)), rng.normal(0, 0.05, (NI, 3))],
        "shot_rig_instance": shot_inst, "shot_rig_camera": shot_rc, "shot_camera": shot_cam,
        "points": pts_gt + rng.normal(0, 0.05, pts_gt.shape),
        "obs_shot": obs_shot, "obs_point": obs_point, "obs_xy": obs_xy, "obs_sigma": np.full(len(obs_xy), 0.004),
        "gt_cam": cam_gt, "gt_rig_camera": rc_gt, "gt_rig_instance": inst_gt, "gt_points": pts_gt, "is_outlier": outl, "models": list(models),
    }
    if gps:
        # measured positions g with t = s R(b) g + t_b: the bias maps measurements to the reconstruction frame
        bias_gt = np.tile([0.0, 0.0, 0.02, 0.3, -0.2, 0.1, 1.01], (NC, 1))
        bc = shot_cam[::NR].astype(np.int32)  # the instance's first shot
        g = np.zeros((NI, 3))
        for i in range(NI):
            b = bias_gt[bc[i]]
            g[i] = ((inst_gt[i, 3:6] - b[3:6]) @ _rodrigues(b[:3])) / b[6]  # R^T (t - t_b) / s
        prob.update({"rig_instance_gps": g + rng.normal(0, 0.01, g.shape), "rig_instance_gps_sigma": np.full((NI, 3), 0.5),
                     "rig_instance_bias_camera": bc, "bias": np.tile([0, 0, 0, 0, 0, 0, 1.0], (NC, 1)),
                     "bias_fixed": np.full(NC, 0 if free_bias else 1, np.uint8), "gt_bias": bias_gt})
    if n_gcp:
        pick = rng.choice(NP, min(n_gcp, NP), replace=False)
        pp, ps = np.zeros((NP, 3)), np.zeros((NP, 3))
        pp[pick] = pts_gt[pick] + rng.normal(0, 0.005, (len(pick), 3))
        ps[pick] = [0.01, 0.01, 0.02]
        alt = np.ones(NP, np.uint8)
        alt[pick[::2]] = 0
        prob.update({"point_prior": pp, "point_prior_sigma": ps, "point_prior_has_altitude": alt})
    if up_vectors:
        up = np.zeros((S, 3))
        for s in range(S):  # the up vector in camera coordinates: R_shot^T e_z with R_shot = R_i R_rc (camera to world)
            Rcw = _rodrigues(inst_gt[shot_inst[s], :3]) @ _rodrigues(rc_gt[shot_rc[s], :3])
            up[s] = Rcw.T @ np.array([0.0, 0.0, 1.0]) + rng.normal(0, 1e-3, 3)
        prob.update({"shot_up": up, "shot_up_sigma": np.full(S, 1e-2)})
    return prob


def make_general_ba_scene(n_shots: int, n_points: int, track_len: int = 10, model: str = "brown", n_gcp: int = 20, gps_bias: bool = True,
                          seed: int = 42, ragged: bool = False) -> dict:
    """The street scene of ``make_ba_scene`` as ``BAHelpers::Bundle`` hands a CALIBRATED data set over (``osfm_bundle_problem`` layout): one
    shared camera of ``model`` (Brown: 9 native parameters) with FREE intrinsics and their priors, a constant identity rig camera,
    one shot per rig instance, position priors through a FREE per-camera similarity bias (``bundle_compensate_gps_bias``), ``n_gcp``
    ground control points = points with position priors.  Scales to BASELINE.json configs[4] (5 000 / 500 000 / 5 000 000)."""
    par = np.asarray(BUNDLE_TEST_CAMERAS[model], float)
    base = make_ba_scene(n_shots, n_points, track_len, seed=seed, model=model if model in GENERIC_MODELS else "perspective",
                         generic_params=par if model in GENERIC_MODELS else None, ragged=ragged)
    if model not in GENERIC_MODELS and model != "perspective":
        raise ValueError("make_general_ba_scene: model %r" % model)
    rng = np.random.default_rng(seed + 1)
    nk = len(par) if model in GENERIC_MODELS else 3
    cam_gt = np.zeros((1, 16))
    cam_gt[0, :nk] = par if model in GENERIC_MODELS else base["gt_cam"]
    cam0 = cam_gt.copy()
    cam0[0, :nk] *= 1.0 + rng.normal(0, 0.01, nk)
    S, NP = n_shots, n_points
    prob = {
        "cam_model": np.asarray([MODEL_IDS[model]], np.int32), "cam_params": cam0, "cam_prior": cam_gt.copy(), "cam_sigma": np.full((1, 16), 0.01),
        "cam_fixed": np.zeros(1, np.uint8),
        "rig_camera_pose": np.zeros((1, 6)), "rig_camera_fixed": np.ones(1, np.uint8),
        "rig_instance_pose": base["shot_pose"], "shot_rig_instance": np.arange(S, dtype=np.int32), "shot_rig_camera": np.zeros(S, np.int32),
        "shot_camera": np.zeros(S, np.int32), "points": base["points"], "obs_shot": base["obs_shot"], "obs_point": base["obs_point"],
        "obs_xy": base["obs_xy"], "obs_sigma": base["obs_sigma"], "is_outlier": base["is_outlier"], "gt_points": base["gt_points"],
        "gt_rig_instance": base["gt_pose"], "gt_cam": cam_gt,
    }
    if gps_bias:  # measured positions g with t = s R(b) g + t_b: the bias maps the measurements into the reconstruction frame
        b = np.array([0.0, 0.0, 0.02, 0.3, -0.2, 0.1, 1.01])
        g = ((base["gt_pose"][:, 3:6] - b[3:6]) @ _rodrigues(b[:3])) / b[6]
        prob.update({"rig_instance_gps": g + rng.normal(0, 0.05, g.shape), "rig_instance_gps_sigma": np.full((S, 3), 0.5),
                     "rig_instance_bias_camera": np.zeros(S, np.int32), "bias": np.tile([0, 0, 0, 0, 0, 0, 1.0], (1, 1)),
                     "bias_fixed": np.zeros(1, np.uint8), "gt_bias": b[None, :]})
    if n_gcp:
        pick = rng.choice(NP, min(n_gcp, NP), replace=False)
        pp, ps = np.zeros((NP, 3)), np.zeros((NP, 3))
        pp[pick] = base["gt_points"][pick] + rng.normal(0, 0.005, (len(pick), 3))
        ps[pick] = [0.01, 0.01, 0.02]
        alt = np.ones(NP, np.uint8)
        alt[pick[::2]] = 0
        prob.update({"point_prior": pp, "point_prior_sigma": ps, "point_prior_has_altitude": alt})
    return prob
