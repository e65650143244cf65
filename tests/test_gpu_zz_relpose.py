"""Calibrated robust matching on the GPU (row M-a9): osfm_pixel_bearings / osfm_relpose_pairs through the C ABI
against the CPU oracle.  The sampler, the five-point solver, the scoring and the local optimisation use + - * / sqrt
only, so the RANSAC stage has to match bit for bit (it does: profiles/r01_relpose_bringup.txt).  The refinement goes
through sin / cos / atan2 (device math library vs glibc) and three rounds of Levenberg-Marquardt amplify the last-bit
differences to ~1e-10 in R (measured; an accept/reject or stop decision of the 10-iteration LM may also flip), so poses
after it are compared to 1e-6 and inlier sets exactly.

The file sorts last on purpose: it is the newest kernel (first run on an MI355X at the very end of round 1)."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rodrigues(r):
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def _scene(rng, n, outliers=0.3, noise=1e-3):
    R = _rodrigues(rng.normal(0, 0.3, 3))
    t = rng.normal(0, 1, 3)
    t /= np.linalg.norm(t)
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)]
    b1 = X + rng.normal(0, noise, X.shape)
    X2 = X @ R.T + t + rng.normal(0, noise, X.shape)
    bad = rng.random(n) < outliers
    X2[bad] = np.c_[rng.uniform(-2, 2, bad.sum()), rng.uniform(-2, 2, bad.sum()), rng.uniform(4, 9, bad.sum())]
    b1 /= np.linalg.norm(b1, axis=1, keepdims=True)
    b2 = X2 / np.linalg.norm(X2, axis=1, keepdims=True)
    return np.ascontiguousarray(b1), np.ascontiguousarray(b2), ~bad


def _batch(rng, sizes, outliers):
    parts = [_scene(rng, n, o) for n, o in zip(sizes, outliers)]
    off = np.r_[0, np.cumsum(sizes)].astype(np.int64)
    b1 = np.concatenate([p[0] for p in parts]) if sum(sizes) else np.zeros((0, 3))
    b2 = np.concatenate([p[1] for p in parts]) if sum(sizes) else np.zeros((0, 3))
    return b1, b2, off


def test_pixel_bearings(oracle_lib):
    from opensfm_amd import matching

    rng = np.random.default_rng(0)
    px = rng.uniform(-0.6, 0.6, (3000, 2))
    px[0] = 0.0
    for name, model, tol in (("perspective", 0, 0.0), ("fisheye", 1, 4e-16)):
        cam = SimpleNamespace(projection_type=name, k1=-0.1, k2=0.01, focal=0.9)
        got = matching.pixel_bearing_many(cam, px)
        want = oracle_lib.pixel_bearings(model, [cam.k1, cam.k2, cam.focal], px)
        assert np.abs(got - want).max() <= tol
    with pytest.raises(NotImplementedError):
        matching.pixel_bearing_many(SimpleNamespace(projection_type="pushbroom", k1=0, k2=0, focal=1), px)


def test_ransac_relative_pose_batch_bits(oracle_lib):
    from opensfm_amd import matching

    rng = np.random.default_rng(1)
    sizes = [30, 5, 4, 0, 200, 9, 500, 120, 64, 65]
    outl = [0.3, 0.0, 0.0, 0.0, 0.5, 0.2, 0.2, 0.9, 0.1, 0.4]
    b1, b2, off = _batch(rng, sizes, outl)
    for iters, use_lo in ((1000, True), (37, True), (150, False)):
        res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "ransac", iters, 0.99, use_lo, 10)
        for p, n in enumerate(sizes):
            s = slice(off[p], off[p + 1])
            want = oracle_lib.ransac_relative_pose(b1[s], b2[s], 0.004, iters, 0.99, use_lo, 10)
            assert (res[p]["score"], res[p]["iterations"]) == (want["score"], want["iterations"]), (p, n, iters)
            assert np.array_equal(np.flatnonzero(mask[s]), want["inliers"])
            assert np.array_equal(res[p]["model"].view(np.uint64), want["model"].view(np.uint64))
            assert np.array_equal(res[p]["lo_model"].view(np.uint64), want["lo_model"].view(np.uint64))


def test_robust_match_calibrated_batch(oracle_lib):
    from opensfm_amd import matching

    rng = np.random.default_rng(2)
    sizes = [7, 8, 40, 300, 1000, 150, 2000]
    outl = [0.0, 0.0, 0.3, 0.4, 0.6, 0.97, 0.3]
    b1, b2, off = _batch(rng, sizes, outl)
    res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "match", 1000, 0.99, True, 10, 10)
    for p, n in enumerate(sizes):
        s = slice(off[p], off[p + 1])
        want = oracle_lib.robust_match_calibrated_bearings(b1[s], b2[s], 0.004, 1000, 0.99, True, 10, 10)
        assert (res[p]["score"], res[p]["iterations"]) == (want["score"], want["iterations"])
        assert np.array_equal(mask[s], want["mask"]), p
        assert res[p]["n_inliers"] == want["mask"].sum()
        if res[p]["n_inliers"]:
            assert np.abs(res[p]["R"] - want["R"]).max() < 1e-6 and np.abs(res[p]["t"] - want["t"]).max() < 1e-6


def test_robust_match_calibrated_leaf(oracle_lib):
    """Same call as the reference's: pixels + cameras + matches + config in, matches[inliers] out."""
    from opensfm_amd import matching

    rng = np.random.default_rng(3)
    n = 600
    b1, b2, good = _scene(rng, n, outliers=0.35)
    cams = [SimpleNamespace(projection_type="fisheye", k1=-0.05, k2=0.01, focal=0.7),
            SimpleNamespace(projection_type="perspective", k1=-0.1, k2=0.02, focal=0.85)]

    def project(cam, b):  # forward projection of the two models (camera_projections_functions.h) to make pixel inputs
        if cam.projection_type == "fisheye":
            l = np.hypot(b[:, 0], b[:, 1])
            theta = np.arctan2(l, b[:, 2])
            u = b[:, :2] * (theta / np.maximum(l, 1e-300))[:, None]
        else:
            u = b[:, :2] / b[:, 2:3]
        r2 = (u**2).sum(1)
        return cam.focal * u * (1 + r2 * (cam.k1 + cam.k2 * r2))[:, None]

    perm = rng.permutation(n)
    p1 = project(cams[0], b1)
    p2 = project(cams[1], b2)[perm]
    inv = np.argsort(perm)
    matches = np.c_[np.arange(n), inv]  # feature i of image 1 <-> feature inv[i] of image 2
    cfg = {"robust_matching_calib_threshold": 0.004, "five_point_refine_match_iterations": 10}
    got = matching.robust_match_calibrated(p1, p2, cams[0], cams[1], matches, cfg)
    want = oracle_lib.robust_match_calibrated(p1, p2, [cams[0].k1, cams[0].k2, cams[0].focal], [cams[1].k1, cams[1].k2, cams[1].focal],
                                              "fisheye", "perspective", matches, 0.004, 10)
    assert np.array_equal(np.asarray(got), np.asarray(want))
    assert good[np.asarray(got)[:, 0]].mean() > 0.98 and len(got) > 0.8 * good.sum()
    assert len(matching.robust_match_calibrated(p1, p2, cams[0], cams[1], matches[:7], cfg)) == 0


def test_match_pairs_calibrated_pipeline(oracle_lib, gpu_ctx):
    """All pairs of a small street scene whose cameras are (slightly) distorted perspectives, so every pair takes the
    calibrated branch: GPU descriptor stage -> gate -> bearings -> one relpose launch -> gate, against the same flow on
    the oracle (matching.py:563-634 with robust_match_calibrated)."""
    from opensfm_amd import matching, synthetic

    sc = synthetic.make_matching_scene(8, 600, seed=33, ragged=True)
    pairs = synthetic.all_pairs(8)
    cam = SimpleNamespace(projection_type="perspective", k1=1e-3, k2=0.0, focal=0.85)
    pts = [sc.pts[sc.offsets[i]: sc.offsets[i + 1]] for i in range(8)]
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    counts, m = matching.match_pairs_calibrated(store, pairs, [cam] * 8, pts, {})
    got = matching.split_matches(counts, m)
    stage0 = oracle_lib.match_pairs(sc.desc.astype(np.float32), sc.pts, sc.offsets, pairs, stage=0)
    survivors = 0
    for (a, b), g, m0 in zip(pairs, got, stage0):
        want = np.zeros((0, 2), np.int32)
        if len(m0) >= 20:
            b1 = oracle_lib.pixel_bearings(0, [cam.k1, cam.k2, cam.focal], pts[a][m0[:, 0], :2])
            b2 = oracle_lib.pixel_bearings(0, [cam.k1, cam.k2, cam.focal], pts[b][m0[:, 1], :2])
            r = oracle_lib.robust_match_calibrated_bearings(b1, b2, 0.004, 1000, 0.99, True, 10, 10)
            if r["mask"].sum() >= 20:
                want = m0[r["mask"]]
        assert np.array_equal(g, want), (a, b)
        survivors += len(want) > 0
    assert survivors >= 7  # neighbouring cameras of the street share points


def test_pixel_bearings_every_projection_type(oracle_lib):
    """ProjectGeneric::Backward of the nine 2-D models and the spherical one.  Affine / distortion stages are + - * / sqrt
    only; the projection stage goes through the device sin / cos / tan, hence 1e-12."""
    import test_oracle_relpose as cams
    from opensfm_amd import matching

    rng = np.random.default_rng(5)
    names = {"brown": ("k1", "k2", "k3", "p1", "p2", "focal", "aspect_ratio", "cx", "cy"),
             "fisheye_opencv": ("k1", "k2", "k3", "k4", "focal", "aspect_ratio", "cx", "cy"),
             "fisheye62": ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "focal", "aspect_ratio", "cx", "cy"),
             "fisheye624": ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s0", "s1", "s2", "s3", "focal", "aspect_ratio", "cx", "cy"),
             "dual": ("transition", "k1", "k2", "focal"), "radial": ("k1", "k2", "focal", "aspect_ratio", "cx", "cy"),
             "simple_radial": ("k1", "focal", "aspect_ratio", "cx", "cy"), "perspective": ("k1", "k2", "focal"),
             "fisheye": ("k1", "k2", "focal"), "spherical": ()}
    for model, par in list(cams._BEARING_CAMERAS.items()) + [("spherical", [])]:
        attrs = dict(zip(names[model], par))
        cam = SimpleNamespace(projection_type=model, principal_point=[attrs.pop("cx", 0.0), attrs.pop("cy", 0.0)], **attrs)
        ang, phi = rng.uniform(0, 0.9, 2000), rng.uniform(0, 2 * np.pi, 2000)
        X = np.c_[np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)]
        px = rng.uniform(-0.3, 0.3, (2000, 2)) if model == "spherical" else cams._forward(model, par, X)
        got = matching.pixel_bearing_many(cam, px)
        want = oracle_lib.pixel_bearings_generic(model, par, px)
        assert np.abs(got - want).max() < 1e-12, model
        if model != "spherical":
            assert np.abs(got - X).max() < 2e-6, model


# ---- guided (epipolar-masked) matching: osfm_match_guided ------------------------------------------------------------


class _Pose:
    """the three methods of pygeometry.Pose the guided route uses (world-to-camera rotation R, origin o)"""

    def __init__(self, R, o):
        self.R, self.o = np.asarray(R, float), np.asarray(o, float)

    def relative_to(self, base):  # pose.h:133-144: T_this_w * T_w_base
        return _Pose(self.R @ base.R.T, base.R @ (self.o - base.o))

    def get_R_cam_to_world(self):
        return self.R.T.copy()

    def get_origin(self):
        return self.o.copy()


def test_masked_and_guided_leaf(oracle_lib):
    """match_brute_force[_symmetric](f1, f2, config, maskij) with explicit masks, and the fused epipolar mask of guided matching,
    against the oracle (same cases as the host emulation, tests/test_guided_host.py)."""
    import test_guided_host as gh
    from opensfm_amd import matching, synthetic

    rng = np.random.default_rng(0)
    sc = synthetic.make_matching_scene(2, 300, seed=9, ragged=True)
    f1 = sc.desc[sc.offsets[0]: sc.offsets[1]].astype(np.float32)
    f2 = sc.desc[sc.offsets[1]: sc.offsets[2]].astype(np.float32)
    for density in (1.0, 0.5, 0.05, 0.004, 0.0):
        mask = rng.random((len(f1), len(f2))) < density
        for ratio in (0.8, 0.999):
            cfg = {"lowes_ratio": ratio}
            want = oracle_lib.match_brute_force_masked(f1, f2, mask, ratio, symmetric=True)
            assert matching.match_brute_force_symmetric(f1, f2, cfg, mask) == [tuple(map(int, m)) for m in want]
            want = oracle_lib.match_brute_force_masked(f1, f2, mask, ratio, symmetric=False)
            assert matching.match_brute_force(f1, f2, cfg, mask) == [tuple(map(int, m)) for m in want]
    for n, thr in ((60, 0.005), (150, 0.02), (97, 0.3), (1000, 0.006)):
        d1, d2, b1, b2, R, o, perm = gh.guided_scene(rng, n)
        mask, _ = oracle_lib.epipolar_mask(b1, b2, R, o, thr)
        want = oracle_lib.match_brute_force_masked(d1, d2, mask, 0.8, symmetric=True)
        rel = _Pose(R.T, o)  # get_R_cam_to_world() = R, get_origin() = o
        got = matching.match_guided(d1, d2, b1, b2, rel, {"lowes_ratio": 0.8, "guided_matching_threshold": thr})
        assert np.array_equal(got, want), (n, thr)
    assert (perm[got[:, 1]] == got[:, 0]).mean() > 0.95 and len(got) > 1200


def test_guided_match_images_with_pairs(oracle_lib):
    """match_images_with_pairs(..., poses): guided descriptor stage -> gate -> robust_match -> gate -> unfilter
    (matching.py:204-207, 563-634) for three views of a repetitive scene with slightly distorted perspective cameras."""
    import test_guided_host as gh
    from opensfm_amd import matching

    rng = np.random.default_rng(4)
    cam = SimpleNamespace(projection_type="perspective", k1=1e-3, k2=0.0, focal=0.85)
    n = 400
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)]
    base = rng.integers(0, 255, (n // 2, 128))
    images, poses, feats, masks, order = ["a", "b", "c"], {}, {}, {}, {}
    for im in images:
        R = gh._rodrigues(rng.normal(0, 0.1, 3))
        o = rng.normal(0, 0.5, 3)
        poses[im] = _Pose(R, o)
        Y = (X - o) @ R.T
        u = Y[:, :2] / Y[:, 2:3]
        r2 = (u**2).sum(1)
        px = cam.focal * u * (1 + r2 * (cam.k1 + cam.k2 * r2))[:, None] + rng.normal(0, 2e-4, (n, 2))
        perm = rng.permutation(n)
        order[im] = perm
        desc = np.clip(np.concatenate([base, base]) + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.float32)
        feats[im] = SimpleNamespace(points=np.c_[px[perm], np.ones((n, 2))], descriptors=desc[perm])
        masks[im] = rng.random(n) > 0.1
    data = SimpleNamespace(config={"matcher_type": "BRUTEFORCE", "robust_matching_min_match": 20, "robust_matching_calib_threshold": 0.004,
                                   "five_point_refine_match_iterations": 10, "lowes_ratio": 0.8, "symmetric_matching": True,
                                   "guided_matching_threshold": 0.006},
                           load_camera_models=lambda: {"cam": cam}, load_features=lambda im: feats[im],
                           load_features_mask=lambda im, pts: masks[im])
    exifs = {im: {"camera": "cam"} for im in images}
    pairs = [("a", "b"), ("a", "c"), ("b", "c")]
    got = matching.match_images_with_pairs(data, {}, exifs, pairs, poses)
    par = [cam.k1, cam.k2, cam.focal]
    for (ia, ib) in pairs:
        pa, pb = feats[ia].points[masks[ia]], feats[ib].points[masks[ib]]
        b1 = oracle_lib.pixel_bearings(0, par, pa[:, :2])
        b2 = oracle_lib.pixel_bearings(0, par, pb[:, :2])
        rel = poses[ib].relative_to(poses[ia])
        emask, _ = oracle_lib.epipolar_mask(b1, b2, rel.get_R_cam_to_world(), rel.get_origin(), 0.006)
        m = oracle_lib.match_brute_force_masked(feats[ia].descriptors[masks[ia]], feats[ib].descriptors[masks[ib]], emask, 0.8, symmetric=True)
        assert len(m) >= 100  # the plain matcher finds (almost) nothing here: every descriptor exists twice
        r = oracle_lib.robust_match_calibrated_bearings(b1[m[:, 0]], b2[m[:, 1]], 0.004, 1000, 0.99, True, 10, 10)
        assert r["mask"].sum() >= 20
        want = matching.unfilter_matches(m[r["mask"]], masks[ia], masks[ib])
        assert np.array_equal(got[ia, ib], np.asarray(want))
        assert (order[ia][got[ia, ib][:, 0]] == order[ib][got[ia, ib][:, 1]]).mean() > 0.97


def test_large_batch_through_the_rounds(oracle_lib):
    """400 pairs of mixed sizes in one call: the work lists of the solver kernels span many wavefronts and most pairs finish in
    different rounds.  RANSAC mode bit for bit on every pair; MATCH mode inlier sets on a sample."""
    from opensfm_amd import matching

    rng = np.random.default_rng(66)
    sizes = [int(v) for v in rng.choice([0, 4, 5, 8, 9, 12, 30, 64, 65, 150, 300, 700], 400)]
    outl = [float(v) for v in rng.choice([0.0, 0.2, 0.4, 0.6, 0.9], 400)]
    b1, b2, off = _batch(rng, sizes, outl)
    res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "ransac", 1000, 0.99, True, 10)
    for p, n in enumerate(sizes):
        s = slice(off[p], off[p + 1])
        if n < 5:
            assert res[p]["score"] == 0 and res[p]["iterations"] == 0 and not mask[s].any()
            continue
        want = oracle_lib.ransac_relative_pose(b1[s], b2[s], 0.004, 1000, 0.99, True, 10)
        assert (res[p]["score"], res[p]["iterations"]) == (want["score"], want["iterations"]), (p, n)
        assert np.array_equal(np.flatnonzero(mask[s]), want["inliers"])
        assert np.array_equal(res[p]["model"].view(np.uint64), want["model"].view(np.uint64))
        assert np.array_equal(res[p]["lo_model"].view(np.uint64), want["lo_model"].view(np.uint64))
    res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "match", 1000, 0.99, True, 10, 10)
    for p in range(0, 400, 7):
        s = slice(off[p], off[p + 1])
        want = oracle_lib.robust_match_calibrated_bearings(b1[s], b2[s], 0.004, 1000, 0.99, True, 10, 10)
        assert np.array_equal(mask[s], want["mask"]), (p, sizes[p])


def test_batched_guided_descriptor_stage_equals_oracle(oracle_lib):
    """osfm_match_pairs_guided, descriptor stage (robust = 0): several pairs of different sizes over one store, thresholds from a narrow
    band to "everything allowed", both the symmetric and the one-way matcher -- against the oracle's epipolar mask + masked matcher.
    The kernel tests (a + b) / 2 < c* instead of pi/2 - acos(.) < threshold: the mask it implies must be the oracle's."""
    import test_guided_host as gh
    from opensfm_amd import matching

    rng = np.random.default_rng(11)
    sizes = [130, 700, 64, 333, 1000]
    scenes = [gh.guided_scene(rng, n // 2) for n in sizes]
    # image 2k = first view of scene k, image 2k + 1 = its second view
    descs, bears, pairs, rels = [], [], [], []
    for k, (d1, d2, b1, b2, R, o, perm) in enumerate(scenes):
        descs += [d1, d2]
        bears += [b1, b2]
        pairs.append((2 * k, 2 * k + 1))
        rels.append(np.concatenate([np.asarray(R).reshape(9), np.asarray(o).reshape(3)]))
    pairs.append((1, 0))  # a pair listed the other way round: inverse relative pose
    R0, o0 = np.asarray(scenes[0][4]), np.asarray(scenes[0][5])
    rels.append(np.concatenate([R0.T.reshape(9), (-R0.T @ o0).reshape(3)]))
    pts = [np.zeros((len(d), 2)) for d in descs]
    store = matching.DescriptorStore(descs, pts)
    for thr in (0.002, 0.006, 0.05, 2.0):
        for sym in (True, False):
            cfg = {"lowes_ratio": 0.8, "guided_matching_threshold": thr, "symmetric_matching": sym}
            counts, m = matching.match_pairs_guided(store, np.asarray(pairs, np.int32), bears, rels, cfg, robust=False)
            got = matching.split_matches(counts, m)
            for (a, b), rel, g in zip(pairs, rels, got):
                emask, _ = oracle_lib.epipolar_mask(bears[a], bears[b], rel[:9].reshape(3, 3), rel[9:], thr)
                want = oracle_lib.match_brute_force_masked(descs[a], descs[b], emask, 0.8, symmetric=sym)
                assert np.array_equal(g, want), (thr, sym, a, b, len(g), len(want))
            if thr == 0.006 and sym:
                assert counts.sum() > 1000  # the band separates the two copies of every descriptor
    store.close()


def test_batched_guided_with_fundamental_ransac(oracle_lib):
    """pinhole cameras: guided descriptor stage + gate + fundamental-matrix RANSAC + gate in one call == the oracle chain"""
    import test_guided_host as gh
    from opensfm_amd import matching

    rng = np.random.default_rng(12)
    n, focal = 600, 0.85
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)]
    base = rng.integers(0, 255, (n // 2, 128))
    descs, pts, bears, poses = [], [], [], []
    for _ in range(3):
        R = gh._rodrigues(rng.normal(0, 0.1, 3))
        o = rng.normal(0, 0.5, 3)
        Y = (X - o) @ R.T
        px = focal * Y[:, :2] / Y[:, 2:3] + rng.normal(0, 2e-4, (n, 2))
        perm = rng.permutation(n)
        d = np.clip(np.concatenate([base, base]) + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.float32)
        descs.append(d[perm])
        pts.append(px[perm])
        bears.append(np.asarray(oracle_lib.pixel_bearings(0, [0.0, 0.0, focal], px[perm]), np.float32))
        poses.append(_Pose(R, o))
    pairs = [(0, 1), (0, 2), (1, 2)]
    rels = [poses[b].relative_to(poses[a]) for a, b in pairs]
    store = matching.DescriptorStore(descs, pts)
    cfg = {"lowes_ratio": 0.8, "guided_matching_threshold": 0.006, "robust_matching_min_match": 20, "robust_matching_threshold": 0.004}
    counts, m = matching.match_pairs_guided(store, np.asarray(pairs, np.int32), bears, rels, cfg, robust=True)
    for (a, b), rel, g in zip(pairs, rels, matching.split_matches(counts, m)):
        emask, _ = oracle_lib.epipolar_mask(bears[a], bears[b], rel.get_R_cam_to_world(), rel.get_origin(), 0.006)
        mm = oracle_lib.match_brute_force_masked(descs[a], descs[b], emask, 0.8, symmetric=True)
        assert len(mm) >= 100
        F, mask, _ = oracle_lib.find_fundamental_ransac(pts[a][mm[:, 0]], pts[b][mm[:, 1]], 0.004, 0.9999)
        want = mm[mask] if F is not None and F[2, 2] != 0.0 else np.zeros((0, 2), np.int32)
        assert np.array_equal(g, want if len(want) >= 20 else np.zeros((0, 2), np.int32))
        assert len(g) > 80
    store.close()


def test_batched_guided_on_float_descriptors(oracle_lib):
    """root-SIFT style float descriptors (feature_root, features.py:292-298) through osfm_match_pairs_guided: the candidates the epipolar
    mask lets through are ranked by the float distance in the oracle's accumulation order (oracle/guided_oracle.c l2sqr =
    oracle/match_oracle.c l2sqr_f32), read from the store's float rows -- not from the 8-bit quantisation the unguided matcher bounds with"""
    import test_guided_host as gh
    from opensfm_amd import matching

    rng = np.random.default_rng(23)
    sizes = [200, 640, 90]
    descs, bears, pairs, rels = [], [], [], []
    for k, n in enumerate(sizes):
        d1, d2, b1, b2, R, o, perm = gh.guided_scene(rng, n // 2)

        def root(d):  # L1-normalise, square root (features.root_feature): genuinely non-integer values
            d = np.asarray(d, np.float32) + rng.uniform(0, 0.5, d.shape).astype(np.float32)
            return np.sqrt(d / d.sum(axis=1, keepdims=True)).astype(np.float32)

        descs += [root(d1), root(d2)]
        bears += [b1, b2]
        pairs.append((2 * k, 2 * k + 1))
        rels.append(np.concatenate([np.asarray(R).reshape(9), np.asarray(o).reshape(3)]))
    store = matching.DescriptorStore(descs, [np.zeros((len(d), 2)) for d in descs])
    total = 0
    for thr in (0.004, 0.02, 2.0):
        for sym in (True, False):
            cfg = {"lowes_ratio": 0.85, "guided_matching_threshold": thr, "symmetric_matching": sym}
            counts, m = matching.match_pairs_guided(store, np.asarray(pairs, np.int32), bears, rels, cfg, robust=False)
            for (a, b), rel, g in zip(pairs, rels, matching.split_matches(counts, m)):
                emask, _ = oracle_lib.epipolar_mask(bears[a], bears[b], rel[:9].reshape(3, 3), rel[9:], thr)
                want = oracle_lib.match_brute_force_masked(descs[a], descs[b], emask, 0.85, symmetric=sym)
                assert np.array_equal(g, want), (thr, sym, a, b, len(g), len(want))
                total += len(want)
    assert total > 500
    store.close()
