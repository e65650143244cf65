"""The reference's OWN Python on this path, executed: opensfm/matching.py and opensfm/multiview.py are loaded from /root/reference
with stand-in modules for what they import -- the compiled pybind modules (pygeometry, pyrobust ...) and cv2 are not available in
this image, so those leaves are served by the CPU oracle's numerics -- and their control flow (ratio test, set intersection, gates,
the 4-2-1 relaxation loop with its pose conversions in numpy, the F / E dispatch, unfilter_matches) is compared with the oracle's
and the product's host-side restatements on the same inputs.  Skipped where /root/reference is not mounted (the GPU box)."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference/opensfm"


class _Stub(types.ModuleType):
    """a module whose unknown attributes are empty classes (the reference's type annotations are evaluated at import time)"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {})
        setattr(self, name, cls)
        return cls

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference is not mounted")


@pytest.fixture(scope="module")
def ref(oracle_lib):
    """(reference matching module, reference multiview module) with oracle-backed leaves"""
    import ctypes as C

    o = oracle_lib
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "cv2" or k == "opensfm" or k.startswith("opensfm.")}

    def P(a, t):
        return a.ctypes.data_as(C.POINTER(t))

    # ---- cv2: the two entry points the path uses ----
    cv2 = _Stub("cv2")
    cv2.FM_RANSAC = 8

    class DMatch:
        def __init__(self, q, t, d):
            self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(np.float32(d))

    class Matcher:
        def __init__(self, kind):
            assert kind == "BruteForce"
            self.train = None

        def add(self, descs):
            self.train = np.ascontiguousarray(descs[0], np.float32)

        def knnMatch(self, f1, k=2, masks=None):
            assert k == 2
            f1 = np.ascontiguousarray(f1, np.float32)
            mask = None if masks is None else np.ascontiguousarray(masks[0], np.uint8)
            idx = np.zeros((len(f1), 2), np.int32)
            dist = np.zeros((len(f1), 2), np.float32)
            o.lib().oracle_knn2_masked(P(f1, C.c_float), len(f1), P(self.train, C.c_float), len(self.train), f1.shape[1],
                                       P(mask, C.c_uint8) if mask is not None else None, P(idx, C.c_int32), P(dist, C.c_float))
            return [[DMatch(i, j, d) for j, d in zip(idx[i], dist[i]) if j >= 0] for i in range(len(f1))]

    cv2.DescriptorMatcher_create = Matcher

    def findFundamentalMat(p1, p2, method, thr, conf):
        F, mask, _ = o.find_fundamental_ransac(p1, p2, thr, conf)
        return F, mask.astype(np.uint8).reshape(-1, 1)

    cv2.findFundamentalMat = findFundamentalMat

    # ---- compiled opensfm modules ----
    pkg = _Stub("opensfm")
    pkg.__path__ = [REF]
    pygeometry = _Stub("opensfm.pygeometry")
    pygeometry.Camera = type("Camera", (), {"is_panorama": staticmethod(lambda t: t in ("equirectangular", "spherical"))})
    pygeometry.Pose = object

    def tri_many(b1, b2, R, t):
        b1, b2 = np.ascontiguousarray(b1, np.float64), np.ascontiguousarray(b2, np.float64)
        R, t = np.ascontiguousarray(R, np.float64), np.ascontiguousarray(t, np.float64)
        ok = np.zeros(len(b1), np.uint8)
        X = np.zeros((len(b1), 3))
        o.lib().oracle_triangulate_two_bearings_midpoint_many(P(b1, C.c_double), P(b2, C.c_double), len(b1), P(R, C.c_double), P(t, C.c_double),
                                                              P(ok, C.c_uint8), P(X, C.c_double))
        return [(bool(k), x) for k, x in zip(ok, X)]

    pygeometry.triangulate_two_bearings_midpoint_many = tri_many
    pygeometry.relative_pose_refinement = lambda Rt, b1, b2, it: o.relative_pose_refinement(Rt, b1, b2, it)[0]
    pygeometry.epipolar_angle_two_bearings_many = lambda b1, b2, R, t: o.epipolar_mask(b1, b2, R, t, 0.0)[1]
    pyrobust = _Stub("opensfm.pyrobust")

    class RobustEstimatorParams:
        iterations, probability, use_local_optimization, use_iteration_reduction, local_optimization_iterations = 100, 0.99, True, True, 10

    pyrobust.RobustEstimatorParams = RobustEstimatorParams
    pyrobust.RansacType = types.SimpleNamespace(RANSAC=0)

    def ransac_relative_pose(b1, b2, threshold, params, kind):
        r = o.ransac_relative_pose(b1, b2, threshold, params.iterations, params.probability, params.use_local_optimization,
                                   params.local_optimization_iterations)
        return types.SimpleNamespace(lo_model=r["lo_model"], model=r["model"], inliers_indices=r["inliers"], score=r["score"])

    pyrobust.ransac_relative_pose = ransac_relative_pose
    mods = {"cv2": cv2, "opensfm": pkg, "opensfm.pygeometry": pygeometry, "opensfm.pyrobust": pyrobust}
    for name in ("pymap", "pyfeatures", "context", "feature_loader", "log", "pairs_selection", "dataset_base"):
        m = _Stub("opensfm." + name)
        mods["opensfm." + name] = m
    mods["opensfm.dataset_base"].DataSetBase = object
    for name, m in mods.items():
        sys.modules[name] = m
        if name.startswith("opensfm."):
            setattr(pkg, name.split(".")[1], m)
    loaded = {}
    for name in ("transformations", "multiview", "matching"):
        spec = importlib.util.spec_from_file_location("opensfm." + name, os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["opensfm." + name] = mod
        setattr(pkg, name, mod)
        spec.loader.exec_module(mod)
        loaded[name] = mod
    yield loaded["matching"], loaded["multiview"]
    for k in [k for k in sys.modules if k == "cv2" or k == "opensfm" or k.startswith("opensfm.")]:
        del sys.modules[k]
    sys.modules.update({k: v for k, v in saved.items() if v is not None})


def _camera(oracle_lib, model, k1, k2, focal):
    return types.SimpleNamespace(projection_type=model, k1=k1, k2=k2, focal=focal,
                                 pixel_bearing_many=lambda p: oracle_lib.pixel_bearings(model, [k1, k2, focal], p))


def test_descriptor_matching_flow(ref, oracle_lib):
    """match_brute_force / match_brute_force_symmetric (matching.py:723-777), with and without maskij"""
    from opensfm_amd import synthetic

    matching, _ = ref
    sc = synthetic.make_matching_scene(2, 300, seed=13, ragged=True)
    f1 = sc.desc[sc.offsets[0]: sc.offsets[1]].astype(np.float32)
    f2 = sc.desc[sc.offsets[1]: sc.offsets[2]].astype(np.float32)
    rng = np.random.default_rng(0)
    for ratio in (0.8, 0.95):
        cfg = {"lowes_ratio": ratio}
        assert sorted(matching.match_brute_force(f1, f2, cfg)) == [tuple(m) for m in oracle_lib.match_brute_force(f1, f2, ratio)]
        assert sorted(matching.match_brute_force_symmetric(f1, f2, cfg)) == [tuple(m) for m in oracle_lib.match_brute_force_symmetric(f1, f2, ratio)]
        mask = rng.random((len(f1), len(f2))) < 0.3
        assert sorted(matching.match_brute_force(f1, f2, cfg, mask)) == [tuple(m) for m in oracle_lib.match_brute_force_masked(f1, f2, mask, ratio, False)]
        assert sorted(matching.match_brute_force_symmetric(f1, f2, cfg, mask)) == [tuple(m) for m in oracle_lib.match_brute_force_masked(f1, f2, mask, ratio, True)]


def test_robust_matching_flow(ref, oracle_lib):
    """robust_match -> robust_match_fundamental / robust_match_calibrated (matching.py:780-802, 871-929) with multiview.relative_pose_ransac,
    compute_inliers_bearings and relative_pose_optimize_nonlinear (multiview.py:494-553) run from the reference's files."""
    import test_relpose_core_host as rp

    matching, multiview = ref
    rng = np.random.default_rng(3)
    cfg = {"robust_matching_threshold": 0.004, "robust_matching_calib_threshold": 0.004, "five_point_refine_match_iterations": 10}
    for n, outl, models in ((300, 0.3, ("fisheye", "perspective")), (120, 0.5, ("perspective", "perspective")), (7, 0.0, ("fisheye", "fisheye")),
                            (60, 0.97, ("fisheye", "perspective"))):
        b1, b2, good = rp._scene(rng, n, outliers=outl)
        cams = [_camera(oracle_lib, models[0], -0.05, 0.01, 0.7), _camera(oracle_lib, models[1], -0.1, 0.02, 0.85)]

        def project(cam, b):
            if cam.projection_type == "fisheye":
                l = np.hypot(b[:, 0], b[:, 1])
                u = b[:, :2] * (np.arctan2(l, b[:, 2]) / np.maximum(l, 1e-300))[:, None]
            else:
                u = b[:, :2] / b[:, 2:3]
            r2 = (u**2).sum(1)
            return cam.focal * u * (1 + r2 * (cam.k1 + cam.k2 * r2))[:, None]

        p1, p2 = project(cams[0], b1), project(cams[1], b2)
        matches = np.c_[np.arange(n), np.arange(n)]
        got = matching.robust_match(p1, p2, cams[0], cams[1], matches, cfg)
        want = oracle_lib.robust_match_calibrated(p1, p2, [cams[0].k1, cams[0].k2, cams[0].focal], [cams[1].k1, cams[1].k2, cams[1].focal],
                                                  models[0], models[1], matches, 0.004, 10)
        assert np.array_equal(np.asarray(got).reshape(-1, 2), np.asarray(want).reshape(-1, 2)), (n, outl)
        # the all-C flow the GPU kernel is tested against gives the same inliers
        bb1, bb2 = cams[0].pixel_bearing_many(p1), cams[1].pixel_bearing_many(p2)
        c = oracle_lib.robust_match_calibrated_bearings(bb1, bb2, 0.004, 1000, 0.99, True, 10, 10)
        assert np.array_equal(matches[c["mask"]], np.asarray(want).reshape(-1, 2))
    # undistorted perspective cameras take the fundamental-matrix branch
    pin = [_camera(oracle_lib, "perspective", 0.0, 0.0, 0.85)] * 2
    b1, b2, good = rp._scene(rng, 200, outliers=0.3)
    p1, p2 = 0.85 * b1[:, :2] / b1[:, 2:3], 0.85 * b2[:, :2] / b2[:, 2:3]
    matches = np.c_[np.arange(200), np.arange(200)]
    got = matching.robust_match(p1, p2, pin[0], pin[1], matches, cfg)
    F, mask, _ = oracle_lib.find_fundamental_ransac(p1, p2, 0.004, 0.9999)
    assert F is not None and np.array_equal(got, matches[mask]) and good[got[:, 0]].mean() > 0.97


def test_inlier_and_unfilter_helpers(ref, oracle_lib):
    import test_relpose_core_host as rp
    from opensfm_amd import matching as product

    matching, _ = ref
    rng = np.random.default_rng(5)
    b1, b2, _ = rp._scene(rng, 500)
    R, t = rp._rodrigues(rng.normal(0, 0.2, 3)), rng.normal(0, 1, 3)
    for thr in (0.004, 0.05, 0.5):
        assert np.array_equal(np.asarray(matching.compute_inliers_bearings(b1, b2, R, t, thr), bool), oracle_lib.inliers_bearings(b1, b2, R, t, thr))
    m1, m2 = rng.random(50) > 0.3, rng.random(60) > 0.3
    m = np.c_[rng.integers(0, m1.sum(), 20), rng.integers(0, m2.sum(), 20)]
    assert np.array_equal(matching.unfilter_matches(m, m1, m2), product.unfilter_matches(m, m1, m2))


class _RefPose:
    """pygeometry.Pose as far as match_unwrap_args / compute_inliers_bearing_epipolar use it (world-to-camera R, origin o)"""

    def __init__(self, R, o):
        self.R, self.o = np.asarray(R, float), np.asarray(o, float)

    def relative_to(self, base):
        return _RefPose(self.R @ base.R.T, base.R @ (self.o - base.o))

    def get_R_cam_to_world(self):
        return self.R.T.copy()

    def get_origin(self):
        return self.o.copy()


def _collection(oracle_lib, rng, guided):
    """4 images of one repetitive scene (every descriptor exists twice when `guided`), mixed cameras, feature masks"""
    import test_relpose_core_host as rp

    cams = {"pin": _camera(oracle_lib, "perspective", 0.0, 0.0, 0.8), "fish": _camera(oracle_lib, "fisheye", -0.05, 0.01, 0.7),
            "dist": _camera(oracle_lib, "perspective", -0.1, 0.02, 0.85)}
    images = ["a", "b", "c", "d"]
    cam_of = {"a": "pin", "b": "pin", "c": "fish", "d": "dist"}
    n = 300
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)]
    base = rng.integers(0, 255, (n // 2 if guided else n, 128))
    feats, masks, poses = {}, {}, {}
    for im in images:
        R, o = rp._rodrigues(rng.normal(0, 0.1, 3)), rng.normal(0, 0.4, 3)
        poses[im] = _RefPose(R, o)
        Y = (X - o) @ R.T + rng.normal(0, 3e-4, X.shape)
        cam = cams[cam_of[im]]
        b = Y / np.linalg.norm(Y, axis=1, keepdims=True)
        if cam.projection_type == "fisheye":
            l = np.hypot(b[:, 0], b[:, 1])
            u = b[:, :2] * (np.arctan2(l, b[:, 2]) / np.maximum(l, 1e-300))[:, None]
        else:
            u = b[:, :2] / b[:, 2:3]
        r2 = (u**2).sum(1)
        px = cam.focal * u * (1 + r2 * (cam.k1 + cam.k2 * r2))[:, None]
        perm = rng.permutation(n)
        desc = np.clip((np.concatenate([base, base]) if guided else base) + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.float32)
        feats[im] = types.SimpleNamespace(points=np.c_[px[perm], np.ones((n, 2))], descriptors=desc[perm], point_index=perm)
        masks[im] = rng.random(n) > 0.1
    config = {"matcher_type": "BRUTEFORCE", "symmetric_matching": True, "lowes_ratio": 0.8, "robust_matching_min_match": 20,
              "robust_matching_threshold": 0.004, "robust_matching_calib_threshold": 0.004, "five_point_refine_match_iterations": 10,
              "matching_use_filters": False, "matching_use_segmentation": False, "guided_matching_threshold": 0.006, "processes": 1}
    return images, cams, cam_of, feats, masks, poses, config


def test_adhoc_filters_equal_the_reference(ref):
    """apply_adhoc_filters (matching.py:939-1064) executed from the reference's file against the product's vectorised version: static
    matches (below and above the 85 % rule), panorama poles, Vermont and BlackVue watermarks"""
    from opensfm_amd import matching as product

    matching, _ = ref
    rng = np.random.default_rng(8)
    makes = {"plain": {"make": "Canon", "model": "X"}, "vt": {"make": "VTrans_Camera", "model": "VTrans_Camera"}, "bv": {"make": "BlackVue", "model": "DR900"},
             "vt_other": {"make": "VTrans_Camera", "model": "other"}}
    data = types.SimpleNamespace(load_exif=lambda im: makes[im])
    cam = lambda t: types.SimpleNamespace(projection_type=t)  # noqa: E731
    n = 400
    for static_frac in (0.0, 0.3, 0.9):
        for t1, t2 in (("perspective", "perspective"), ("spherical", "perspective"), ("fisheye", "equirectangular"), ("spherical", "spherical")):
            for im1, im2 in (("plain", "plain"), ("vt", "plain"), ("plain", "bv"), ("bv", "vt"), ("vt_other", "plain")):
                p1 = np.c_[rng.uniform(-0.5, 0.5, n), rng.uniform(-0.375, 0.375, n), rng.uniform(0, 1, n)]
                p2 = np.c_[rng.uniform(-0.5, 0.5, n), rng.uniform(-0.375, 0.375, n), rng.uniform(0, 1, n)]
                m = np.c_[rng.permutation(n)[:250], rng.permutation(n)[:250]]
                k = int(static_frac * len(m))
                p2[m[:k, 1], :2] = p1[m[:k, 0], :2] + rng.uniform(-5e-4, 5e-4, (k, 2))  # (almost) static matches
                want = matching.apply_adhoc_filters(data, [tuple(x) for x in m], im1, cam(t1), p1, im2, cam(t2), p2)
                got = product.apply_adhoc_filters(data, m, im1, cam(t1), p1, im2, cam(t2), p2)
                assert [tuple(int(v) for v in g) for g in got] == [tuple(int(v) for v in w) for w in want], (static_frac, t1, t2, im1, im2)
    assert len(product.apply_adhoc_filters(data, np.zeros((0, 2), int), "plain", cam("perspective"), p1, "plain", cam("perspective"), p2)) == 0


def emulate_product_leaves(monkeypatch, oracle_lib):
    """opensfm_amd.matching with every C-ABI call redirected to the host emulations / the oracle (no GPU): what the CPU flow tests run,
    and what the GPU flow tests compare the device against (tests/test_gpu_flow.py)"""
    import test_guided_host as gh
    import test_relpose_core_host as rp
    from opensfm_amd import matching as product

    bearings, relpose_pairs = rp._emulated_calls(rp.build_host())
    ghost = gh.build_host()

    def guided_leaf(f1, f2, ratio, symmetric, maskij=None, bearings1=None, bearings2=None, R=None, t=None, threshold=0.0, ctx=None):
        return gh.host_match(ghost, f1, f2, b1=bearings1, b2=bearings2, R=R, t=t, threshold=threshold, ratio=ratio, symmetric=symmetric)

    class FakeStore:
        ctx = None

        def __init__(self, descs, pts, ctx=None, hamming=False):
            assert not hamming
            self.off = np.r_[0, np.cumsum([len(d) for d in descs])].astype(np.int64)
            self.desc = np.concatenate(descs).astype(np.float32)
            self.pts = np.concatenate([np.asarray(p, float)[:, :2] for p in pts])

        def close(self):
            pass

    def fake_match_pairs(store, ipairs, cfg=None, robust=True, timings=None):
        per = oracle_lib.match_pairs(store.desc, store.pts, store.off, np.asarray(ipairs, np.int32), ratio=0.8, min_match=20, thr=0.004,
                                     conf=0.9999, stage=1 if robust else 0)
        return np.asarray([len(m) for m in per], np.int32), (np.concatenate(per) if len(per) else np.zeros((0, 2), np.int32))

    def fundamental_leaf(p1, p2, threshold, confidence=0.9999, max_iters=1000, ctx=None):
        F, mask, _ = oracle_lib.find_fundamental_ransac(p1, p2, threshold, confidence, max_iters)
        return F, mask.astype(np.uint8).reshape(-1, 1)

    monkeypatch.setattr(product, "DescriptorStore", FakeStore)
    monkeypatch.setattr(product, "default_context", lambda device=None: None)
    monkeypatch.setattr(product, "match_pairs", fake_match_pairs)
    monkeypatch.setattr(product, "pixel_bearing_many", bearings)
    monkeypatch.setattr(product, "relpose_pairs", relpose_pairs)
    # osfm_match_pairs_calibrated composed on the host from its (emulated) stages
    monkeypatch.setattr(product, "match_pairs_calibrated", rp._composed_match_pairs_calibrated(
        product, lambda st: [st.pts[st.off[i]: st.off[i + 1]] for i in range(len(st.off) - 1)]))
    monkeypatch.setattr(product, "_match_guided_leaf", guided_leaf)
    monkeypatch.setattr(product, "find_fundamental_ransac", fundamental_leaf)
    # osfm_match_pairs_guided composed on the host from the same (emulated) leaves
    split = lambda st, arr: [arr[st.off[i]: st.off[i + 1]] for i in range(len(st.off) - 1)]
    monkeypatch.setattr(product, "match_pairs_guided", gh.composed_match_pairs_guided(product, guided_leaf, lambda st: split(st, st.desc),
                                                                                       lambda st: split(st, st.pts)))


@pytest.mark.parametrize("guided,filters,lmeds", [(False, False, False), (True, False, False), (False, True, False), (True, True, False),
                                                   (False, False, True), (True, False, True)])
def test_match_flow_equals_the_product_host_flow(ref, oracle_lib, monkeypatch, guided, filters, lmeds):
    """The reference's match_unwrap_args -> match() (matching.py:182-214, 563-634; guided: 260-337) executed from its own file for
    every pair of a mixed collection, against opensfm_amd.matching.match_images_with_pairs with its C-ABI calls redirected to the
    host emulations: same gates, same dispatch, same unfiltered result for every pair.  lmeds: robust_matching_min_match = 8 and an image
    with a dozen features, so the pinhole pair reaches cv2.findFundamentalMat with 8 <= n < 15 correspondences -- cv2's LMedS branch, which
    the batch path sends through the leaf (rounds 1-5: NotImplementedError)."""
    import test_guided_host as gh
    import test_relpose_core_host as rp
    from opensfm_amd import matching as product

    matching, _ = ref
    rng = np.random.default_rng(17 if guided else 16)
    images, cams, cam_of, feats, masks, poses, config = _collection(oracle_lib, rng, guided)
    exifs = {im: {"camera": cam_of[im]} for im in images}
    pairs = [(a, b) for i, a in enumerate(images) for b in images[i + 1:]]
    config["matching_use_filters"] = filters  # the ad-hoc filters between the descriptor stage and the gates (matching.py:323-334,399-411)
    if lmeds:
        config["robust_matching_min_match"] = 8
        keep = np.flatnonzero(masks["b"])[:12]
        masks["b"][:] = False
        masks["b"][keep] = True
        # LMedS on a dozen correspondences depends on their ORDER (its subsets are drawn by index).  The reference hands cv2 the order a python
        # set of tuples happens to iterate in (matching.py:777); the product hands every robust stage the matches sorted by (i, j) (the one order
        # a batched kernel can produce, and the oracle pipeline's).  Same matches, the product's order, for the reference's robust stage:
        orig_sym = matching.match_brute_force_symmetric
        monkeypatch.setattr(matching, "match_brute_force_symmetric", lambda *a, **k: sorted(orig_sym(*a, **k)))
    exif_of = {"a": {"make": "BlackVue", "model": "DR900"}, "b": {"make": "Canon", "model": "X"}, "c": {"make": "VTrans_Camera", "model": "VTrans_Camera"},
               "d": {"make": "blackvue", "model": "x"}}
    # ---- the reference side ----
    feats_masked = {im: types.SimpleNamespace(points=feats[im].points[masks[im]], descriptors=feats[im].descriptors[masks[im]]) for im in images}
    loader = types.SimpleNamespace(
        load_all_data=lambda data, im, masked=True, segmentation_in_descriptor=False: feats_masked[im],
        load_mask=lambda data, im: masks[im],
        load_bearings=lambda data, im, masked=True, camera=None: camera.pixel_bearing_many(np.array(feats_masked[im].points[:, :2], dtype=float)))
    monkeypatch.setattr(matching.feature_loader, "instance", loader, raising=False)
    monkeypatch.setattr(matching.log, "setup", lambda: None, raising=False)
    data = types.SimpleNamespace(config=config, load_camera_models=lambda: cams, load_features=lambda im: feats[im],
                                 load_features_mask=lambda im, pts: masks[im], load_exif=lambda im: exif_of[im])
    want = {}
    for im1, im2 in pairs:
        _, _, m = matching.match_unwrap_args((im1, im2, cams, exifs, data, {}, poses if guided else None))
        want[im1, im2] = np.asarray(m)
    # ---- the product side, leaves on the emulations ----
    emulate_product_leaves(monkeypatch, oracle_lib)
    got = product.match_images_with_pairs(data, {}, exifs, pairs, poses if guided else None)
    survivors = 0
    def rows(a):  # the reference returns a python set's order (matching.py:777), the product sorts by (i, j): compare as sets of rows
        a = np.asarray(a).reshape(-1, 2)
        return a[np.lexsort((a[:, 1], a[:, 0]))]

    for pair in pairs:
        assert np.array_equal(rows(got[pair]), rows(want[pair])), pair
        survivors += len(want[pair]) > 0
    assert survivors >= (3 if lmeds else 5)
    if lmeds and not guided:
        assert 8 <= len(want["a", "b"]) < 15  # the pinhole pair went through LMedS and survived


def test_segmentation_in_descriptor_flow(ref, oracle_lib, monkeypatch):
    """matching_use_segmentation: the reference's FeatureLoader.load_all_data / _add_segmentation_in_descriptor / mask handling
    (feature_loading.py:106-173) executed from its own file feed the reference's match() with 129-column descriptors; the product's
    match_images_with_pairs builds the same column from the same FeaturesData stand-ins.  The scene is repetitive (every descriptor
    exists twice, so Lowe's test rejects everything) and the two copies carry different classes: only the column lets pairs match."""
    from opensfm_amd import matching as product

    matching, _ = ref
    rng = np.random.default_rng(23)
    images, cams, cam_of, feats, masks, poses, config = _collection(oracle_lib, rng, True)
    config.update(matching_use_segmentation=True, feature_type="HAHOG", hahog_normalize_to_uchar=True)
    exifs = {im: {"camera": cam_of[im]} for im in images}
    pairs = [(a, b) for i, a in enumerate(images) for b in images[i + 1:]]
    n = len(feats["a"].points)
    label_of_point = np.r_[np.zeros(n // 2, int), 1 + rng.integers(0, 3, n - n // 2)]  # a scene point keeps its class in every image
    # ---- the reference's feature_loading.py, from its file ----
    ft = sys.modules["opensfm.features"] = _Stub("opensfm.features")

    class FeaturesData:
        def __init__(self, points, descriptors, colors, semantic):
            self.points, self.descriptors, self.colors, self.semantic = points, descriptors, colors, semantic

        def get_segmentation(self):
            return None if not self.semantic else self.semantic.segmentation

        def mask(self, mask):
            sem = None if not self.semantic else types.SimpleNamespace(segmentation=self.semantic.segmentation[mask])
            return FeaturesData(self.points[mask], self.descriptors[mask], None, sem)

    ft.FeaturesData = FeaturesData
    sys.modules["opensfm"].features = ft
    spec = importlib.util.spec_from_file_location("opensfm.feature_loading", os.path.join(REF, "feature_loading.py"))
    fl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fl)
    full = {}
    for im in images:
        lab = label_of_point[feats[im].point_index]
        full[im] = FeaturesData(feats[im].points, feats[im].descriptors, None, types.SimpleNamespace(segmentation=lab))
    Data = type("Data", (), {})  # hashable by identity: the reference's loaders sit behind lru_cache
    data = Data()
    data.config, data.load_camera_models, data.load_features = config, (lambda: cams), (lambda im: full[im])
    data.load_features_mask, data.load_exif = (lambda im, pts: masks[im]), (lambda im: {"make": "Canon", "model": "X"})
    loader = fl.FeatureLoader()
    monkeypatch.setattr(loader, "_load_features_nocache", lambda d, im: full[im], raising=False)
    monkeypatch.setattr(loader, "load_mask", lambda d, im: masks[im], raising=False)
    monkeypatch.setattr(matching.feature_loader, "instance", loader, raising=False)
    monkeypatch.setattr(matching.log, "setup", lambda: None, raising=False)
    aug = loader.load_all_data(data, "a", masked=True, segmentation_in_descriptor=True)
    assert aug.descriptors.shape[1] == 129 and set(np.unique(aug.descriptors[:, -1])) == {0.0, 35.0, 70.0, 105.0}
    want = {}
    for im1, im2 in pairs:
        _, _, m = matching.match_unwrap_args((im1, im2, cams, exifs, data, {}, None))
        want[im1, im2] = np.asarray(m)
    emulate_product_leaves(monkeypatch, oracle_lib)
    got = product.match_images_with_pairs(data, {}, exifs, pairs, None)

    def rows(a):
        a = np.asarray(a).reshape(-1, 2)
        return a[np.lexsort((a[:, 1], a[:, 0]))]

    for pair in pairs:
        assert np.array_equal(rows(got[pair]), rows(want[pair])), pair
    assert sum(len(w) > 0 for w in want.values()) >= 5
    # without the column nothing passes Lowe's test: the labels decide
    config["matching_use_segmentation"] = False
    assert all(len(m) == 0 for m in product.match_images_with_pairs(data, {}, exifs, pairs, None).values())
    # the reference's guard (feature_loading.py:126-133)
    config.update(matching_use_segmentation=True, feature_type="SIFT")
    with pytest.raises(RuntimeError):
        product.match_images_with_pairs(data, {}, exifs, pairs, None)


def test_matches_files_are_read_by_the_reference_dataset(tmp_path, oracle_lib):
    """The reference's own DataSet.save_matches / load_matches / find_matches (opensfm/dataset.py:344-404, loaded from its file with
    stand-ins for the modules it imports) read what opensfm_amd.matching.write_matches_files wrote, and vice versa; grouping as
    matching.save_matches (matching.py:128-157)."""
    import gzip
    import pickle

    from opensfm_amd import matching as product

    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "opensfm" or k.startswith("opensfm.") or k.startswith("PIL")}
    pkg = _Stub("opensfm")
    pkg.__path__ = [REF]
    sys.modules["opensfm"] = pkg
    for name in ("config", "features", "geo", "io", "masking", "pygeometry", "pymap", "rig", "types", "dataset_base"):
        m = _Stub("opensfm." + name)
        sys.modules["opensfm." + name] = m
        setattr(pkg, name, m)
    sys.modules["opensfm.dataset_base"].DataSetBase = object
    for name in ("PIL", "PIL.PngImagePlugin"):
        sys.modules.setdefault(name, _Stub(name))
    try:
        spec = importlib.util.spec_from_file_location("opensfm.dataset", os.path.join(REF, "dataset.py"))
        dataset = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(dataset)
        io_handler = types.SimpleNamespace(isfile=os.path.isfile, open_rb=lambda p: open(p, "rb"), open_wb=lambda p: open(p, "wb"),
                                           mkdir_p=lambda p: os.makedirs(p, exist_ok=True))
        ds = dataset.DataSet.__new__(dataset.DataSet)
        ds.data_path, ds.io_handler = str(tmp_path), io_handler
        rng = np.random.default_rng(0)
        images = ["a.jpg", "b.jpg", "c.jpg", "d.jpg"]
        pairs = np.array([[0, 1], [0, 2], [1, 2], [2, 3], [1, 3]], np.int32)
        counts = np.array([30, 0, 25, 40, 0], np.int32)
        matches = rng.integers(0, 500, (int(counts.sum()), 2)).astype(np.int32)
        files = product.write_matches_files(str(tmp_path), images, pairs, counts, matches, images_ref=images[:3])
        assert sorted(os.path.basename(f) for f in files) == ["a.jpg_matches.pkl.gz", "b.jpg_matches.pkl.gz", "c.jpg_matches.pkl.gz"]
        per = product.split_matches(counts, matches)
        for (a, b), m in zip(pairs, per):
            got = ds.find_matches(images[a], images[b])  # the reference's restricted unpickler accepts the files
            assert np.array_equal(np.asarray(got).reshape(-1, 2), m) and (len(m) == 0 or got.dtype == np.array([1], dtype=int).dtype)
            assert np.array_equal(np.asarray(ds.find_matches(images[b], images[a])).reshape(-1, 2), m[:, ::-1])
        # byte-level: the reference's save_matches on the same dictionaries produces the same pickles
        for im in images[:3]:
            ours = gzip.decompress(open(os.path.join(str(tmp_path), "matches", im + "_matches.pkl.gz"), "rb").read())
            d = pickle.loads(ours)
            ds.data_path = str(tmp_path / "ref")
            ds.save_matches(im, d)
            theirs = gzip.decompress(open(os.path.join(str(tmp_path), "ref", "matches", im + "_matches.pkl.gz"), "rb").read())
            assert ours == theirs
            ds.data_path = str(tmp_path)
        # the dict-based mirror of matching.save_matches groups the same way
        recorded = {}
        fake = types.SimpleNamespace(save_matches=lambda im, d: recorded.__setitem__(im, d))
        product.save_matches(fake, images[:3], {(images[a], images[b]): m for (a, b), m in zip(pairs, per)})
        assert {im: sorted(d) for im, d in recorded.items()} == {"a.jpg": ["b.jpg", "c.jpg"], "b.jpg": ["c.jpg", "d.jpg"], "c.jpg": ["d.jpg"]}
        with pytest.raises(RuntimeError):
            product.save_matches(fake, ["a.jpg"], {("c.jpg", "d.jpg"): per[3]})
    finally:
        for k in [k for k in sys.modules if k == "opensfm" or k.startswith("opensfm.")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})
