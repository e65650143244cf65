"""The product's calibrated-matching numerics (opensfm_amd/csrc/relpose_core.h) and the round-based orchestration of the LO-RANSAC
(relpose_rounds.h: walk / solve5 / solveN over work lists), compiled for the host with loops in place of kernels
(tests/native/relpose_core_host.cpp), against the CPU oracle -- bit for bit.  This pins everything of relpose.hip except the GPU
wave policy, the kernel wrappers and the host loop that launches the rounds."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def build_host():
    src = os.path.join(HERE, "native", "relpose_core_host.cpp")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "relpose_core_host.so")
    deps = [src] + [os.path.join(HERE, "..", "opensfm_amd", "csrc", h) for h in ("relpose_core.h", "relpose_rounds.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-o", so, src])
    return C.CDLL(so)


@pytest.fixture(scope="module")
def host():
    return build_host()


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


def _composed_match_pairs_calibrated(matching, pts_of_store):
    """What osfm_match_pairs_calibrated does on the device, composed on the host from its stages (descriptor stage, gate, bearings,
    gather, relative-pose batch, compaction, gate): stands in for the C-ABI call where the tests redirect the stages to emulations."""

    def run(store, pairs, cameras, points=None, config=None, timings=None):
        points = points if points is not None else pts_of_store(store)
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        min_match = int(matching._cfg(config, "robust_matching_min_match"))
        counts, matches = matching.match_pairs(store, pairs, config, robust=False)
        per_pair = matching.split_matches(counts, matches)
        keep = [p for p, m in enumerate(per_pair) if len(m) >= max(min_match, 1)]
        out_counts = np.zeros(len(pairs), np.int32)
        if not keep:
            return out_counts, np.zeros((0, 2), np.int32)
        bearings = {im: matching.pixel_bearing_many(cameras[im], np.asarray(points[im], np.float64)[:, :2], store.ctx)
                    for im in sorted({int(i) for p in keep for i in pairs[p]})}
        b1 = np.concatenate([bearings[int(pairs[p, 0])][per_pair[p][:, 0]] for p in keep])
        b2 = np.concatenate([bearings[int(pairs[p, 1])][per_pair[p][:, 1]] for p in keep])
        off = np.r_[0, np.cumsum([len(per_pair[p]) for p in keep])].astype(np.int64)
        _, mask, _ = matching.relpose_pairs(b1, b2, off, matching._cfg(config, "robust_matching_calib_threshold"), "match", 1000, 0.99, True, 10,
                                            matching._cfg(config, "five_point_refine_match_iterations"), store.ctx)
        chunks = []
        for k, p in enumerate(keep):
            rm = per_pair[p][mask[off[k]: off[k + 1]]]
            if len(rm) >= min_match and len(rm) > 0:
                out_counts[p] = len(rm)
                chunks.append(rm)
        return out_counts, (np.concatenate(chunks) if chunks else np.zeros((0, 2), np.int32))

    return run


def _emulated_calls(host):
    """(pixel_bearing_many, relpose_pairs) of opensfm_amd.matching served by the host emulation instead of the C ABI."""

    def bearings(camera, points, ctx=None):
        from opensfm_amd import matching

        model, par = matching.camera_parameters(camera)
        px = np.ascontiguousarray(np.asarray(points, np.float64)[:, :2])
        out = np.zeros((len(px), 3))
        host.host_pixel_bearings_generic(model, _p(par, C.c_double), _p(px, C.c_double), len(px), _p(out, C.c_double))
        return out

    def relpose_pairs(b1, b2, offsets, threshold, mode="match", iterations=1000, probability=0.99, use_lo=True, lo_iterations=10,
                      refine_iterations=10, ctx=None):
        assert mode == "match"
        mask = np.zeros(len(b1), bool)
        for p in range(len(offsets) - 1):
            s = slice(int(offsets[p]), int(offsets[p + 1]))
            x, y = np.ascontiguousarray(b1[s]), np.ascontiguousarray(b2[s])
            k = len(x)
            R, t, models, info = np.zeros(9), np.zeros(3), np.zeros(24), np.zeros(2, np.int32)
            m = np.zeros(max(k, 1), np.uint8)
            host.host_robust_match_calibrated(_p(x, C.c_double), _p(y, C.c_double), k, C.c_double(threshold), iterations,
                                              C.c_double(probability), int(use_lo), lo_iterations, refine_iterations, _p(R, C.c_double),
                                              _p(t, C.c_double), _p(m, C.c_uint8), _p(models, C.c_double), _p(info, C.c_int32))
            mask[s] = m[:k].astype(bool)
        return [{}] * (len(offsets) - 1), mask, 0.0

    return bearings, relpose_pairs


def test_five_point_and_pose_bits(host, oracle_lib):
    rng = np.random.default_rng(0)
    for trial in range(200):
        b1, b2, _ = _scene(rng, 5, outliers=0.0, noise=1e-2 if trial % 2 else 0.0)
        ref = oracle_lib.essential_five_points(b1, b2)
        Es = np.zeros(90)
        k = host.host_essential_five_points(_p(b1, C.c_double), _p(b2, C.c_double), _p(Es, C.c_double))
        assert k == len(ref)
        got = Es[: 9 * k].reshape(k, 3, 3)
        assert np.array_equal(got.view(np.uint64), np.asarray(ref).reshape(k, 3, 3).view(np.uint64))
        for E in got:
            want = oracle_lib.relative_pose_from_essential(E, b1, b2)
            RT = np.zeros(12)
            ok = host.host_relative_pose_from_essential(_p(np.ascontiguousarray(E), C.c_double), _p(b1, C.c_double), _p(b2, C.c_double), 5,
                                                        _p(RT, C.c_double))
            assert bool(ok) == (want is not None)
            if ok:
                assert np.array_equal(RT.view(np.uint64), np.ascontiguousarray(want).reshape(-1).view(np.uint64))
    # degenerate input must not loop or crash
    z = np.zeros((5, 3))
    assert host.host_essential_five_points(_p(z, C.c_double), _p(z, C.c_double), _p(np.zeros(90), C.c_double)) == 0


def test_bearings_inliers_and_picks_bits(host, oracle_lib):
    rng = np.random.default_rng(1)
    for model in (0, 1):
        cam = np.array([-0.1, 0.01, 0.9])
        px = np.ascontiguousarray(rng.uniform(-0.6, 0.6, (500, 2)))
        px[0] = 0.0
        out = np.zeros((500, 3))
        host.host_pixel_bearings(model, _p(cam, C.c_double), _p(px, C.c_double), 500, _p(out, C.c_double))
        assert np.array_equal(out.view(np.uint64), oracle_lib.pixel_bearings(model, cam, px).view(np.uint64))
    import test_oracle_relpose as cams

    for model, par in list(cams._BEARING_CAMERAS.items()) + [("spherical", [])]:  # every projection type, bit for bit
        ang, phi = rng.uniform(0, 0.9, 300), rng.uniform(0, 2 * np.pi, 300)
        X = np.c_[np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)]
        px = np.ascontiguousarray(rng.uniform(-0.3, 0.3, (300, 2)) if model == "spherical" else cams._forward(model, par, X))
        px[0] = 0.0
        p16 = np.r_[np.asarray(par, float), np.zeros(16 - len(par))]
        out = np.zeros((300, 3))
        host.host_pixel_bearings_generic(oracle_lib.BEARING_MODELS[model], _p(p16, C.c_double), _p(px, C.c_double), 300, _p(out, C.c_double))
        want = oracle_lib.pixel_bearings_generic(model, par, px)
        assert np.array_equal(out.view(np.uint64), want.view(np.uint64)), model  # NaN at the centre of the dual model included
    b1, b2, _ = _scene(rng, 400)
    R = _rodrigues(rng.normal(0, 0.2, 3))
    t = rng.normal(0, 1, 3)
    for thr in (0.004, 0.05, 0.5):
        mask = np.zeros(400, np.uint8)
        host.host_inliers_bearings(_p(b1, C.c_double), _p(b2, C.c_double), 400, _p(np.ascontiguousarray(R), C.c_double), _p(t, C.c_double),
                                   C.c_double(thr), _p(mask, C.c_uint8))
        assert np.array_equal(mask.astype(bool), oracle_lib.inliers_bearings(b1, b2, R, t, thr))
    # the 100 correspondences the refinement looks at: float(rand()) / RAND_MAX * n after srand(42)
    r = oracle_lib.glibc_rand(42, 100)
    for n in (8, 100, 1777):
        picked = np.zeros(100, np.int32)
        host.host_refinement_picks(n, _p(picked, C.c_int32))
        want = np.minimum((r.astype(np.float32) / np.float32(2147483647) * np.float32(n)).astype(np.int64), n - 1)
        assert np.array_equal(picked, want)


@pytest.mark.parametrize("width", [1, 7, 64])
def test_ransac_decision_sequence_bits(host, oracle_lib, width):
    """width = cap on how many iterations are sampled and solved speculatively per round: 1 is the reference's sequential loop;
    7 and 64 (= the kernel's 16) exercise discarded speculation when the local optimisation fires in the middle of a batch."""
    rng = np.random.default_rng(2)
    cases = [(5, 0.0), (6, 0.0), (9, 0.2), (30, 0.3), (200, 0.5), (500, 0.2), (120, 0.9)]
    for n, outl in cases:
        for use_lo, iters in ((1, 1000), (0, 150), (1, 37)):
            b1, b2, _ = _scene(rng, n, outliers=outl)
            want = oracle_lib.ransac_relative_pose(b1, b2, 0.004, iters, 0.99, bool(use_lo), 10)
            model, lo = np.zeros(12), np.zeros(12)
            inl = np.zeros(n, np.int32)
            it = C.c_int(0)
            score = host.host_ransac_relative_pose(width, _p(b1, C.c_double), _p(b2, C.c_double), n, C.c_double(0.004), iters, C.c_double(0.99),
                                                   use_lo, 10, _p(model, C.c_double), _p(lo, C.c_double), _p(inl, C.c_int32), C.byref(it))
            assert (score, it.value) == (want["score"], want["iterations"]), (n, outl, use_lo, iters)
            assert np.array_equal(inl[:score], want["inliers"])
            assert np.array_equal(model.view(np.uint64), want["model"].reshape(-1).view(np.uint64))
            assert np.array_equal(lo.view(np.uint64), want["lo_model"].reshape(-1).view(np.uint64))
    # fewer than five correspondences: nothing runs
    b1, b2, _ = _scene(rng, 4)
    it = C.c_int(7)
    assert host.host_ransac_relative_pose(width, _p(b1, C.c_double), _p(b2, C.c_double), 4, C.c_double(0.004), 100, C.c_double(0.99), 1, 10,
                                          _p(np.zeros(12), C.c_double), _p(np.zeros(12), C.c_double), _p(np.zeros(4, np.int32), C.c_int32),
                                          C.byref(it)) == 0 and it.value == 0


def test_sampler_on_the_tabulated_stream(host, oracle_lib):
    """draw_sample_tab on the table of raw std::mt19937(42) outputs == the oracle's sampler (itself pinned against the reference's
    random_sampler.h compiled on the build box)"""
    for n, size in ((5, 5), (6, 5), (13, 5), (300, 5), (40, 12), (12, 12), (100000, 5)):
        out = np.zeros((200, size), np.int32)
        host.host_draw_samples(n, size, 200, _p(out, C.c_int32))
        assert np.array_equal(out, oracle_lib.ransac_draws(n, size, 200))


def test_a_batch_of_pairs_goes_through_the_rounds_together(host, oracle_lib):
    """many pairs of different sizes share the work lists: every pair still gets the oracle's result, in any speculation width"""
    rng = np.random.default_rng(22)
    sizes = [4, 5, 9, 30, 77, 200, 400, 64, 130, 12, 0, 300]
    outl = [0.0, 0.0, 0.2, 0.3, 0.5, 0.5, 0.8, 0.95, 0.1, 0.0, 0.0, 0.4]
    scenes = [_scene(rng, n, outliers=o) if n else (np.zeros((0, 3)), np.zeros((0, 3)), None) for n, o in zip(sizes, outl)]
    b1 = np.ascontiguousarray(np.concatenate([s[0] for s in scenes]))
    b2 = np.ascontiguousarray(np.concatenate([s[1] for s in scenes]))
    off = np.r_[0, np.cumsum(sizes)].astype(np.int64)
    want = [oracle_lib.ransac_relative_pose(s[0], s[1], 0.004, 1000, 0.99, True, 10) if n >= 5 else None for s, n in zip(scenes, sizes)]
    try:
        for width in (1, 3, 16):
            host.host_set_max_width(width)
            scores, iters = np.zeros(len(sizes), np.int32), np.zeros(len(sizes), np.int32)
            models, mask = np.zeros((len(sizes), 24)), np.zeros(max(len(b1), 1), np.uint8)
            rounds = host.host_rounds_ransac_batch(_p(b1, C.c_double), _p(b2, C.c_double), _p(off, C.c_int64), len(sizes), C.c_double(0.004), 1000,
                                                   C.c_double(0.99), 1, 10, _p(scores, C.c_int32), _p(iters, C.c_int32), _p(models, C.c_double),
                                                   _p(mask, C.c_uint8))
            assert rounds > 1
            for p, w in enumerate(want):
                if w is None:
                    assert scores[p] == 0 and iters[p] == 0 and not mask[off[p]:off[p + 1]].any()
                    continue
                assert (scores[p], iters[p]) == (w["score"], w["iterations"]), (p, width)
                assert np.array_equal(np.flatnonzero(mask[off[p]:off[p + 1]]), w["inliers"])
                assert np.array_equal(models[p, :12].view(np.uint64), w["model"].reshape(-1).view(np.uint64))
                assert np.array_equal(models[p, 12:].view(np.uint64), w["lo_model"].reshape(-1).view(np.uint64))
    finally:
        host.host_set_max_width(16)


def test_group_eigenvalues_equal_the_single_lane_ones(host, oracle_lib):
    """stage B1 as the GPU runs it (real_eigenvalues10_group: sixteen lanes share the matrix, lane j owns column j / row j; the harness
    runs its lane loops in turn) against stage B1 by one lane, on every five-point problem of a batch: counts and bits"""
    host.host_eig_problems.restype = C.c_long
    host.host_eig_mismatches.restype = C.c_long
    before = host.host_eig_problems()
    rng = np.random.default_rng(5)
    sizes = [40, 200, 300, 64, 9, 150]
    scenes = [_scene(rng, n, outliers=o, noise=nz) for n, o, nz in zip(sizes, (0.3, 0.6, 0.8, 0.0, 0.2, 0.5), (1e-3, 1e-3, 1e-2, 0.0, 1e-4, 3e-3))]
    b1 = np.ascontiguousarray(np.concatenate([s[0] for s in scenes]))
    b2 = np.ascontiguousarray(np.concatenate([s[1] for s in scenes]))
    off = np.r_[0, np.cumsum(sizes)].astype(np.int64)
    scores, iters = np.zeros(len(sizes), np.int32), np.zeros(len(sizes), np.int32)
    models, mask = np.zeros((len(sizes), 24)), np.zeros(len(b1), np.uint8)
    host.host_rounds_ransac_batch(_p(b1, C.c_double), _p(b2, C.c_double), _p(off, C.c_int64), len(sizes), C.c_double(0.004), 1000, C.c_double(0.99),
                                  1, 10, _p(scores, C.c_int32), _p(iters, C.c_int32), _p(models, C.c_double), _p(mask, C.c_uint8))
    assert host.host_eig_problems() - before > 500
    assert host.host_eig_mismatches() == 0


def test_refinement_bits(host, oracle_lib):
    rng = np.random.default_rng(3)
    for trial in range(20):
        n = int(rng.integers(8, 400))
        b1, b2, good = _scene(rng, n, outliers=0.0, noise=2e-3)
        r = oracle_lib.ransac_relative_pose(b1, b2, 0.004, 200)
        RT0 = r["lo_model"].copy()
        if trial % 3 == 0:  # a perturbed start: more LM iterations, rejected steps
            RT0[:, :3] = _rodrigues(rng.normal(0, 0.02, 3)) @ RT0[:, :3]
        want, it_w, costs_w = oracle_lib.relative_pose_refinement(RT0, b1, b2, 10)
        RT = np.ascontiguousarray(RT0.reshape(-1).copy())
        costs = np.zeros(2)
        it = host.host_relative_pose_refinement(_p(RT, C.c_double), _p(b1, C.c_double), _p(b2, C.c_double), n, 10, _p(costs, C.c_double))
        assert it == it_w
        assert np.array_equal(RT.view(np.uint64), want.reshape(-1).view(np.uint64))
        assert tuple(costs) == costs_w


def test_robust_match_calibrated_bits(host, oracle_lib):
    rng = np.random.default_rng(4)
    for n, outl in ((7, 0.0), (8, 0.0), (40, 0.3), (300, 0.4), (1000, 0.6), (150, 0.97)):
        b1, b2, good = _scene(rng, n, outliers=outl)
        want = oracle_lib.robust_match_calibrated_bearings(b1, b2, 0.004, 1000, 0.99, True, 10, 10)
        R, t, models, info = np.zeros(9), np.zeros(3), np.zeros(24), np.zeros(2, np.int32)
        mask = np.zeros(n, np.uint8)
        cnt = host.host_robust_match_calibrated(_p(b1, C.c_double), _p(b2, C.c_double), n, C.c_double(0.004), 1000, C.c_double(0.99), 1, 10, 10,
                                                _p(R, C.c_double), _p(t, C.c_double), _p(mask, C.c_uint8), _p(models, C.c_double),
                                                _p(info, C.c_int32))
        assert cnt == want["mask"].sum()
        assert np.array_equal(mask.astype(bool), want["mask"])
        assert (int(info[0]), int(info[1])) == (want["score"], want["iterations"])
        if cnt:
            assert np.array_equal(R.view(np.uint64), want["R"].reshape(-1).view(np.uint64))
            assert np.array_equal(t.view(np.uint64), want["t"].view(np.uint64))
            if outl < 0.9:  # the inliers are the true correspondences (a few noisy ones may fall outside the threshold)
                assert (want["mask"] & ~good).sum() <= 0.02 * n and (want["mask"] & good).sum() >= 0.8 * good.sum()


def test_product_host_function_on_the_emulation(host, oracle_lib, monkeypatch):
    """opensfm_amd.matching.robust_match_calibrated (the reference's signature) with its two C-ABI calls redirected to the
    host emulation: checks the Python glue and the logic of the GPU leaf test without a GPU."""
    from opensfm_amd import matching

    import test_gpu_zz_relpose as gpu_tests

    bearings, relpose_pairs = _emulated_calls(host)
    monkeypatch.setattr(matching, "pixel_bearing_many", bearings)
    monkeypatch.setattr(matching, "relpose_pairs", relpose_pairs)
    gpu_tests.test_robust_match_calibrated_leaf(oracle_lib)


def test_match_images_with_pairs_routes_calibrated_pairs(host, oracle_lib, monkeypatch):
    """match_images_with_pairs on a mixed collection (two pinhole cameras, a fisheye, a distorted perspective): pinhole
    pairs go to the fused launch, the others through match_pairs_calibrated = the reference's match() flow
    (matching.py:563-634) with robust_match_calibrated; the C-ABI calls are redirected (descriptor stage: canned matches,
    bearings / relative pose: host emulation) so that the host logic -- gates, gathering, unfiltering -- runs without a GPU."""
    from types import SimpleNamespace

    from opensfm_amd import matching

    import test_gpu_zz_relpose as gpu_tests

    rng = np.random.default_rng(7)
    cams = {"pin": SimpleNamespace(projection_type="perspective", k1=0.0, k2=0.0, focal=0.8),
            "fish": SimpleNamespace(projection_type="fisheye", k1=-0.05, k2=0.01, focal=0.7),
            "dist": SimpleNamespace(projection_type="perspective", k1=-0.1, k2=0.02, focal=0.85)}
    images = ["a", "b", "c", "d"]
    cam_of = {"a": "pin", "b": "pin", "c": "fish", "d": "dist"}
    n = 260
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)]

    def forward(cam, b):
        if cam.projection_type == "fisheye":
            l = np.hypot(b[:, 0], b[:, 1])
            u = b[:, :2] * (np.arctan2(l, b[:, 2]) / np.maximum(l, 1e-300))[:, None]
        else:
            u = b[:, :2] / b[:, 2:3]
        r2 = (u**2).sum(1)
        return cam.focal * u * (1 + r2 * (cam.k1 + cam.k2 * r2))[:, None]

    feats, masks, order = {}, {}, {}
    for im in images:
        R = _rodrigues(rng.normal(0, 0.15, 3))
        t = rng.normal(0, 0.4, 3)
        Y = X @ R.T + t + rng.normal(0, 1e-3, X.shape)
        perm = rng.permutation(n)
        order[im] = perm  # feature f of the image shows point perm[f]
        feats[im] = SimpleNamespace(points=np.c_[forward(cams[cam_of[im]], Y / np.linalg.norm(Y, axis=1, keepdims=True))[perm], np.ones((n, 2))],
                                    descriptors=np.zeros((n, 128), np.float32))
        masks[im] = rng.random(n) > 0.1
    data = SimpleNamespace(config={"matcher_type": "BRUTEFORCE", "robust_matching_min_match": 20, "robust_matching_calib_threshold": 0.004,
                                   "five_point_refine_match_iterations": 10, "lowes_ratio": 0.8, "symmetric_matching": True},
                           load_camera_models=lambda: cams, load_features=lambda im: feats[im],
                           load_features_mask=lambda im, pts: masks[im])
    exifs = {im: {"camera": cam_of[im]} for im in images}
    pairs = [(a, b) for i, a in enumerate(images) for b in images[i + 1:]]

    class FakeStore:
        ctx = None

        def __init__(self, descs, pts, ctx=None):
            self.pts = pts

        def close(self):
            pass

    stage = {}

    def fake_match_pairs(store, ipairs, config=None, robust=True, timings=None):
        """descriptor stage: the true correspondences among the unmasked features, 25 % replaced by wrong ones; the pair (a, c)
        gets only 15 so that the first min-match gate fires."""
        counts, chunks = [], []
        for a, b in np.asarray(ipairs):
            ia, ib = images[a], images[b]
            fa = np.flatnonzero(masks[ia])
            fb = np.flatnonzero(masks[ib])
            pos_b = {int(order[ib][f]): k for k, f in enumerate(fb)}
            m = np.array([(k, pos_b[int(order[ia][f])]) for k, f in enumerate(fa) if int(order[ia][f]) in pos_b], np.int32)
            wrong = rng.random(len(m)) < 0.25
            m[wrong, 1] = rng.integers(0, len(fb), wrong.sum())
            if (ia, ib) == ("a", "c"):
                m = m[:15]
            stage[(ia, ib, bool(robust))] = m
            counts.append(len(m))
            chunks.append(m)
        return np.asarray(counts, np.int32), np.concatenate(chunks) if chunks else np.zeros((0, 2), np.int32)

    bearings, relpose_pairs = _emulated_calls(host)
    monkeypatch.setattr(matching, "DescriptorStore", FakeStore)
    monkeypatch.setattr(matching, "match_pairs", fake_match_pairs)
    monkeypatch.setattr(matching, "pixel_bearing_many", bearings)
    monkeypatch.setattr(matching, "relpose_pairs", relpose_pairs)
    monkeypatch.setattr(matching, "match_pairs_calibrated", _composed_match_pairs_calibrated(matching, lambda st: st.pts))
    got = matching.match_images_with_pairs(data, {}, exifs, pairs)
    assert set(got) == set(pairs)
    assert ("a", "b", True) in stage and all((a, b, False) in stage for a, b in pairs if (a, b) != ("a", "b"))
    for (ia, ib) in pairs:
        if (ia, ib) == ("a", "b"):  # pinhole pair: whatever the fused launch returned, unfiltered
            want = matching.unfilter_matches(stage[ia, ib, True], masks[ia], masks[ib])
            assert np.array_equal(got[ia, ib], want)
            continue
        m = stage[ia, ib, False]
        if len(m) < 20:
            assert (ia, ib) == ("a", "c") and len(got[ia, ib]) == 0
            continue
        c1, c2 = cams[cam_of[ia]], cams[cam_of[ib]]
        rm = oracle_lib.robust_match_calibrated(feats[ia].points[masks[ia]], feats[ib].points[masks[ib]], [c1.k1, c1.k2, c1.focal],
                                                [c2.k1, c2.k2, c2.focal], c1.projection_type, c2.projection_type, m, 0.004, 10)
        assert len(rm) >= 20
        want = matching.unfilter_matches(rm, masks[ia], masks[ib])
        assert np.array_equal(got[ia, ib], np.asarray(want))
        # and they are true correspondences
        assert (order[ia][got[ia, ib][:, 0]] == order[ib][got[ia, ib][:, 1]]).mean() > 0.98


def test_gpu_pipeline_test_logic_on_the_emulation(host, oracle_lib, monkeypatch):
    """tests/test_gpu_zz_relpose.py::test_match_pairs_calibrated_pipeline with the descriptor stage served by the oracle and the
    bearing / relative-pose calls by the host emulation: the expectations of the GPU test are themselves checked on CPU."""
    from opensfm_amd import matching

    import test_gpu_zz_relpose as gpu_tests

    class FakeStore:
        ctx = None

        @classmethod
        def from_packed(cls, desc, pts, offsets, ctx=None):
            s = cls()
            s.desc, s.pts, s.offsets = desc, pts, offsets
            return s

    def fake_match_pairs(store, ipairs, config=None, robust=True, timings=None):
        assert not robust
        per = oracle_lib.match_pairs(store.desc.astype(np.float32), store.pts, store.offsets, ipairs, stage=0)
        return np.asarray([len(m) for m in per], np.int32), np.concatenate(per)

    bearings, relpose_pairs = _emulated_calls(host)
    monkeypatch.setattr(matching, "DescriptorStore", FakeStore)
    monkeypatch.setattr(matching, "match_pairs", fake_match_pairs)
    monkeypatch.setattr(matching, "pixel_bearing_many", bearings)
    monkeypatch.setattr(matching, "relpose_pairs", relpose_pairs)
    monkeypatch.setattr(matching, "match_pairs_calibrated", _composed_match_pairs_calibrated(
        matching, lambda st: [st.pts[st.offsets[i]: st.offsets[i + 1]] for i in range(len(st.offsets) - 1)]))
    gpu_tests.test_match_pairs_calibrated_pipeline(oracle_lib, None)


def test_gpu_bearing_tests_logic_on_the_emulation(host, oracle_lib, monkeypatch):
    """The two bearing tests of tests/test_gpu_zz_relpose.py with pixel_bearing_many served by the host emulation (this also runs
    matching.camera_parameters: attribute names and native parameter order of every projection type)."""
    from opensfm_amd import matching

    import test_gpu_zz_relpose as gpu_tests

    bearings, _ = _emulated_calls(host)

    def checked(camera, points, ctx=None):
        if camera.projection_type not in matching._BEARING_MODELS:
            raise NotImplementedError(camera.projection_type)
        return bearings(camera, points, ctx)

    monkeypatch.setattr(matching, "pixel_bearing_many", checked)
    gpu_tests.test_pixel_bearings(oracle_lib)
    gpu_tests.test_pixel_bearings_every_projection_type(oracle_lib)
