"""CPU: BA oracle -- residual/Jacobian pinned on the reference's own test inputs (golden fixture),
loss functions, LM behaviour on synthetic scenes (style of opensfm/test/test_bundle.py)."""
import json
import os

import numpy as np
import pytest

from opensfm_amd import synthetic

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reprojection_golden.json")


def test_projection_and_jacobian_match_golden(oracle_lib):
    cases = json.load(open(GOLD))
    assert len(cases) >= 18 and len({c.get("model", "perspective") for c in cases}) == 9  # every 2-D projection type
    for c in cases:
        res, Jp, Jc, Jk = oracle_lib.ba_project(c["X"], c["pose"], c["cam"], c["obs"], c["sd"], c.get("model", "perspective"))
        # the reference asserts analytic == autodiff to 1e-14 on O(1) entries (reprojection_errors_test.cc:53)
        small_angle = max(abs(v) for v in c["pose"][:3]) < 1e-3
        for got, key in ((res, "residual"), (Jp, "Jp"), (Jc, "Jc"), (Jk, "Jk")):
            if key not in c:  # constant-camera models: no intrinsics Jacobian
                continue
            want = np.asarray(c[key])
            tol = (2e-13 if small_angle else 1e-13) * max(1.0, np.abs(want).max())
            assert np.allclose(got, want, rtol=0, atol=tol), (key, c["pose"])


def test_losses_match_ceres_definitions(oracle_lib):
    for s in [0.0, 0.3, 1.0, 7.0, 1e4]:
        a = 1.3
        b = a * a
        rho, rho1 = oracle_lib.ba_loss("SoftLOneLoss", a, s)
        assert rho == pytest.approx(2 * b * (np.sqrt(1 + s / b) - 1)) and rho1 == pytest.approx(1 / np.sqrt(1 + s / b))
        rho, rho1 = oracle_lib.ba_loss("CauchyLoss", a, s)
        assert rho == pytest.approx(b * np.log(1 + s / b)) and rho1 == pytest.approx(1 / (1 + s / b))
        rho, rho1 = oracle_lib.ba_loss("HuberLoss", a, s)
        assert rho == pytest.approx(s if s <= b else 2 * a * np.sqrt(s) - b)
        rho, rho1 = oracle_lib.ba_loss("TrivialLoss", a, s)
        assert rho == s and rho1 == 1.0


def test_ba_converges_to_noise_floor(oracle_lib):
    """test_bundle.py:116-165 asserts std of reprojection errors < 5e-3 on its synthetic scene;
    here: inlier RMSE reaches the injected 1 px noise level."""
    pr = synthetic.make_ba_scene(40, 800, 6, seed=3)
    r = oracle_lib.ba_solve(pr, max_iterations=50)
    inl = ~pr["is_outlier"]
    rmse_px = np.sqrt((r["reproj_err"][inl] ** 2).sum(1).mean()) * 2000
    assert rmse_px < 2.2
    assert r["final_cost"] < 0.2 * r["initial_cost"]
    assert np.all(np.diff(r["cost_history"][: r["iterations"] + 1]) <= 1e-9)  # monotone (rejected steps repeat)
    assert np.std(r["reproj_err"][inl]) < 5e-3
    # intrinsics stay near their prior (sd 0.01)
    assert np.allclose(r["cam_params"], pr["cam_params"], atol=0.02)


def test_fixed_blocks_stay_fixed(oracle_lib):
    pr = synthetic.make_ba_scene(12, 200, 5, seed=4)
    pr["cam_fixed"] = np.ones(1, np.uint8)
    pr["shot_fixed"] = np.zeros(12, np.uint8)
    pr["shot_fixed"][:2] = 1
    pr["point_fixed"] = np.zeros(200, np.uint8)
    pr["point_fixed"][::7] = 1
    r = oracle_lib.ba_solve(pr, max_iterations=10)
    assert np.array_equal(r["cam_params"], pr["cam_params"])
    assert np.array_equal(r["shot_pose"][:2], pr["shot_pose"][:2])
    assert np.array_equal(r["points"][::7], pr["points"][::7])
    assert not np.array_equal(r["shot_pose"][2:], pr["shot_pose"][2:])
    assert r["final_cost"] < r["initial_cost"]


def test_zero_iterations_returns_input(oracle_lib):
    pr = synthetic.make_ba_scene(8, 100, 4, seed=5)
    r = oracle_lib.ba_solve(pr, max_iterations=0)
    assert r["iterations"] == 0 and np.array_equal(r["points"], pr["points"])
    assert r["final_cost"] == r["initial_cost"]


def test_up_vector_residual_and_jacobian(oracle_lib):
    oracle = oracle_lib
    """UpVectorError (absolute_motion_errors.h:12-39): r = (R(rot) up/|up| - e_z)/sd; the closed-form
    Jacobian against central differences, and the rotation against Rodrigues' formula."""
    rng = np.random.default_rng(5)
    for _ in range(5):
        pose = np.r_[rng.normal(0, 0.7, 3), rng.normal(0, 1, 3)]
        up = rng.normal(0, 1, 3)
        r, J = oracle.ba_up(pose, up, 0.05)
        a = pose[:3]
        th = np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K
        want = (R @ (up / np.linalg.norm(up)) - np.array([0, 0, 1.0])) / 0.05
        assert np.allclose(r, want, atol=1e-12)
        num = np.zeros((3, 3))
        for k in range(3):
            h = 1e-6
            pp, pm = pose.copy(), pose.copy()
            pp[k] += h
            pm[k] -= h
            num[:, k] = (oracle.ba_up(pp, up, 0.05)[0] - oracle.ba_up(pm, up, 0.05)[0]) / (2 * h)
        assert np.allclose(J, num, atol=1e-6)


def test_up_vector_prior_levels_the_cameras(oracle_lib):
    oracle = oracle_lib
    """With a strong up-vector prior the optimised rotations map the prior vector onto +z."""
    pr = synthetic.make_ba_scene(12, 200, 5, seed=3, outlier_frac=0.0)
    S = len(pr["shot_pose"])
    pr["shot_up"] = np.tile([0.0, -1.0, 0.0], (S, 1))
    pr["shot_up_sigma"] = np.full(S, 1e-3)
    o = oracle.ba_solve(pr, max_iterations=30)
    assert o["final_cost"] < o["initial_cost"]
    res = np.array([np.linalg.norm(oracle.ba_up(p, [0, -1.0, 0], 1.0)[0]) for p in o["shot_pose"]])
    res0 = np.array([np.linalg.norm(oracle.ba_up(p, [0, -1.0, 0], 1.0)[0]) for p in pr["shot_pose"]])
    assert res.mean() <= res0.mean() + 1e-9


def test_projection_values_against_the_reference_forward_functions(oracle_lib):
    """The reference's own PROJ::Forward / DISTO::Forward compiled on this box (oracle/_ref/libcamera_ref.so, built from
    /root/reference by oracle/Makefile): the BA oracle's projection is within 2 ulp of it for the six 2-D camera models whose
    reference code is plain scalar C++ (the oracle multiplies by 1 / z where the reference divides, hence not bit-identical)."""
    import pytest

    import test_oracle_relpose as cams

    if oracle_lib.camera_ref_lib() is None:
        pytest.skip("oracle/_ref/libcamera_ref.so is absent and /root/reference is not mounted")
    rng = np.random.default_rng(8)
    covered = 0
    for model, par in cams._BEARING_CAMERAS.items():
        ang, phi = rng.uniform(0, 0.9, 300), rng.uniform(0, 2 * np.pi, 300)
        X = np.c_[np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)] * rng.uniform(0.5, 20, 300)[:, None]
        ref = oracle_lib.ref_camera(model, par, X, backward=False)
        if ref is None:
            continue
        mine = np.array([oracle_lib.ba_project(x, np.zeros(6), np.array(par), np.zeros(2), 1.0, model)[0] for x in X])
        assert np.abs(mine - ref).max() <= 4.5e-16, model
        covered += 1
    assert covered == 6


def test_jacobians_against_the_reference_derivative_functions(oracle_lib):
    """PerspectiveProjection / FisheyeProjection ::ForwardDerivatives and Disto24::ForwardDerivatives compiled from the reference
    (oracle/_ref/libcamera_ref.so; the chain rule and the focal scaling written out in the adapter): the BA oracle's point and
    intrinsics Jacobians of the two optimised camera models agree to a few ulp (pose = identity)."""
    import pytest

    if oracle_lib.camera_ref_lib() is None:
        pytest.skip("oracle/_ref/libcamera_ref.so is absent and /root/reference is not mounted")
    rng = np.random.default_rng(9)
    for model, par in (("perspective", [-0.1, 0.01, 0.85]), ("fisheye", [-0.05, 0.004, 0.45]), ("perspective", [0.3, 0.1, -0.03])):
        ang, phi = rng.uniform(0, 0.9, 200), rng.uniform(0, 2 * np.pi, 200)
        X = np.c_[np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)] * rng.uniform(0.5, 20, 200)[:, None]
        Jx, Jk = oracle_lib.ref_camera_jacobian(model, par, X)
        for i, x in enumerate(X):
            _, Jp, _, Jkk = oracle_lib.ba_project(x, np.zeros(6), np.array(par), np.zeros(2), 1.0, model)
            assert np.abs(Jp - Jx[i]).max() <= 4e-15 * np.abs(Jx[i]).max()
            assert np.abs(Jkk - Jk[i]).max() <= 4e-15 * np.abs(Jk[i]).max()


def test_full_intrinsics_jacobian_matches_golden(oracle_lib):
    """Groundwork for optimising the intrinsics of the 4-16 parameter cameras: d residual / d (every native parameter), against the
    50-digit golden vectors at the reference's own test camera arrays (brown, fisheye_opencv, fisheye62, fisheye624, dual, radial,
    simple_radial)."""
    gold = json.load(open(GOLD))
    used = 0
    for c in gold:
        if "Jk_full" not in c:
            continue
        res, Jk = oracle_lib.ba_project_intrinsics(c["X"], c["pose"], c["cam"], c["obs"], c["sd"], c["model"])
        want = np.array(c["Jk_full"])
        assert np.allclose(res, c["residual"], rtol=1e-12, atol=1e-12)
        assert Jk.shape == want.shape and np.abs(Jk - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), c["model"]
        used += 1
    assert used == 9


def test_parallel_schur_elimination_reproduces_the_serial_one(oracle_lib):
    """The all-cores elimination (every entry of the reduced system owned by one thread, contributions in the serial order; only the
    camera x camera block and the camera rows of the right-hand side from fixed-chunk partial sums) against the serial loops, and its
    independence of the thread count: identical bits for 1, 3 and all threads."""
    from opensfm_amd import synthetic

    pr = synthetic.make_ba_scene(60, 3000, 6, seed=5)
    kw = dict(max_iterations=6, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    n_all = oracle_lib.num_threads()
    try:
        oracle_lib.ba_set_parallel(False)
        serial = oracle_lib.ba_solve(pr, **kw)
        oracle_lib.ba_set_parallel(True)
        runs = []
        for nt in (1, 3, n_all):
            oracle_lib.set_num_threads(nt)
            runs.append(oracle_lib.ba_solve(pr, **kw))
    finally:
        oracle_lib.set_num_threads(n_all)
        oracle_lib.ba_set_parallel(True)
    for r in runs[1:]:
        assert np.array_equal(r["points"], runs[0]["points"]) and np.array_equal(r["shot_pose"], runs[0]["shot_pose"])
        assert np.allclose(r["cost_history"], runs[0]["cost_history"], rtol=1e-13)  # the cost itself is an OpenMP reduction
    assert np.allclose(serial["cost_history"], runs[0]["cost_history"], rtol=1e-11)
    assert np.abs(serial["points"] - runs[0]["points"]).max() < 1e-9 and np.abs(serial["shot_pose"] - runs[0]["shot_pose"]).max() < 1e-9


def test_ragged_tracks_scene(oracle_lib):
    """make_ba_scene(ragged=True): track lengths 2 + Poisson, sightings missing inside a window -- the shape a feature tracker leaves.
    The plain scene of the same seed is untouched, every point keeps two sightings, shot sets are no longer shared, and the oracle
    converges on it as on the plain one."""
    from opensfm_amd import synthetic

    plain = synthetic.make_ba_scene(40, 900, 6, seed=3)
    again = synthetic.make_ba_scene(40, 900, 6, seed=3)
    assert all(np.array_equal(plain[k], again[k]) for k in plain)
    pr = synthetic.make_ba_scene(40, 900, 6, seed=3, ragged=True)
    n = np.bincount(pr["obs_point"], minlength=900)
    assert n.min() >= 2 and n.max() > 6 and abs(n.mean() - 0.85 * 6) < 1.0
    assert (np.diff(pr["obs_shot"]) >= 0).all()  # shot-major
    sets = {pr["obs_shot"][pr["obs_point"] == p].tobytes() for p in range(900)}
    assert len(sets) > 300  # the plain scene has at most 35 (one per window start)
    r = oracle_lib.ba_solve(pr, max_iterations=30)
    assert r["final_cost"] < 0.2 * r["initial_cost"]


def test_parallel_skyline_factor_equals_the_serial_one(oracle_lib):
    """The oracle's right-looking skyline Cholesky on all cores (used for wide profiles: block surveys) performs, per entry, the serial
    up-looking factor's operations in the same order: the whole solve -- every point and pose after four LM iterations -- is identical bit for bit whether it is
    forced on (ba_set_parallel(2)) or not, on a sequence band and on a block survey whose shots are not numbered along the band."""
    from opensfm_amd import synthetic

    kw = dict(max_iterations=4, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    n_all = oracle_lib.num_threads()
    try:
        for pr in (synthetic.make_ba_scene(50, 2000, 6, seed=9), synthetic.make_ba_scene_grid(6, 14, 1500, 9, seed=3)):
            oracle_lib.set_num_threads(max(2, min(n_all, 7)))
            oracle_lib.ba_set_parallel(1)
            a = oracle_lib.ba_solve(pr, **kw)
            oracle_lib.ba_set_parallel(2)
            b = oracle_lib.ba_solve(pr, **kw)
            assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["shot_pose"], b["shot_pose"])
            assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-13)  # the cost itself is an OpenMP reduction
    finally:
        oracle_lib.set_num_threads(n_all)
        oracle_lib.ba_set_parallel(1)
