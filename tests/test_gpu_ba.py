"""GPU parity: Schur-PCG Levenberg-Marquardt == CPU oracle (exact Schur + Cholesky) within the
north-star tolerance: reprojection RMSE within 1e-4 px after the same LM iteration count."""
import numpy as np
import pytest

from opensfm_amd import synthetic

pytestmark = pytest.mark.gpu
PX = 2000.0  # max(w, h) of the synthetic camera (synthetic_scene.py:31-32): normalized -> pixels
NO_TOL = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)


def _rmse_px(err, mask=None):
    e = err if mask is None else err[mask]
    return float(np.sqrt((e**2).sum(1).mean()) * PX)


def test_zero_iterations_reprojection_errors(oracle_lib, gpu_ctx):
    """ComputeReprojectionErrors (bundle_adjuster.cc:1196-1208) with no optimisation: pure residual
    kernel against the CPU statement, ~1 ulp."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(20, 300, 5, seed=11)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 0})
    o = oracle_lib.ba_solve(pr, max_iterations=0)
    assert g["iterations"] == 0
    assert np.allclose(g["reproj_err"], o["reproj_err"], rtol=0, atol=1e-14)
    assert g["initial_cost"] == pytest.approx(o["initial_cost"], rel=1e-13)
    assert np.array_equal(g["points"], pr["points"])


@pytest.mark.parametrize("shots,points,track,seed,loss", [(30, 600, 6, 1, "SoftLOneLoss"), (60, 1500, 8, 2, "SoftLOneLoss"),
                                                          (25, 400, 5, 3, "TrivialLoss"), (25, 400, 5, 4, "HuberLoss"),
                                                          (25, 400, 5, 5, "CauchyLoss")])
def test_lm_trajectory_equals_oracle(oracle_lib, gpu_ctx, shots, points, track, seed, loss):
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(shots, points, track, seed=seed)
    iters = 12
    cfg = {"loss_function": loss, "loss_function_threshold": 1.0, "bundle_max_iterations": iters}
    g = bundle.bundle_arrays(pr, cfg, **NO_TOL)
    o = oracle_lib.ba_solve(pr, loss=loss, max_iterations=iters, **NO_TOL)
    assert g["iterations"] == o["iterations"] == iters
    assert g["successful_steps"] == o["successful_steps"]
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    # north star: RMSE within 1e-4 px after the same LM iteration count
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    assert abs(g["rmse_final"] - o["rmse_final"]) * PX < 1e-4
    assert np.allclose(g["cam_params"], o["cam_params"], atol=1e-6)
    assert np.allclose(g["shot_pose"], o["shot_pose"], atol=1e-5)
    if loss != "TrivialLoss":  # a robust loss is needed to shrug off the 5 % gross outliers
        assert _rmse_px(g["reproj_err"], ~pr["is_outlier"]) < 2.5


def test_default_termination_matches_oracle(oracle_lib, gpu_ctx):
    """Ceres default tolerances (function 1e-6): same termination reason, same iteration count +-1."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(30, 600, 6, seed=21, outlier_frac=0.0)
    g = bundle.bundle_arrays(pr, {"loss_function": "TrivialLoss"})
    o = oracle_lib.ba_solve(pr, loss="TrivialLoss")
    assert g["termination"] == o["termination"] == 1
    assert abs(g["iterations"] - o["iterations"]) <= 1
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


def test_fixed_blocks(oracle_lib, gpu_ctx):
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(12, 200, 5, seed=4)
    pr["cam_fixed"] = np.ones(1, np.uint8)
    pr["shot_fixed"] = np.zeros(12, np.uint8)
    pr["shot_fixed"][:2] = 1
    pr["point_fixed"] = np.zeros(200, np.uint8)
    pr["point_fixed"][::7] = 1
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=8, **NO_TOL)
    assert np.array_equal(g["cam_params"], pr["cam_params"])
    assert np.array_equal(g["shot_pose"][:2], pr["shot_pose"][:2])
    assert np.array_equal(g["points"][::7], pr["points"][::7])
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)


def test_no_gps_free_gauge(oracle_lib, gpu_ctx):
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(20, 400, 5, seed=6, use_gps=False)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 10}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=10, **NO_TOL)
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-6)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


def test_lund_scale_config(oracle_lib, gpu_ctx):
    """BASELINE.json configs[2]: 500 cams / 50k points / 300k observations, 20 LM iterations."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(500, 50000, 6, seed=42)
    assert len(pr["obs_shot"]) == 300000
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 20}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=20, **NO_TOL)
    assert g["iterations"] == o["iterations"] == 20
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    # once the cost moves by < 1e-8 relative per step, accept/reject decisions are rounding noise
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-6)
    assert g["num_images"] == 500 and g["num_points"] == 50000 and g["num_reprojections"] == 300000


def test_bundle_adjuster_builder_api(gpu_ctx):
    """pybundle.BundleAdjuster use as BAHelpers does it (ba_helpers.cc:590-741): ids, run(), getters; a perspective problem with one
    constant identity rig camera goes to the streaming solver."""
    from opensfm_amd import bundle
    from opensfm_amd.geometry_types import Camera, Pose

    pr = synthetic.make_ba_scene(6, 60, 4, seed=9, outlier_frac=0.0)
    ba = bundle.BundleAdjuster()
    k1, k2, f = pr["cam_params"][0]
    k1p, k2p, fp = pr["cam_prior"][0]
    ba.add_camera("cam", Camera.create_perspective(f, k1, k2), Camera.create_perspective(fp, k1p, k2p), False)
    ba.add_rig_camera("cam", Pose(), Pose(), True)
    for s in range(6):
        ba.add_rig_instance(f"inst{s}", Pose.from_cam_to_world(pr["shot_pose"][s, :3], pr["shot_pose"][s, 3:]), {f"shot{s}": "cam"}, {f"shot{s}": "cam"}, False)
        ba.add_rig_instance_position_prior(f"inst{s}", pr["shot_gps"][s], np.full(3, 5.0), "")
    for p in range(60):
        ba.add_point(f"p{p}", pr["points"][p], False)
    for s, p, xy, sd in zip(pr["obs_shot"], pr["obs_point"], pr["obs_xy"], pr["obs_sigma"]):
        ba.add_point_projection_observation(f"shot{s}", f"p{p}", xy, sd)
    ba.set_point_projection_loss_function("SoftLOneLoss", 1)
    ba.set_internal_parameters_prior_sd(0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01)
    ba.set_num_threads(1)
    ba.set_max_num_iterations(30)
    ba.set_linear_solver_type("SPARSE_SCHUR")
    ba.run()
    assert ba.solver == "osfm_ba_solve"
    errs = np.array([e for p in range(60) for e in ba.get_point(f"p{p}").reprojection_errors.values()])
    assert len(errs) == len(pr["obs_shot"])
    assert np.sqrt((errs**2).sum(1).mean()) * PX < 2.0
    assert "iterations" in ba.brief_report()
    assert ba.get_rig_instance_pose("inst0").get_origin().shape == (3,)
    assert ba.get_camera("cam").focal != f  # free intrinsics moved
    with pytest.raises(RuntimeError):
        bundle.BundleAdjuster().set_linear_solver_type("NOPE")
    with pytest.raises(RuntimeError):
        bundle.make_options({"loss_function": "NoSuchLoss"})


def test_banded_and_jacobi_preconditioners_agree(oracle_lib, gpu_ctx):
    """Same LM trajectory whichever preconditioner drives the Schur-PCG (both solve to 1e-10)."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(80, 2000, 7, seed=13)
    a = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, preconditioner=0, **NO_TOL)
    b = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, preconditioner=1, **NO_TOL)
    assert a["preconditioner_bandwidth"] == a["shot_bandwidth"] == 6 and b["preconditioner_bandwidth"] == 0
    assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-8)
    assert a["pcg_iterations"] < b["pcg_iterations"]
    assert abs(_rmse_px(a["reproj_err"]) - _rmse_px(b["reproj_err"])) < 1e-4


def test_wide_bandwidth_falls_back_or_truncates(oracle_lib, gpu_ctx):
    """Loop closure: the first and last shots share points, the band is truncated (or dropped) and
    CG must still converge to the oracle's trajectory."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(40, 800, 5, seed=14)
    rng = np.random.default_rng(0)
    extra_pts = np.arange(0, 40)  # points of the first shots, also seen by the last shot
    cam = pr["gt_cam"]
    xy = synthetic.project_perspective(pr["gt_points"][extra_pts], pr["gt_pose"][39], cam) + rng.normal(0, 5e-4, (40, 2))
    pr["obs_shot"] = np.concatenate([pr["obs_shot"], np.full(40, 39, np.int32)])
    pr["obs_point"] = np.concatenate([pr["obs_point"], extra_pts.astype(np.int32)])
    pr["obs_xy"] = np.concatenate([pr["obs_xy"], xy])
    pr["obs_sigma"] = np.concatenate([pr["obs_sigma"], np.full(40, 0.004)])
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=8, **NO_TOL)
    # the caller's order has a wide band; the solver may renumber the shots (RCM) to shrink it
    assert max(g["shot_bandwidth"], g["shot_bandwidth_input"]) > 15
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)


def test_grid_topology_matches_oracle(oracle_lib, gpu_ctx):
    """A block survey (rows x cols cameras numbered line after line, every point seen from three lines): the co-visibility half-width is
    ~2 x cols, whatever the numbering -- the banded preconditioner cannot hold it, CG must still follow the oracle's LM trajectory"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene_grid(12, 30, 6000, 9, seed=4)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=8, **NO_TOL)
    assert g["shot_bandwidth"] > 15  # no renumbering brings a 2-D block under the band the preconditioner holds
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


def test_grid_topology_takes_the_wide_band_solver(oracle_lib, gpu_ctx):
    """The block survey again, through what the solver reports: the shots are renumbered by the sweep along the long side, the exact
    band (half-width above the 15 shots of the streaming band, below the wide solver's limit) is factorised directly, and CG needs one
    or two iterations per LM iteration -- the preconditioner is the reduced matrix itself (a truncated band needed ~1000)."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene_grid(12, 30, 6000, 9, seed=4)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)
    assert g["shots_reordered"] and g["shot_bandwidth_input"] == 62
    assert 15 < g["shot_bandwidth"] <= 36 and g["preconditioner_bandwidth"] == g["shot_bandwidth"]
    assert g["pcg_iterations"] <= 3 * g["iterations"]


def test_two_free_cameras_on_a_grid(oracle_lib, gpu_ctx):
    """Two free cameras shared by the shots of a block survey: the camera border has six columns, which go through the wide band's walk
    in two groups together with the solve's own right-hand side."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene_grid(10, 24, 4000, 9, seed=6)
    S = len(pr["shot_pose"])
    base = pr["cam_params"][0]
    pr["cam_params"] = np.array([base, base * [1.05, 0.95, 1.01]])
    pr["cam_prior"] = np.tile(pr["cam_prior"][0], (2, 1))
    pr["cam_sigma"] = np.tile(pr["cam_sigma"][0], (2, 1))
    pr["cam_fixed"] = np.zeros(2, np.uint8)
    pr["shot_camera"] = (np.arange(S) % 2).astype(np.int32)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=8, **NO_TOL)
    assert g["preconditioner_bandwidth"] > 15
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert np.allclose(g["cam_params"], o["cam_params"], atol=1e-6)
    assert g["pcg_iterations"] <= 3 * g["iterations"]


def test_long_tracks_take_the_strided_matvec_path(oracle_lib, gpu_ctx):
    """Tracks longer than the cooperative mat-vec tile (128 observations) use the strided
    workgroup path; the shot band is far wider than the preconditioner can hold."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(170, 120, 150, seed=15, outlier_frac=0.0)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 6}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=6, **NO_TOL)
    assert max(g["shot_bandwidth"], g["shot_bandwidth_input"]) > 128
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


def test_tracks_longer_than_the_cooperative_tile(oracle_lib, gpu_ctx):
    """points seen from more than 256 shots (the tile of the cooperative kernels): the evaluation, pass A of the mat-vec, the camera border's
    pass A, the right-hand side and the back-substitution take their strided paths; the exact band of such a scene is the whole matrix"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(300, 40, 290, seed=15, outlier_frac=0.0)
    assert np.bincount(pr["obs_point"]).max() > 256
    o = oracle_lib.ba_solve(pr, max_iterations=5, **NO_TOL)
    for pre in (0, 1):
        g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 5}, preconditioner=pre, **NO_TOL)
        assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7), pre
        assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


def test_up_vector_prior_matches_oracle(oracle_lib, gpu_ctx):
    """Absolute up-vector prior (align_method orientation_prior, ba_helpers.cc:609-621,688-692;
    UpVectorError + CauchyLoss(1), bundle_adjuster.cc:955-970): a dense 3x3 prior block on every
    shot rotation under its own robust loss."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(30, 600, 6, seed=1)
    S = len(pr["shot_pose"])
    pr["shot_up"] = np.tile([0.0, -2.0, 0.0], (S, 1))  # normalised by the library, as the reference does
    pr["shot_up_sigma"] = np.full(S, 1e-3)
    pr["shot_up_sigma"][::7] = 0.0  # some shots without the prior
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 10}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=10, **NO_TOL)
    assert g["successful_steps"] == o["successful_steps"]
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    assert np.allclose(g["shot_pose"], o["shot_pose"], atol=1e-6)
    b = bundle.bundle_arrays(pr, {"bundle_max_iterations": 10}, preconditioner=1, **NO_TOL)  # block-Jacobi path
    assert np.allclose(b["cost_history"], o["cost_history"], rtol=1e-7)


def test_several_cameras_one_fixed(oracle_lib, gpu_ctx):
    """Three camera models shared cyclically by the shots (one of them held constant): the camera
    border of the reduced system has 6 free unknowns coupled to every shot."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(45, 900, 6, seed=23)
    S = len(pr["shot_pose"])
    base = pr["cam_params"][0]
    pr["cam_params"] = np.array([base, base * [1.1, 0.9, 1.02], base * [0.9, 1.2, 0.99]])
    pr["cam_prior"] = np.tile(pr["cam_prior"][0], (3, 1))
    pr["cam_sigma"] = np.tile(pr["cam_sigma"][0], (3, 1))
    pr["cam_fixed"] = np.array([0, 0, 1], np.uint8)
    pr["shot_camera"] = (np.arange(S) % 3).astype(np.int32)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 10}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=10, **NO_TOL)
    assert g["successful_steps"] == o["successful_steps"]
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    assert np.allclose(g["cam_params"], o["cam_params"], atol=1e-6)
    assert np.array_equal(g["cam_params"][2], pr["cam_params"][2])


def test_bundle_local_and_shot_poses(oracle_lib, gpu_ctx):
    """BundleLocal / BundleShotPoses (ba_helpers.cc:117-311,408-579): neighbourhood selection, the
    sub-problem (boundary shots, cameras [and points] constant), 10 iterations -- against the oracle
    on the same sub-problem."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(60, 1500, 8, seed=31)
    cfg = {"local_bundle_radius": 3, "local_bundle_min_common_points": 20, "local_bundle_max_shots": 9}
    interior, boundary = bundle.shot_neighborhood(pr, 30, 3, 20, 9)
    assert interior[30] and 1 < interior.sum() <= 9 and boundary.sum() > 0 and not (interior & boundary).any()
    pt_ids, rep = bundle.bundle_local_arrays(pr, 30, cfg)
    sub = rep["sub_problem"]
    assert sub["cam_fixed"].all() and rep["num_interior_images"] == interior.sum() and rep["num_boundary_images"] == boundary.sum()
    o = oracle_lib.ba_solve(sub, max_iterations=10)
    assert rep["iterations"] == o["iterations"]
    assert np.allclose(rep["cost_history"], o["cost_history"], rtol=1e-7)
    moved = np.abs(rep["shot_pose"] - pr["shot_pose"]).max(axis=1) > 0
    assert np.array_equal(moved, interior)  # boundary and other shots untouched
    assert np.allclose(rep["points"][pt_ids], o["points"], atol=1e-7)
    sp = bundle.bundle_shot_poses_arrays(pr, [10, 11], cfg)
    o2 = oracle_lib.ba_solve(sp["sub_problem"], max_iterations=10)
    assert np.allclose(sp["cost_history"], o2["cost_history"], rtol=1e-7)
    assert np.allclose(sp["shot_pose"][[10, 11]], o2["shot_pose"], atol=1e-7)
    other = np.ones(60, bool)
    other[[10, 11]] = False
    assert np.array_equal(sp["shot_pose"][other], pr["shot_pose"][other])


@pytest.mark.parametrize("shots,track", [(48, 10), (30, 6), (14, 5), (3, 3)])
def test_one_workgroup_band_factor_equals_the_cyclic_reduction(oracle_lib, gpu_ctx, monkeypatch, shots, track):
    """Few shots: sband_factor_kernel / sband_solve_kernel (the band factorised and applied by one workgroup) against the cyclic reduction
    (OSFM_BA_NO_SBAND) and the oracle, on a local problem (constant cameras: the band is the reduced matrix) and on the full one (the shared
    camera's exact border: its columns go through the one-workgroup solve four right-hand sides at a time).  Both factorisations are exact."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(shots, 40 * shots, track, seed=100 + shots)
    local = bundle.local_problem(pr, shots // 2, {"local_bundle_radius": 3, "local_bundle_min_common_points": 10, "local_bundle_max_shots": 30})[0]
    for prob in (local, pr):
        res = []
        for no_sband in (False, True):
            if no_sband:
                monkeypatch.setenv("OSFM_BA_NO_SBAND", "1")
            else:
                monkeypatch.delenv("OSFM_BA_NO_SBAND", raising=False)
            res.append(bundle.bundle_arrays(prob, {"bundle_max_iterations": 6}, **NO_TOL))
        a, b = res
        o = oracle_lib.ba_solve(prob, max_iterations=6, **NO_TOL)
        assert a["pcg_iterations"] == b["pcg_iterations"] == a["iterations"]
        assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-12)
        assert np.allclose(a["cost_history"], o["cost_history"], rtol=1e-10)
        # (the full problem's gauge is held by the GPS priors alone: two exact solves differ by rounding times the gauge's conditioning)
        tol = 1e-9 if prob is local else 1e-6
        assert np.abs(a["shot_pose"] - b["shot_pose"]).max() < tol and np.abs(a["points"] - b["points"]).max() < tol


def test_fisheye_camera_model(oracle_lib, gpu_ctx):
    """FisheyeCamera = <FisheyeProjection, Disto24, UniformScale> (camera_instances.h:187): same
    [k1, k2, focal] parameters, equidistant projection; mixed with a perspective camera."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(30, 600, 6, seed=41, model="fisheye")
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 10}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=10, **NO_TOL)
    assert g["successful_steps"] == o["successful_steps"]
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    assert np.allclose(g["cam_params"], o["cam_params"], atol=1e-6)
    assert _rmse_px(g["reproj_err"], ~pr["is_outlier"]) < 2.5
    # solving the fisheye data with the perspective model must NOT fit (the model flag is honoured)
    pr_wrong = dict(pr)
    pr_wrong.pop("cam_model")
    w = bundle.bundle_arrays(pr_wrong, {"bundle_max_iterations": 10}, **NO_TOL)
    assert w["final_cost"] > 1.05 * g["final_cost"]  # narrow field of view: Disto24 absorbs most of the difference
    z = bundle.bundle_arrays(pr, {"bundle_max_iterations": 0})
    assert np.allclose(z["reproj_err"], oracle_lib.ba_solve(pr, max_iterations=0)["reproj_err"], rtol=0, atol=1e-13)


def test_unordered_shots_are_renumbered(oracle_lib, gpu_ctx):
    """An unordered collection (the street sequence with its shots shuffled): the co-visibility band
    in the caller's order is as wide as the problem, the solver renumbers the shots by reverse
    Cuthill-McKee, runs the exact banded path, and returns everything in the caller's order."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(90, 2500, 7, seed=51)
    rng = np.random.default_rng(1)
    perm = rng.permutation(90)  # new index -> old index
    inv = np.argsort(perm)
    sh = dict(pr)
    for k in ("shot_pose", "shot_camera", "shot_gps", "shot_gps_sigma", "gt_pose"):
        sh[k] = pr[k][perm]
    sh["obs_shot"] = inv[pr["obs_shot"]].astype(np.int32)
    g = bundle.bundle_arrays(sh, {"bundle_max_iterations": 8}, **NO_TOL)
    o = oracle_lib.ba_solve(sh, max_iterations=8, **NO_TOL)
    ref = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)  # the same problem in sequence order
    assert g["shots_reordered"] and g["shot_bandwidth_input"] > 40 and g["shot_bandwidth"] <= 10
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"]
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert np.allclose(g["shot_pose"], o["shot_pose"], atol=1e-6)
    assert np.allclose(g["shot_pose"], ref["shot_pose"][perm], atol=1e-6)
    assert g["pcg_iterations"] <= 2 * ref["pcg_iterations"]
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


GENERIC_PARAMS = {
    "brown": [-0.12, 0.03, -0.004, 0.001, -0.0007, 0.72, 1.003, 0.004, -0.006],
    "fisheye_opencv": [-0.03, 0.004, -0.0006, 0.0001, 0.45, 0.999, 0.002, -0.001],
    "fisheye62": [-0.03, 0.004, -0.0006, 0.0001, 0.00002, -0.000004, 0.0004, -0.0003, 0.45, 0.999, 0.002, -0.001],
    "fisheye624": [-0.03, 0.004, -0.0006, 0.0001, 0.00002, -0.000004, 0.0004, -0.0003, 0.0002, -0.0001, 0.0003, 0.00005, 0.45, 0.999, 0.002,
                   -0.001],
    "dual": [0.4, -0.05, 0.004, 0.6],
    "radial": [-0.1, 0.01, 0.7, 0.998, -0.003, 0.002],
    "simple_radial": [-0.08, 0.7, 1.01, 0.001, 0.002],
}


@pytest.mark.parametrize("model", sorted(GENERIC_PARAMS))
def test_constant_cameras_of_the_other_2d_models(oracle_lib, gpu_ctx, model):
    """BROWN / FISHEYE_OPENCV / FISHEYE62 / FISHEYE624 / DUAL / RADIAL / SIMPLE_RADIAL as CONSTANT cameras (how
    BundleLocal and BundleShotPoses use every camera): poses and points are optimised through their projection."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(24, 400, 6, seed=61, model=model, generic_params=GENERIC_PARAMS[model])
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=8, **NO_TOL)
    assert g["successful_steps"] == o["successful_steps"]
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    assert np.allclose(g["shot_pose"], o["shot_pose"], atol=1e-6)
    assert _rmse_px(g["reproj_err"], ~pr["is_outlier"]) < 2.5
    z = bundle.bundle_arrays(pr, {"bundle_max_iterations": 0})
    assert np.allclose(z["reproj_err"], oracle_lib.ba_solve(pr, max_iterations=0)["reproj_err"], rtol=0, atol=1e-13)


def test_other_models_must_be_constant(gpu_ctx):
    from opensfm_amd import bundle
    from opensfm_amd._lib import OsfmError

    pr = synthetic.make_ba_scene(10, 100, 4, seed=62, model="brown", generic_params=GENERIC_PARAMS["brown"])
    pr["cam_fixed"] = np.zeros(1, np.uint8)
    with pytest.raises(OsfmError):
        bundle.bundle_arrays(pr, {"bundle_max_iterations": 2})


@pytest.mark.parametrize("shots,points,track,ragged,seed", [(40, 800, 6, False, 3), (200, 6000, 10, False, 4), (120, 4000, 5, True, 5),
                                                             (64, 900, 2, False, 6), (300, 5000, 6, True, 7), (90, 2500, 12, False, 8),
                                                             (50, 1200, 16, False, 9)])
def test_band_by_windows_equals_band_per_shot(oracle_lib, gpu_ctx, monkeypatch, shots, points, track, ragged, seed):
    """band_mfma_kernel (one f64 matrix-core GEMM per first shot of the tracks, no atomics) against band_assemble_kernel (per shot, LDS
    atomics): OSFM_BA_CHECK_BAND makes every LM iteration assemble the band both ways and fail above 1e-10 of the largest entry.
    Ragged tracks (no two points share a shot set), tracks of two, every tile count of the kernel (tracks of 2 .. 16 shots)."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(shots, points, track, seed=seed, ragged=ragged)
    monkeypatch.setenv("OSFM_BA_CHECK_BAND", "1")
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, **NO_TOL)
    monkeypatch.delenv("OSFM_BA_CHECK_BAND")
    assert g["iterations"] == 4
    assert g["preconditioner_bandwidth"] <= 15  # wider bands go to the direct block LDL^T, not to this kernel
    monkeypatch.setenv("OSFM_BA_BAND_PER_SHOT", "1")
    h = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, **NO_TOL)
    assert np.allclose(g["cost_history"], h["cost_history"], rtol=1e-12)
    assert g["pcg_iterations"] <= h["pcg_iterations"] + 2
    # the matrix-core assembly has no atomics: two runs agree bit for bit
    monkeypatch.delenv("OSFM_BA_BAND_PER_SHOT")
    g2 = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, **NO_TOL)
    assert np.array_equal(g["cost_history"], g2["cost_history"]) and np.array_equal(g["shot_pose"], g2["shot_pose"])


@pytest.mark.parametrize("shots,points,track,seed", [(150, 5000, 12, 8), (400, 12000, 20, 9)])
def test_ragged_tracks_match_oracle(oracle_lib, gpu_ctx, shots, points, track, seed):
    """`make_ba_scene(ragged=True)`: track lengths 2 + Poisson with 15 % of the sightings missing -- no two tracks share a shot set and the
    half-width is the longest track, beyond the cluster-tridiagonal band: the exact band goes through the wide solver (cyclic reduction over
    dense clusters) and CG confirms in one or two iterations per LM iteration."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(shots, points, track, seed=seed, ragged=True)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 8}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=8, **NO_TOL)
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"] > 10
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-9)
    assert np.abs(g["shot_pose"] - o["shot_pose"]).max() < 1e-7
    assert g["pcg_iterations"] <= 3 * g["iterations"]


def test_dense_cyclic_reduction_equals_block_ldlt(oracle_lib, gpu_ctx, monkeypatch):
    """The two factorisations of the wide band (round 4: cyclic reduction over dense clusters; round 3: block LDL^T chain, kept under
    OSFM_BA_WIDE_LDLT) are the same preconditioner: the same CG iteration counts (give or take a step at the tolerance), the same trajectory."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene_grid(12, 30, 6000, 9, seed=4)
    a = bundle.bundle_arrays(pr, {"bundle_max_iterations": 6}, **NO_TOL)
    monkeypatch.setenv("OSFM_BA_WIDE_LDLT", "1")
    b = bundle.bundle_arrays(pr, {"bundle_max_iterations": 6}, **NO_TOL)
    monkeypatch.delenv("OSFM_BA_WIDE_LDLT")
    assert a["preconditioner_bandwidth"] == b["preconditioner_bandwidth"] > 15
    assert abs(a["pcg_iterations"] - b["pcg_iterations"]) <= 2  # a residual at the tolerance can take one more step on either side
    assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-10)


def test_half_width_beyond_one_assembly_pass(gpu_ctx):
    """Tracks 560 shots long: the exact band (half-width 559) is assembled in two passes of band columns (one workgroup's LDS holds 520)
    and factorised by the dense-cluster cyclic reduction; CG confirms in one or two iterations per LM iteration, and the trajectory is the
    one block-Jacobi PCG reaches with hundreds of iterations."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(700, 260, 560, seed=21)
    a = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, **NO_TOL)
    assert a["shot_bandwidth"] == 559 and a["preconditioner_bandwidth"] == 559
    assert a["pcg_iterations"] <= 3 * a["iterations"]
    b = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, preconditioner=1, pcg_max_iterations=5000, **NO_TOL)
    assert b["preconditioner_bandwidth"] == 0 and b["pcg_iterations"] > 10 * a["pcg_iterations"]
    assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-6)


def test_lookahead_variant_of_the_wide_band_inversion(gpu_ctx, monkeypatch):
    """OSFM_BA_LOOKAHEAD=1: the pivot chain of the blocked Gauss-Jordan on a second stream (the next panel's pivot block formed and inverted
    beside the trailing product of the current one) -- measured slower than the single-stream order and off by default, kept correct here:
    same CG counts give or take a step, same trajectory."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene_grid(12, 30, 6000, 9, seed=4)
    a = bundle.bundle_arrays(pr, {"bundle_max_iterations": 5}, **NO_TOL)
    monkeypatch.setenv("OSFM_BA_LOOKAHEAD", "1")
    b = bundle.bundle_arrays(pr, {"bundle_max_iterations": 5}, **NO_TOL)
    monkeypatch.delenv("OSFM_BA_LOOKAHEAD")
    assert a["preconditioner_bandwidth"] == b["preconditioner_bandwidth"] > 15
    assert abs(a["pcg_iterations"] - b["pcg_iterations"]) <= 2
    assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-10)


def test_results_do_not_depend_on_what_the_cached_slabs_held(gpu_ctx):
    """the solve's device arrays are sub-allocated from slabs the context keeps between calls (round 5): whatever a solve reads it must
    have written in the same call.  Two different problems alternate on one context -- each run is bit for bit the run on a fresh
    context (whose slabs come straight from the driver)."""
    from opensfm_amd import _lib, bundle

    problems = [synthetic.make_ba_scene(60, 2500, 7, seed=1), synthetic.make_ba_scene(45, 4000, 5, seed=2, ragged=True),
                synthetic.make_ba_scene(60, 2500, 7, seed=3)]
    fresh = []
    for pr in problems:
        c = _lib.Context(0)
        fresh.append(bundle.bundle_arrays(pr, {"bundle_max_iterations": 5}, ctx=c, **NO_TOL))
        c.close()
    shared = _lib.Context(0)
    try:
        for rep in range(2):
            for pr, f in zip(problems, fresh):
                g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 5}, ctx=shared, **NO_TOL)
                assert np.array_equal(g["cost_history"], f["cost_history"]) and g["pcg_iterations"] == f["pcg_iterations"]
                for k in ("shot_pose", "points", "cam_params", "reproj_err"):
                    assert np.array_equal(g[k], f[k]), (rep, k)
    finally:
        shared.close()
