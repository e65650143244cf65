"""GPU parity of the general bundle adjustment (osfm_bundle_solve = the streaming Schur solver in its generic mode,
opensfm_amd/csrc/ba_generic.inc: analytic Jacobians, points eliminated on the fly, exact band + exactly eliminated border, PCG) against
the CPU oracle (oracle/bundle_general_oracle.cc: jets; full dense normal equations, or -- many points -- the same equations with the
points eliminated first): same cost after the same number of LM iterations, reprojection RMSE within 1e-4 px (north_star), per camera
family and per residual family, and at BASELINE configs[2] size.  1e-4 px at the synthetic 2000-px image = 5e-8 in normalized coordinates."""
import numpy as np
import pytest

from opensfm_amd import synthetic

pytestmark = pytest.mark.gpu

NO_TOL = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)


def _rmse_px(err):
    return float(np.sqrt((np.asarray(err) ** 2).sum(1).mean()) * 2000.0)


def _compare(oracle_lib, gpu_ctx, pr, iters=8):
    from opensfm_amd import bundle

    g = bundle.bundle_general_arrays(pr, {"bundle_max_iterations": iters}, ctx=gpu_ctx, **NO_TOL)
    o = oracle_lib.bundle_general(pr, max_iterations=iters, **NO_TOL)
    assert g["iterations"] == o["iterations"] == iters
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-7), (g["cost_history"], o["cost_history"])
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    for k in ("cam_params", "rig_camera_pose", "rig_instance_pose", "points"):
        assert np.allclose(g[k], o[k], atol=1e-6), k
    if "bias" in o:
        assert np.allclose(g["bias"], o["bias"], atol=1e-6)
    assert g["final_cost"] < g["initial_cost"]
    return g, o


@pytest.mark.parametrize("model", ["perspective", "fisheye", "brown", "fisheye_opencv", "fisheye62", "fisheye624", "dual", "radial", "simple_radial"])
def test_every_camera_family_with_free_intrinsics(oracle_lib, gpu_ctx, model):
    """all native parameters of the model are optimised (bundle_adjuster.cc:568-593 priors, logarithmic focal / aspect ratio)"""
    pr = synthetic.make_bundle_scene(models=(model,), n_instances=8, n_points=120, rig=False, gps=False, n_gcp=0, up_vectors=False, seed=7)
    g, _ = _compare(oracle_lib, gpu_ctx, pr)
    nk = len(synthetic.BUNDLE_TEST_CAMERAS[model])
    assert np.abs(g["cam_params"][0, :nk] - pr["cam_params"][0, :nk]).max() > 0  # the intrinsics moved


def test_spherical_camera_3d_residual(oracle_lib, gpu_ctx):
    """ReprojectionError3D (projection_errors.h:208-246): unit bearing minus the observed bearing, three residuals per observation"""
    pr = synthetic.make_bundle_scene(models=("spherical",), n_instances=8, n_points=120, rig=False, gps=False, n_gcp=0, up_vectors=False, seed=8)
    g, _ = _compare(oracle_lib, gpu_ctx, pr)
    assert np.abs(g["reproj_err"][:, 2]).max() > 0


def test_rig_bias_control_points_up_vectors(oracle_lib, gpu_ctx):
    """everything BAHelpers::Bundle wires at once: two cameras of different models on a two-camera rig whose second rig camera is free
    (with its pose prior), GPS priors through free per-camera biases, control points with and without altitude, up vectors"""
    pr = synthetic.make_bundle_scene(models=("perspective", "brown"), n_instances=10, n_points=140, seed=5)
    g, _ = _compare(oracle_lib, gpu_ctx, pr, iters=10)
    assert np.abs(g["rig_camera_pose"][1] - pr["rig_camera_pose"][1]).max() > 0
    assert np.abs(g["bias"] - pr["bias"]).max() > 0


def depth_priors_for(pr, rng, frac=0.4, noise=0.01):
    """depth priors (map::Depth) for a fraction of the observations: the true depth of the point in its camera, radial or along z,
    perturbed -- RelativeDepthError (relative_depth_error.h) next to the reprojection of the same observation"""
    from scipy.spatial.transform import Rotation

    M = len(pr["obs_shot"])
    depth, sd, radial = np.zeros(M), np.zeros(M), np.ones(M, np.uint8)
    for o in range(M):
        if rng.random() > frac:
            continue
        s = pr["obs_shot"][o]
        Xc = np.asarray(pr["points"][pr["obs_point"][o]], float)
        for pose in (pr["rig_instance_pose"][pr["shot_rig_instance"][s]], pr["rig_camera_pose"][pr["shot_rig_camera"][s]]):
            Xc = Rotation.from_rotvec(-np.asarray(pose[:3])).apply(Xc - pose[3:])  # WorldToLocal of a CAM_TO_WORLD pose
        radial[o] = rng.random() < 0.5
        depth[o] = (np.linalg.norm(Xc) if radial[o] else Xc[2]) * (1 + noise * rng.normal())
        sd[o] = 0.05 * (1 + rng.random())
    return {"obs_depth": depth, "obs_depth_sigma": sd, "obs_depth_radial": radial}


@pytest.mark.parametrize("models,rig", [(("perspective",), False), (("brown", "fisheye"), True), (("spherical",), False)])
def test_depth_priors(oracle_lib, gpu_ctx, models, rig):
    """RelativeDepthError (bundle_adjuster.cc:497-528,812; ba_helpers.cc:695-696 passes obs.depth_prior): radial and z depths, with and
    without a useful rig camera, under the same robust loss as the reprojections"""
    pr = synthetic.make_bundle_scene(models=models, n_instances=8, n_points=100, rig=rig, gps=False, n_gcp=0, up_vectors=False, seed=13)
    pr.update(depth_priors_for(pr, np.random.default_rng(3)))
    g, o = _compare(oracle_lib, gpu_ctx, pr)
    plain = {k: v for k, v in pr.items() if not k.startswith("obs_depth")}
    o0 = oracle_lib.bundle_general(plain, max_iterations=8, **NO_TOL)
    assert o["initial_cost"] > o0["initial_cost"] and not np.allclose(o["points"], o0["points"], atol=1e-9)  # the priors count
    assert g["reproj_err"].shape == (len(pr["obs_shot"]), 3)
    bad = dict(pr, obs_depth=np.where(np.arange(len(pr["obs_shot"])) == int(np.flatnonzero(pr["obs_depth_sigma"] > 0)[0]), np.nan, pr["obs_depth"]))
    from opensfm_amd import bundle

    with pytest.raises(Exception):
        bundle.bundle_general_arrays(bad, {"bundle_max_iterations": 2}, ctx=gpu_ctx, **NO_TOL)


def test_constant_blocks(oracle_lib, gpu_ctx):
    """constant cameras, constant rig cameras (non-identity: still part of the projection), constant biases, some constant instances and
    points: none of them moves, the rest agrees with the oracle"""
    pr = synthetic.make_bundle_scene(models=("fisheye", "radial"), n_instances=9, n_points=120, seed=9, free_cameras=False, free_rig_camera=False,
                                     free_bias=False)
    pr["rig_instance_fixed"] = np.zeros(9, np.uint8)
    pr["rig_instance_fixed"][[0, 4]] = 1
    pr["point_fixed"] = (np.arange(len(pr["points"])) % 7 == 0).astype(np.uint8)
    g, _ = _compare(oracle_lib, gpu_ctx, pr)
    assert np.array_equal(g["cam_params"], pr["cam_params"]) and np.array_equal(g["rig_camera_pose"], pr["rig_camera_pose"])
    assert np.array_equal(g["rig_instance_pose"][[0, 4]], pr["rig_instance_pose"][[0, 4]])
    assert np.array_equal(g["points"][pr["point_fixed"] == 1], pr["points"][pr["point_fixed"] == 1])


def test_general_solver_equals_the_streaming_solver(oracle_lib, gpu_ctx):
    """on the domain both cover (perspective [k1 k2 focal], identity rig, GPS priors) osfm_bundle_solve and osfm_ba_solve walk the same
    LM trajectory: the generic rows of the streaming solver (ba_generic.inc) vs its specialised [k1 k2 focal] kernels, both implicit Schur-PCG at 1e-10"""
    import test_oracle_bundle_general as og

    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(40, 900, 6, seed=12)
    a = bundle.bundle_arrays(pr, {"bundle_max_iterations": 6}, ctx=gpu_ctx, **NO_TOL)
    b = bundle.bundle_general_arrays(og._as_general(pr), {"bundle_max_iterations": 6}, ctx=gpu_ctx, **NO_TOL)
    assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-7)
    assert np.allclose(a["shot_pose"], b["rig_instance_pose"], atol=1e-7)
    assert abs(_rmse_px(a["reproj_err"]) - _rmse_px(b["reproj_err"][:, :2])) < 1e-4


def test_generic_mode_at_configs2_size_equals_the_schur_oracle(oracle_lib, gpu_ctx):
    """BASELINE configs[2] (500 cams / 50 k points / 300 k observations, 20 LM iterations) through osfm_bundle_solve: the generic rows
    walk the trajectory of oracle/ba_oracle.c (analytic derivatives, Schur + skyline Cholesky) -- RMSE within 1e-4 px after the same
    iteration count"""
    import test_oracle_bundle_general as og

    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(500, 50000, 6, seed=42)
    g = bundle.bundle_general_arrays(og._as_general(pr), {"bundle_max_iterations": 20}, ctx=gpu_ctx, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=20, **NO_TOL)
    assert g["iterations"] == o["iterations"] == 20
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-9)
    assert abs(_rmse_px(g["reproj_err"][:, :2]) - _rmse_px(o["reproj_err"])) < 1e-4
    assert np.allclose(g["rig_instance_pose"], o["shot_pose"], atol=1e-7) and np.allclose(g["cam_params"][:, :3], o["cam_params"], atol=1e-9)
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"] and g["pcg_iterations"] <= 3 * 20  # exact band + exact border: CG only confirms


@pytest.mark.parametrize("ragged", [False, True])
def test_brown_camera_free_bias_control_points_at_scale(oracle_lib, gpu_ctx, ragged):
    """what BAHelpers::Bundle builds on a calibrated data set -- a BROWN camera with nine free intrinsics, position priors through a free
    similarity bias, 20 control points -- beyond what a dense reduced system could carry comfortably (1 500 reduced unknowns, 8 000 points):
    the oracle eliminates the points and factorises the reduced system densely.  ragged: the wide band (cyclic reduction over dense
    clusters) under the same border"""
    from opensfm_amd import bundle

    pr = synthetic.make_general_ba_scene(250, 8000, 8, model="brown", n_gcp=20, gps_bias=True, seed=5, ragged=ragged)
    g = bundle.bundle_general_arrays(pr, {"bundle_max_iterations": 10}, ctx=gpu_ctx, **NO_TOL)
    o = oracle_lib.bundle_general(pr, max_iterations=10, **NO_TOL)
    assert g["iterations"] == o["iterations"] == 10
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-8), (g["cost_history"], o["cost_history"])
    assert abs(_rmse_px(g["reproj_err"][:, :2]) - _rmse_px(o["reproj_err"][:, :2])) < 1e-4
    for k in ("cam_params", "rig_instance_pose", "points", "bias"):
        assert np.allclose(g[k], o[k], atol=1e-6), k
    assert np.abs(g["bias"][0] - pr["bias"][0]).max() > 1e-3 and np.abs(g["cam_params"][0, :9] - pr["cam_params"][0, :9]).max() > 0
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"] and (g["shot_bandwidth"] > 10) == ragged


def test_brown_camera_at_configs2_size_twenty_iterations(oracle_lib, gpu_ctx):
    """BASELINE configs[2] (500 cams / 50 k points / 300 k observations, 20 LM iterations) with what BAHelpers::Bundle builds on a calibrated
    data set: a BROWN camera with nine free intrinsics and their priors, GPS priors through a free similarity bias, 20 control points.
    osfm_bundle_solve against oracle/bundle_general_oracle.cc (jets, arrow-form normal equations, points eliminated, dense Cholesky of
    the 3 016 reduced unknowns): the same 20 steps, reprojection RMSE within 1e-4 px, the same parameters"""
    from opensfm_amd import bundle

    pr = synthetic.make_general_ba_scene(500, 50000, 6, model="brown", n_gcp=20, gps_bias=True, seed=42)
    g = bundle.bundle_general_arrays(pr, {"bundle_max_iterations": 20}, ctx=gpu_ctx, **NO_TOL)
    o = oracle_lib.bundle_general(pr, max_iterations=20, **NO_TOL)
    assert g["iterations"] == o["iterations"] == 20 and len(pr["obs_shot"]) == 300000
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-8), (g["cost_history"], o["cost_history"])
    assert abs(_rmse_px(g["reproj_err"][:, :2]) - _rmse_px(o["reproj_err"][:, :2])) < 1e-4
    assert np.abs(g["reproj_err"][:, :2] - o["reproj_err"][:, :2]).max() * 2000.0 < 1e-4  # every observation, in pixels of a 2000-px image
    for k in ("cam_params", "rig_instance_pose", "points", "bias"):
        assert np.allclose(g[k], o[k], atol=1e-6), k
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"] and g["pcg_iterations"] <= 2 * 20


@pytest.mark.parametrize("model", ["brown", "fisheye_opencv", "perspective"])
def test_compact_rows_equal_the_rows_with_border_slots(gpu_ctx, monkeypatch, model):
    """One 2-D projection type, reprojection rows only, no free rig camera: the rows keep (Xc, wt) instead of their 2 KW border slots
    (gen_eval_kernel's COMPACT layout, round 6) and pass A of the mat-vec, the back-substitution and the border's point pass rebuild the slots with
    project_full on the same Xc -- the expressions of the evaluation kernel, operation by operation (bit for bit on the host emulation and on fresh
    contexts, `tools/r06_compact_diag.py`; here to 1e-10: the prior blocks are accumulated with atomics, whose order is not fixed between two runs).
    Long tracks (the strided path of the three kernels) ride along in the second scene."""
    from opensfm_amd import bundle

    scenes = [synthetic.make_general_ba_scene(60, 1500, 6, model=model, n_gcp=5, gps_bias=True, seed=11),
              synthetic.make_general_ba_scene(300, 40, 290, model=model, n_gcp=0, gps_bias=True, seed=12)]
    for pr in scenes:
        res = []
        for full in (False, True):
            if full:
                monkeypatch.setenv("OSFM_BA_GEN_FULL_ROWS", "1")
            else:
                monkeypatch.delenv("OSFM_BA_GEN_FULL_ROWS", raising=False)
            res.append(bundle.bundle_general_arrays(pr, {"bundle_max_iterations": 5}, ctx=gpu_ctx, **NO_TOL))
        a, b = res
        assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-10, atol=0)
        for k in ("cam_params", "rig_instance_pose", "points", "bias"):
            assert np.allclose(a[k], b[k], rtol=0, atol=1e-7), k


def test_nine_free_cameras_keep_the_exact_border(oracle_lib, gpu_ctx):
    """nine free BROWN cameras = 81 border unknowns: inside the kGenMaxNB = 100 the exact border elimination carries since round 6 (64 until round
    5, when this scene fell back to Jacobi-preconditioned border rows and more than four CG iterations per LM iteration): the preconditioner
    is the reduced matrix, CG confirms in one or two iterations -- same trajectory"""
    pr = synthetic.make_bundle_scene(models=("brown",) * 9, n_instances=27, n_points=400, rig=False, gps=False, n_gcp=0, up_vectors=False, seed=3)
    g, _ = _compare(oracle_lib, gpu_ctx, pr, iters=6)
    assert g["pcg_iterations"] <= 6 * 2


def test_priors_of_a_border_wider_than_the_prior_kernels_lds_copy(oracle_lib, gpu_ctx):
    """twelve free BROWN cameras = 108 border unknowns: gen_prior_kernel's per-workgroup copy of the border x border prior block (NB^2 + NB
    doubles) would not fit the 64 KB a launch gets by default -- round 5 launched it anyway, the launch failed unnoticed and the camera
    priors dropped out of the normal equations.  Beyond kGenPriorLdsMaxNB the kernel adds to the global arrays directly; the trajectory
    (camera priors included: they are what holds nine intrinsics per camera on 3 shots each) is the oracle's"""
    pr = synthetic.make_bundle_scene(models=("brown",) * 12, n_instances=36, n_points=500, rig=False, gps=False, n_gcp=0, up_vectors=False, seed=5)
    g, o = _compare(oracle_lib, gpu_ctx, pr, iters=5)
    assert np.abs(g["cam_params"][:, :9] - pr["cam_params"][:, :9]).max() > 0
    # ... and 108 unknowns are beyond the exact border elimination (kGenMaxNB = 100): the band still preconditions the instance block, the border
    # rows are Jacobi-preconditioned, and CG carries the coupling
    assert g["pcg_iterations"] > 5 * 3
