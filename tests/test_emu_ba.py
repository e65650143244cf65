"""The product's bundle-adjustment kernels and LM driver (opensfm_amd/csrc/ba.hip with ba_generic.inc / ba_generic_host.inc) executed on the HOST by the HIP
emulation of tests/native/hipemu -- every kernel, launch, LDS exchange, wavefront shuffle and fp64 MFMA of the real sources, workgroup by
workgroup -- against the CPU oracle.  This is the `-m "not gpu"` twin of tests/test_gpu_ba.py / test_gpu_bundle_general.py at sizes the
emulation finishes in seconds: it catches an indexing slip or an uninitialised read (emulated device memory and dynamic LDS are
poisoned with NaNs) before a GPU lease is spent on it.  The emulated library is test infrastructure; the product never loads it."""
import os

import numpy as np
import pytest

from emu_util import emulated
from opensfm_amd import synthetic

NO_TOL = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)


def _rmse_px(err):
    return float(np.sqrt((np.asarray(err)[:, :2] ** 2).sum(1).mean()) * 2000.0)


def test_streaming_solver_narrow_band_matches_oracle(oracle_lib):
    """sequence, half-width 4: matrix-core band assembly, cyclic reduction in LDS, exact camera border, PCG"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(20, 300, 5, seed=11)
    with emulated() as lib:
        g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, **NO_TOL)
        assert lib.hipemu_launch_count() > 100
    o = oracle_lib.ba_solve(pr, max_iterations=4, **NO_TOL)
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"] and g["shot_bandwidth"] >= 2
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-10)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


def _compare_general(oracle_lib, pr, iters=5, rtol=1e-9):
    from opensfm_amd import bundle

    with emulated():
        g = bundle.bundle_general_arrays(pr, {"bundle_max_iterations": iters}, **NO_TOL)
    o = oracle_lib.bundle_general(pr, max_iterations=iters, **NO_TOL)
    assert g["iterations"] == o["iterations"] == iters
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=rtol), (g["cost_history"], o["cost_history"])
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
    for k in ("cam_params", "rig_camera_pose", "rig_instance_pose", "points"):
        assert np.allclose(g[k], o[k], atol=1e-8), k
    if "bias" in o:
        assert np.allclose(g["bias"], o["bias"], atol=1e-8)
    return g, o


@pytest.mark.parametrize("model", ["brown", "fisheye_opencv", "spherical"])
def test_generic_streaming_solver_camera_families(oracle_lib, model):
    """osfm_bundle_solve = the streaming solver in its generic mode (ba_generic.inc): free native intrinsics in the border (9 columns for
    Brown), the 3-row bearing residual of a spherical camera"""
    pr = synthetic.make_bundle_scene(models=(model,), n_instances=8, n_points=100, rig=False, gps=False, n_gcp=0, up_vectors=False, seed=7)
    g, _ = _compare_general(oracle_lib, pr)
    if model != "spherical":
        assert np.abs(g["cam_params"][0, :8] - pr["cam_params"][0, :8]).max() > 0


def test_generic_streaming_solver_rig_bias_control_points_up_vectors(oracle_lib):
    """everything BAHelpers::Bundle wires at once: two camera models on a two-camera rig with a free rig camera, position priors through
    free biases, control points with and without altitude, up vectors -- border of 6 + 3 + 9 + 14 unknowns, two views per instance"""
    pr = synthetic.make_bundle_scene(models=("perspective", "brown"), n_instances=8, n_points=90, seed=5)
    g, _ = _compare_general(oracle_lib, pr, iters=4)
    assert np.abs(g["rig_camera_pose"][1] - pr["rig_camera_pose"][1]).max() > 0 and np.abs(g["bias"] - pr["bias"]).max() > 0


def test_generic_priors_with_a_border_beyond_the_lds_copy(oracle_lib):
    """twelve free BROWN cameras = 108 border unknowns: the prior kernel's workgroup copy of the border block is only used up to
    kGenPriorLdsMaxNB unknowns; beyond, the priors go to the global arrays directly -- the trajectory with the camera priors is the oracle's"""
    pr = synthetic.make_bundle_scene(models=("brown",) * 12, n_instances=24, n_points=160, rig=False, gps=False, n_gcp=0, up_vectors=False, seed=5)
    _compare_general(oracle_lib, pr, iters=3, rtol=1e-8)


def test_straight_line_iteration_equals_the_careful_path(monkeypatch):
    """round 6: an LM iteration whose preconditioner is expected to be exact is queued in one piece (start of PCG, its one iteration, the
    back-substitution, the candidate and its linearisation) and verified by a single round trip.  Same kernels on the same inputs as the
    careful path (OSFM_BA_NO_FAST): the same bits -- also when the verification fails (forced here at LM iteration 2: the blocks go back, the old
    point is linearised again, the iteration is redone on the careful path), for the [k1 k2 focal] kernels, a local problem (constant
    cameras) and the generic rows"""
    from opensfm_amd import bundle

    def runs(solve, pr, iters):
        out = []
        for env in ({"OSFM_BA_NO_FAST": "1"}, {}, {"OSFM_BA_FAST_FAIL_AT": "2"}):
            with monkeypatch.context() as mp:
                for k, v in env.items():
                    mp.setenv(k, v)
                with emulated():
                    out.append(solve(pr, {"bundle_max_iterations": iters}, **NO_TOL))
        return out

    pr = synthetic.make_ba_scene(20, 300, 5, seed=11)
    careful, fast, failed = runs(bundle.bundle_arrays, pr, 4)
    assert careful["pcg_iterations"] >= 4
    for g in (fast, failed):
        assert np.array_equal(g["cost_history"], careful["cost_history"]) and np.array_equal(g["shot_pose"], careful["shot_pose"])
        assert np.array_equal(g["points"], careful["points"]) and np.array_equal(g["cam_params"], careful["cam_params"])
    sub = bundle.local_problem(synthetic.make_ba_scene(60, 900, 6, seed=7), 30)[0]
    careful, fast, failed = runs(bundle.bundle_arrays, sub, 3)
    for g in (fast, failed):
        assert np.array_equal(g["cost_history"], careful["cost_history"]) and np.array_equal(g["shot_pose"], careful["shot_pose"])
    prg = synthetic.make_bundle_scene(models=("brown",), n_instances=8, n_points=100, rig=False, gps=False, n_gcp=0, up_vectors=False, seed=7)
    careful, fast, failed = runs(bundle.bundle_general_arrays, prg, 3)
    for g in (fast, failed):
        assert np.array_equal(g["cost_history"], careful["cost_history"]) and np.array_equal(g["rig_instance_pose"], careful["rig_instance_pose"])


@pytest.mark.skipif(not os.environ.get("OSFM_SLOW_TESTS"), reason="seven minutes on the emulation (hundreds of block-Jacobi CG iterations, a barrier tree per mat-vec); "
                    "tests/test_gpu_ba.py::test_tracks_longer_than_the_cooperative_tile runs the same on the device")
def test_tracks_longer_than_the_cooperative_tile(oracle_lib):
    """a point seen from more than 256 shots: the evaluation (its point block is added from the stored rows by one thread), the mat-vec's pass A,
    the right-hand side and the back-substitution take their strided whole-workgroup paths; block-Jacobi preconditioner (the band of such a
    scene is the whole matrix)"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(300, 12, 290, seed=15, outlier_frac=0.0)
    assert np.bincount(pr["obs_point"]).max() > 256
    with emulated():
        g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 3}, preconditioner=1, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=3, **NO_TOL)
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-8), (g["cost_history"], o["cost_history"])
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4


def test_generic_mode_equals_the_specialised_kernels(oracle_lib):
    """on the domain both cover ([k1 k2 focal] perspective cameras, identity rig) the generic rows and the specialised ones walk the same
    trajectory"""
    import test_oracle_bundle_general as og

    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(16, 200, 5, seed=12)
    with emulated():
        a = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, **NO_TOL)
        b = bundle.bundle_general_arrays(og._as_general(pr), {"bundle_max_iterations": 4}, **NO_TOL)
    assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-10)
    assert np.allclose(a["shot_pose"], b["rig_instance_pose"], atol=1e-9)


def test_generic_mode_constant_blocks_count_in_the_cost(oracle_lib):
    """constant cameras / rig cameras / biases / some instances and points: none moves, and their priors still count in the cost (Ceres
    folds residual blocks over constant parameter blocks into its fixed cost) -- the first GPU run of the generic mode had dropped them"""
    pr = synthetic.make_bundle_scene(models=("fisheye", "radial"), n_instances=9, n_points=100, seed=9, free_cameras=False, free_rig_camera=False,
                                     free_bias=False)
    pr["rig_instance_fixed"] = np.zeros(9, np.uint8)
    pr["rig_instance_fixed"][[0, 4]] = 1
    pr["point_fixed"] = (np.arange(len(pr["points"])) % 7 == 0).astype(np.uint8)
    g, _ = _compare_general(oracle_lib, pr, iters=3)
    assert np.array_equal(g["cam_params"], pr["cam_params"]) and np.array_equal(g["rig_instance_pose"][[0, 4]], pr["rig_instance_pose"][[0, 4]])
    assert np.array_equal(g["points"][pr["point_fixed"] == 1], pr["points"][pr["point_fixed"] == 1])


def test_wide_band_on_the_hand_written_gemm_matches_oracle(oracle_lib):
    """ragged tracks: co-visibility half-width beyond the cluster-tridiagonal band -> cyclic reduction over dense clusters, every batched
    product on dgemm_mfma_kernel (NN, NT and TN forms, edges that are not multiples of the 64 x 64 tile or of the K step of 32)"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(90, 800, 8, seed=3, ragged=True)
    with emulated():
        g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 2}, **NO_TOL)
    o = oracle_lib.ba_solve(pr, max_iterations=2, **NO_TOL)
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"] > 10
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-10)
    # the factorisation is exact (the pivot blocks' inverses by 16 x 16 pivots on the matrix cores, panels of 96 + 18 unknowns here): CG
    # confirms in one iteration per LM step -- a wrong inverse would still converge, only slower
    assert g["pcg_iterations"] <= g["iterations"] + 1


def test_local_bundle_adjustment_problem_matches_oracle(oracle_lib):
    """BAHelpers::BundleLocal's problem (interior free, boundary and cameras constant): the band alone is the preconditioner, the camera
    rows are inert -- the per-shot block-Jacobi blocks are skipped (round 5) -- and CG confirms in one iteration per LM step"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(40, 500, 6, seed=7)
    sub = bundle.local_problem(pr, 20, {"local_bundle_radius": 3, "local_bundle_min_common_points": 20, "local_bundle_max_shots": 8})[0]
    assert sub["cam_fixed"].all() and 0 < sub["shot_fixed"].sum() < len(sub["shot_fixed"])
    with emulated():
        g = bundle.bundle_arrays(sub, {"bundle_max_iterations": 4}, **NO_TOL)
    o = oracle_lib.ba_solve(sub, max_iterations=4, **NO_TOL)
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-11) and g["pcg_iterations"] <= 4
    assert np.array_equal(g["shot_pose"][sub["shot_fixed"] == 1], sub["shot_pose"][sub["shot_fixed"] == 1])


def test_compact_generic_rows_equal_the_rows_with_border_slots(monkeypatch):
    """gen_eval_kernel's COMPACT layout (round 6: (Xc, wt) instead of the 2 KW border slots; pass A of the mat-vec, the back-substitution and the border's
    point pass rebuild the slots with project_full on the same Xc) against OSFM_BA_GEN_FULL_ROWS: the same expressions, the same bits"""
    from opensfm_amd import bundle

    pr = synthetic.make_general_ba_scene(20, 300, 5, model="brown", n_gcp=3, gps_bias=True, seed=11)
    res = []
    for full in (False, True):
        if full:
            monkeypatch.setenv("OSFM_BA_GEN_FULL_ROWS", "1")
        else:
            monkeypatch.delenv("OSFM_BA_GEN_FULL_ROWS", raising=False)
        with emulated():
            res.append(bundle.bundle_general_arrays(pr, {"bundle_max_iterations": 3}, **NO_TOL))
    a, b = res
    assert np.array_equal(a["cost_history"], b["cost_history"])
    for k in ("cam_params", "rig_instance_pose", "points", "bias"):
        assert np.array_equal(a[k], b[k]), k


def test_one_workgroup_band_factor_equals_the_cyclic_reduction(monkeypatch):
    """Few shots: the band is factorised by ONE workgroup (sband_factor_kernel: block LDL^T with 6 x 6 pivots, the next pivot inverted by wavefront 0
    beside the trailing update) and applied by sband_solve_kernel -- against the cyclic reduction of the same band (OSFM_BA_NO_SBAND): both are exact,
    CG confirms in one iteration per LM step, the trajectories agree to rounding.  A local problem (constant cameras) and a full one (the shared
    camera's exact border: four right-hand sides through the one-workgroup solve)."""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(40, 500, 6, seed=7)
    sub = bundle.local_problem(pr, 20, {"local_bundle_radius": 3, "local_bundle_min_common_points": 20, "local_bundle_max_shots": 8})[0]
    full = synthetic.make_ba_scene(14, 300, 5, seed=3)
    for prob in (sub, full):
        res = []
        for no_sband in (False, True):
            if no_sband:
                monkeypatch.setenv("OSFM_BA_NO_SBAND", "1")
            else:
                monkeypatch.delenv("OSFM_BA_NO_SBAND", raising=False)
            with emulated():
                res.append(bundle.bundle_arrays(prob, {"bundle_max_iterations": 4}, **NO_TOL))
        a, b = res
        assert a["pcg_iterations"] == b["pcg_iterations"] == a["iterations"]
        assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-12)
        # (the full problem has its gauge held by the GPS priors alone: two exact solves differ by rounding times the gauge's conditioning)
        assert np.abs(a["shot_pose"] - b["shot_pose"]).max() < 1e-8 and np.abs(a["points"] - b["points"]).max() < 1e-8


def test_pivot_block_inverse_on_the_matrix_cores():
    """dgj_pivot_kernel's inverse (16 x 16 pivots, v_mfma_f64_16x16x4 updates in LDS, the next pivot block inverted by wavefront 0 beside the
    trailing tiles) on random SPD blocks of every order the panels take: max |A A^-1 - I| at rounding level"""
    import re
    import subprocess

    import emu_util

    build_emu = emu_util._builder()
    build_emu.build()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")
    exe = os.path.join(here, "_build", "gj16_harness")
    root = os.path.dirname(os.path.dirname(here))
    subprocess.run([build_emu.CLANG, "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-unknown-attributes", "-Wno-unused-value", "-I", os.path.join(here, "hipemu"),
                    "-I", os.path.join(root, "opensfm_amd", "csrc"), "-I", os.path.join(root, "include"), "-I", os.path.join(here, "_build"),
                    os.path.join(here, "gj16_harness.cpp"), os.path.join(here, "emu_ctx.cpp"), "-o", exe, "-lpthread"], check=True, timeout=600)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    rows = re.findall(r"w (\d+): max \|A inv - I\| = (\S+) status (\d+)", out)
    assert [int(r[0]) for r in rows] == [90, 96, 72, 36, 18, 6], out
    assert all(float(r[1]) < 1e-13 and r[2] == "0" for r in rows), out


def test_wide_band_falls_back_to_the_ldlt_chain_when_the_dense_clusters_do_not_fit(capfd, monkeypatch):
    """the dense-cluster blocks of the wide band are 2-3 x the memory of the block LDL^T window: when the device cannot hold them the
    solver takes the LDL^T chain instead of returning OSFM_E_NOMEM (OSFM_BA_DENSE_CR_BUDGET stands in for the free memory) -- the same
    exact band, the same trajectory"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(60, 500, 8, seed=3, ragged=True)
    monkeypatch.setenv("OSFM_BA_TRACE", "1")
    with emulated():
        a = bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, **NO_TOL)
        err_a = capfd.readouterr().err
        monkeypatch.setenv("OSFM_BA_DENSE_CR_BUDGET", "100000")
        b = bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, **NO_TOL)
        err_b = capfd.readouterr().err
    assert "wide 1 dense 1" in err_a and "wide 1 dense 0" in err_b
    assert np.allclose(a["cost_history"], b["cost_history"], rtol=1e-11) and b["pcg_iterations"] <= b["iterations"] + 1
