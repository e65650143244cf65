"""The bundle-adjustment seam on the MI355X: the reference's own test_bundle.py cases and the adapter scenes of ``bundle_cases`` through
the real solvers, and the facade's results against the CPU oracle on the same flattened problem."""
import numpy as np
import pytest

import bundle_cases as cases
from opensfm_amd import bundle, opensfm_adapter

pytestmark = pytest.mark.gpu


def test_reference_singleton(gpu_ctx):
    sa = cases.case_singleton()
    assert sa.solver == "osfm_bundle_solve"  # no observations: the general solver


def test_reference_singleton_pan_tilt_roll(gpu_ctx):
    cases.case_singleton_pan_tilt_roll()


def test_pair_with_points_priors(gpu_ctx, oracle_lib):
    sa = cases.case_pair_with_points_priors()
    assert np.allclose(sa.get_rig_instance_pose("1").translation, [0.5, -2, 2], atol=1e-2)
    assert np.allclose(sa.get_rig_instance_pose("2").translation, [-1.5, -2, 2], atol=1e-2)
    assert np.allclose(sa.get_point("p1").p, [-0.5, 2, 2], atol=1e-6)
    assert np.allclose(sa.get_point("p2").p, [1.5, 2, 2], atol=1e-6)


def test_pair_with_depth_priors(gpu_ctx):
    """depth priors through the facade on the real solver (RelativeDepthError, bundle_adjuster.cc:497-528)"""
    sa, (z1, r2) = cases.case_pair_with_depth_priors()
    assert np.allclose(sa.get_rig_instance_pose("1").translation, [0.5, -2, 2], atol=1e-2)
    assert np.allclose(sa.get_point("p1").p, [-0.5, 2, 2], atol=1e-4)
    sb, _ = cases.case_pair_with_depth_priors(contradict=True)
    pose = sb.get_rig_instance_pose("2")
    got = np.linalg.norm(pose.get_R_world_to_cam() @ sb.get_point("p1").p + pose.get_t_world_to_cam())
    assert abs(got - 1.5 * r2) < 1e-2 * r2


def test_reference_void_gps_ignored(gpu_ctx):
    cases.case_void_gps_ignored()


def test_reference_alignment_prior(gpu_ctx):
    cases.case_alignment_prior()


def test_adapter_fixed_internals_takes_the_streaming_solver(gpu_ctx, monkeypatch):
    used = []
    real = bundle.bundle_arrays
    monkeypatch.setattr(bundle, "bundle_arrays", lambda *a, **k: used.append(1) or real(*a, **k))
    cases.case_adapter_fixed_internals()
    assert used  # perspective cameras, identity rig camera, no biases: osfm_ba_solve


def test_streaming_and_general_solvers_agree(gpu_ctx, monkeypatch):
    r1, _ = cases.case_adapter_fixed_internals()
    monkeypatch.setattr(bundle.BundleAdjuster, "_streaming_form", staticmethod(lambda prob: None))
    r2, _ = cases.case_adapter_fixed_internals()
    a = np.array([p.coordinates for p in r1.points.values()])
    b = np.array([p.coordinates for p in r2.points.values()])
    assert np.abs(a - b).max() < 1e-4
    o1 = np.array([s.pose.get_origin() for s in r1.shots.values()])
    o2 = np.array([s.pose.get_origin() for s in r2.shots.values()])
    assert np.abs(o1 - o2).max() < 1e-4


def test_adapter_rig_gps_bias_gcp_matches_the_oracle(gpu_ctx, oracle_lib, monkeypatch):
    """the whole BAHelpers::Bundle flow (rigs, two camera models, biases, control points) on the GPU == the same flattened problem in the oracle"""
    prob, r, rep = cases.case_adapter_rig_gps_bias_gcp()
    gt = prob["gt_points"]
    est = np.array([r.points["p%d" % p].coordinates for p in range(len(gt))])
    assert np.sqrt(((est - gt) ** 2).sum(1).mean()) < 0.05
    assert "iterations" in rep["brief_report"]
    monkeypatch.setattr(bundle, "bundle_general_arrays", cases.oracle_solver(oracle_lib))
    _, r_o, _ = cases.case_adapter_rig_gps_bias_gcp()
    est_o = np.array([r_o.points["p%d" % p].coordinates for p in range(len(gt))])
    assert np.abs(est - est_o).max() < 1e-6
    for k in r.rig_instances:
        assert np.allclose(r.rig_instances[k].pose.cam_to_world_parameters(), r_o.rig_instances[k].pose.cam_to_world_parameters(), atol=1e-7)
    for k in r.cameras:
        assert np.allclose(r.cameras[k].get_parameters_values(), r_o.cameras[k].get_parameters_values(), atol=1e-8)
        assert np.allclose(r.biases[k].parameters(), r_o.biases[k].parameters(), atol=1e-7)
    assert np.allclose(r.rig_cameras["rc1"].pose.cam_to_world_parameters(), r_o.rig_cameras["rc1"].pose.cam_to_world_parameters(), atol=1e-7)


def test_triangulate_gcp_uses_the_device_bearings(gpu_ctx):
    from opensfm_amd.geometry_types import GroundControlPoint, GroundControlPointObservation

    models = ("brown",)
    prob = cases.scene(models, rig=False, gps=False, free_cameras=False, px_noise=0.0, outlier_frac=0.0, n_instances=8, n_points=120)
    prob["rig_instance_pose"] = prob["gt_rig_instance"].copy()
    r = cases.reconstruction_from_problem(prob, models)
    p = 11
    point = GroundControlPoint("g", {}, True)
    for s, xy in zip(prob["obs_shot"][prob["obs_point"] == p], prob["obs_xy"][prob["obs_point"] == p]):
        point.observations.append(GroundControlPointObservation("s%03d" % s, xy))
    ok, X = opensfm_adapter.triangulate_gcp(point, r.shots)
    assert ok and np.linalg.norm(X - prob["gt_points"][p]) < 1e-6


def test_pan_tilt_roll_priors_match_the_oracle(gpu_ctx, oracle_lib, monkeypatch):
    g1, g2 = cases.case_singleton_pan_tilt_roll(), cases.case_singleton()
    monkeypatch.setattr(bundle, "bundle_general_arrays", cases.oracle_solver(oracle_lib))
    o1, o2 = cases.case_singleton_pan_tilt_roll(), cases.case_singleton()
    for g, o in ((g1, o1), (g2, o2)):
        assert np.allclose(g.get_rig_instance_pose("1").cam_to_world_parameters(), o.get_rig_instance_pose("1").cam_to_world_parameters(), atol=1e-9)
        assert g._report["iterations"] == o._report["iterations"]
        assert np.allclose(g._report["cost_history"], o._report["cost_history"], rtol=1e-9, atol=1e-18)


def test_bundle_local_and_shot_poses_over_map_objects(gpu_ctx, oracle_lib, monkeypatch):
    """pysfm.BAHelpers.bundle_local / bundle_shot_poses (reconstruction.py:89-127) on the real solver == the same flows with the oracle"""
    rg, rep_g, interior, _ = cases.case_bundle_local()
    sg, _ = cases.case_bundle_shot_poses()
    monkeypatch.setattr(bundle, "bundle_general_arrays", cases.oracle_solver(oracle_lib))
    monkeypatch.setattr(bundle.BundleAdjuster, "_streaming_form", staticmethod(lambda prob: None))
    ro, rep_o, interior_o, _ = cases.case_bundle_local()
    so, _ = cases.case_bundle_shot_poses()
    assert interior == interior_o and rep_g["num_reprojections"] == rep_o["num_reprojections"]
    for a, b in ((rg, ro), (sg, so)):
        for k in a.rig_instances:
            assert np.allclose(a.rig_instances[k].pose.cam_to_world_parameters(), b.rig_instances[k].pose.cam_to_world_parameters(), atol=1e-7)
        for k in a.points:
            assert np.allclose(a.points[k].coordinates, b.points[k].coordinates, atol=1e-7)
