"""Host logic of the bundle-adjustment seam on CPU: the ``BundleAdjuster`` facade (string ids -> osfm_bundle_problem arrays, the
reference's sigma rules) and ``opensfm_adapter.bundle`` (= BAHelpers::Bundle + BundleToMap).  The numerical solver is replaced by
the CPU oracle here -- the same case bodies run against the real HIP solver in ``test_gpu_bundle_facade.py``."""
import numpy as np
import pytest

import bundle_cases as cases
from opensfm_amd import bundle, opensfm_adapter
from opensfm_amd.geometry_types import Camera, GroundControlPoint, GroundControlPointObservation, Pose, Similarity


@pytest.fixture()
def cpu_solver(monkeypatch, oracle_lib):
    monkeypatch.setattr(bundle, "bundle_general_arrays", cases.oracle_solver(oracle_lib))
    monkeypatch.setattr(bundle.BundleAdjuster, "_streaming_form", staticmethod(lambda prob: None))

    def bearing(camera, projection):  # Camera::Bearing on the host = the oracle's, the product's is osfm_pixel_bearings
        from opensfm_amd.matching import camera_parameters

        if hasattr(camera, "pixel_bearing"):
            return np.asarray(camera.pixel_bearing(projection), float)
        model, par = camera_parameters(camera)
        return oracle_lib.pixel_bearings_generic(model, par, np.asarray(projection, float).reshape(1, 2))[0]

    monkeypatch.setattr(opensfm_adapter, "_bearing", bearing)


def test_reference_unicode_ids():
    cases.case_unicode_ids()


def test_reference_singleton(cpu_solver):
    sa = cases.case_singleton()
    assert sa.solver == "osfm_bundle_solve"


def test_reference_singleton_pan_tilt_roll(cpu_solver):
    cases.case_singleton_pan_tilt_roll()


def test_pair_with_points_priors(cpu_solver):
    sa = cases.case_pair_with_points_priors()
    assert np.allclose(sa.get_rig_instance_pose("1").translation, [0.5, -2, 2], atol=1e-2)
    assert np.allclose(sa.get_rig_instance_pose("2").translation, [-1.5, -2, 2], atol=1e-2)
    assert np.allclose(sa.get_point("p1").p, [-0.5, 2, 2], atol=1e-6)
    assert np.allclose(sa.get_point("p2").p, [1.5, 2, 2], atol=1e-6)
    assert set(sa.get_point("p1").reprojection_errors) == {"1", "2"}


def test_pair_with_depth_priors(cpu_solver):
    sa, (z1, r2) = cases.case_pair_with_depth_priors()
    assert z1 > 1.0 and r2 > z1
    assert np.allclose(sa.get_rig_instance_pose("1").translation, [0.5, -2, 2], atol=1e-2)
    assert np.allclose(sa.get_point("p1").p, [-0.5, 2, 2], atol=1e-4)
    assert set(sa.get_point("p1").reprojection_errors) == {"1", "2"}
    sb, _ = cases.case_pair_with_depth_priors(contradict=True)
    pose = sb.get_rig_instance_pose("2")
    got = np.linalg.norm(pose.get_R_world_to_cam() @ sb.get_point("p1").p + pose.get_t_world_to_cam())
    assert abs(got - 1.5 * r2) < 1e-2 * r2  # the strong prior is met
    sa.add_point_projection_observation("1", "p2", np.array([0, 0]), 1, cases.Depth(float("nan"), True, 1.0))
    with pytest.raises(RuntimeError, match="1 has non-finite depth prior"):  # thrown from Run with the shot id, bundle_adjuster.cc:508-511
        sa.run()
    with pytest.raises(ValueError):  # a standard deviation <= 0 is refused, not silently read as "no prior"
        sa.add_point_projection_observation("1", "p2", np.array([0, 0]), 1, cases.Depth(1.0, True, 0.0))


def test_reference_void_gps_ignored(cpu_solver):
    cases.case_void_gps_ignored()


def test_reference_alignment_prior(cpu_solver):
    cases.case_alignment_prior()


def test_adapter_fixed_internals(cpu_solver):
    cases.case_adapter_fixed_internals()


def test_adapter_rig_gps_bias_gcp(cpu_solver):
    prob, r, rep = cases.case_adapter_rig_gps_bias_gcp()
    gt = prob["gt_points"]
    est = np.array([r.points["p%d" % p].coordinates for p in range(len(gt))])
    assert np.sqrt(((est - gt) ** 2).sum(1).mean()) < 0.05  # anchored by GPS through the bias + control points
    assert not r.biases["c0"].scale == 1.0  # the free bias moved
    assert not np.allclose(r.rig_cameras["rc1"].pose.cam_to_world_parameters(), prob["rig_camera_pose"][1])  # 24 shots per rig camera: free


# ---- flattening rules ----
def _two_camera_adjuster():
    ba = bundle.BundleAdjuster()
    brown = Camera.create_brown(0.9, 1.01, [0.01, -0.02], [0.05, 0.01, 0.001, 0.002, -0.001])
    f624 = Camera.create_fisheye624(0.6, 1.0, [0.0, 0.0], [0.01] * 12)
    ba.add_camera("b", brown, brown, False)
    ba.add_camera("f", f624, f624, False)
    ba.add_rig_camera("rc", Pose(), Pose(), True)
    ba.add_rig_camera("rc2", Pose(np.array([0.0, 0.1, 0]), np.array([0.3, 0, 0])), Pose(), False)
    ba.add_rig_instance("i", Pose(), {"s2": "f", "s1": "b"}, {"s1": "rc", "s2": "rc2"}, False)
    return ba


def test_camera_sigma_follows_the_reference_table():
    """GetDefaultCameraSigma (bundle_adjuster.cc:47-69): focal / aspect ratio / principal point / k1 k2 k3 / p1 p2 from
    SetInternalParametersPriorSD, 0 for the parameter types the table has no entry for (k4 .. k6, s0 .. s3)"""
    ba = _two_camera_adjuster()
    ba.set_internal_parameters_prior_sd(0.11, 0.12, 0.13, 0.14, 0.15, 0.16, 0.17, 0.18, 0.19)
    prob = ba._problem()
    # brown: k1 k2 k3 p1 p2 focal aspect cx cy
    assert np.allclose(prob["cam_sigma"][0, :9], [0.14, 0.15, 0.18, 0.16, 0.17, 0.11, 0.12, 0.13, 0.13])
    # fisheye624: k1..k6 p1 p2 s0..s3 focal aspect cx cy
    assert np.allclose(prob["cam_sigma"][1], [0.14, 0.15, 0.18, 0, 0, 0, 0.16, 0.17, 0, 0, 0, 0, 0.11, 0.12, 0.13, 0.13])
    assert list(prob["cam_model"]) == [2, 5]


def test_rig_sigma_and_pose_layout():
    ba = _two_camera_adjuster()
    ba.set_rig_parameters_prior_sd(0.3, 0.7)
    prob = ba._problem()
    assert np.allclose(prob["rig_camera_sigma"], [[0.7, 0.7, 0.7, 0.3, 0.3, 0.3]] * 2)
    # bundle::Pose data is camera-to-world: [-rotation, origin]
    p = Pose(np.array([0.0, 0.1, 0]), np.array([0.3, 0, 0]))
    assert np.allclose(prob["rig_camera_pose"][1], np.r_[-p.rotation, p.get_origin()])
    assert np.allclose(prob["rig_camera_prior"][1], 0)
    assert list(prob["rig_camera_fixed"]) == [1, 0]


def test_bias_rules():
    """AddCamera creates a constant identity bias, SetCameraBias a free one; the prior of an instance goes through the bias of the
    camera of its first shot (std::map order)"""
    ba = _two_camera_adjuster()
    ba.add_rig_instance_position_prior("i", np.array([1.0, 2, 3]), np.array([2.0, 2, 2]), "")
    ba.set_camera_bias("b", Similarity(np.array([0, 0, 0.1]), np.array([1.0, 0, 0]), 1.1))
    prob = ba._problem()
    assert list(prob["bias_fixed"]) == [0, 1]
    assert np.allclose(prob["bias"][0], [0, 0, 0.1, 1, 0, 0, 1.1]) and np.allclose(prob["bias"][1], [0, 0, 0, 0, 0, 0, 1])
    assert prob["rig_instance_bias_camera"][0] == 0  # "s1" < "s2" -> camera "b"
    with pytest.raises(RuntimeError):
        ba.set_camera_bias("nope", Similarity())


def test_unknown_ids_raise():
    ba = _two_camera_adjuster()
    with pytest.raises(RuntimeError):
        ba.add_rig_instance("j", Pose(), {"s": "missing"}, {"s": "rc"}, False)
    with pytest.raises(RuntimeError):
        ba.add_rig_instance_position_prior("missing", np.zeros(3), np.ones(3), "")
    with pytest.raises(IndexError):
        ba.add_point_projection_observation("s1", "missing", np.zeros(2), 1.0)
    with pytest.raises(RuntimeError):
        ba.add_rig_camera("rc", Pose(), Pose(), True)
    with pytest.raises(NotImplementedError):
        ba.add_relative_motion(None)
    with pytest.raises(RuntimeError):
        ba.set_linear_solver_type("NOPE")


def test_streaming_dispatch_rule():
    """the streaming solver only gets the problems it can represent"""
    from opensfm_amd import synthetic

    prob = cases.scene(("perspective",), rig=False, gps=True, n_instances=6, n_points=60)
    prob = {k: v for k, v in prob.items() if not k.startswith("gt_") and k not in ("is_outlier", "models")}
    prob["rig_camera_fixed"] = np.ones(1, np.uint8)
    s = bundle.BundleAdjuster._streaming_form(prob)
    assert s is not None and s["shot_pose"].shape == (6, 6) and s["cam_params"].shape == (1, 3) and "shot_gps" in s
    for change in ({"cam_model": np.array([2], np.int32)}, {"bias_fixed": np.zeros(1, np.uint8)}, {"rig_camera_fixed": np.zeros(1, np.uint8)},
                   {"shot_pan": np.zeros(6), "shot_pan_sigma": np.ones(6)}, {"point_prior": np.zeros((len(prob["points"]), 3))}):
        assert bundle.BundleAdjuster._streaming_form(dict(prob, **change)) is None
    rig = cases.scene(("perspective",), rig=True, n_instances=6, n_points=60)
    assert bundle.BundleAdjuster._streaming_form(rig) is None


# ---- adapter pieces ----
def test_triangulate_gcp_and_alignment_detection():
    models = ("perspective",)
    prob = cases.scene(models, rig=False, gps=True, free_cameras=False, px_noise=0.0, outlier_frac=0.0, n_instances=8, n_points=120)
    prob["rig_instance_pose"] = prob["gt_rig_instance"].copy()
    prob["cam_params"] = prob["gt_cam"].copy()
    r = cases.reconstruction_from_problem(prob, models, gps_accuracy=1.0)
    p = 17
    point = GroundControlPoint("g", {}, True)
    for s, xy in zip(prob["obs_shot"][prob["obs_point"] == p], prob["obs_xy"][prob["obs_point"] == p]):
        point.observations.append(GroundControlPointObservation("s%03d" % s, xy))
    point.observations.append(GroundControlPointObservation("not-a-shot", np.zeros(2)))

    class HostCamera:  # TriangulateGCP needs Camera::Bearing; here the closed form of an undistorted perspective camera
        def __init__(self, cam):
            self.cam = cam

        def pixel_bearing(self, px):
            v = np.array([px[0], px[1], self.cam.focal])
            return v / np.linalg.norm(v)

    prob_cam = r.cameras["c0"]
    prob_cam.k1 = prob_cam.k2 = 0.0
    for shot in r.shots.values():
        shot.camera = HostCamera(prob_cam)
    ok, X = opensfm_adapter.triangulate_gcp(point, r.shots)
    # the observations were generated with distortion; without it the rays still meet within centimetres
    assert ok and np.linalg.norm(X - prob["gt_points"][p]) < 0.2
    lonely = GroundControlPoint("h", {}, True)
    lonely.observations.append(point.observations[0])
    assert opensfm_adapter.triangulate_gcp(lonely, r.shots)[0] is False
    # instances along a line: "orientation_prior"; with fewer than three constraints as well
    for k, shot in enumerate(r.shots.values()):
        shot.rig_instance.pose.set_origin(np.array([0.5 * k, 0.0, 0.0]))
    assert opensfm_adapter.detect_alignment_constraints(r, {"bundle_use_gps": True}, []) == "orientation_prior"
    for k, shot in enumerate(r.shots.values()):
        shot.rig_instance.pose.set_origin(shot.rig_instance.pose.get_origin() + np.array([0.0, (k % 3) * 2.0, (k % 2) * 1.0]))
    assert opensfm_adapter.detect_alignment_constraints(r, {"bundle_use_gps": True}, []) == "naive"
    assert opensfm_adapter.detect_alignment_constraints(r, {"bundle_use_gps": False}, []) == "orientation_prior"


def test_midpoint_triangulation_closed_form():
    rng = np.random.default_rng(0)
    X = np.array([0.3, -0.2, 5.0])
    centers = rng.normal(0, 1.0, (4, 3))
    bearings = X - centers
    bearings /= np.linalg.norm(bearings, axis=1)[:, None]
    ok, Y = opensfm_adapter.triangulate_bearings_midpoint(centers, bearings, [1.0] * 4, 0.1 * np.pi / 180, 1e-3)
    assert ok and np.allclose(Y, X, atol=1e-12)
    ok, _ = opensfm_adapter.triangulate_bearings_midpoint(centers[:2], np.tile(bearings[0], (2, 1)), [1.0, 1.0], 0.1 * np.pi / 180, 1e-3)
    assert not ok  # parallel rays
    ok, _ = opensfm_adapter.triangulate_bearings_midpoint(centers, -bearings, [1.0] * 4, 0.1 * np.pi / 180, 1e-3)
    assert not ok  # behind the cameras


def test_gps_accuracy_must_be_positive(cpu_solver):
    r, camera, shot = cases._single_shot_reconstruction(np.random.default_rng(4))
    shot.metadata.gps_position, shot.metadata.gps_accuracy = np.zeros(3), 0.0
    with pytest.raises(RuntimeError, match="accuracy <= 0"):
        opensfm_adapter.bundle(r, {camera.id: camera}, dict(r.rig_cameras.items()), [], None)


def test_gcp_weights(cpu_solver):
    """AddGCPToBundle (ba_helpers.cc:348-405): global weight = gcp_global_weight * dominant terms / total control-point terms; the
    point prior gets sd / weight, the observations 0.001 / weight"""
    prob, r, rep = cases.case_adapter_rig_gps_bias_gcp(n_instances=6)
    calls = []

    class Spy(bundle.BundleAdjuster):
        def add_point_prior(self, *a):
            calls.append(("prior", a))
            super().add_point_prior(*a)

        def add_point_projection_observation(self, shot, point, obs, sd, depth_prior=None):
            if str(point).startswith("gcp-"):
                calls.append(("obs", sd))
            super().add_point_projection_observation(shot, point, obs, sd, depth_prior)

    cams, rigs = cases.priors_from_problem(prob, ("perspective", "brown"))
    point = GroundControlPoint("g", {"latitude": 1.0, "longitude": 2.0, "altitude": 3.0}, False)
    shots = sorted(r.shots)[:3]
    for s in shots:
        point.observations.append(GroundControlPointObservation(s, np.array([0.01, 0.02])))
    point.observations.append(GroundControlPointObservation("unknown", np.zeros(2)))
    opensfm_adapter.bundle(r, cams, rigs, [point], {"bundle_use_gcp": True, "align_method": "naive", "bundle_max_iterations": 2}, adjuster=Spy())
    n_obs = sum(len(s.observations) for s in r.shots.values())
    weight = 0.01 * (len(r.rig_instances) + n_obs) / (1 + 3)
    prior = [c for c in calls if c[0] == "prior"][0][1]
    assert prior[0] == "gcp-g" and np.allclose(prior[1], [1, 2, 0]) and np.allclose(prior[2], np.array([0.01, 0.01, 0.1]) / weight) and prior[3] is False
    assert [c[1] for c in calls if c[0] == "obs"] == [pytest.approx(0.001 / weight)] * 3


def test_bundle_local_over_map_objects(cpu_solver):
    """pysfm.BAHelpers.bundle_local / shot_neighborhood_ids as reconstruction.py:107-149 call them (the solver: the CPU oracle)"""
    cases.case_bundle_local()


def test_bundle_shot_poses_over_map_objects(cpu_solver):
    """pysfm.BAHelpers.bundle_shot_poses as reconstruction.py:89-104 calls it"""
    cases.case_bundle_shot_poses()


def test_the_references_reconstruction_module_drives_the_adapter(cpu_solver):
    """opensfm/reconstruction.py of the reference, loaded as it is: its bundle / bundle_local / bundle_shot_poses (and its log_bundle_stats)
    on top of ``opensfm_amd.compat.pysfm`` -- the same results as the adapter called directly"""
    ref = cases.load_reference_reconstruction()
    if ref is None:
        pytest.skip("/root/reference is not mounted")
    assert cases.case_reference_reconstruction_module(ref)


def test_the_references_test_bundle_functions_pass(cpu_solver):
    """opensfm/test/test_bundle.py of the reference, executed: unicode ids, the singleton, pan / tilt / roll, void GPS values, the
    alignment prior -- its assertions and tolerances, its config.default_config(), its reconstruction.bundle"""
    t = cases.load_reference_test_bundle()
    if t is None:
        pytest.skip("/root/reference is not mounted")
    assert cases.case_reference_test_bundle(t)
