"""Bodies of the bundle-adjustment facade / adapter tests, shared by the CPU run (the facade's host logic, with the oracle standing
in for the solver) and the GPU run (the real ``osfm_bundle_solve`` / ``osfm_ba_solve``).  The first group restates the reference's
own ``opensfm/test/test_bundle.py`` cases that ``BAHelpers::Bundle`` can reach (no relative motions / heatmaps / linear motion)."""
import numpy as np

from opensfm_amd import bundle, opensfm_adapter, synthetic
from opensfm_amd.geometry_types import (Camera, Depth, GroundControlPoint, GroundControlPointObservation, Observation, Pose, Reconstruction,
                                        RigCamera, RigInstance, Similarity)


def oracle_solver(oracle_lib):
    """drop-in for ``bundle.bundle_general_arrays`` that runs the CPU oracle (tests only)"""

    def solve(problem, config=None, ctx=None, **overrides):
        cfg = dict(config or {})
        out = oracle_lib.bundle_general(problem, loss=cfg.get("loss_function", "SoftLOneLoss"), loss_threshold=cfg.get("loss_function_threshold", 1.0),
                                        max_iterations=cfg.get("bundle_max_iterations", 100), **overrides)
        if "bias" not in out:
            out["bias"] = np.tile([0, 0, 0, 0, 0, 0, 1.0], (len(problem["cam_model"]), 1))
        out["brief_report"] = "oracle: iterations %d" % out["iterations"]
        return out

    return solve


# ---- pan / tilt / roll of a world-to-camera rotation (opensfm/geometry.py:22-49) ----
def ptr_from_rotation(R):
    ez, ex = R.T @ [0, 0, 1.0], R.T @ [1.0, 0, 0]
    pan = np.arctan2(ez[0], ez[1])
    tilt = np.arctan2(-ez[2], np.linalg.norm(ez[:2]))
    a = np.cross(ez, [0, 0, 1.0])
    a /= np.linalg.norm(a)
    return pan, tilt, np.arcsin(ez @ np.cross(ex, a))


def _adjuster():
    """the reference's ``bundle_adjuster`` fixture (test_bundle.py:38-44)"""
    ba = bundle.BundleAdjuster()
    camera = Camera.create_perspective(1.0, 0.0, 0.0)
    ba.add_camera("cam1", camera, camera, True)
    ba.add_rig_camera("rig_cam1", Pose(), Pose(), True)
    return ba


def case_unicode_ids():
    """test_bundle.py:21-35"""
    ba = bundle.BundleAdjuster()
    camera = Camera.create_perspective(0.4, 0.1, -0.01)
    ba.add_camera("A\xb2", camera, camera, True)
    ba.add_camera(b"A_2", camera, camera, True)


def case_singleton():
    """test_bundle.py:47-76 (test_sigleton)"""
    sa = _adjuster()
    sa.add_rig_instance("1", Pose(np.array([0.5, 0, 0]), np.array([0, 0, 0])), {"1": "cam1"}, {"1": "rig_cam1"}, False)
    sa.add_rig_instance_position_prior("1", np.array([1, 0, 0]), np.array([1, 1, 1]), "")
    sa.add_absolute_up_vector("1", np.array([0, -1, 0]), 1)
    sa.add_absolute_pan("1", np.radians(180), 1)
    sa.run()
    s1 = sa.get_rig_instance_pose("1")
    assert np.allclose(s1.translation, [1, 0, 0], atol=1e-6)
    return sa


def case_singleton_pan_tilt_roll():
    """test_bundle.py:79-111"""
    pan, tilt, roll = 1, 0.3, 0.2
    sa = _adjuster()
    sa.add_rig_instance("1", Pose(np.array([0.5, 0, 0]), np.array([0, 0, 0])), {"1": "cam1"}, {"1": "rig_cam1"}, False)
    sa.add_rig_instance_position_prior("1", np.array([1, 0, 0]), np.array([1, 1, 1]), "")
    sa.add_absolute_pan("1", pan, 1)
    sa.add_absolute_tilt("1", tilt, 1)
    sa.add_absolute_roll("1", roll, 1)
    sa.run()
    pose = sa.get_rig_instance_pose("1")
    assert np.allclose(pose.get_origin(), [1, 0, 0], atol=1e-6)
    assert np.allclose(ptr_from_rotation(pose.get_rotation_matrix()), (pan, tilt, roll))
    return sa


def case_pair_with_points_priors():
    """the part of test_bundle.py:240-341 that BAHelpers can reach: two instances anchored by two control points with position priors and
    the pan / tilt / roll priors of an identity rotation (the reference case also ties the pair with a relative motion, which only
    fixes what the GPS-free case leaves free: the second rotation, pinned here by the same angles)"""
    sa = _adjuster()
    for i in range(2):
        sa.add_rig_instance(str(i + 1), Pose(np.array([1e-3, 1e-3, 1e-3]), np.array([1e-3, 1e-3, 1e-3])), {str(i + 1): "cam1"},
                            {str(i + 1): "rig_cam1"}, False)
    sa.add_point("p1", np.array([0, 0, 0]), False)
    sa.add_point("p2", np.array([0, 0, 0]), False)
    for s in ("1", "2"):
        sa.add_absolute_roll(s, np.radians(90), 1)
        sa.add_absolute_pan(s, -np.radians(90), 1)
        sa.add_absolute_tilt(s, -np.radians(90), 1)
    std_dev = np.array([1, 1, 1])
    sa.add_point_projection_observation(shot="1", point="p1", observation=np.array([0, 0]), std_deviation=1)
    sa.add_point_projection_observation(shot="2", point="p1", observation=np.array([-0.5, 0]), std_deviation=1)
    sa.add_point_prior("p1", np.array([-0.5, 2, 2]), std_dev, True)
    sa.add_point_projection_observation(shot="2", point="p2", observation=np.array([0, 0]), std_deviation=1)
    sa.add_point_projection_observation(shot="1", point="p2", observation=np.array([0.5, 0]), std_deviation=1)
    sa.add_point_prior("p2", np.array([1.5, 2, 2]), std_dev, True)
    sa.run()
    return sa


def case_pair_with_depth_priors(contradict=False):
    """the pair above with depth priors on p1 (RelativeDepthError, bundle_adjuster.cc:497-528): the z depth in shot 1 and the radial
    depth in shot 2, taken from the optimum of the pair above -- the optimum stays; `contradict`: a strong radial prior 1.5 x too long in
    shot 2, which the point follows (its position prior and reprojections give way)"""
    ref = case_pair_with_points_priors()

    def in_camera(shot):
        pose = ref.get_rig_instance_pose(shot)
        return pose.get_R_world_to_cam() @ ref.get_point("p1").p + pose.get_t_world_to_cam()

    z1, r2 = in_camera("1")[2], np.linalg.norm(in_camera("2"))
    sa = _adjuster()
    for i in range(2):
        sa.add_rig_instance(str(i + 1), Pose(np.array([1e-3, 1e-3, 1e-3]), np.array([1e-3, 1e-3, 1e-3])), {str(i + 1): "cam1"},
                            {str(i + 1): "rig_cam1"}, False)
    sa.add_point("p1", np.array([0, 0, 0]), False)
    sa.add_point("p2", np.array([0, 0, 0]), False)
    for s in ("1", "2"):
        sa.add_absolute_roll(s, np.radians(90), 1)
        sa.add_absolute_pan(s, -np.radians(90), 1)
        sa.add_absolute_tilt(s, -np.radians(90), 1)
    std_dev = np.array([1, 1, 1])
    sa.add_point_projection_observation("1", "p1", np.array([0, 0]), 1, Depth(z1, False, 0.1))
    sa.add_point_projection_observation("2", "p1", np.array([-0.5, 0]), 1, Depth(1.5 * r2, True, 1e-3) if contradict else Depth(r2, True, 0.1))
    sa.add_point_prior("p1", np.array([-0.5, 2, 2]), std_dev, True)
    sa.add_point_projection_observation("2", "p2", np.array([0, 0]), 1)
    sa.add_point_projection_observation("1", "p2", np.array([0.5, 0]), 1)
    sa.add_point_prior("p2", np.array([1.5, 2, 2]), std_dev, True)
    sa.run()
    return sa, (z1, r2)


def _single_shot_reconstruction(rng):
    camera = Camera.create_perspective(1.0, 0.0, 0.0)
    camera.id = "camera1"
    r = Reconstruction()
    r.add_camera(camera)
    shot = r.create_shot("1", camera.id, Pose(rng.random(3), rng.random(3)))
    return r, camera, shot


def case_void_gps_ignored():
    """test_bundle.py:686-730"""
    r, camera, shot = _single_shot_reconstruction(np.random.default_rng(1))
    camera_priors, rig_priors = {camera.id: camera}, dict(r.rig_cameras.items())
    for pos, acc, expected in ((None, 1, np.ones(3)), (np.zeros(3), None, np.ones(3)), (np.zeros(3), 1, np.zeros(3))):
        shot.metadata.gps_position, shot.metadata.gps_accuracy = pos, acc
        shot.rig_instance.pose.set_origin(np.ones(3))
        opensfm_adapter.bundle(r, camera_priors, rig_priors, [], None)
        assert np.allclose(r.shots["1"].pose.get_origin(), expected, atol=1e-6)


def case_alignment_prior():
    """test_bundle.py:733-759: one shot, no constraints -> "orientation_prior", cameras end with the Y axis pointing down"""
    r, camera, shot = _single_shot_reconstruction(np.random.default_rng(2))
    shot.metadata.gps_position, shot.metadata.gps_accuracy = np.array([0.0, 0, 0]), 1
    opensfm_adapter.bundle(r, {camera.id: camera}, dict(r.rig_cameras.items()), [], None)
    shot = r.shots["1"]
    assert np.allclose(shot.pose.translation, np.zeros(3), atol=1e-6)
    assert np.allclose(shot.pose.transform([0, 0, 1]), [0, -1, 0], atol=1e-7)


# ---- scenes through the adapter ----
def reconstruction_from_problem(prob, models, gps_accuracy=None):
    """geometry_types objects of a ``synthetic.make_bundle_scene`` problem (rig cameras "rc<k>", instances "i<k>", shots "s<k>" ...)"""
    r = Reconstruction()
    for c, m in enumerate(models):
        cam = Camera("spherical" if m == "spherical" else m)
        from opensfm_amd.geometry_types import set_camera_parameter_values

        set_camera_parameter_values(cam, prob["cam_params"][c])
        cam.id = "c%d" % c
        r.add_camera(cam)
    for k, v in enumerate(prob["rig_camera_pose"]):
        r.add_rig_camera(RigCamera("rc%d" % k, Pose.from_cam_to_world(v[:3], v[3:])))
    for k, v in enumerate(prob["rig_instance_pose"]):
        r.add_rig_instance(RigInstance("i%d" % k, Pose.from_cam_to_world(v[:3], v[3:])))
    for s in range(len(prob["shot_camera"])):
        shot = r.create_shot("s%03d" % s, "c%d" % prob["shot_camera"][s], None, "rc%d" % prob["shot_rig_camera"][s], "i%d" % prob["shot_rig_instance"][s])
        if "rig_instance_gps" in prob and gps_accuracy is not None:
            shot.metadata.gps_position = prob["rig_instance_gps"][prob["shot_rig_instance"][s]]
            shot.metadata.gps_accuracy = gps_accuracy
    for p, X in enumerate(prob["points"]):
        r.create_point("p%d" % p, X)
    for s, p, xy, sd in zip(prob["obs_shot"], prob["obs_point"], prob["obs_xy"], prob["obs_sigma"]):
        r.add_observation("s%03d" % s, "p%d" % p, Observation(xy[0], xy[1], sd))
    return r


def priors_from_problem(prob, models):
    from opensfm_amd.geometry_types import set_camera_parameter_values

    cams = {}
    for c, m in enumerate(models):
        cam = Camera(m)
        set_camera_parameter_values(cam, prob["cam_prior"][c])
        cam.id = "c%d" % c
        cams[cam.id] = cam
    rigs = {"rc%d" % k: RigCamera("rc%d" % k, Pose.from_cam_to_world(v[:3], v[3:])) for k, v in enumerate(prob["rig_camera_prior"])}
    return cams, rigs


def scene(models=("perspective", "brown"), **kw):
    kw.setdefault("n_gcp", 0)
    kw.setdefault("up_vectors", False)
    kw.setdefault("free_bias", False)
    return synthetic.make_bundle_scene(models=models, **kw)


def rmse(errors_by_point):
    e = np.array([np.asarray(v)[:2] for pt in errors_by_point for v in pt.reprojection_errors.values()])
    return float(np.sqrt((e**2).sum(1).mean()))


def case_adapter_fixed_internals():
    """test_bundle.py:120-168 (test_bundle_projection_fixed_internals): with optimize_camera_parameters off the cameras do not move and the
    reprojection errors collapse"""
    models = ("perspective",)
    prob = scene(models, rig=False, gps=False, free_cameras=False, outlier_frac=0.0, n_instances=8, n_points=120)
    r = reconstruction_from_problem(prob, models)
    cams, rigs = priors_from_problem(prob, models)
    before = r.cameras["c0"].get_parameters_values().copy()
    rep = opensfm_adapter.bundle(r, cams, rigs, [], {"bundle_use_gps": False, "optimize_camera_parameters": False})
    all_errors = [e for p in r.points.values() for v in p.reprojection_errors.values() for e in v]
    assert np.std(all_errors) < 5e-3
    assert np.array_equal(r.cameras["c0"].get_parameters_values(), before)
    assert rep["num_images"] == 8 and rep["num_points"] == len(prob["points"]) and rep["num_reprojections"] == len(prob["obs_shot"])
    assert set(rep["wall_times"]) == {"setup", "run", "teardown"}
    return r, rep


def case_adapter_rig_gps_bias_gcp(n_instances=24):
    """rigs (two rig cameras, the second free once there are > 10 shots per rig camera), two camera models with free intrinsics, GPS
    priors through free biases, ground control points with position priors: every block BAHelpers::Bundle can create"""
    models = ("perspective", "brown")
    prob = synthetic.make_bundle_scene(models=models, n_instances=n_instances, n_points=200, n_gcp=0, up_vectors=False, free_bias=True, seed=5)
    r = reconstruction_from_problem(prob, models, gps_accuracy=0.5)
    cams, rigs = priors_from_problem(prob, models)
    for c in r.cameras:
        r.biases[c] = Similarity()
    # control points: three ground-truth points seen from several shots, with their positions as "lla" through a flat converter
    class Flat:
        def to_topocentric(self, lat, lon, alt):
            return np.array([lat, lon, alt], float)

    r.reference = Flat()
    gcp = []
    gt = prob["gt_points"]
    for g, p in enumerate((3, 40, 90)):
        point = GroundControlPoint("g%d" % g, {"latitude": gt[p, 0], "longitude": gt[p, 1], "altitude": gt[p, 2]}, True)
        for s, xy in zip(prob["obs_shot"][prob["obs_point"] == p], prob["obs_xy"][prob["obs_point"] == p]):
            point.observations.append(GroundControlPointObservation("s%03d" % s, xy))
        gcp.append(point)
    cfg = {"bundle_use_gcp": True, "bundle_compensate_gps_bias": True, "align_method": "naive", "bundle_max_iterations": 50}
    rep = opensfm_adapter.bundle(r, cams, rigs, gcp, cfg)
    return prob, r, rep
