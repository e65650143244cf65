"""``pysfm.BAHelpers.bundle`` over reconstruction objects (``opensfm/src/sfm/src/ba_helpers.cc:580-762`` = ``BAHelpers::Bundle``,
``:764-830`` = ``BundleToMap``): the seam ``reconstruction.bundle`` (``opensfm/reconstruction.py:70-89``) calls.

The reconstruction is duck-typed on the attributes the reference's ``types.Reconstruction`` / ``pymap`` objects have
(``cameras``, ``shots``, ``points``, ``rig_cameras``, ``rig_instances``, ``biases``, ``reference``; shot ``.camera`` / ``.pose`` /
``.metadata.gps_position`` / ``.metadata.gps_accuracy``; landmark ``.coordinates`` / ``.reprojection_errors`` ...), so the object a
maintainer passes is the one they already hold; ``opensfm_amd.geometry_types`` has plain-Python stand-ins.  Everything here is host
glue that builds the ``BundleAdjuster`` calls in the reference's order; the numerics run in ``osfm_bundle_solve`` / ``osfm_ba_solve``.
"""
import time
from typing import Any, Dict, Iterable, List, Optional, Tuple

import numpy as np

from . import bundle as _bundle
from .geometry_types import optional_value

# config.py:default values of the keys BAHelpers::Bundle reads
DEFAULTS: Dict[str, Any] = {
    "optimize_camera_parameters": True, "bundle_analytic_derivatives": True, "align_method": "auto", "align_orientation_prior": "horizontal",
    "bundle_use_gps": True, "bundle_use_gcp": False, "bundle_compensate_gps_bias": False, "loss_function": "SoftLOneLoss",
    "loss_function_threshold": 1.0, "exif_focal_sd": 0.01, "aspect_ratio_sd": 0.01, "principal_point_sd": 0.01,
    "radial_distortion_k1_sd": 0.01, "radial_distortion_k2_sd": 0.01, "radial_distortion_k3_sd": 0.01, "radial_distortion_k4_sd": 0.01,
    "tangential_distortion_p1_sd": 0.01, "tangential_distortion_p2_sd": 0.01, "rig_translation_sd": 0.1, "rig_rotation_sd": 0.1,
    "processes": 1, "bundle_max_iterations": 100, "gcp_horizontal_sd": 0.01, "gcp_vertical_sd": 0.1, "gcp_global_weight": 0.01,
}


def _cfg(config: Optional[Dict[str, Any]], key: str):
    if config is not None and key in config:
        return config[key]
    return DEFAULTS[key]


def _shot_observations(shot) -> Iterable[Tuple[str, Any]]:
    """(landmark id, observation) of a shot: ``Shot::GetLandmarkObservations`` through whatever the object exposes"""
    if hasattr(shot, "get_landmark_observations"):
        for lm, obs in shot.get_landmark_observations().items():
            yield (lm if isinstance(lm, str) else lm.id), obs
    else:  # pymap.Shot
        for lm in shot.get_valid_landmarks():
            yield lm.id, shot.get_landmark_observation(lm)


def _bearing(camera, projection) -> np.ndarray:
    if hasattr(camera, "pixel_bearing"):
        return np.asarray(camera.pixel_bearing(projection), float)
    from .matching import pixel_bearing_many

    return pixel_bearing_many(camera, np.asarray(projection, float).reshape(1, 2))[0]


def _angle(u, v) -> float:
    """``geometry::AngleBetweenVectors`` (triangulation.cc:66-73)"""
    c = float(u @ v) / np.sqrt(float(u @ u) * float(v @ v))
    return 0.0 if abs(c) >= 1.0 else float(np.arccos(c))


def triangulate_bearings_midpoint(centers, bearings, thresholds, min_angle, min_depth) -> Tuple[bool, np.ndarray]:
    """``geometry::TriangulateBearingsMidpoint`` (triangulation.cc:139-178 with the closed form of triangulation.h:58-82)"""
    os_, bs = np.asarray(centers, float), np.asarray(bearings, float)
    n = len(os_)
    if len(thresholds) < n:
        return False, np.zeros(3)
    if not any(min_angle <= _angle(bs[i], bs[j]) <= np.pi - min_angle for i in range(n) for j in range(i)):
        return False, np.zeros(3)
    BBt = bs.T @ bs
    BBtA = sum(np.outer(b, b) @ o for b, o in zip(bs, os_))
    A = os_.sum(0)
    Cinv = np.linalg.inv(n * np.eye(3) - BBt)
    X = (np.eye(3) + BBt @ Cinv) @ A / n - Cinv @ BBtA
    for o, b, th in zip(os_, bs, thresholds):
        if _angle(X - o, b) > th or float((X - o) @ b) < min_depth:
            return False, np.zeros(3)
    return True, X


def triangulate_gcp(point, shots) -> Tuple[bool, np.ndarray]:
    """``BAHelpers::TriangulateGCP`` (ba_helpers.cc:313-345): midpoint of the control point's rays, 1 rad reprojection threshold"""
    os_, bs = [], []
    for obs in point.observations:
        shot = shots.get(obs.shot_id) if hasattr(shots, "get") else (shots[obs.shot_id] if obs.shot_id in shots else None)
        if shot is not None:
            pose = shot.pose
            bs.append(np.asarray(pose.get_R_cam_to_world(), float) @ _bearing(shot.camera, obs.projection))
            os_.append(np.asarray(pose.get_origin(), float))
    if len(os_) >= 2:
        return triangulate_bearings_midpoint(os_, bs, [1.0] * len(os_), 0.1 * np.pi / 180.0, 1e-3)
    return False, np.zeros(3)


def _lla_vec(point) -> np.ndarray:
    """``GroundControlPoint::GetLlaVec3d`` (map/ground_control_points.h:52-58)"""
    return np.array([point.lla["latitude"], point.lla["longitude"], point.lla["altitude"] if point.has_altitude else 0.0])


def _gps(shot) -> Tuple[Optional[np.ndarray], Optional[float]]:
    m = shot.metadata
    pos, acc = optional_value(m.gps_position), optional_value(m.gps_accuracy)
    return (None if pos is None else np.asarray(pos, float)), (None if acc is None else float(acc))


def alignment_constraints(reconstruction, config, gcp) -> Tuple[np.ndarray, np.ndarray]:
    """``BAHelpers::AlignmentConstraints`` (ba_helpers.cc:832-878): (Xp measured, X reconstructed)"""
    Xp, X = [], []
    if gcp and _cfg(config, "bundle_use_gcp"):
        for point in gcp:
            if not point.lla:
                continue
            ok, coords = triangulate_gcp(point, reconstruction.shots)
            if ok:
                Xp.append(reconstruction.reference.to_topocentric(*_lla_vec(point)))
                X.append(coords)
    if _cfg(config, "bundle_use_gps"):
        for shot in reconstruction.shots.values():
            pos, _ = _gps(shot)
            if pos is not None:
                Xp.append(pos)
                X.append(np.asarray(shot.pose.get_origin(), float))
    return np.array(Xp).reshape(-1, 3), np.array(X).reshape(-1, 3)


def detect_alignment_constraints(reconstruction, config, gcp) -> str:
    """``BAHelpers::DetectAlignmentConstraints`` (ba_helpers.cc:880-907): "orientation_prior" for < 3 or collinear constraints"""
    _, X = alignment_constraints(reconstruction, config, gcp)
    if len(X) < 3:
        return "orientation_prior"
    Xz = X - X.mean(0)
    evals = np.linalg.eigvalsh(Xz.T @ Xz)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = abs(evals[2] / evals[1])
    if int((evals < 1e-10).sum()) > 1 or ratio > 5e3:
        return "orientation_prior"
    return "naive"


def add_gcp_to_bundle(ba, reconstruction, gcp, config) -> int:
    """``BAHelpers::AddGCPToBundle`` (ba_helpers.cc:348-405)"""
    shots = reconstruction.shots
    dominant_terms = len(ba.get_rig_instances()) + ba.get_projections_count() + ba.get_relative_motions_count()
    total_terms = 0
    for point in gcp:
        ok, _ = triangulate_gcp(point, shots)
        if ok or point.lla:
            total_terms += 1
        total_terms += sum(1 for obs in point.observations if obs.shot_id in shots)
    global_weight = _cfg(config, "gcp_global_weight") * dominant_terms / max(1, total_terms)
    added = 0
    for point in gcp:
        point_id = "gcp-" + str(point.id)
        ok, coordinates = triangulate_gcp(point, shots)
        if not ok:
            if point.lla:
                coordinates = reconstruction.reference.to_topocentric(*_lla_vec(point))
            else:
                continue
        ba.add_point(point_id, coordinates, False)
        if point.lla:
            sd = np.array([_cfg(config, "gcp_horizontal_sd"), _cfg(config, "gcp_horizontal_sd"), _cfg(config, "gcp_vertical_sd")])
            ba.add_point_prior(point_id, reconstruction.reference.to_topocentric(*_lla_vec(point)), sd / global_weight, point.has_altitude)
        for obs in point.observations:
            if obs.shot_id in shots:
                ba.add_point_projection_observation(obs.shot_id, point_id, obs.projection, 0.001 / global_weight)
                added += 1
    return added


def bundle_to_map(ba, reconstruction, update_cameras: bool) -> None:
    """``BAHelpers::BundleToMap`` (ba_helpers.cc:764-830)"""
    from .geometry_types import camera_parameter_values, set_camera_parameter_values

    if update_cameras:
        for cam_id, cam in reconstruction.cameras.items():
            set_camera_parameter_values(cam, camera_parameter_values(ba.get_camera(cam_id)))
    biases = getattr(reconstruction, "biases", None)
    if biases is not None:
        for bias_id in list(biases.keys()):
            b = ba.get_bias(bias_id)
            if not np.all(np.isfinite(np.r_[b.rotation, b.translation, b.scale])):
                raise RuntimeError("Bias " + str(bias_id) + " has either NaN or INF values.")
            if hasattr(reconstruction, "set_bias"):
                reconstruction.set_bias(bias_id, b)
            else:
                biases[bias_id] = b
    for instance_id, instance in reconstruction.rig_instances.items():
        pose = ba.get_rig_instance_pose(instance_id)
        if not (np.all(np.isfinite(pose.rotation)) and np.all(np.isfinite(pose.translation))):
            raise RuntimeError("Rig Instance " + str(instance_id) + " has either NaN or INF values.")
        instance.pose = pose
    for rig_camera_id, rig_camera in reconstruction.rig_cameras.items():
        pose = ba.get_rig_camera_pose(rig_camera_id)
        if not (np.all(np.isfinite(pose.rotation)) and np.all(np.isfinite(pose.translation))):
            raise RuntimeError("Rig Camera " + str(rig_camera_id) + " has either NaN or INF values.")
        rig_camera.pose = pose
    for point_id, point in reconstruction.points.items():
        pt = ba.get_point(point_id)
        if not np.all(np.isfinite(pt.p)):
            raise RuntimeError("Point " + str(point_id) + " has either NaN or INF values.")
        point.coordinates = pt.p
        point.reprojection_errors = pt.reprojection_errors


def bundle(reconstruction, camera_priors: Dict[str, Any], rig_camera_priors: Dict[str, Any], gcp: Optional[List[Any]],
           config: Optional[Dict[str, Any]] = None, adjuster=None) -> Dict[str, Any]:
    """``pysfm.BAHelpers.bundle(map, camera_priors, rig_camera_priors, gcp, config)``: global bundle adjustment of a reconstruction in
    place; returns the report dict of ``ba_helpers.cc:743-762``.  ``adjuster`` is for tests (a pre-built ``BundleAdjuster``)."""
    start = time.perf_counter()
    gcp = list(gcp or [])
    ba = adjuster or _bundle.BundleAdjuster()
    fix_cameras = not _cfg(config, "optimize_camera_parameters")
    ba.set_use_analytic_derivatives(bool(_cfg(config, "bundle_analytic_derivatives")))
    all_cameras = reconstruction.cameras
    for cam_id, cam in all_cameras.items():
        ba.add_camera(cam_id, cam, camera_priors[cam_id], fix_cameras)
    for pt_id, pt in reconstruction.points.items():
        ba.add_point(pt_id, pt.coordinates, False)

    align_method = _cfg(config, "align_method")
    if align_method == "auto":
        align_method = detect_alignment_constraints(reconstruction, config, gcp)
    up_vector = None
    if align_method == "orientation_prior":
        prior = _cfg(config, "align_orientation_prior")
        if prior == "vertical":
            up_vector = np.array([0.0, 0.0, -1.0])
        elif prior == "horizontal":
            up_vector = np.array([0.0, -1.0, 0.0])

    n_rig_cameras = len(reconstruction.rig_cameras)
    shots_per_rig_camera = len(reconstruction.shots) // n_rig_cameras if n_rig_cameras > 0 else 1
    lock_rig_camera = shots_per_rig_camera <= 10  # kMinRigInstanceForAdjust
    for rig_camera_id, rig_camera in reconstruction.rig_cameras.items():
        is_leverarm = rig_camera_id in all_cameras
        ba.add_rig_camera(rig_camera_id, rig_camera.pose, rig_camera_priors[rig_camera_id].pose, is_leverarm or lock_rig_camera)

    use_gps = bool(_cfg(config, "bundle_use_gps"))
    for rig_instance_id, instance in reconstruction.rig_instances.items():
        average_position, average_std, gps_count = np.zeros(3), 0.0, 0
        shot_cameras, shot_rig_cameras = {}, {}
        for shot_id, rig_camera_id in instance.rig_camera_ids.items():
            shot = reconstruction.shots[shot_id]
            shot_cameras[shot_id] = shot.camera.id
            shot_rig_cameras[shot_id] = rig_camera_id
            if use_gps:
                pos, acc = _gps(shot)
                if pos is not None and acc is not None:
                    if acc <= 0:
                        raise RuntimeError("Shot " + str(shot_id) + " has an accuracy <= 0: " + str(acc) +
                                           ". Try modifying your input parser to filter such values.")
                    average_position += pos
                    average_std += acc
                    gps_count += 1
        ba.add_rig_instance(rig_instance_id, instance.pose, shot_cameras, shot_rig_cameras, False)
        if use_gps and gps_count > 0:
            ba.add_rig_instance_position_prior(rig_instance_id, average_position / gps_count, np.full(3, average_std / gps_count), "dummy")

    added_reprojections = 0
    for shot_id, shot in reconstruction.shots.items():
        if up_vector is not None:
            ba.add_absolute_up_vector(shot_id, up_vector, 1e-3)
        for lm_id, obs in _shot_observations(shot):
            ba.add_point_projection_observation(shot_id, lm_id, obs.point, obs.scale, optional_value(getattr(obs, "depth_prior", None)))
            added_reprojections += 1

    if _cfg(config, "bundle_use_gcp") and gcp:
        add_gcp_to_bundle(ba, reconstruction, gcp, config)
    if _cfg(config, "bundle_compensate_gps_bias"):
        for camera_id in all_cameras.keys():
            ba.set_camera_bias(camera_id, reconstruction.biases[camera_id])

    ba.set_point_projection_loss_function(_cfg(config, "loss_function"), _cfg(config, "loss_function_threshold"))
    ba.set_internal_parameters_prior_sd(
        _cfg(config, "exif_focal_sd"), _cfg(config, "aspect_ratio_sd"), _cfg(config, "principal_point_sd"),
        _cfg(config, "radial_distortion_k1_sd"), _cfg(config, "radial_distortion_k2_sd"), _cfg(config, "tangential_distortion_p1_sd"),
        _cfg(config, "tangential_distortion_p2_sd"), _cfg(config, "radial_distortion_k3_sd"), _cfg(config, "radial_distortion_k4_sd"))
    ba.set_rig_parameters_prior_sd(_cfg(config, "rig_translation_sd"), _cfg(config, "rig_rotation_sd"))
    ba.set_num_threads(int(_cfg(config, "processes")))
    ba.set_max_num_iterations(int(_cfg(config, "bundle_max_iterations")))
    ba.set_linear_solver_type("SPARSE_SCHUR")
    timer_setup = time.perf_counter()
    ba.run()
    timer_run = time.perf_counter()
    bundle_to_map(ba, reconstruction, not fix_cameras)
    timer_teardown = time.perf_counter()
    return {
        "brief_report": ba.brief_report(),
        "wall_times": {"setup": timer_setup - start, "run": timer_run - timer_setup, "teardown": timer_teardown - timer_run},
        "num_images": len(reconstruction.shots), "num_points": len(reconstruction.points), "num_reprojections": added_reprojections,
    }


# ------------------------------------------------------------------------------------------------
# local / pose-only bundle adjustment over reconstruction objects (reconstruction.py:89-149 calls these)
# ------------------------------------------------------------------------------------------------
DEFAULTS.update({"local_bundle_radius": 3, "local_bundle_min_common_points": 20, "local_bundle_max_shots": 30})  # config.py:296-300


def _instance_id(shot) -> str:
    return shot.rig_instance_id if hasattr(shot, "rig_instance_id") else shot.rig_instance.id


def _rig_camera_id(shot) -> str:
    return shot.rig_camera_id if hasattr(shot, "rig_camera_id") else shot.rig_camera.id


def _instance_shot_ids(reconstruction, shot) -> List[str]:
    """``RigInstance::GetShotIDs`` of the instance a shot belongs to"""
    inst = reconstruction.rig_instances[_instance_id(shot)]
    if hasattr(inst, "rig_camera_ids"):
        return list(inst.rig_camera_ids.keys())
    return list(inst.shots.keys())


def _landmark_shots(reconstruction) -> Dict[str, List[str]]:
    """landmark id -> the shots observing it (``Landmark::GetObservations``), from the shots' side of the map"""
    out: Dict[str, List[str]] = {}
    for shot_id, shot in reconstruction.shots.items():
        for lm_id, _ in _shot_observations(shot):
            out.setdefault(lm_id, []).append(shot_id)
    return out


def _direct_shot_neighbors(reconstruction, shot_ids: set, min_common_points: int, max_neighbors: int, lm_shots: Dict[str, List[str]]) -> set:
    """``BAHelpers::DirectShotNeighbors`` (ba_helpers.cc:68-115): the shots outside ``shot_ids`` ranked by the number of points they share
    with it, the first ``max_neighbors`` with at least ``min_common_points`` -- each with every shot of its rig instance.  (The
    reference sorts the entries of an unordered_map: equal counts come out in an unspecified order there; here ties go to the
    smaller shot id.)"""
    points = set()
    for sid in shot_ids:
        for lm_id, _ in _shot_observations(reconstruction.shots[sid]):
            points.add(lm_id)
    common: Dict[str, int] = {}
    for lm_id in points:
        for sid in lm_shots.get(lm_id, ()):
            if sid not in shot_ids:
                common[sid] = common.get(sid, 0) + 1
    pairs = sorted(common.items(), key=lambda kv: (-kv[1], kv[0]))
    max_n = min(max_neighbors, len(pairs))
    neighbors = set()
    for idx, (sid, n) in enumerate(pairs):
        if n >= min_common_points and idx < max_n:
            neighbors.update(_instance_shot_ids(reconstruction, reconstruction.shots[sid]))
        else:
            break
    return neighbors


def shot_neighborhood_ids(reconstruction, central_shot_id: str, radius: int, min_common_points: int, max_interior_size: int,
                          lm_shots: Optional[Dict[str, List[str]]] = None) -> Tuple[set, set]:
    """``pysfm.BAHelpers.shot_neighborhood_ids`` (ba_helpers.cc:17-66): (interior, boundary) shot ids around a shot -- the central shot (and
    its rig instance) at distance 0, shots at distance n + 1 share at least ``min_common_points`` points with those at distance n, up to
    ``radius`` / ``max_interior_size``; the boundary = everything else sharing a point with the interior"""
    lm_shots = lm_shots if lm_shots is not None else _landmark_shots(reconstruction)
    central = reconstruction.shots[central_shot_id]
    interior = set(_instance_shot_ids(reconstruction, central))
    interior.add(central_shot_id)
    distance = 1
    while distance < radius and len(interior) < max_interior_size:
        interior |= _direct_shot_neighbors(reconstruction, interior, min_common_points, max_interior_size - len(interior), lm_shots)
        distance += 1
    boundary = _direct_shot_neighbors(reconstruction, interior, 1, 1000000, lm_shots)
    return interior, boundary


def _set_internal_priors_and_loss(ba, config) -> None:
    ba.set_point_projection_loss_function(_cfg(config, "loss_function"), _cfg(config, "loss_function_threshold"))
    ba.set_internal_parameters_prior_sd(
        _cfg(config, "exif_focal_sd"), _cfg(config, "aspect_ratio_sd"), _cfg(config, "principal_point_sd"),
        _cfg(config, "radial_distortion_k1_sd"), _cfg(config, "radial_distortion_k2_sd"), _cfg(config, "tangential_distortion_p1_sd"),
        _cfg(config, "tangential_distortion_p2_sd"), _cfg(config, "radial_distortion_k3_sd"), _cfg(config, "radial_distortion_k4_sd"))
    ba.set_rig_parameters_prior_sd(_cfg(config, "rig_translation_sd"), _cfg(config, "rig_rotation_sd"))
    ba.set_num_threads(int(_cfg(config, "processes")))


def _add_instances(ba, reconstruction, rig_instance_ids, is_free_shot, config) -> None:
    """the rig-instance loop BundleLocal and BundleShotPoses share (ba_helpers.cc:176-224, 467-511): an instance is constant as soon as
    one of its shots is not to be optimised; moving instances get the average GPS position / accuracy of their free shots as a prior.

    Known divergence, on purpose: BundleShotPoses (ba_helpers.cc:467-511) calls AddRigInstance and AddRigInstancePositionPrior INSIDE
    its loop over the instance's shots, dividing the running sums by the running count each time -- for an instance with three or
    more free shots (or free and fixed shots mixed) the prior it ends with depends on the iteration order of an unordered_map.
    BundleLocal (:176-224) adds once, after the loop, which is what this function does for both; instances of one or two shots
    (every data set without a multi-camera rig) get identical priors either way."""
    use_gps = bool(_cfg(config, "bundle_use_gps"))
    for rig_instance_id in rig_instance_ids:
        instance = reconstruction.rig_instances[rig_instance_id]
        shot_cameras, shot_rig_cameras = {}, {}
        average_position, average_std, gps_count, fix_instance = np.zeros(3), 0.0, 0, False
        for shot_id, rig_camera_id in instance.rig_camera_ids.items():
            shot = reconstruction.shots[shot_id]
            shot_cameras[shot_id] = shot.camera.id
            shot_rig_cameras[shot_id] = rig_camera_id
            if is_free_shot(shot_id):
                pos, acc = _gps(shot)
                if use_gps and pos is not None and acc is not None:
                    average_position += pos
                    average_std += acc
                    gps_count += 1
            else:
                fix_instance = True
        ba.add_rig_instance(rig_instance_id, instance.pose, shot_cameras, shot_rig_cameras, fix_instance)
        if not fix_instance and gps_count > 0:
            ba.add_rig_instance_position_prior(rig_instance_id, average_position / gps_count, np.full(3, average_std / gps_count), "dummy")


def bundle_local(reconstruction, camera_priors: Dict[str, Any], rig_camera_priors: Dict[str, Any], gcp: Optional[List[Any]], central_shot_id: str,
                 config: Optional[Dict[str, Any]] = None) -> Tuple[List[str], Dict[str, Any]]:
    """``pysfm.BAHelpers.bundle_local`` (ba_helpers.cc:117-311; ``reconstruction.bundle_local``, reconstruction.py:107-127): ten LM
    iterations over the rig instances of the interior of a shot's neighbourhood and every point they see, the boundary shots and all
    cameras / rig cameras constant; updates the reconstruction in place and returns (ids of the adjusted points, report)"""
    start = time.perf_counter()
    gcp = list(gcp or [])
    lm_shots = _landmark_shots(reconstruction)
    interior, boundary = shot_neighborhood_ids(reconstruction, central_shot_id, int(_cfg(config, "local_bundle_radius")),
                                               int(_cfg(config, "local_bundle_min_common_points")), int(_cfg(config, "local_bundle_max_shots")), lm_shots)
    ba = _bundle.BundleAdjuster()
    ba.set_use_analytic_derivatives(bool(_cfg(config, "bundle_analytic_derivatives")))
    for cam_id, cam in reconstruction.cameras.items():
        ba.add_camera(cam_id, cam, camera_priors[cam_id], True)
    int_and_bound = sorted(interior | boundary)
    rig_camera_ids = sorted({_rig_camera_id(reconstruction.shots[s]) for s in int_and_bound})
    rig_instance_ids = sorted({_instance_id(reconstruction.shots[s]) for s in int_and_bound})
    for rig_camera_id in rig_camera_ids:
        ba.add_rig_camera(rig_camera_id, reconstruction.rig_cameras[rig_camera_id].pose, rig_camera_priors[rig_camera_id].pose, True)
    _add_instances(ba, reconstruction, rig_instance_ids, lambda sid: sid not in boundary, config)
    pt_ids: List[str] = []
    points = set()
    added_reprojections = 0
    for shot_id in sorted(interior):
        for lm_id, obs in _shot_observations(reconstruction.shots[shot_id]):
            if lm_id not in points:
                points.add(lm_id)
                pt_ids.append(lm_id)
                ba.add_point(lm_id, reconstruction.points[lm_id].coordinates, False)
            ba.add_point_projection_observation(shot_id, lm_id, obs.point, obs.scale, optional_value(getattr(obs, "depth_prior", None)))
            added_reprojections += 1
    for shot_id in sorted(boundary):
        for lm_id, obs in _shot_observations(reconstruction.shots[shot_id]):
            if lm_id in points:
                ba.add_point_projection_observation(shot_id, lm_id, obs.point, obs.scale, optional_value(getattr(obs, "depth_prior", None)))
                added_reprojections += 1
    if _cfg(config, "bundle_use_gcp") and gcp:
        add_gcp_to_bundle(ba, reconstruction, gcp, config)
    _set_internal_priors_and_loss(ba, config)
    ba.set_max_num_iterations(10)
    ba.set_linear_solver_type("DENSE_SCHUR")
    timer_setup = time.perf_counter()
    ba.run()
    timer_run = time.perf_counter()
    for rig_instance_id in rig_instance_ids:
        reconstruction.rig_instances[rig_instance_id].pose = ba.get_rig_instance_pose(rig_instance_id)
    for lm_id in pt_ids:
        pt = ba.get_point(lm_id)
        reconstruction.points[lm_id].coordinates = pt.p
        reconstruction.points[lm_id].reprojection_errors = pt.reprojection_errors
    timer_teardown = time.perf_counter()
    report = {"brief_report": ba.brief_report(),
              "wall_times": {"setup": timer_setup - start, "run": timer_run - timer_setup, "teardown": timer_teardown - timer_run},
              "num_images": len(interior), "num_interior_images": len(interior), "num_boundary_images": len(boundary),
              "num_other_images": len(reconstruction.shots) - len(interior) - len(boundary), "num_points": len(pt_ids),
              "num_reprojections": added_reprojections}
    return pt_ids, report


def bundle_shot_poses(reconstruction, shot_ids, camera_priors: Dict[str, Any], rig_camera_priors: Dict[str, Any],
                      config: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """``pysfm.BAHelpers.bundle_shot_poses`` (ba_helpers.cc:408-579; ``reconstruction.bundle_shot_poses``, reconstruction.py:89-104):
    ten LM iterations over the poses of the rig instances of ``shot_ids`` only -- cameras, rig cameras and every point they see
    constant (resection refinement of a newly added image); updates the reconstruction in place and returns the report"""
    start = time.perf_counter()
    shot_ids = set(shot_ids)
    ba = _bundle.BundleAdjuster()
    ba.set_use_analytic_derivatives(bool(_cfg(config, "bundle_analytic_derivatives")))
    rig_instance_ids = sorted({_instance_id(reconstruction.shots[s]) for s in shot_ids})
    rig_camera_ids, camera_ids = set(), set()
    for rig_instance_id in rig_instance_ids:
        for shot_id, rig_camera_id in reconstruction.rig_instances[rig_instance_id].rig_camera_ids.items():
            rig_camera_ids.add(rig_camera_id)
            camera_ids.add(reconstruction.shots[shot_id].camera.id)
    for rig_camera_id in sorted(rig_camera_ids):
        ba.add_rig_camera(rig_camera_id, reconstruction.rig_cameras[rig_camera_id].pose, rig_camera_priors[rig_camera_id].pose, True)
    for camera_id in sorted(camera_ids):
        ba.add_camera(camera_id, reconstruction.cameras[camera_id], camera_priors[camera_id], True)
    landmarks = set()
    for shot_id in sorted(shot_ids):
        for lm_id, _ in _shot_observations(reconstruction.shots[shot_id]):
            if lm_id not in landmarks:
                landmarks.add(lm_id)
                ba.add_point(lm_id, reconstruction.points[lm_id].coordinates, True)
    _add_instances(ba, reconstruction, rig_instance_ids, lambda sid: sid in shot_ids, config)
    for shot_id in sorted(shot_ids):
        for lm_id, obs in _shot_observations(reconstruction.shots[shot_id]):
            ba.add_point_projection_observation(shot_id, lm_id, obs.point, obs.scale, optional_value(getattr(obs, "depth_prior", None)))
    _set_internal_priors_and_loss(ba, config)
    ba.set_max_num_iterations(10)
    ba.set_linear_solver_type("DENSE_QR")
    timer_setup = time.perf_counter()
    ba.run()
    timer_run = time.perf_counter()
    for rig_instance_id in rig_instance_ids:
        reconstruction.rig_instances[rig_instance_id].pose = ba.get_rig_instance_pose(rig_instance_id)
    timer_teardown = time.perf_counter()
    return {"brief_report": ba.brief_report(),
            "wall_times": {"setup": timer_setup - start, "run": timer_run - timer_setup, "teardown": timer_teardown - timer_run}}
