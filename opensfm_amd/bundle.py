"""Host-side mirror of the reference's bundle-adjustment interface, backed by the HIP solver.

================================================  ==================================================
this module                                       reference
================================================  ==================================================
``bundle_arrays(problem, config)``                flat-array form of ``pysfm.BAHelpers.bundle``
                                                  (``sfm/src/ba_helpers.cc:581-763``)
``BundleAdjuster`` (subset, same method names)    ``pybundle.BundleAdjuster``
                                                  (``bundle/python/pybind.cc:45-117``)
``bundle(reconstruction_like, ...)``              ``opensfm/reconstruction.py:69-86``
================================================  ==================================================

Only what ``BAHelpers::Bundle`` adds on a plain (single rig-camera, perspective) reconstruction is
implemented: cameras with intrinsics priors, shots (rig instances) with optional position priors,
points, reprojection observations with a shared robust loss.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Any, Dict, List, Optional

import numpy as np

from . import _lib
from ._ba_abi import BaOptions, BaProblem, BaReport, fill_bundle_problem
from ._lib import check, default_context

LOSSES = {"TrivialLoss": 0, "SoftLOneLoss": 1, "HuberLoss": 2, "CauchyLoss": 3}
TERMINATION = {0: "NO_CONVERGENCE (max iterations)", 1: "CONVERGENCE (function tolerance)", 2: "CONVERGENCE (gradient tolerance)",
               3: "CONVERGENCE (parameter tolerance)", 4: "CONVERGENCE (min trust region radius)", -1: "FAILURE"}

DEFAULT_CONFIG: Dict[str, Any] = {
    # opensfm/config.py:241-263,283
    "loss_function": "SoftLOneLoss",
    "loss_function_threshold": 1.0,
    "bundle_max_iterations": 100,
    "exif_focal_sd": 0.01,
    "radial_distortion_k1_sd": 0.01,
    "radial_distortion_k2_sd": 0.01,
    "optimize_camera_parameters": True,
    "bundle_use_gps": True,
    # opensfm/config.py:296-300
    "local_bundle_radius": 3,
    "local_bundle_min_common_points": 20,
    "local_bundle_max_shots": 30,
}


def _cfg(config, key):
    return config[key] if config is not None and key in config else DEFAULT_CONFIG[key]


def _dp(a, t):
    return a.ctypes.data_as(C.POINTER(t))


BA_TIME_MATVEC = 2  # options.verbose bit: time ten Schur mat-vecs after the solve (report['ms_per_matvec']; the bench's roofline block)


def make_options(config: Optional[Dict[str, Any]] = None, **overrides) -> BaOptions:
    o = BaOptions()
    _lib.load().osfm_ba_options_default(C.byref(o))
    name = _cfg(config, "loss_function")
    if name not in LOSSES:
        # bundle_adjuster.cc:427
        raise RuntimeError("ceres::LossFunction with name " + str(name) + " not found.")
    o.loss = LOSSES[name]
    o.loss_threshold = float(_cfg(config, "loss_function_threshold"))
    o.max_iterations = int(_cfg(config, "bundle_max_iterations"))
    for k, v in overrides.items():
        setattr(o, k, v)
    return o


def bundle_arrays(problem: Dict[str, np.ndarray], config: Optional[Dict[str, Any]] = None, ctx=None, **overrides) -> Dict[str, Any]:
    """Global BA over flat arrays (layout: ``opensfm_amd.synthetic.make_ba_scene``).  Inputs are not
    modified; returns the optimised ``cam_params``/``shot_pose``/``points``, per-observation
    ``reproj_err`` (sigma 1, ``bundle_adjuster.cc:1196-1208``) and a report whose keys mirror
    ``ba_helpers.cc:743-762``."""
    ctx = ctx or default_context()
    lib = _lib.load()
    t0 = time.perf_counter()
    f64 = lambda k: np.ascontiguousarray(problem[k], np.float64).copy()
    cams, poses, pts = f64("cam_params").reshape(-1, 3), f64("shot_pose").reshape(-1, 6), f64("points").reshape(-1, 3)
    nc = len(cams)
    cam_prior = np.ascontiguousarray(problem.get("cam_prior", cams), np.float64)
    if "cam_sigma" in problem:
        cam_sigma = np.ascontiguousarray(problem["cam_sigma"], np.float64)
    else:  # SetInternalParametersPriorSD (ba_helpers.cc:716-726): k1, k2, focal
        cam_sigma = np.tile([_cfg(config, "radial_distortion_k1_sd"), _cfg(config, "radial_distortion_k2_sd"),
                             _cfg(config, "exif_focal_sd")], (nc, 1)).astype(np.float64)
    if "cam_fixed" in problem:
        cam_fixed = np.ascontiguousarray(problem["cam_fixed"], np.uint8)
    else:  # ba_helpers.cc:595-599: fix = !optimize_camera_parameters
        cam_fixed = np.full(nc, 0 if _cfg(config, "optimize_camera_parameters") else 1, np.uint8)
    shot_camera = np.ascontiguousarray(problem["shot_camera"], np.int32)
    obs_shot = np.ascontiguousarray(problem["obs_shot"], np.int32)
    obs_point = np.ascontiguousarray(problem["obs_point"], np.int32)
    obs_xy = np.ascontiguousarray(problem["obs_xy"], np.float64)
    obs_sigma = np.ascontiguousarray(problem["obs_sigma"], np.float64)
    reproj = np.zeros((len(obs_shot), 2), np.float64)
    keep: List[np.ndarray] = []
    P = BaProblem()
    P.n_cameras, P.n_shots, P.n_points, P.n_obs = nc, len(poses), len(pts), len(obs_shot)
    P.cam_params, P.cam_prior, P.cam_sigma = _dp(cams, C.c_double), _dp(cam_prior, C.c_double), _dp(cam_sigma, C.c_double)
    P.cam_fixed = _dp(cam_fixed, C.c_uint8)
    P.shot_pose, P.shot_camera = _dp(poses, C.c_double), _dp(shot_camera, C.c_int32)
    use_gps = bool(_cfg(config, "bundle_use_gps"))
    for key, t, ct in (("shot_fixed", np.uint8, C.c_uint8), ("point_fixed", np.uint8, C.c_uint8),
                       ("shot_gps", np.float64, C.c_double), ("shot_gps_sigma", np.float64, C.c_double),
                       ("shot_up", np.float64, C.c_double), ("shot_up_sigma", np.float64, C.c_double),
                       ("cam_model", np.int32, C.c_int32), ("cam_ext", np.float64, C.c_double)):
        if problem.get(key) is not None and (use_gps or not key.startswith("shot_gps")):
            arr = np.ascontiguousarray(problem[key], t)
            keep.append(arr)
            setattr(P, key, _dp(arr, ct))
    P.points = _dp(pts, C.c_double)
    P.obs_shot, P.obs_point = _dp(obs_shot, C.c_int32), _dp(obs_point, C.c_int32)
    P.obs_xy, P.obs_sigma, P.reproj_err = _dp(obs_xy, C.c_double), _dp(obs_sigma, C.c_double), _dp(reproj, C.c_double)
    O = make_options(config, **overrides)
    R = BaReport()
    t1 = time.perf_counter()
    check(lib.osfm_ba_solve(ctx.handle, C.byref(P), C.byref(O), C.byref(R)), "osfm_ba_solve")
    t2 = time.perf_counter()
    brief = ("osfm-mi355 LM: iterations %d (successful %d), initial cost %.6e, final cost %.6e, termination: %s, "
             "Schur-PCG iterations %d" % (R.iterations, R.successful_steps, R.initial_cost, R.final_cost,
                                          TERMINATION.get(R.termination, str(R.termination)), R.pcg_iterations_total))
    return {
        "cam_params": cams, "shot_pose": poses, "points": pts, "reproj_err": reproj,
        "iterations": R.iterations, "successful_steps": R.successful_steps, "termination": R.termination,
        "initial_cost": R.initial_cost, "final_cost": R.final_cost,
        "rmse_initial": R.rmse_normalized_initial, "rmse_final": R.rmse_normalized_final,
        "cost_history": np.array(R.cost_history[: min(R.iterations, 255) + 1]),
        "pcg_iterations": int(R.pcg_iterations_total),
        "shot_bandwidth": int(R.shot_bandwidth), "preconditioner_bandwidth": int(R.preconditioner_bandwidth),
        "shots_reordered": bool(R.shots_reordered), "shot_bandwidth_input": int(R.shot_bandwidth_input),
        "seconds_solver": R.seconds_total, "seconds_linear_solver": R.seconds_linear_solver,
        "seconds_setup": R.seconds_setup, "seconds_run": R.seconds_run, "seconds_teardown": R.seconds_teardown,
        "ms_per_matvec": (R.ms_matvec_total / R.matvec_calls) if R.matvec_calls else None,
        # report dict of BAHelpers::Bundle (ba_helpers.cc:743-762)
        "brief_report": brief,
        "wall_times": {"setup": (t1 - t0) + R.seconds_setup, "run": R.seconds_run, "teardown": R.seconds_teardown},
        "num_images": len(poses), "num_points": len(pts), "num_reprojections": len(obs_shot),
    }


def bundle_general_arrays(problem: Dict[str, Any], config: Optional[Dict[str, Any]] = None, ctx=None, **overrides) -> Dict[str, Any]:
    """The general bundle adjustment over flat arrays (``osfm_bundle_solve``; field names and shapes: ``_ba_abi.BUNDLE_FIELDS`` =
    ``osfm_bundle_problem`` of ``include/osfm_mi355.h``): every camera model with free or constant intrinsics, rig cameras, rig
    instances, GPS priors through per-camera biases, point priors (ground control points), up vectors, depth priors -- on the
    streaming Schur solver (its generic mode, ``csrc/ba_generic.inc``): linear in the observations, no dense reduced system.
    Inputs are not modified; returns the optimised parameter arrays, ``reproj_err`` (n_obs x 3, sigma 1) and the report."""
    ctx = ctx or default_context()
    lib = _lib.load()
    P, arr = fill_bundle_problem(problem)
    O = make_options(config, **overrides)
    R = BaReport()
    check(lib.osfm_bundle_solve(ctx.handle, C.byref(P), C.byref(O), C.byref(R)), "osfm_bundle_solve")
    out = {k: arr[k] for k in ("cam_params", "rig_camera_pose", "rig_instance_pose", "points") if k in arr}
    if "bias" in arr:
        out["bias"] = arr["bias"]
    out.update({
        "reproj_err": arr["reproj_err"][: int(P.n_obs)],
        "iterations": R.iterations, "successful_steps": R.successful_steps, "termination": R.termination,
        "initial_cost": R.initial_cost, "final_cost": R.final_cost,
        "cost_history": np.array(R.cost_history[: min(R.iterations, 255) + 1]),
        "seconds_setup": R.seconds_setup, "seconds_run": R.seconds_run, "seconds_teardown": R.seconds_teardown,
        "seconds_linear_solver": R.seconds_linear_solver, "seconds_solver": R.seconds_total,
        "pcg_iterations": int(R.pcg_iterations_total), "ms_per_matvec": (R.ms_matvec_total / R.matvec_calls) if R.matvec_calls else None,
        "shot_bandwidth": int(R.shot_bandwidth), "preconditioner_bandwidth": int(R.preconditioner_bandwidth),
        "shots_reordered": int(R.shots_reordered), "shot_bandwidth_input": int(R.shot_bandwidth_input),
        "brief_report": "osfm-mi355 LM (streaming Schur, generic rows): iterations %d (successful %d), initial cost %.6e, final cost %.6e, termination: %s, "
                        "pcg iterations %d"
                        % (R.iterations, R.successful_steps, R.initial_cost, R.final_cost, TERMINATION.get(R.termination, str(R.termination)),
                           R.pcg_iterations_total),
        "wall_times": {"setup": R.seconds_setup, "run": R.seconds_run, "teardown": R.seconds_teardown},
        "num_images": int(P.n_shots), "num_points": int(P.n_points), "num_reprojections": int(P.n_obs),
    })
    return out


# ------------------------------------------------------------------------------------------------
# local bundle adjustment / pose-only bundle adjustment
# ------------------------------------------------------------------------------------------------
def _direct_shot_neighbors(obs_shot, obs_point, n_shots, n_points, inside: np.ndarray, min_common: int, max_neighbors: int):
    """``BAHelpers::DirectShotNeighbors`` (ba_helpers.cc:68-115): shots outside ``inside`` ranked by
    the number of points they share with it (the reference sorts an unordered_map, so ties come out
    in an unspecified order there; here ties go to the lower shot index)."""
    pt_in = np.zeros(n_points, bool)
    pt_in[obs_point[inside[obs_shot]]] = True
    sel = pt_in[obs_point] & ~inside[obs_shot]
    common = np.bincount(obs_shot[sel], minlength=n_shots)
    order = np.argsort(-common, kind="stable")
    order = order[common[order] >= max(1, min_common)][:max_neighbors]
    out = np.zeros(n_shots, bool)
    out[order] = True
    return out


def shot_neighborhood(problem: Dict[str, np.ndarray], central_shot: int, radius: int, min_common_points: int,
                      max_interior_size: int):
    """``BAHelpers::ShotNeighborhood`` (ba_helpers.cc:36-66) on flat arrays -> (interior, boundary)
    boolean masks over the shots (one shot per rig instance)."""
    obs_shot = np.asarray(problem["obs_shot"], np.int64)
    obs_point = np.asarray(problem["obs_point"], np.int64)
    n_shots, n_points = len(problem["shot_pose"]), len(problem["points"])
    interior = np.zeros(n_shots, bool)
    interior[central_shot] = True
    distance = 1
    while distance < radius and interior.sum() < max_interior_size:
        remaining = max_interior_size - int(interior.sum())
        interior |= _direct_shot_neighbors(obs_shot, obs_point, n_shots, n_points, interior, min_common_points, remaining)
        distance += 1
    boundary = _direct_shot_neighbors(obs_shot, obs_point, n_shots, n_points, interior, 1, 1000000)
    return interior, boundary


def _sub_problem(problem, shots_free: np.ndarray, shots_fixed: np.ndarray, point_mask: np.ndarray, obs_mask: np.ndarray,
                 points_fixed: bool, use_gps: bool):
    shot_ids = np.flatnonzero(shots_free | shots_fixed)
    pt_ids = np.flatnonzero(point_mask)
    smap = -np.ones(len(problem["shot_pose"]), np.int64)
    smap[shot_ids] = np.arange(len(shot_ids))
    pmap = -np.ones(len(problem["points"]), np.int64)
    pmap[pt_ids] = np.arange(len(pt_ids))
    sub = {
        "cam_params": np.asarray(problem["cam_params"], np.float64),
        "cam_prior": np.asarray(problem.get("cam_prior", problem["cam_params"]), np.float64),
        "cam_fixed": np.ones(len(problem["cam_params"]), np.uint8),  # constexpr bool fix_cameras{true}
        "shot_pose": np.asarray(problem["shot_pose"], np.float64)[shot_ids],
        "shot_camera": np.asarray(problem["shot_camera"], np.int32)[shot_ids],
        "shot_fixed": shots_fixed[shot_ids].astype(np.uint8),
        "points": np.asarray(problem["points"], np.float64)[pt_ids],
        "point_fixed": np.full(len(pt_ids), 1 if points_fixed else 0, np.uint8),
        "obs_shot": smap[np.asarray(problem["obs_shot"], np.int64)[obs_mask]].astype(np.int32),
        "obs_point": pmap[np.asarray(problem["obs_point"], np.int64)[obs_mask]].astype(np.int32),
        "obs_xy": np.asarray(problem["obs_xy"], np.float64)[obs_mask],
        "obs_sigma": np.asarray(problem["obs_sigma"], np.float64)[obs_mask],
    }
    for key in ("cam_sigma", "cam_model", "cam_ext"):
        if key in problem:
            sub[key] = problem[key]
    if use_gps and problem.get("shot_gps") is not None and problem.get("shot_gps_sigma") is not None:
        sub["shot_gps"] = np.asarray(problem["shot_gps"], np.float64)[shot_ids]
        sub["shot_gps_sigma"] = np.where(shots_free[shot_ids], np.asarray(problem["shot_gps_sigma"], np.float64)[shot_ids], 0.0)
    return sub, shot_ids, pt_ids


def local_problem(problem: Dict[str, np.ndarray], central_shot: int, config: Optional[Dict[str, Any]] = None):
    """The problem ``BAHelpers::BundleLocal`` builds (ba_helpers.cc:117-222): interior shots free,
    boundary shots and every camera constant, the points seen from the interior free, all their
    observations from interior and boundary shots, position priors on interior shots only."""
    interior, boundary = shot_neighborhood(problem, central_shot, int(_cfg(config, "local_bundle_radius")),
                                           int(_cfg(config, "local_bundle_min_common_points")),
                                           int(_cfg(config, "local_bundle_max_shots")))
    obs_shot = np.asarray(problem["obs_shot"], np.int64)
    obs_point = np.asarray(problem["obs_point"], np.int64)
    point_mask = np.zeros(len(problem["points"]), bool)
    point_mask[obs_point[interior[obs_shot]]] = True
    obs_mask = interior[obs_shot] | (boundary[obs_shot] & point_mask[obs_point])
    sub, shot_ids, pt_ids = _sub_problem(problem, interior, boundary, point_mask, obs_mask, False, bool(_cfg(config, "bundle_use_gps")))
    return sub, shot_ids, pt_ids, interior, boundary


def bundle_local_arrays(problem: Dict[str, np.ndarray], central_shot: int, config: Optional[Dict[str, Any]] = None, ctx=None,
                        **overrides):
    """``pysfm.BAHelpers.bundle_local`` (ba_helpers.cc:117-311) on flat arrays: 10 LM iterations over the
    neighbourhood of ``central_shot``.  Returns (point ids, report); ``report["shot_pose"]`` /
    ``report["points"]`` are full-size copies of the inputs with the adjusted blocks written back."""
    sub, shot_ids, pt_ids, interior, boundary = local_problem(problem, central_shot, config)
    cfg = dict(config or {})
    cfg["bundle_max_iterations"] = 10  # ba.SetMaxNumIterations(10), ba_helpers.cc:259
    r = bundle_arrays(sub, cfg, ctx=ctx, **overrides)
    poses = np.array(problem["shot_pose"], np.float64, copy=True)
    pts = np.array(problem["points"], np.float64, copy=True)
    poses[shot_ids] = r["shot_pose"]
    pts[pt_ids] = r["points"]
    report = {
        "brief_report": r["brief_report"], "wall_times": r["wall_times"],
        "num_images": int(interior.sum()), "num_interior_images": int(interior.sum()),
        "num_boundary_images": int(boundary.sum()),
        "num_other_images": int(len(poses) - interior.sum() - boundary.sum()),
        "num_points": int(len(pt_ids)), "num_reprojections": int(len(sub["obs_shot"])),
        "shot_pose": poses, "points": pts, "reproj_err": r["reproj_err"], "sub_problem": sub,
        "iterations": r["iterations"], "cost_history": r["cost_history"],
    }
    return pt_ids, report


def bundle_shot_poses_arrays(problem: Dict[str, np.ndarray], shot_ids, config: Optional[Dict[str, Any]] = None, ctx=None,
                             **overrides):
    """``pysfm.BAHelpers.bundle_shot_poses`` (ba_helpers.cc:408-579): only the poses of ``shot_ids`` move;
    cameras and points are constant; 10 LM iterations."""
    free = np.zeros(len(problem["shot_pose"]), bool)
    free[np.asarray(shot_ids, np.int64)] = True
    obs_shot = np.asarray(problem["obs_shot"], np.int64)
    obs_point = np.asarray(problem["obs_point"], np.int64)
    obs_mask = free[obs_shot]
    point_mask = np.zeros(len(problem["points"]), bool)
    point_mask[obs_point[obs_mask]] = True
    sub, sids, pids = _sub_problem(problem, free, np.zeros_like(free), point_mask, obs_mask, True, bool(_cfg(config, "bundle_use_gps")))
    cfg = dict(config or {})
    cfg["bundle_max_iterations"] = 10
    r = bundle_arrays(sub, cfg, ctx=ctx, **overrides)
    poses = np.array(problem["shot_pose"], np.float64, copy=True)
    poses[sids] = r["shot_pose"]
    return {"brief_report": r["brief_report"], "wall_times": r["wall_times"], "shot_pose": poses, "sub_problem": sub,
            "iterations": r["iterations"], "cost_history": r["cost_history"]}


class _BAPoint:
    """``bundle::Point`` as ``get_point`` returns it: ``id``, ``p`` and ``reprojection_errors`` (shot id -> residual with sigma 1)"""

    def __init__(self, point_id, p, errors):
        self.id = point_id
        self.p = np.asarray(p, float)
        self.reprojection_errors = errors


def _pose_c2w(pose) -> np.ndarray:
    """bundle::Pose data [rx ry rz tx ty tz] (CAM_TO_WORLD, bundle/data/pose.h:34-43) of a pygeometry.Pose-like object"""
    if hasattr(pose, "cam_to_world_parameters"):
        return np.asarray(pose.cam_to_world_parameters(), float)
    rot = np.asarray(pose.rotation, float).reshape(3)
    return np.r_[-rot, np.asarray(pose.get_origin(), float).reshape(3)]


class BundleAdjuster:
    """``pybundle.BundleAdjuster`` (``bundle/python/pybind.cc:45-117``, ``bundle_adjuster.h:178-300``): same method names, argument
    order and meaning for everything ``BAHelpers::Bundle`` / ``BundleLocal`` / ``BundleShotPoses`` call, plus the absolute pan / tilt /
    roll priors.  String ids become dense indices at ``run()``, which hands the problem to ``osfm_bundle_solve`` (every camera
    model, rigs, biases, control points) -- or to the streaming ``osfm_ba_solve`` when the problem is in its domain (perspective /
    fisheye cameras, one constant identity rig camera, one shot per instance, identity biases, no point priors).
    Cameras / poses / similarities are any objects with the reference's attributes (``geometry_types`` has plain ones).
    Not implemented (raise): relative motions / rotations, common positions, heatmaps, linear motion, reconstructions with shared
    scales, covariances -- none of which ``BAHelpers`` uses."""

    def __init__(self):
        self._cams: Dict[str, Dict[str, Any]] = {}
        self._bias: Dict[str, Any] = {}
        self._rig_cameras: Dict[str, Dict[str, Any]] = {}
        self._instances: Dict[str, Dict[str, Any]] = {}
        self._shots: Dict[str, Dict[str, Any]] = {}
        self._points: Dict[str, Dict[str, Any]] = {}
        self._obs: List[Any] = []
        self._loss = ("CauchyLoss", 1.0)  # the constructor's default, bundle_adjuster.cc:25
        self._sd = dict(focal=1.0, aspect_ratio=1.0, c=1.0, k1=1.0, k2=1.0, p1=1.0, p2=1.0, k3=1.0, k4=1.0)
        self._rig_sd = (1.0, 1.0)  # translation, rotation (bundle_adjuster.h:350-351)
        self._max_iter = 500
        self._report: Optional[Dict[str, Any]] = None
        self.force_general = False  # tests: never take the streaming solver

    # ---- variables ----
    def add_camera(self, cam_id, camera, camera_prior, constant):
        self._cams[cam_id] = {"camera": camera, "prior": camera_prior, "fixed": bool(constant)}
        self._bias.setdefault(cam_id, None)  # identity, constant (bundle_adjuster.cc:98-106)

    def set_camera_bias(self, camera_id, bias):
        if camera_id not in self._cams:
            raise RuntimeError("Camera " + str(camera_id) + " doesn't exist.")
        self._bias[camera_id] = bias  # a fresh Similarity data block: free (SetCameraBias, bundle_adjuster.cc:109-116)

    def add_rig_camera(self, rig_camera_id, pose, pose_prior, fixed):
        if rig_camera_id in self._rig_cameras:
            raise RuntimeError("Rig model " + str(rig_camera_id) + " already exist.")
        self._rig_cameras[rig_camera_id] = {"v": _pose_c2w(pose), "prior": _pose_c2w(pose_prior), "fixed": bool(fixed)}

    def add_rig_instance(self, rig_instance_id, rig_instance_pose, shot_cameras: Dict[str, str], shot_rig_cameras: Dict[str, str], fixed):
        self._instances[rig_instance_id] = {"v": _pose_c2w(rig_instance_pose), "fixed": bool(fixed), "gps": None, "gps_sd": None,
                                            "shots": list(shot_cameras)}
        for shot_id, cam_id in shot_cameras.items():
            rc_id = shot_rig_cameras[shot_id]
            if cam_id not in self._cams:
                raise RuntimeError("Camera " + str(cam_id) + " doesn't exist.")
            if rc_id not in self._rig_cameras:
                raise RuntimeError("Rig camera " + str(rc_id) + " doesn't exist.")
            self._shots[shot_id] = {"instance": rig_instance_id, "camera": cam_id, "rig_camera": rc_id}

    def add_rig_instance_position_prior(self, instance_id, position, std_deviation, scale_group=""):
        if instance_id not in self._instances:
            raise RuntimeError("Rig instance " + str(instance_id) + " doesn't exist.")
        self._instances[instance_id]["gps"] = np.asarray(position, float).reshape(3)
        self._instances[instance_id]["gps_sd"] = np.broadcast_to(np.asarray(std_deviation, float), (3,)).copy()

    def add_point(self, point_id, position, constant):
        self._points[point_id] = {"p": np.asarray(position, float).reshape(3).copy(), "fixed": bool(constant), "prior": None}

    def add_point_prior(self, point_id, position, std_deviation, has_altitude_prior):
        if point_id not in self._points:
            raise RuntimeError("Point " + str(point_id) + " doesn't exist.")
        self._points[point_id]["prior"] = (np.asarray(position, float).reshape(3), np.asarray(std_deviation, float).reshape(3), bool(has_altitude_prior))

    def has_point(self, point_id):
        return point_id in self._points

    def add_point_projection_observation(self, shot, point, observation, std_deviation, depth_prior=None):
        """``BundleAdjuster::AddPointProjectionObservation`` (bundle_adjuster.cc:236-250); ``depth_prior``: a ``map::Depth``
        (``.value``, ``.is_radial``, ``.std_deviation``, observation.h:10-18) -> a RelativeDepthError next to the reprojection"""
        if shot not in self._shots or point not in self._points:
            raise IndexError("unknown shot or point")  # std::map::at
        depth = None
        if depth_prior is not None:
            depth = (float(depth_prior.value), float(depth_prior.std_deviation), bool(depth_prior.is_radial))
            if not depth[1] > 0:  # the reference scales the residual by 1 / sd whatever it is; the C ABI reads sd <= 0 as "no prior": say so
                raise ValueError("depth prior of shot %s / point %s has std_deviation %r (must be > 0)" % (shot, point, depth[1]))
        self._obs.append((shot, point, float(observation[0]), float(observation[1]), float(std_deviation), depth))

    def _shot_prior(self, key, shot_id, value, std_deviation):
        self._shots[shot_id][key] = (value, float(std_deviation))

    def add_absolute_up_vector(self, shot_id, up_vector, std_deviation):
        """``BundleAdjuster::AddAbsoluteUpVector`` (bundle_adjuster.cc:300-308)."""
        self._shot_prior("up", shot_id, np.asarray(up_vector, float).reshape(3), std_deviation)

    def add_absolute_pan(self, shot_id, angle, std_deviation):
        self._shot_prior("pan", shot_id, float(angle), std_deviation)

    def add_absolute_tilt(self, shot_id, angle, std_deviation):
        self._shot_prior("tilt", shot_id, float(angle), std_deviation)

    def add_absolute_roll(self, shot_id, angle, std_deviation):
        self._shot_prior("roll", shot_id, float(angle), std_deviation)

    # ---- the rest of the reference's builder surface: not on this path ----
    def _unsupported(self, *a, **k):
        raise NotImplementedError("this residual family is not implemented on the GPU path (BAHelpers does not use it)")

    add_relative_motion = add_relative_rotation = add_common_position = add_heatmap = add_absolute_position_heatmap = _unsupported
    add_linear_motion = add_reconstruction = add_reconstruction_instance = set_scale_sharing = set_gauge_fix_shots = _unsupported

    # ---- minimisation setup ----
    def set_point_projection_loss_function(self, name, threshold):
        self._loss = (name, float(threshold))

    def set_relative_motion_loss_function(self, name, threshold):
        pass  # no relative motion residuals on this path

    def set_internal_parameters_prior_sd(self, focal_sd, aspect_ratio_sd, c_sd, k1_sd, k2_sd, p1_sd, p2_sd, k3_sd, k4_sd):
        self._sd = dict(focal=focal_sd, aspect_ratio=aspect_ratio_sd, c=c_sd, k1=k1_sd, k2=k2_sd, p1=p1_sd, p2=p2_sd, k3=k3_sd, k4=k4_sd)

    def set_rig_parameters_prior_sd(self, rig_translation_sd, rig_rotation_sd):
        self._rig_sd = (float(rig_translation_sd), float(rig_rotation_sd))

    def set_max_num_iterations(self, n):
        self._max_iter = int(n)

    def set_num_threads(self, n):  # the GPU path has no thread knob
        pass

    def set_linear_solver_type(self, t):
        if t not in ("SPARSE_SCHUR", "DENSE_SCHUR", "ITERATIVE_SCHUR", "DENSE_QR", "SPARSE_NORMAL_CHOLESKY"):
            raise RuntimeError("Linear solver type " + str(t) + " doesn't exist.")  # bundle_adjuster.cc:1107

    def set_use_analytic_derivatives(self, v):
        pass

    def set_compute_reprojection_errors(self, v):
        pass

    def set_compute_covariances(self, v):
        if v:
            raise NotImplementedError("covariance estimation is not on the GPU path")

    def set_adjust_absolute_position_std(self, v):
        if v:
            raise NotImplementedError("adjusting the GPS standard deviations is not on the GPU path")

    def get_covariance_estimation_valid(self):
        return False

    # ---- flattening ----
    def _camera_sigma(self, camera) -> np.ndarray:
        """GetDefaultCameraSigma (bundle_adjuster.cc:47-69): parameter types without an entry (k4 .. k6, s0 .. s3) get sigma 0, i.e. the
        prior pins them -- exactly what the reference's std::unordered_map::operator[] default does"""
        from .geometry_types import CAMERA_PARAMETERS

        key = {"focal": "focal", "aspect_ratio": "aspect_ratio", "cx": "c", "cy": "c", "k1": "k1", "k2": "k2", "k3": "k3", "p1": "p1", "p2": "p2"}
        names = CAMERA_PARAMETERS["spherical" if camera.projection_type == "equirectangular" else camera.projection_type]
        out = np.ones(16)
        for k, n in enumerate(names):
            out[k] = 1.0 if n == "transition" else (self._sd[key[n]] if n in key else 0.0)
        return out

    def _problem(self) -> Dict[str, Any]:
        from .geometry_types import CAMERA_MODEL_IDS, camera_parameter_values

        cam_ids, rc_ids, inst_ids = list(self._cams), list(self._rig_cameras), list(self._instances)
        shot_ids, pt_ids = list(self._shots), list(self._points)
        ci, ri, ii = ({k: n for n, k in enumerate(x)} for x in (cam_ids, rc_ids, inst_ids))
        si, pi = ({k: n for n, k in enumerate(x)} for x in (shot_ids, pt_ids))
        self._index = dict(cam=cam_ids, rc=rc_ids, inst=inst_ids, shot=shot_ids, pt=pt_ids)
        NC, NR, NI, S, NP = len(cam_ids), len(rc_ids), len(inst_ids), len(shot_ids), len(pt_ids)
        bias = np.tile([0, 0, 0, 0, 0, 0, 1.0], (NC, 1))
        bias_fixed = np.ones(NC, np.uint8)
        for k, b in self._bias.items():
            if b is not None:
                bias[ci[k]] = np.r_[np.asarray(b.rotation, float), np.asarray(b.translation, float), float(b.scale)]
                bias_fixed[ci[k]] = 0
        gps, gps_sd, bias_cam = np.zeros((NI, 3)), np.zeros((NI, 3)), np.zeros(NI, np.int32)
        for k, inst in self._instances.items():
            if inst["gps"] is not None:
                gps[ii[k]], gps_sd[ii[k]] = inst["gps"], inst["gps_sd"]
                if not inst["shots"]:
                    raise RuntimeError("Reference camera of RigInstance " + str(k) + " doesn't have associated Bias")
                # rig_instance.shot_cameras.begin(): the smallest shot id of the std::map / first of the unordered_map
                bias_cam[ii[k]] = ci[self._shots[sorted(inst["shots"])[0]]["camera"]]
        tr_sd, rot_sd = self._rig_sd
        prob: Dict[str, Any] = {
            "cam_model": np.array([CAMERA_MODEL_IDS[self._cams[k]["camera"].projection_type] for k in cam_ids], np.int32),
            "cam_params": np.array([camera_parameter_values(self._cams[k]["camera"]) for k in cam_ids]).reshape(NC, 16),
            "cam_prior": np.array([camera_parameter_values(self._cams[k]["prior"]) for k in cam_ids]).reshape(NC, 16),
            "cam_sigma": np.array([self._camera_sigma(self._cams[k]["camera"]) for k in cam_ids]).reshape(NC, 16),
            "cam_fixed": np.array([self._cams[k]["fixed"] for k in cam_ids], np.uint8),
            "bias": bias, "bias_fixed": bias_fixed,
            "rig_camera_pose": np.array([self._rig_cameras[k]["v"] for k in rc_ids]).reshape(NR, 6),
            "rig_camera_prior": np.array([self._rig_cameras[k]["prior"] for k in rc_ids]).reshape(NR, 6),
            "rig_camera_sigma": np.tile([rot_sd] * 3 + [tr_sd] * 3, (NR, 1)).astype(float),
            "rig_camera_fixed": np.array([self._rig_cameras[k]["fixed"] for k in rc_ids], np.uint8),
            "rig_instance_pose": np.array([self._instances[k]["v"] for k in inst_ids]).reshape(NI, 6),
            "rig_instance_fixed": np.array([self._instances[k]["fixed"] for k in inst_ids], np.uint8),
            "shot_rig_instance": np.array([ii[self._shots[k]["instance"]] for k in shot_ids], np.int32),
            "shot_rig_camera": np.array([ri[self._shots[k]["rig_camera"]] for k in shot_ids], np.int32),
            "shot_camera": np.array([ci[self._shots[k]["camera"]] for k in shot_ids], np.int32),
            "points": np.array([self._points[k]["p"] for k in pt_ids]).reshape(NP, 3),
            "point_fixed": np.array([self._points[k]["fixed"] for k in pt_ids], np.uint8),
            "obs_shot": np.array([si[o[0]] for o in self._obs], np.int32), "obs_point": np.array([pi[o[1]] for o in self._obs], np.int32),
            "obs_xy": np.array([[o[2], o[3]] for o in self._obs]).reshape(-1, 2), "obs_sigma": np.array([o[4] for o in self._obs]),
        }
        if any(o[5] is not None for o in self._obs):  # depth priors (RelativeDepthError): sd 0 = none
            prob["obs_depth"] = np.array([o[5][0] if o[5] else 0.0 for o in self._obs])
            prob["obs_depth_sigma"] = np.array([o[5][1] if o[5] else 0.0 for o in self._obs])
            prob["obs_depth_radial"] = np.array([o[5][2] if o[5] else 1 for o in self._obs], np.uint8)
        if gps_sd.max() > 0:
            prob.update(rig_instance_gps=gps, rig_instance_gps_sigma=gps_sd, rig_instance_bias_camera=bias_cam)
        if any(p["prior"] is not None for p in self._points.values()):
            pp, ps, alt = np.zeros((NP, 3)), np.zeros((NP, 3)), np.ones(NP, np.uint8)
            for k, p in self._points.items():
                if p["prior"] is not None:
                    pp[pi[k]], ps[pi[k]], alt[pi[k]] = p["prior"][0], np.maximum(p["prior"][1], 1e-300), p["prior"][2]
            prob.update(point_prior=pp, point_prior_sigma=ps, point_prior_has_altitude=alt)
        if any("up" in sh for sh in self._shots.values()):
            prob["shot_up"] = np.array([self._shots[k]["up"][0] if "up" in self._shots[k] else np.zeros(3) for k in shot_ids]).reshape(S, 3)
            prob["shot_up_sigma"] = np.array([self._shots[k]["up"][1] if "up" in self._shots[k] else 0.0 for k in shot_ids])
        for key in ("pan", "tilt", "roll"):
            if any(key in sh for sh in self._shots.values()):
                prob["shot_" + key] = np.array([self._shots[k][key][0] if key in self._shots[k] else 0.0 for k in shot_ids])
                prob["shot_" + key + "_sigma"] = np.array([self._shots[k][key][1] if key in self._shots[k] else 0.0 for k in shot_ids])
        return prob

    @staticmethod
    def _streaming_form(prob: Dict[str, Any]) -> Optional[Dict[str, Any]]:
        """the same problem in the layout of ``osfm_ba_solve`` when it lies in that solver's domain, else None"""
        NI, S = len(prob["rig_instance_pose"]), len(prob["shot_camera"])
        if (S != NI or len(prob["rig_camera_pose"]) != 1 or not prob["rig_camera_fixed"][0] or prob["rig_camera_pose"].any()
                or (prob["cam_model"] > 1).any() or not prob["bias_fixed"].all() or (prob["bias"] != [0, 0, 0, 0, 0, 0, 1.0]).any()
                or "point_prior" in prob or "obs_depth" in prob or any(k in prob for k in ("shot_pan", "shot_tilt", "shot_roll"))
                or not np.array_equal(np.sort(prob["shot_rig_instance"]), np.arange(NI)) or len(prob["obs_shot"]) == 0):
            return None
        if "rig_instance_gps_sigma" in prob:
            sd = prob["rig_instance_gps_sigma"]
            if not (np.allclose(sd[:, 0], sd[:, 1]) and np.allclose(sd[:, 0], sd[:, 2])):
                return None  # the streaming solver takes one sd per position prior
        order = np.argsort(prob["shot_rig_instance"])  # shot s of the streaming problem = the shot of instance s
        inv = np.empty(S, np.int64)
        inv[order] = np.arange(S)
        out = {"cam_params": prob["cam_params"][:, :3], "cam_prior": prob["cam_prior"][:, :3], "cam_sigma": prob["cam_sigma"][:, :3],
               "cam_fixed": prob["cam_fixed"], "cam_model": prob["cam_model"], "shot_pose": prob["rig_instance_pose"],
               "shot_camera": prob["shot_camera"][order], "shot_fixed": prob.get("rig_instance_fixed", np.zeros(NI, np.uint8)), "points": prob["points"],
               "point_fixed": prob.get("point_fixed", np.zeros(len(prob["points"]), np.uint8)), "obs_shot": inv[prob["obs_shot"]].astype(np.int32), "obs_point": prob["obs_point"],
               "obs_xy": prob["obs_xy"], "obs_sigma": prob["obs_sigma"]}
        if "rig_instance_gps" in prob:
            out["shot_gps"], out["shot_gps_sigma"] = prob["rig_instance_gps"], prob["rig_instance_gps_sigma"][:, 0]
        if "shot_up" in prob:
            out["shot_up"], out["shot_up_sigma"] = prob["shot_up"][order], prob["shot_up_sigma"][order]
        return out

    def run(self):
        from .geometry_types import set_camera_parameter_values

        if not self._cams or not self._rig_cameras or not self._instances or not self._shots:
            raise RuntimeError("BundleAdjuster.run: nothing to adjust")
        for o in self._obs:  # thrown from Run, with the shot id, as bundle_adjuster.cc:508-511 does
            if o[5] is not None and not np.isfinite(o[5][0]):
                raise RuntimeError(str(o[0]) + " has non-finite depth prior")
        prob = self._problem()
        cfg = {"loss_function": self._loss[0], "loss_function_threshold": self._loss[1], "bundle_max_iterations": self._max_iter}
        stream = None if self.force_general else self._streaming_form(prob)
        if stream is not None:
            r = bundle_arrays(stream, dict(cfg, bundle_use_gps=True))
            cam = prob["cam_params"].copy()
            cam[:, :3] = r["cam_params"]
            res = {"cam_params": cam, "bias": prob["bias"], "rig_camera_pose": prob["rig_camera_pose"], "rig_instance_pose": r["shot_pose"],
                   "points": r["points"], "reproj_err": np.c_[r["reproj_err"], np.zeros(len(r["reproj_err"]))]}
            self.solver = "osfm_ba_solve"
        else:
            r = bundle_general_arrays(prob, cfg)
            res = r
            self.solver = "osfm_bundle_solve"
        ix = self._index
        for n, k in enumerate(ix["cam"]):
            cam = self._cams[k]["camera"]
            cam = cam.copy() if hasattr(cam, "copy") else __import__("copy").deepcopy(cam)
            set_camera_parameter_values(cam, res["cam_params"][n])
            self._cams[k]["result"] = cam
            self._cams[k]["bias_result"] = res["bias"][n]
        for n, k in enumerate(ix["rc"]):
            self._rig_cameras[k]["v"] = np.asarray(res["rig_camera_pose"][n])
        for n, k in enumerate(ix["inst"]):
            self._instances[k]["v"] = np.asarray(res["rig_instance_pose"][n])
        for n, k in enumerate(ix["pt"]):
            self._points[k]["p"] = np.asarray(res["points"][n])
            self._points[k]["errors"] = {}
        spherical = {k for k in ix["cam"] if self._cams[k]["camera"].projection_type in ("spherical", "equirectangular")}
        for (shot, point, *_), e in zip(self._obs, res["reproj_err"]):
            self._points[point]["errors"][shot] = np.asarray(e if self._shots[shot]["camera"] in spherical else e[:2])
        self._report = r

    # ---- getters ----
    def get_camera(self, cam_id):
        c = self._cams[cam_id]
        return c.get("result", c["camera"])

    def get_bias(self, cam_id):
        from .geometry_types import Similarity

        b = self._cams[cam_id].get("bias_result")
        if b is None:
            return self._bias[cam_id] or Similarity()
        return Similarity(b[:3], b[3:6], b[6])

    def _pose_object(self, v):
        from .geometry_types import Pose

        return Pose.from_cam_to_world(v[:3], v[3:6])

    def get_rig_camera_pose(self, rig_camera_id):
        return self._pose_object(self._rig_cameras[rig_camera_id]["v"])

    def get_rig_instance_pose(self, rig_instance_id):
        return self._pose_object(self._instances[rig_instance_id]["v"])

    def get_point(self, point_id):
        return _BAPoint(point_id, self._points[point_id]["p"], self._points[point_id].get("errors", {}))

    def get_projections_count(self):
        return len(self._obs)

    def get_relative_motions_count(self):
        return 0

    def get_rig_instances(self):
        return {k: self.get_rig_instance_pose(k) for k in self._instances}

    def brief_report(self):
        return self._report["brief_report"] if self._report else ""

    def full_report(self):
        return self.brief_report()
