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
    instances, GPS priors through per-camera biases, point priors (ground control points), up vectors.  Inputs are not modified;
    returns the optimised parameter arrays, ``reproj_err`` (n_obs x 3, sigma 1) and the report."""
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
        "seconds_linear_solver": R.seconds_linear_solver,
        "brief_report": "osfm-mi355 LM (dense Schur): iterations %d (successful %d), initial cost %.6e, final cost %.6e, termination: %s"
                        % (R.iterations, R.successful_steps, R.initial_cost, R.final_cost, TERMINATION.get(R.termination, str(R.termination))),
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


class _Point:
    def __init__(self, pid, p):
        self.id = pid
        self.p = np.asarray(p, float)
        self.reprojection_errors: Dict[str, np.ndarray] = {}


class _Pose:
    """What ``get_rig_instance_pose`` returns in the reference is a ``pygeometry.Pose``; here a light
    value object with the same two accessors the callers use."""

    def __init__(self, rt):
        self.rotation = np.asarray(rt[:3], float)  # angle-axis, camera -> world (bundle/data/pose.h:34-43)
        self._origin = np.asarray(rt[3:6], float)

    def get_origin(self):
        return self._origin


class BundleAdjuster:
    """Builder with the method names of ``pybundle.BundleAdjuster`` (``bundle/python/pybind.cc:45-117``)
    for the subset ``BAHelpers::Bundle`` uses; string ids are turned into dense indices at ``run()``.

    Cameras are given as ``(k1, k2, focal)`` triples of a PERSPECTIVE camera (the reference passes a
    ``pygeometry.Camera``; any object with ``projection_type``, ``k1``, ``k2``, ``focal`` works)."""

    def __init__(self):
        self._cams: Dict[str, Dict[str, Any]] = {}
        self._shots: Dict[str, Dict[str, Any]] = {}
        self._points: Dict[str, Dict[str, Any]] = {}
        self._obs: List[Any] = []
        self._loss = ("SoftLOneLoss", 1.0)
        self._sd = {"focal": 0.01, "k1": 0.01, "k2": 0.01}
        self._max_iter = 100
        self._report: Optional[Dict[str, Any]] = None

    @staticmethod
    def _cam_values(cam):
        if hasattr(cam, "projection_type"):
            if cam.projection_type != "perspective":
                raise NotImplementedError("only PERSPECTIVE cameras are on the GPU path")
            return np.array([cam.k1, cam.k2, cam.focal], float)
        return np.asarray(cam, float).reshape(3)

    def add_camera(self, cam_id, camera, camera_prior, constant):
        self._cams[cam_id] = {"v": self._cam_values(camera), "prior": self._cam_values(camera_prior), "fixed": bool(constant)}

    def add_rig_instance(self, rig_instance_id, rotation, origin, shot_cameras: Dict[str, str], fixed=False):
        """One shot per rig instance (identity rig camera, the plain-reconstruction case)."""
        if len(shot_cameras) != 1:
            raise NotImplementedError("multi-camera rigs are not on the GPU path")
        (shot_id, cam_id), = shot_cameras.items()
        self._shots[shot_id] = {"rt": np.concatenate([np.asarray(rotation, float), np.asarray(origin, float)]),
                                "camera": cam_id, "fixed": bool(fixed), "gps": None, "gps_sd": 0.0,
                                "instance": rig_instance_id}

    def add_rig_instance_position_prior(self, instance_id, position, std_deviation, scale_group=""):
        for sh in self._shots.values():
            if sh["instance"] == instance_id:
                sh["gps"] = np.asarray(position, float)
                sh["gps_sd"] = float(np.mean(std_deviation))

    def add_absolute_up_vector(self, shot_id, up_vector, std_deviation):
        """``BundleAdjuster::AddAbsoluteUpVector`` (bundle_adjuster.cc:300-308)."""
        self._shots[shot_id]["up"] = np.asarray(up_vector, float)
        self._shots[shot_id]["up_sd"] = float(std_deviation)

    def add_point(self, point_id, position, constant):
        self._points[point_id] = {"p": np.asarray(position, float), "fixed": bool(constant)}

    def has_point(self, point_id):
        return point_id in self._points

    def add_point_projection_observation(self, shot, point, observation, std_deviation, depth_prior=None):
        if depth_prior is not None:
            raise NotImplementedError("depth priors are not on the GPU path")
        self._obs.append((shot, point, float(observation[0]), float(observation[1]), float(std_deviation)))

    def set_point_projection_loss_function(self, name, threshold):
        self._loss = (name, float(threshold))

    def set_internal_parameters_prior_sd(self, focal_sd, aspect_ratio_sd, c_sd, k1_sd, k2_sd, p1_sd, p2_sd, k3_sd, k4_sd):
        self._sd = {"focal": focal_sd, "k1": k1_sd, "k2": k2_sd}

    def set_max_num_iterations(self, n):
        self._max_iter = int(n)

    def set_num_threads(self, n):  # the GPU path has no thread knob
        pass

    def set_linear_solver_type(self, t):
        if t not in ("SPARSE_SCHUR", "DENSE_SCHUR", "ITERATIVE_SCHUR"):
            raise RuntimeError("Linear solver type " + str(t) + " doesn't exist.")  # bundle_adjuster.cc:1107

    def set_use_analytic_derivatives(self, v):
        pass

    def set_compute_reprojection_errors(self, v):
        pass

    def run(self):
        cam_ids, shot_ids, pt_ids = list(self._cams), list(self._shots), list(self._points)
        ci = {k: i for i, k in enumerate(cam_ids)}
        si = {k: i for i, k in enumerate(shot_ids)}
        pi = {k: i for i, k in enumerate(pt_ids)}
        gps = np.zeros((len(shot_ids), 3))
        gps_sd = np.zeros(len(shot_ids))
        for k, s in self._shots.items():
            if s["gps"] is not None:
                gps[si[k]] = s["gps"]
                gps_sd[si[k]] = s["gps_sd"]
        prob = {
            "cam_params": np.array([self._cams[k]["v"] for k in cam_ids]),
            "cam_prior": np.array([self._cams[k]["prior"] for k in cam_ids]),
            "cam_sigma": np.tile([self._sd["k1"], self._sd["k2"], self._sd["focal"]], (len(cam_ids), 1)).astype(float),
            "cam_fixed": np.array([self._cams[k]["fixed"] for k in cam_ids], np.uint8),
            "shot_pose": np.array([self._shots[k]["rt"] for k in shot_ids]),
            "shot_camera": np.array([ci[self._shots[k]["camera"]] for k in shot_ids], np.int32),
            "shot_fixed": np.array([self._shots[k]["fixed"] for k in shot_ids], np.uint8),
            "points": np.array([self._points[k]["p"] for k in pt_ids]),
            "point_fixed": np.array([self._points[k]["fixed"] for k in pt_ids], np.uint8),
            "obs_shot": np.array([si[o[0]] for o in self._obs], np.int32),
            "obs_point": np.array([pi[o[1]] for o in self._obs], np.int32),
            "obs_xy": np.array([[o[2], o[3]] for o in self._obs]),
            "obs_sigma": np.array([o[4] for o in self._obs]),
        }
        if gps_sd.max() > 0:
            prob["shot_gps"], prob["shot_gps_sigma"] = gps, gps_sd
        if any("up" in s for s in self._shots.values()):
            prob["shot_up"] = np.array([self._shots[k].get("up", np.zeros(3)) for k in shot_ids])
            prob["shot_up_sigma"] = np.array([self._shots[k].get("up_sd", 0.0) for k in shot_ids])
        cfg = {"loss_function": self._loss[0], "loss_function_threshold": self._loss[1], "bundle_max_iterations": self._max_iter}
        r = bundle_arrays(prob, cfg)
        for k in cam_ids:
            self._cams[k]["v"] = r["cam_params"][ci[k]]
        for k in shot_ids:
            self._shots[k]["rt"] = r["shot_pose"][si[k]]
        for k in pt_ids:
            self._points[k]["p"] = r["points"][pi[k]]
            self._points[k]["errors"] = {}
        for (shot, point, *_), e in zip(self._obs, r["reproj_err"]):
            self._points[point]["errors"][shot] = e
        self._report = r

    def get_camera(self, cam_id):
        return self._cams[cam_id]["v"]

    def get_rig_instance_pose(self, rig_instance_id):
        for s in self._shots.values():
            if s["instance"] == rig_instance_id:
                return _Pose(s["rt"])
        raise KeyError(rig_instance_id)

    def get_point(self, point_id):
        p = _Point(point_id, self._points[point_id]["p"])
        p.reprojection_errors = self._points[point_id].get("errors", {})
        return p

    def brief_report(self):
        return self._report["brief_report"] if self._report else ""

    def full_report(self):
        return self.brief_report()
