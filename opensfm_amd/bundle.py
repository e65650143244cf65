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
from ._ba_abi import BaOptions, BaProblem, BaReport
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
                       ("shot_up", np.float64, C.c_double), ("shot_up_sigma", np.float64, C.c_double)):
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
        "seconds_solver": R.seconds_total, "seconds_linear_solver": R.seconds_linear_solver,
        "seconds_setup": R.seconds_setup, "seconds_run": R.seconds_run, "seconds_teardown": R.seconds_teardown,
        "ms_per_matvec": (R.ms_matvec_total / R.matvec_calls) if R.matvec_calls else None,
        # report dict of BAHelpers::Bundle (ba_helpers.cc:743-762)
        "brief_report": brief,
        "wall_times": {"setup": (t1 - t0) + R.seconds_setup, "run": R.seconds_run, "teardown": R.seconds_teardown},
        "num_images": len(poses), "num_points": len(pts), "num_reprojections": len(obs_shot),
    }


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
