"""ctypes structs/signatures of the bundle-adjustment part of include/osfm_mi355.h."""
import ctypes as C


class BaProblem(C.Structure):
    _fields_ = [
        ("n_cameras", C.c_int32), ("n_shots", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int64),
        ("cam_params", C.POINTER(C.c_double)), ("cam_prior", C.POINTER(C.c_double)), ("cam_sigma", C.POINTER(C.c_double)),
        ("cam_fixed", C.POINTER(C.c_uint8)),
        ("shot_pose", C.POINTER(C.c_double)), ("shot_camera", C.POINTER(C.c_int32)), ("shot_fixed", C.POINTER(C.c_uint8)),
        ("shot_gps", C.POINTER(C.c_double)), ("shot_gps_sigma", C.POINTER(C.c_double)),
        ("points", C.POINTER(C.c_double)), ("point_fixed", C.POINTER(C.c_uint8)),
        ("obs_shot", C.POINTER(C.c_int32)), ("obs_point", C.POINTER(C.c_int32)),
        ("obs_xy", C.POINTER(C.c_double)), ("obs_sigma", C.POINTER(C.c_double)), ("reproj_err", C.POINTER(C.c_double)),
        ("shot_up", C.POINTER(C.c_double)), ("shot_up_sigma", C.POINTER(C.c_double)),
        ("cam_model", C.POINTER(C.c_int32)), ("cam_ext", C.POINTER(C.c_double)),
    ]


class BaOptions(C.Structure):
    _fields_ = [
        ("loss", C.c_int32), ("loss_threshold", C.c_double), ("max_iterations", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double), ("verbose", C.c_int32), ("pcg_tolerance", C.c_double),
        ("pcg_max_iterations", C.c_int32), ("preconditioner", C.c_int32), ("pcg_direct_tolerance", C.c_double),
    ]


class BaReport(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("rmse_normalized_initial", C.c_double), ("rmse_normalized_final", C.c_double),
        ("seconds_total", C.c_double), ("seconds_linear_solver", C.c_double), ("cost_history", C.c_double * 256),
        ("pcg_iterations_total", C.c_int64), ("ms_matvec_total", C.c_double), ("matvec_calls", C.c_int64),
        ("shot_bandwidth", C.c_int32), ("preconditioner_bandwidth", C.c_int32),
        ("seconds_setup", C.c_double), ("seconds_run", C.c_double), ("seconds_teardown", C.c_double),
        ("shots_reordered", C.c_int32), ("shot_bandwidth_input", C.c_int32),
    ]


class BundleProblem(C.Structure):
    """osfm_bundle_problem (include/osfm_mi355.h): the general bundle adjustment"""
    _fields_ = [
        ("n_cameras", C.c_int32), ("cam_model", C.POINTER(C.c_int32)), ("cam_params", C.POINTER(C.c_double)),
        ("cam_prior", C.POINTER(C.c_double)), ("cam_sigma", C.POINTER(C.c_double)), ("cam_fixed", C.POINTER(C.c_uint8)),
        ("bias", C.POINTER(C.c_double)), ("bias_fixed", C.POINTER(C.c_uint8)),
        ("n_rig_cameras", C.c_int32), ("rig_camera_pose", C.POINTER(C.c_double)), ("rig_camera_prior", C.POINTER(C.c_double)),
        ("rig_camera_sigma", C.POINTER(C.c_double)), ("rig_camera_fixed", C.POINTER(C.c_uint8)),
        ("n_rig_instances", C.c_int32), ("rig_instance_pose", C.POINTER(C.c_double)), ("rig_instance_fixed", C.POINTER(C.c_uint8)),
        ("rig_instance_gps", C.POINTER(C.c_double)), ("rig_instance_gps_sigma", C.POINTER(C.c_double)),
        ("rig_instance_bias_camera", C.POINTER(C.c_int32)),
        ("n_shots", C.c_int32), ("shot_rig_instance", C.POINTER(C.c_int32)), ("shot_rig_camera", C.POINTER(C.c_int32)),
        ("shot_camera", C.POINTER(C.c_int32)), ("shot_up", C.POINTER(C.c_double)), ("shot_up_sigma", C.POINTER(C.c_double)),
        ("n_points", C.c_int32), ("points", C.POINTER(C.c_double)), ("point_fixed", C.POINTER(C.c_uint8)),
        ("point_prior", C.POINTER(C.c_double)), ("point_prior_sigma", C.POINTER(C.c_double)), ("point_prior_has_altitude", C.POINTER(C.c_uint8)),
        ("n_obs", C.c_int64), ("obs_shot", C.POINTER(C.c_int32)), ("obs_point", C.POINTER(C.c_int32)), ("obs_xy", C.POINTER(C.c_double)),
        ("obs_sigma", C.POINTER(C.c_double)), ("reproj_err", C.POINTER(C.c_double)),
        ("shot_pan", C.POINTER(C.c_double)), ("shot_pan_sigma", C.POINTER(C.c_double)), ("shot_tilt", C.POINTER(C.c_double)),
        ("shot_tilt_sigma", C.POINTER(C.c_double)), ("shot_roll", C.POINTER(C.c_double)), ("shot_roll_sigma", C.POINTER(C.c_double)),
        ("obs_depth", C.POINTER(C.c_double)), ("obs_depth_sigma", C.POINTER(C.c_double)), ("obs_depth_radial", C.POINTER(C.c_uint8)),
    ]


# (field, numpy dtype, columns or None for 1-D, required, in/out) of the dict form of a BundleProblem
BUNDLE_FIELDS = [
    ("cam_model", "int32", None, True, False), ("cam_params", "float64", 16, True, True), ("cam_prior", "float64", 16, True, False),
    ("cam_sigma", "float64", 16, True, False), ("cam_fixed", "uint8", None, True, False),
    ("bias", "float64", 7, False, True), ("bias_fixed", "uint8", None, False, False),
    ("rig_camera_pose", "float64", 6, True, True), ("rig_camera_prior", "float64", 6, False, False),
    ("rig_camera_sigma", "float64", 6, False, False), ("rig_camera_fixed", "uint8", None, True, False),
    ("rig_instance_pose", "float64", 6, True, True), ("rig_instance_fixed", "uint8", None, False, False),
    ("rig_instance_gps", "float64", 3, False, False), ("rig_instance_gps_sigma", "float64", 3, False, False),
    ("rig_instance_bias_camera", "int32", None, False, False),
    ("shot_rig_instance", "int32", None, True, False), ("shot_rig_camera", "int32", None, True, False), ("shot_camera", "int32", None, True, False),
    ("shot_up", "float64", 3, False, False), ("shot_up_sigma", "float64", None, False, False),
    ("points", "float64", 3, True, True), ("point_fixed", "uint8", None, False, False), ("point_prior", "float64", 3, False, False),
    ("point_prior_sigma", "float64", 3, False, False), ("point_prior_has_altitude", "uint8", None, False, False),
    ("obs_shot", "int32", None, True, False), ("obs_point", "int32", None, True, False), ("obs_xy", "float64", 2, True, False),
    ("obs_sigma", "float64", None, True, False),
    ("shot_pan", "float64", None, False, False), ("shot_pan_sigma", "float64", None, False, False),
    ("shot_tilt", "float64", None, False, False), ("shot_tilt_sigma", "float64", None, False, False),
    ("shot_roll", "float64", None, False, False), ("shot_roll_sigma", "float64", None, False, False),
    ("obs_depth", "float64", None, False, False), ("obs_depth_sigma", "float64", None, False, False), ("obs_depth_radial", "uint8", None, False, False),
]
_CT = {"int32": C.c_int32, "float64": C.c_double, "uint8": C.c_uint8}


def fill_bundle_problem(problem, struct_cls=BundleProblem):
    """dict of arrays -> (ctypes struct, arrays kept alive / to read the results from).  In/out arrays are COPIES of the inputs."""
    import numpy as np

    P = struct_cls()
    arrays = {}
    for name, dt, cols, required, inout in BUNDLE_FIELDS:
        v = problem.get(name)
        if v is None:
            if required:
                raise KeyError(f"bundle problem needs {name!r}")
            continue
        a = np.array(v, dtype=dt, copy=True) if inout else np.ascontiguousarray(v, dtype=dt)
        a = np.ascontiguousarray(a.reshape(-1, cols) if cols else a.reshape(-1))
        arrays[name] = a
        setattr(P, name, a.ctypes.data_as(C.POINTER(_CT[dt])))
    P.n_cameras = len(arrays["cam_model"])
    P.n_rig_cameras = len(arrays["rig_camera_pose"])
    P.n_rig_instances = len(arrays["rig_instance_pose"])
    P.n_shots = len(arrays["shot_camera"])
    P.n_points = len(arrays["points"])
    P.n_obs = len(arrays["obs_shot"])
    for name, n in (("cam_params", P.n_cameras), ("cam_prior", P.n_cameras), ("cam_sigma", P.n_cameras), ("cam_fixed", P.n_cameras),
                    ("bias", P.n_cameras), ("bias_fixed", P.n_cameras), ("rig_camera_prior", P.n_rig_cameras), ("rig_camera_sigma", P.n_rig_cameras),
                    ("rig_camera_fixed", P.n_rig_cameras), ("rig_instance_fixed", P.n_rig_instances), ("rig_instance_gps", P.n_rig_instances),
                    ("rig_instance_gps_sigma", P.n_rig_instances), ("rig_instance_bias_camera", P.n_rig_instances),
                    ("shot_rig_instance", P.n_shots), ("shot_rig_camera", P.n_shots), ("shot_up", P.n_shots), ("shot_up_sigma", P.n_shots),
                    ("shot_pan", P.n_shots), ("shot_pan_sigma", P.n_shots), ("shot_tilt", P.n_shots), ("shot_tilt_sigma", P.n_shots),
                    ("shot_roll", P.n_shots), ("shot_roll_sigma", P.n_shots),
                    ("point_fixed", P.n_points), ("point_prior", P.n_points), ("point_prior_sigma", P.n_points),
                    ("point_prior_has_altitude", P.n_points), ("obs_point", P.n_obs), ("obs_xy", P.n_obs), ("obs_sigma", P.n_obs),
                    ("obs_depth", P.n_obs), ("obs_depth_sigma", P.n_obs), ("obs_depth_radial", P.n_obs)):
        if name in arrays and len(arrays[name]) != n:
            raise ValueError(f"bundle problem: {name!r} has {len(arrays[name])} rows, expected {n}")
    reproj = np.zeros((max(P.n_obs, 1), 3), np.float64)
    arrays["reproj_err"] = reproj
    P.reproj_err = reproj.ctypes.data_as(C.POINTER(C.c_double))
    return P, arrays


SIGNATURES = {
    "osfm_bundle_solve": (C.c_int, [C.c_void_p, C.POINTER(BundleProblem), C.POINTER(BaOptions), C.POINTER(BaReport)]),
    "osfm_ba_options_default": (None, [C.POINTER(BaOptions)]),
    "osfm_ba_solve": (C.c_int, [C.c_void_p, C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(BaReport)]),
    "osfm_ba_shot_order": (C.c_int, [C.POINTER(BaProblem), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}
