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
        ("pcg_max_iterations", C.c_int32), ("preconditioner", C.c_int32),
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


SIGNATURES = {
    "osfm_ba_options_default": (None, [C.POINTER(BaOptions)]),
    "osfm_ba_solve": (C.c_int, [C.c_void_p, C.POINTER(BaProblem), C.POINTER(BaOptions), C.POINTER(BaReport)]),
    "osfm_ba_shot_order": (C.c_int, [C.POINTER(BaProblem), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}
