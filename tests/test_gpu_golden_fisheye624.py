"""The Fisheye624 golden pixels of the reference's own test (opensfm/src/geometry/test/camera_test.cc:119-172, tests/golden/
fisheye624_camera_test.json) on the device, through the C ABI: the generic bundle adjustment's projection (osfm_bundle_solve with zero
iterations returns the reprojection errors against the golden pixels) and osfm_pixel_bearings (the golden pixels' bearings are the points'
directions).  The CPU twin, on the oracle: tests/test_golden_fisheye624.py."""
import ctypes as C

import numpy as np
import pytest

from fisheye624_golden import MODEL_FISHEYE624, is_approx, load

pytestmark = pytest.mark.gpu


def _problem(par, points, pixels):
    n = len(points)
    cam = np.zeros((1, 16))
    cam[0, : len(par)] = par
    return {
        "cam_model": np.array([MODEL_FISHEYE624], np.int32), "cam_params": cam, "cam_prior": cam.copy(), "cam_sigma": np.ones((1, 16)),
        "cam_fixed": np.ones(1, np.uint8),
        "rig_camera_pose": np.zeros((1, 6)), "rig_camera_prior": np.zeros((1, 6)), "rig_camera_sigma": np.ones((1, 6)), "rig_camera_fixed": np.ones(1, np.uint8),
        "rig_instance_pose": np.zeros((1, 6)), "shot_rig_instance": np.zeros(1, np.int32), "shot_rig_camera": np.zeros(1, np.int32),
        "shot_camera": np.zeros(1, np.int32),
        "points": np.ascontiguousarray(points, np.float64), "obs_shot": np.zeros(n, np.int32), "obs_point": np.arange(n, dtype=np.int32),
        "obs_xy": np.ascontiguousarray(pixels, np.float64), "obs_sigma": np.ones(n),
    }


@pytest.mark.parametrize("case", [0, 1])
def test_generic_bundle_projection_reproduces_the_references_golden_pixels(gpu_ctx, case):
    from opensfm_amd import bundle

    points, cases, prec = load()
    name, par, want = cases[case]
    g = bundle.bundle_general_arrays(_problem(par, points, want), {"bundle_max_iterations": 0}, ctx=gpu_ctx)
    projected = g["reproj_err"][:, :2] + want  # ComputeReprojectionErrors: projected - observed at sigma 1
    assert is_approx(want, projected, prec), (name, projected, want)
    assert np.abs(g["reproj_err"][:, :2]).max() < 5e-6


@pytest.mark.parametrize("case", [0, 1])
def test_pixel_bearings_of_the_golden_pixels_point_at_the_points(oracle_lib, gpu_ctx, case):
    from opensfm_amd import _lib

    points, cases, _ = load()
    _, par, want = cases[case]
    par16 = np.zeros(16)
    par16[: len(par)] = par
    px = np.ascontiguousarray(want, np.float64)
    out = np.zeros((len(px), 3))
    dp = C.POINTER(C.c_double)
    rc = _lib.load().osfm_pixel_bearings(gpu_ctx.handle, MODEL_FISHEYE624, par16.ctypes.data_as(dp), px.ctypes.data_as(dp), len(px), out.ctypes.data_as(dp))
    assert rc == 0
    d = points / np.linalg.norm(points, axis=1, keepdims=True)
    assert np.abs(out - d).max() < 1e-6
    assert np.abs(out - oracle_lib.pixel_bearings_generic(MODEL_FISHEYE624, par16, px)).max() < 1e-13  # and the oracle's, to rounding
