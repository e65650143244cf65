"""Fisheye624 golden vectors held by the reference's own test (opensfm/src/geometry/test/camera_test.cc:119-172: two distortion
vectors -- thin prism alone, tangential + thin prism --, three points, Eigen isApprox 1e-5) against the CPU oracle: the forward projection
of bundle_general_oracle.cc (the BA residual), the reference's own functors compiled from /root/reference where they are present
(oracle/_ref/libcamera_ref.so), and the backward map of relpose_oracle.c (a golden pixel's bearing is the point's direction)."""
import numpy as np
import pytest

from fisheye624_golden import MODEL_FISHEYE624, is_approx, load


@pytest.mark.parametrize("case", [0, 1])
def test_oracle_projection_reproduces_the_references_golden_pixels(oracle_lib, case):
    points, cases, prec = load()
    name, par, want = cases[case]
    got = np.zeros_like(want)
    for i, X in enumerate(points):  # identity pose, no rig camera: the residual against observation (0, 0) is the projection itself
        res, _ = oracle_lib.bundle_reprojection(MODEL_FISHEYE624, par, np.zeros(6), np.zeros(6), False, X, np.zeros(2), 1.0)
        got[i] = res[:2]
    assert is_approx(want, got, prec), (name, got, want)
    assert np.abs(got - want).max() < 5e-6  # (the literals carry eight or nine significant digits)


@pytest.mark.parametrize("case", [0, 1])
def test_oracle_bearings_of_the_golden_pixels_point_at_the_points(oracle_lib, case):
    points, cases, _ = load()
    _, par, want = cases[case]
    b = oracle_lib.pixel_bearings_generic(MODEL_FISHEYE624, par, want)
    d = points / np.linalg.norm(points, axis=1, keepdims=True)
    assert np.abs(b - d).max() < 1e-6, (b, d)  # Newton's undistortion stops at 1e-6 (camera_distortions_functions.h:776-799)


@pytest.mark.parametrize("case", [0, 1])
def test_compiled_reference_undistortion_agrees_where_it_is_built(oracle_lib, case):
    """the reference's own Disto624::Backward + FisheyeProjection::Backward (oracle/_ref/libcamera_ref_eigen.so, compiled from
    /root/reference where that exists) on the golden pixels: the points' directions, and the oracle's bearings bit for bit"""
    points, cases, _ = load()
    _, par, want = cases[case]
    got = oracle_lib.ref_camera_eigen_backward(MODEL_FISHEYE624, par, want)
    if got is None:
        pytest.skip("oracle/_ref/libcamera_ref_eigen.so is not built here (no /root/reference)")
    d = points / np.linalg.norm(points, axis=1, keepdims=True)
    assert np.abs(got - d).max() < 1e-6
    assert np.abs(got - oracle_lib.pixel_bearings_generic(MODEL_FISHEYE624, par, want)).max() < 1e-13
