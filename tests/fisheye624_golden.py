"""The Fisheye624 golden pixels the reference holds as literals (opensfm/src/geometry/test/camera_test.cc:119-172), shared by the CPU
test (oracle) and the GPU test (product)."""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_FISHEYE624 = 5  # OSFM_CAMERA_FISHEYE624, native order [k1..k6 p1 p2 s0 s1 s2 s3 | focal ar cx cy]


def load():
    g = json.load(open(os.path.join(HERE, "golden", "fisheye624_camera_test.json")))
    cam = g["camera"]
    f = cam["width"] / (2 * math.pi)
    cases = []
    for c in g["cases"]:
        par = np.array(list(c["distortion_k1_k6_p1_p2_s0_s3"]) + [f, cam["aspect_ratio"]] + list(cam["principal_point"]), np.float64)
        cases.append((c["name"], par, np.array(c["reference_pixels"], np.float64)))
    return np.array(g["points"], np.float64), cases, float(g["isapprox_precision"])


def is_approx(a: np.ndarray, b: np.ndarray, prec: float) -> bool:
    """Eigen's DenseBase::isApprox: ||a - b||_F <= prec * min(||a||_F, ||b||_F)"""
    return float(np.linalg.norm(a - b)) <= prec * min(float(np.linalg.norm(a)), float(np.linalg.norm(b)))
