"""ctypes front-end of the CPU oracle (``oracle/*.c`` -> ``oracle/liboracle.so``).

TEST INFRASTRUCTURE ONLY.  Imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by ``opensfm_amd`` (the product path fails loudly
when its HIP library is missing; it has no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".cc"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


REFERENCE_ROBUST = "/root/reference/opensfm/src/robust"


def build_ref(force: bool = False) -> Optional[str]:
    """oracle/_ref/librobust_ref.so: the reference's own robust_estimator.h / random_sampler.h / scorer.h compiled from where they
    lie under /root/reference around this oracle's model numerics (ref_adapters/robust_ref.cc).  Rebuilt only where the reference
    is mounted; elsewhere the prebuilt file (it travels with the repo snapshot) is used if present.  -> path or None."""
    so = os.path.join(_HERE, "_ref", "librobust_ref.so")
    if os.path.isdir(REFERENCE_ROBUST):
        deps = [os.path.join(_HERE, "ref_adapters", "robust_ref.cc"), os.path.join(_HERE, "relpose_oracle.c")]
        if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["make", "-C", _HERE, "-B", "_ref/librobust_ref.so"], stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


def build_camera_ref(force: bool = False) -> Optional[str]:
    """oracle/_ref/libcamera_ref.so: the reference's scalar camera functions (camera_projections_functions.h,
    camera_distortions_functions.h, newton_raphson.h) compiled from /root/reference (ref_adapters/camera_ref.cc)."""
    so = os.path.join(_HERE, "_ref", "libcamera_ref.so")
    if os.path.isdir(REFERENCE_ROBUST):
        deps = [os.path.join(_HERE, "ref_adapters", "camera_ref.cc"), os.path.join(_HERE, "ref_adapters", "stubs", "Eigen", "Eigen")]
        if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["make", "-C", _HERE, "-B", "_ref/libcamera_ref.so"], stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


def build_camera_ref_eigen(force: bool = False) -> Optional[str]:
    """oracle/_ref/libcamera_ref_eigen.so: the reference's DistoBrown / Disto62 / Disto624 ::Backward compiled from /root/reference against the
    small functional Eigen stand-in (ref_adapters/camera_ref_eigen.cc, ref_adapters/stubs_small_eigen)."""
    so = os.path.join(_HERE, "_ref", "libcamera_ref_eigen.so")
    if os.path.isdir(REFERENCE_ROBUST):
        deps = [os.path.join(_HERE, "ref_adapters", "camera_ref_eigen.cc"), os.path.join(_HERE, "ref_adapters", "stubs_small_eigen", "Eigen", "Eigen")]
        if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["make", "-C", _HERE, "-B", "_ref/libcamera_ref_eigen.so"], stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


def build_hahog_ref(force: bool = False) -> Optional[str]:
    """oracle/_ref/libhahog_ref.so: the reference's features::hahog (features/src/hahog.cc) compiled unmodified from /root/reference
    with the vendored vlfeat sources it calls (ref_adapters/hahog_ref.cc, stand-ins for its pybind11 types under ref_adapters/stubs)."""
    so = os.path.join(_HERE, "_ref", "libhahog_ref.so")
    if os.path.isdir(REFERENCE_ROBUST):
        deps = [os.path.join(_HERE, "ref_adapters", "hahog_ref.cc"), os.path.join(_HERE, "ref_adapters", "covdet_ref.c"),
                os.path.join(_HERE, "ref_adapters", "stubs", "foundation", "python_types.h")]
        if force or not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["make", "-C", _HERE, "-B", "_ref/libhahog_ref.so"], stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


_HAHOGREF = None


def hahog_ref(image: np.ndarray, peak_threshold: float, edge_threshold: float, target_num_features: int):
    """The REFERENCE's pyfeatures.hahog(image, peak_threshold, edge_threshold, target_num_features) (features/src/hahog.cc:125-206 over
    vlfeat's covdet / sift), image float32 in [0, 1]: (points n x 4 [x, y, size, angle in degrees], descriptors n x 128 float32), or
    None where the compiled reference is not available."""
    global _HAHOGREF
    if _HAHOGREF is None:
        so = build_hahog_ref()
        if so is None:
            return None
        _HAHOGREF = C.CDLL(so)
        _HAHOGREF.hahog_ref.restype = C.c_long
        _HAHOGREF.hahog_ref.argtypes = [C.POINTER(C.c_float), C.c_long, C.c_long, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.c_long]
    im = np.ascontiguousarray(image, np.float32)
    cap = max(16, 4 * max(1, int(target_num_features)) if target_num_features else 1 << 20)
    while True:
        pts = np.zeros((cap, 4), np.float32)
        desc = np.zeros((cap, 128), np.float32)
        n = _HAHOGREF.hahog_ref(_p(im, C.c_float), im.shape[0], im.shape[1], peak_threshold, edge_threshold, int(target_num_features),
                                _p(pts, C.c_float), _p(desc, C.c_float), cap)
        if n < 0:
            return None
        if n <= cap:
            return pts[:n].copy(), desc[:n].copy()
        cap = int(n)


_CAMREF = None


def camera_ref_lib() -> Optional[C.CDLL]:
    global _CAMREF
    if _CAMREF is None:
        so = build_camera_ref()
        if so is None:
            return None
        _CAMREF = C.CDLL(so)
    return _CAMREF


def ref_camera(model, par, pts, backward: bool) -> Optional[np.ndarray]:
    """The REFERENCE's PROJ / DISTO ::Forward (pts: camera-frame points, n x 3 -> n x 2) or ::Backward (pts: normalised image
    coordinates, n x 2 -> bearings n x 3), native parameter order.  None for the models that need real Eigen (brown, fisheye62/624)."""
    mid = int(BEARING_MODELS[model] if isinstance(model, str) else model)
    par = np.ascontiguousarray(np.r_[np.asarray(par, np.float64).reshape(-1), np.zeros(16)][:16])
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2 if backward else 3)
    out = np.zeros((len(pts), 3 if backward else 2))
    ok = camera_ref_lib().ref_camera(mid, int(backward), _p(par, C.c_double), _p(pts, C.c_double), len(pts), _p(out, C.c_double))
    return out if ok else None


_CAMREF_EIGEN = None


def ref_camera_eigen_backward(model, par, px) -> Optional[np.ndarray]:
    """The REFERENCE's bearings of the brown / fisheye62 / fisheye624 cameras (DISTO::Backward = Newton on a 2-vector, then PROJ::Backward),
    native parameter order; None when the library is absent or the model is another one."""
    global _CAMREF_EIGEN
    if _CAMREF_EIGEN is None:
        so = build_camera_ref_eigen()
        if so is None:
            return None
        _CAMREF_EIGEN = C.CDLL(so)
    mid = int(BEARING_MODELS[model] if isinstance(model, str) else model)
    par = np.ascontiguousarray(np.r_[np.asarray(par, np.float64).reshape(-1), np.zeros(16)][:16])
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    out = np.zeros((len(px), 3))
    ok = _CAMREF_EIGEN.ref_camera_eigen_backward(mid, _p(par, C.c_double), _p(px, C.c_double), len(px), _p(out, C.c_double))
    return out if ok else None


def ref_camera_jacobian(model, par, X):
    """(Jx n x 2 x 3, Jk n x 2 x 3) through the REFERENCE's ForwardDerivatives of PROJ and Disto24, models 0 / 1 ([k1, k2, focal])."""
    mid = int(BEARING_MODELS[model] if isinstance(model, str) else model)
    par = np.ascontiguousarray(np.r_[np.asarray(par, np.float64).reshape(-1), np.zeros(16)][:16])
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    Jx, Jk = np.zeros((len(X), 2, 3)), np.zeros((len(X), 2, 3))
    ok = camera_ref_lib().ref_camera_jacobian(mid, _p(par, C.c_double), _p(X, C.c_double), len(X), _p(Jx, C.c_double), _p(Jk, C.c_double))
    return (Jx, Jk) if ok else None


_REF = None


def ref_lib() -> Optional[C.CDLL]:
    global _REF
    if _REF is None:
        so = build_ref()
        if so is None:
            return None
        _REF = C.CDLL(so)
    return _REF


def ref_random_samples(n: int, size: int, count: int) -> np.ndarray:
    """The first `count` index samples RandomSamplesGenerator<std::mt19937>(42) of the REFERENCE hands out (random_sampler.h)."""
    out = np.zeros((count, size), np.int32)
    ref_lib().ref_random_samples(int(n), int(size), int(count), _p(out, C.c_int32))
    return out


def ref_ransac_relative_pose(b1, b2, threshold: float, iterations: int = 1000, probability: float = 0.99, use_lo: bool = True,
                             lo_iterations: int = 10):
    """The REFERENCE's Estimate<RansacScoring, MODEL> (robust_estimator.h:37-119) with this oracle's relative-pose numerics as MODEL."""
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    n = len(b1)
    model, lo = np.zeros(12), np.zeros(12)
    inl = np.zeros(max(n, 1), np.int32)
    score = ref_lib().ref_ransac_relative_pose(_p(b1, C.c_double), _p(b2, C.c_double), n, C.c_double(threshold), int(iterations),
                                               C.c_double(probability), int(use_lo), int(lo_iterations), _p(model, C.c_double),
                                               _p(lo, C.c_double), _p(inl, C.c_int32))
    return {"score": score, "model": model.reshape(3, 4), "lo_model": lo.reshape(3, 4), "inliers": inl[:score].copy()}


def ransac_draws(n: int, size: int, count: int) -> np.ndarray:
    """The oracle's own sampler (mt19937(42) + its restated uniform_int_distribution): first `count` samples of `size` out of n."""
    out = np.zeros((count, size), np.int32)
    lib().oracle_random_samples(int(n), int(size), int(count), _p(out, C.c_int32))
    return out


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_det_log.restype = C.c_double
        _LIB.oracle_det_log.argtypes = [C.c_double]
    return _LIB


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def knn2(f1: np.ndarray, f2: np.ndarray):
    f1 = np.ascontiguousarray(f1, np.float32)
    f2 = np.ascontiguousarray(f2, np.float32)
    n1 = len(f1)
    idx = np.empty(n1, np.int32)
    d1 = np.empty(n1, np.float32)
    d2 = np.empty(n1, np.float32)
    s1 = np.empty(n1, np.float32)
    s2 = np.empty(n1, np.float32)
    lib().oracle_knn2_l2(_p(f1, C.c_float), n1, _p(f2, C.c_float), len(f2), f1.shape[1] if n1 else f2.shape[1],
                         _p(idx, C.c_int), _p(d1, C.c_float), _p(d2, C.c_float), _p(s1, C.c_float), _p(s2, C.c_float))
    return idx, d1, d2, s1, s2


def match_brute_force(f1: np.ndarray, f2: np.ndarray, ratio: float = 0.8, squared: bool = False) -> np.ndarray:
    """``matching.py:723-756``; returns array (K, 2) of (queryIdx, trainIdx)."""
    f1 = np.ascontiguousarray(f1, np.float32)
    f2 = np.ascontiguousarray(f2, np.float32)
    n1 = len(f1)
    good = np.empty(max(n1, 1), np.int32)
    dim = f1.shape[1] if f1.ndim == 2 else 128
    lib().oracle_match_brute_force(_p(f1, C.c_float), n1, _p(f2, C.c_float), len(f2), dim,
                                   C.c_double(ratio), int(squared), _p(good, C.c_int))
    good = good[:n1]
    i = np.flatnonzero(good >= 0)
    return np.stack([i, good[i]], axis=1).astype(np.int32)


def match_brute_force_symmetric(fi: np.ndarray, fj: np.ndarray, ratio: float = 0.8, squared: bool = False) -> np.ndarray:
    """``matching.py:759-777``; (K, 2) sorted by (i, j)."""
    fi = np.ascontiguousarray(fi, np.float32)
    fj = np.ascontiguousarray(fj, np.float32)
    cap = max(1, min(len(fi), len(fj)))
    out = np.empty((cap, 2), np.int32)
    dim = fi.shape[1] if fi.ndim == 2 else 128
    n = lib().oracle_match_brute_force_symmetric(_p(fi, C.c_float), len(fi), _p(fj, C.c_float), len(fj), dim,
                                                 C.c_double(ratio), int(squared), _p(out, C.c_int), cap)
    return out[:n].copy()


def match_hamming(f1: np.ndarray, f2: np.ndarray, ratio: float = 0.8, symmetric: bool = False) -> np.ndarray:
    """``match_brute_force[_symmetric]`` on uint8 bit strings (``matching.py:737-740``: cv2 BruteForce-Hamming); (K, 2) sorted by (i, j)."""
    f1 = np.ascontiguousarray(f1, np.uint8)
    f2 = np.ascontiguousarray(f2, np.uint8)
    w = f1.shape[1]
    if symmetric:
        cap = max(1, min(len(f1), len(f2)))
        out = np.empty((cap, 2), np.int32)
        n = lib().oracle_match_hamming_symmetric(_p(f1, C.c_uint8), len(f1), _p(f2, C.c_uint8), len(f2), w, C.c_double(ratio), _p(out, C.c_int), cap)
        return out[:n].copy()
    good = np.empty(max(len(f1), 1), np.int32)
    lib().oracle_match_hamming(_p(f1, C.c_uint8), len(f1), _p(f2, C.c_uint8), len(f2), w, C.c_double(ratio), _p(good, C.c_int))
    good = good[: len(f1)]
    i = np.flatnonzero(good >= 0)
    return np.stack([i, good[i]], axis=1).astype(np.int32)


def match_flann(f1: np.ndarray, f2: np.ndarray, ratio: float = 0.8) -> np.ndarray:
    """``matching.py:683-697`` with an exact search: index over ``f1``, queries ``f2``; (K, 2) of (index row, query row) in query order."""
    f1 = np.ascontiguousarray(f1, np.float32)
    f2 = np.ascontiguousarray(f2, np.float32)
    cap = max(1, len(f2))
    out = np.empty((cap, 2), np.int32)
    n = lib().oracle_match_flann(_p(f1, C.c_float), len(f1), _p(f2, C.c_float), len(f2), 128, C.c_double(ratio), _p(out, C.c_int), cap)
    return out[:n].copy()


def find_fundamental_ransac(p1: np.ndarray, p2: np.ndarray, thr: float = 0.004, conf: float = 0.9999,
                            max_iters: int = 1000) -> Tuple[Optional[np.ndarray], np.ndarray, int]:
    """``cv2.findFundamentalMat(p1, p2, FM_RANSAC, thr, conf)`` restated; returns (F|None, mask, iters)."""
    p1 = np.ascontiguousarray(p1, np.float64)
    p2 = np.ascontiguousarray(p2, np.float64)
    n = len(p1)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    it = C.c_int(0)
    r = lib().oracle_find_fundamental_ransac(_p(p1, C.c_double), _p(p2, C.c_double), n, C.c_double(thr),
                                             C.c_double(conf), max_iters, _p(F, C.c_double), _p(mask, C.c_uint8), C.byref(it))
    if r < 0:
        raise NotImplementedError("n == 7: cv2 returns the stacked 7-point solutions; the reference requires >= 8 matches")
    return (F.reshape(3, 3) if r == 1 else None), mask[:n].astype(bool), it.value


def run_7point(m1: np.ndarray, m2: np.ndarray) -> np.ndarray:
    m1 = np.ascontiguousarray(m1, np.float64)
    m2 = np.ascontiguousarray(m2, np.float64)
    F = np.zeros(27, np.float64)
    n = lib().oracle_run_7point(_p(m1, C.c_double), _p(m2, C.c_double), _p(F, C.c_double))
    return F.reshape(3, 3, 3)[:n]


def det_log(x: float) -> float:
    return lib().oracle_det_log(C.c_double(x))


def update_num_iters(p: float, ep: float, max_iters: int) -> int:
    return lib().oracle_update_num_iters(C.c_double(p), C.c_double(ep), max_iters)


def cvrng_sequence(seed: int, count: int) -> np.ndarray:
    out = np.empty(count, np.uint32)
    lib().oracle_cvrng_sequence(C.c_uint64(seed & (2**64 - 1)), count, _p(out, C.c_uint))
    return out


def match_pairs(desc_f32: np.ndarray, pts: np.ndarray, offsets: np.ndarray, pairs: np.ndarray,
                ratio: float = 0.8, min_match: int = 20, thr: float = 0.004, conf: float = 0.9999,
                stage: int = 1):
    """``matching.py:63-98`` over a packed store; returns list of (K,2) int32 arrays per pair."""
    desc_f32 = np.ascontiguousarray(desc_f32, np.float32)
    pts = np.ascontiguousarray(pts, np.float64)
    offsets = np.ascontiguousarray(offsets, np.int64)
    pairs = np.ascontiguousarray(pairs, np.int32)
    npairs = len(pairs)
    cap = int(np.max(np.diff(offsets))) if len(offsets) > 1 else 1
    counts = np.zeros(max(npairs, 1), np.int32)
    out = np.zeros((max(npairs, 1), cap, 2), np.int32)
    lib().oracle_match_pairs(_p(desc_f32, C.c_float), _p(pts, C.c_double), _p(offsets, C.c_int64), desc_f32.shape[1],
                             _p(pairs, C.c_int), npairs, C.c_double(ratio), min_match, C.c_double(thr), C.c_double(conf),
                             stage, _p(counts, C.c_int), _p(out, C.c_int), cap)
    return [out[p, : counts[p]].copy() for p in range(npairs)]


def match_pairs_gemm(desc_f32: np.ndarray, offsets: np.ndarray, pairs: np.ndarray, ratio: float = 0.8):
    """the descriptor stage (symmetric ratio matches, before the robust stage) of every pair in GEMM form -- integer-valued descriptors
    only; a blocked AVX micro-kernel, OpenMP over pairs (oracle/match_oracle.c): bench.py's second CPU figure.  Returns the list of
    (K, 2) arrays ``match_pairs(..., stage=0)`` returns."""
    desc_f32 = np.ascontiguousarray(desc_f32, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    pairs = np.ascontiguousarray(pairs, np.int32)
    npairs = len(pairs)
    cap = int(np.max(np.diff(offsets))) if len(offsets) > 1 else 1
    counts = np.zeros(max(npairs, 1), np.int32)
    out = np.zeros((max(npairs, 1), cap, 2), np.int32)
    lib().oracle_match_pairs_gemm(_p(desc_f32, C.c_float), _p(offsets, C.c_int64), desc_f32.shape[1], _p(pairs, C.c_int), npairs, C.c_double(ratio),
                                  _p(counts, C.c_int), _p(out, C.c_int), cap)
    return [out[p, : counts[p]].copy() for p in range(npairs)]


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))


def ba_set_parallel(on) -> None:
    """the BA oracle's Schur elimination on all cores (default) or with the serial loops that define its summation order"""
    lib().oracle_ba_set_parallel(int(on))  # 0 serial, 1 parallel (default), 2 parallel with the right-looking skyline factor forced


# ------------------------------------------------------------------------------------------------
# bundle adjustment oracle (oracle/ba_oracle.c)
# ------------------------------------------------------------------------------------------------
class _BaProblem(C.Structure):
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


class _BaOptions(C.Structure):
    _fields_ = [("loss", C.c_int32), ("loss_threshold", C.c_double), ("max_iterations", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_radius", C.c_double), ("verbose", C.c_int32)]


class _BaReport(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("rmse_normalized_initial", C.c_double), ("rmse_normalized_final", C.c_double),
                ("seconds_total", C.c_double), ("seconds_linear_solver", C.c_double), ("cost_history", C.c_double * 256)]


LOSSES = {"TrivialLoss": 0, "SoftLOneLoss": 1, "HuberLoss": 2, "CauchyLoss": 3}


def ba_solve(problem: dict, loss: str = "SoftLOneLoss", loss_threshold: float = 1.0, max_iterations: int = 100,
             function_tolerance: float = 1e-6, gradient_tolerance: float = 1e-10, parameter_tolerance: float = 1e-8,
             verbose: bool = False) -> dict:
    """Solve a flat BA problem (see opensfm_amd.synthetic.make_ba_scene for the dict layout).
    Returns a dict with the optimised arrays and the report.  Inputs are not modified."""
    f64 = lambda k: np.ascontiguousarray(problem[k], np.float64).copy()
    i32 = lambda k: np.ascontiguousarray(problem[k], np.int32)
    cams, poses, pts = f64("cam_params"), f64("shot_pose"), f64("points")
    cam_prior = np.ascontiguousarray(problem.get("cam_prior", problem["cam_params"]), np.float64)
    cam_sigma = np.ascontiguousarray(problem.get("cam_sigma", np.full_like(cams, 0.01)), np.float64)
    cam_fixed = np.ascontiguousarray(problem.get("cam_fixed", np.zeros(len(cams), np.uint8)), np.uint8)
    shot_camera = i32("shot_camera")
    obs_shot, obs_point = i32("obs_shot"), i32("obs_point")
    obs_xy = np.ascontiguousarray(problem["obs_xy"], np.float64)
    obs_sigma = np.ascontiguousarray(problem["obs_sigma"], np.float64)
    reproj = np.zeros((len(obs_shot), 2), np.float64)
    keep = [cams, poses, pts, cam_prior, cam_sigma, cam_fixed, shot_camera, obs_shot, obs_point, obs_xy, obs_sigma, reproj]
    P = _BaProblem()
    P.n_cameras, P.n_shots, P.n_points, P.n_obs = len(cams), len(poses), len(pts), len(obs_shot)
    P.cam_params, P.cam_prior, P.cam_sigma = _p(cams, C.c_double), _p(cam_prior, C.c_double), _p(cam_sigma, C.c_double)
    P.cam_fixed = _p(cam_fixed, C.c_uint8)
    P.shot_pose, P.shot_camera = _p(poses, C.c_double), _p(shot_camera, C.c_int32)
    for key, fld, t, ct in (("shot_fixed", "shot_fixed", np.uint8, C.c_uint8), ("point_fixed", "point_fixed", np.uint8, C.c_uint8),
                            ("shot_gps", "shot_gps", np.float64, C.c_double), ("shot_gps_sigma", "shot_gps_sigma", np.float64, C.c_double),
                            ("shot_up", "shot_up", np.float64, C.c_double), ("shot_up_sigma", "shot_up_sigma", np.float64, C.c_double),
                            ("cam_model", "cam_model", np.int32, C.c_int32), ("cam_ext", "cam_ext", np.float64, C.c_double)):
        if problem.get(key) is not None:
            arr = np.ascontiguousarray(problem[key], t)
            keep.append(arr)
            setattr(P, fld, _p(arr, ct))
    P.points = _p(pts, C.c_double)
    P.obs_shot, P.obs_point = _p(obs_shot, C.c_int32), _p(obs_point, C.c_int32)
    P.obs_xy, P.obs_sigma, P.reproj_err = _p(obs_xy, C.c_double), _p(obs_sigma, C.c_double), _p(reproj, C.c_double)
    O = _BaOptions(LOSSES[loss], loss_threshold, max_iterations, function_tolerance, gradient_tolerance,
                   parameter_tolerance, 1e4, int(verbose))
    R = _BaReport()
    lib().oracle_ba_solve(C.byref(P), C.byref(O), C.byref(R))
    return {
        "cam_params": cams, "shot_pose": poses, "points": pts, "reproj_err": reproj,
        "iterations": R.iterations, "successful_steps": R.successful_steps, "termination": R.termination,
        "initial_cost": R.initial_cost, "final_cost": R.final_cost,
        "rmse_initial": R.rmse_normalized_initial, "rmse_final": R.rmse_normalized_final,
        "seconds_total": R.seconds_total, "seconds_linear_solver": R.seconds_linear_solver,
        "cost_history": np.array(R.cost_history[: R.iterations + 1]),
    }


def bundle_general(problem: dict, loss: str = "SoftLOneLoss", loss_threshold: float = 1.0, max_iterations: int = 100,
                   function_tolerance: float = 1e-6, gradient_tolerance: float = 1e-10, parameter_tolerance: float = 1e-8) -> dict:
    """oracle_bundle_solve (bundle_general_oracle.cc) on the dict form of an osfm_bundle_problem (field list shared with the product's
    ctypes binding: it describes the C struct, not an algorithm)."""
    from opensfm_amd._ba_abi import BundleProblem, fill_bundle_problem

    P, arr = fill_bundle_problem(problem, BundleProblem)
    O = _BaOptions()
    O.loss = {"TrivialLoss": 0, "SoftLOneLoss": 1, "HuberLoss": 2, "CauchyLoss": 3}[loss]
    O.loss_threshold, O.max_iterations = loss_threshold, max_iterations
    O.function_tolerance, O.gradient_tolerance, O.parameter_tolerance, O.initial_radius = function_tolerance, gradient_tolerance, parameter_tolerance, 1e4

    class _Rep(C.Structure):
        _fields_ = [("iterations", C.c_int32), ("successful_steps", C.c_int32), ("termination", C.c_int32), ("initial_cost", C.c_double),
                    ("final_cost", C.c_double), ("cost_history", C.c_double * 256)]

    R = _Rep()
    f = lib().oracle_bundle_solve
    f.restype = C.c_int
    f(C.byref(P), C.byref(O), C.byref(R))
    out = {k: arr[k] for k in ("cam_params", "rig_camera_pose", "rig_instance_pose", "points", "bias") if k in arr}
    out.update({"reproj_err": arr["reproj_err"][: int(P.n_obs)], "iterations": R.iterations, "successful_steps": R.successful_steps,
                "termination": R.termination, "initial_cost": R.initial_cost, "final_cost": R.final_cost,
                "cost_history": np.array(R.cost_history[: min(R.iterations, 255) + 1])})
    return out


def bundle_reprojection(model: int, cam, inst, rcam, use_rig_camera: bool, pt, obs, sigma: float):
    """one reprojection residual (2 or 3 values) and its Jacobian (nres x 31: camera 16 | rig instance 6 | rig camera 6 | point 3), by jets"""
    cam16 = np.zeros(16)
    cam16[: len(cam)] = cam
    res, J = np.zeros(3), np.zeros(93)
    f = lib().oracle_bundle_reprojection
    f.restype = C.c_int
    n = f(int(model), _p(cam16, C.c_double), _p(np.ascontiguousarray(inst, np.float64), C.c_double), _p(np.ascontiguousarray(rcam, np.float64), C.c_double),
          int(use_rig_camera), _p(np.ascontiguousarray(pt, np.float64), C.c_double), _p(np.ascontiguousarray(obs, np.float64), C.c_double),
          C.c_double(sigma), _p(res, C.c_double), _p(J, C.c_double))
    return res[:n].copy(), J[: 31 * n].reshape(n, 31).copy()


def ba_up(pose, up, sigma):
    """Up-vector residual (3) and its Jacobian w.r.t. the shot rotation (3x3) -- absolute_motion_errors.h:12-39."""
    pose, up = np.ascontiguousarray(pose, np.float64), np.ascontiguousarray(up, np.float64)
    r, J = np.zeros(3), np.zeros(9)
    lib().oracle_ba_up(_p(pose, C.c_double), _p(up, C.c_double), C.c_double(sigma), _p(r, C.c_double), _p(J, C.c_double))
    return r, J.reshape(3, 3)


CAMERA_MODELS = {"perspective": 0, "fisheye": 1, "brown": 2, "fisheye_opencv": 3, "fisheye62": 4, "fisheye624": 5, "dual": 6,
                 "radial": 7, "simple_radial": 8}


def ba_project(X, pose, cam, obs, sigma, model="perspective"):
    """One observation: residual (2), Jp (2x3), Jc (2x6: d/d(rx,ry,rz,tx,ty,tz)), Jk (2x3: d/d(k1,k2,f))."""
    X, pose, cam, obs = (np.ascontiguousarray(a, np.float64) for a in (X, pose, cam, obs))
    cam = np.ascontiguousarray(np.r_[cam, np.zeros(16 - len(cam))])  # models >= 2 read up to 16 native parameters
    res, Jp, Jc, Jk = np.zeros(2), np.zeros(6), np.zeros(12), np.zeros(6)
    lib().oracle_ba_project(_p(X, C.c_double), _p(pose, C.c_double), _p(cam, C.c_double), _p(obs, C.c_double),
                            C.c_double(sigma), _p(res, C.c_double), _p(Jp, C.c_double), _p(Jc, C.c_double), _p(Jk, C.c_double),
                            C.c_int(CAMERA_MODELS[model] if isinstance(model, str) else int(model)))
    return res, Jp.reshape(2, 3), Jc.reshape(2, 6), Jk.reshape(2, 3)


def ba_project_intrinsics(X, pose, par, obs, sigma, model):
    """Residual (2) and its Jacobian w.r.t. every intrinsic parameter of a camera of model 2..8 (2 x n_params, native order)."""
    X, pose, obs = (np.ascontiguousarray(a, np.float64) for a in (X, pose, obs))
    n = len(par)
    par = np.ascontiguousarray(np.r_[np.asarray(par, np.float64), np.zeros(16 - n)])
    res, Jk = np.zeros(2), np.zeros((2, 16))
    lib().oracle_ba_project_intrinsics(C.c_int(CAMERA_MODELS[model] if isinstance(model, str) else int(model)), _p(X, C.c_double),
                                       _p(pose, C.c_double), _p(par, C.c_double), _p(obs, C.c_double), C.c_double(sigma), _p(res, C.c_double),
                                       _p(Jk, C.c_double))
    return res, Jk[:, :n].copy()


def ba_loss(loss: str, a: float, s: float):
    out = np.zeros(2)
    lib().oracle_ba_loss(LOSSES[loss], C.c_double(a), C.c_double(s), _p(out, C.c_double))
    return out[0], out[1]


# ------------------------------------------------------------------------------------------------
# tracks oracle (oracle/tracks_oracle.c)
# ------------------------------------------------------------------------------------------------
def tracks(edge_a, edge_b, node_offsets, min_length: int = 2):
    """-> (n_tracks, obs_track, obs_image, obs_feature) -- tracking.py:82-98 + _good_track."""
    ea = np.ascontiguousarray(edge_a, np.int32)
    eb = np.ascontiguousarray(edge_b, np.int32)
    off = np.ascontiguousarray(node_offsets, np.int64)
    n = int(off[-1])
    ot, oi, of = (np.zeros(max(n, 1), np.int32) for _ in range(3))
    nt = C.c_int64(0)
    f = lib().oracle_tracks
    f.restype = C.c_int64
    nobs = f(_p(ea, C.c_int32), _p(eb, C.c_int32), C.c_int64(len(ea)), _p(off, C.c_int64), C.c_int32(len(off) - 1),
             C.c_int32(min_length), _p(ot, C.c_int32), _p(oi, C.c_int32), _p(of, C.c_int32), C.byref(nt))
    return int(nt.value), ot[:nobs], oi[:nobs], of[:nobs]


# ------------------------------------------------------------------------------------------------
# relative pose oracle -- groundwork for the calibrated robust-matching branch (oracle/relpose_oracle.c)
# ------------------------------------------------------------------------------------------------
def essential_five_points(b1, b2) -> np.ndarray:
    """geometry::EssentialFivePoints (essential.h:99-160): 5 bearing pairs -> (k, 3, 3), x2^T E x1 = 0."""
    b1 = np.ascontiguousarray(b1, np.float64).reshape(5, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(5, 3)
    out = np.zeros(90, np.float64)
    n = lib().oracle_essential_five_points(_p(b1, C.c_double), _p(b2, C.c_double), _p(out, C.c_double))
    return out.reshape(10, 3, 3)[:n]


def relative_pose_from_essential(E, b1, b2):
    """RelativePoseFromEssential (relative_pose.h:12-84) -> 3 x 4 [R | t] with x2 ~ R x1 + t, or None."""
    E = np.ascontiguousarray(E, np.float64).reshape(3, 3)
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    RT = np.zeros(12, np.float64)
    ok = lib().oracle_relative_pose_from_essential(_p(E, C.c_double), _p(b1, C.c_double), _p(b2, C.c_double), len(b1), _p(RT, C.c_double))
    return RT.reshape(3, 4) if ok else None


def ransac_relative_pose(b1, b2, threshold: float, iterations: int = 1000, probability: float = 0.99, use_lo: bool = True,
                         lo_iterations: int = 10):
    """pyrobust.ransac_relative_pose(b1, b2, threshold, params, RANSAC) (instanciations.cc:33-48).  The Python caller
    (multiview.relative_pose_ransac, multiview.py:494-517) only sets params.iterations: probability stays 0.99.
    -> dict(score, model, lo_model, inliers, iterations)"""
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    n = len(b1)
    model, lo = np.zeros(12), np.zeros(12)
    inl = np.zeros(max(n, 1), np.int32)
    it = C.c_int(0)
    score = lib().oracle_ransac_relative_pose(_p(b1, C.c_double), _p(b2, C.c_double), n, C.c_double(threshold), iterations,
                                              C.c_double(probability), int(use_lo), lo_iterations, _p(model, C.c_double), _p(lo, C.c_double),
                                              _p(inl, C.c_int32), C.byref(it))
    return {"score": score, "model": model.reshape(3, 4), "lo_model": lo.reshape(3, 4), "inliers": inl[:score].copy(), "iterations": it.value}


def essential_n_points(b1, b2):
    """pygeometry.essential_n_points (geometry/essential.h:162-192) -> list with 0 or 1 matrices (3 x 3)."""
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    E = np.zeros(9)
    ok = lib().oracle_essential_n_points_contiguous(_p(b1, C.c_double), _p(b2, C.c_double), len(b1), _p(E, C.c_double))
    return [E.reshape(3, 3)] if ok else []


def pixel_bearings(model, cam, px) -> np.ndarray:
    """Camera.pixel_bearing_many for the PERSPECTIVE (0) / FISHEYE (1) models, cam = [k1, k2, focal]."""
    cam = np.ascontiguousarray(cam, np.float64)
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    out = np.zeros((len(px), 3))
    lib().oracle_pixel_bearings(int(CAMERA_MODELS[model] if isinstance(model, str) else model), _p(cam, C.c_double), _p(px, C.c_double),
                                len(px), _p(out, C.c_double))
    return out


BEARING_MODELS = dict(CAMERA_MODELS, spherical=9)


def pixel_bearings_generic(model, par, px) -> np.ndarray:
    """Camera.pixel_bearing_many for every projection type; par in the native order [projection][distortion][affine]
    (models 0 / 1: [k1, k2, focal])."""
    par = np.ascontiguousarray(np.r_[np.asarray(par, np.float64).reshape(-1), np.zeros(16)][:16])
    px = np.ascontiguousarray(px, np.float64).reshape(-1, 2)
    out = np.zeros((len(px), 3))
    lib().oracle_pixel_bearings_generic(int(BEARING_MODELS[model] if isinstance(model, str) else model), _p(par, C.c_double),
                                        _p(px, C.c_double), len(px), _p(out, C.c_double))
    return out


def inliers_bearings(b1, b2, R, t, threshold: float = 0.01) -> np.ndarray:
    """matching.compute_inliers_bearings (matching.py:805-844): R, t from the second image to the first."""
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    mask = np.zeros(max(len(b1), 1), np.uint8)
    lib().oracle_inliers_bearings(_p(b1, C.c_double), _p(b2, C.c_double), len(b1), _p(R, C.c_double), _p(t, C.c_double),
                                  C.c_double(threshold), _p(mask, C.c_uint8))
    return mask[: len(b1)].astype(bool)


def glibc_rand(seed: int, count: int) -> np.ndarray:
    out = np.zeros(count, np.int32)
    lib().oracle_glibc_rand(C.c_uint32(seed), count, _p(out, C.c_int32))
    return out


def relpose_cost(b1, b2, params):
    """RelativePoseCost (relative_pose.h:86-147) at params = [angle-axis of R, centre of the second camera]:
    -> residuals (101), Jacobian (101 x 6)."""
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    par = np.ascontiguousarray(params, np.float64)
    res, jac = np.zeros(101), np.zeros(606)
    lib().oracle_relpose_cost(_p(b1, C.c_double), _p(b2, C.c_double), len(b1), _p(par, C.c_double), _p(res, C.c_double), _p(jac, C.c_double))
    return res, jac.reshape(101, 6)


def relative_pose_refinement(RT, b1, b2, iterations: int):
    """pygeometry.relative_pose_refinement (relative_pose.h:149-183): -> (RT refined 3 x 4, iterations, (cost0, cost1))."""
    RT = np.ascontiguousarray(RT, np.float64).reshape(3, 4).copy()
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    costs = np.zeros(2)
    it = lib().oracle_relative_pose_refinement(_p(RT, C.c_double), _p(b1, C.c_double), _p(b2, C.c_double), len(b1), int(iterations),
                                               _p(costs, C.c_double))
    return RT, it, (float(costs[0]), float(costs[1]))


def robust_match_calibrated(p1, p2, cam1, cam2, model1, model2, matches, threshold: float = 0.004, refine_iterations: int = 10):
    """matching.robust_match_calibrated (matching.py:871-903) for PERSPECTIVE / FISHEYE cameras [k1, k2, focal]:
    bearings -> LO-RANSAC (1000 iterations) -> 3 x (inliers at 4, 2, 1 x threshold -> refinement) -> inliers."""
    matches = np.asarray(matches, np.int64).reshape(-1, 2)
    if len(matches) < 8:
        return np.zeros((0, 2), np.int64)
    b1 = pixel_bearings(model1, cam1, np.asarray(p1, np.float64)[matches[:, 0], :2])
    b2 = pixel_bearings(model2, cam2, np.asarray(p2, np.float64)[matches[:, 1], :2])
    r = ransac_relative_pose(b1, b2, threshold, 1000)
    lo = r["lo_model"]
    R, t = lo[:, :3].T.copy(), -lo[:, :3].T @ lo[:, 3]  # multiview.relative_pose_ransac: pose of camera 2 in camera 1
    for relax in (4, 2, 1):
        inl = inliers_bearings(b1, b2, R, t, relax * threshold)
        if inl.sum() < 8:
            return np.zeros((0, 2), np.int64)
        RT = np.c_[R.T, -R.T @ t]  # relative_pose_optimize_nonlinear (multiview.py:541-553)
        RT, _, _ = relative_pose_refinement(RT, b1[inl], b2[inl], refine_iterations)
        R, t = RT[:, :3].T.copy(), -RT[:, :3].T @ RT[:, 3]
    inl = inliers_bearings(b1, b2, R, t, threshold)
    return matches[inl]


def robust_match_calibrated_bearings(b1, b2, threshold: float = 0.004, iterations: int = 1000, probability: float = 0.99,
                                     use_lo: bool = True, lo_iterations: int = 10, refine_iterations: int = 10):
    """robust_match_calibrated (matching.py:871-903) on bearings, every step in C with written-out sums (what the
    product has to match bit for bit).  -> dict(mask, R, t, model, lo_model, score, iterations)."""
    b1 = np.ascontiguousarray(b1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float64).reshape(-1, 3)
    n = len(b1)
    R, t, models, info = np.zeros(9), np.zeros(3), np.zeros(24), np.zeros(2, np.int32)
    mask = np.zeros(max(n, 1), np.uint8)
    lib().oracle_robust_match_calibrated(_p(b1, C.c_double), _p(b2, C.c_double), n, C.c_double(threshold), int(iterations),
                                         C.c_double(probability), int(use_lo), int(lo_iterations), int(refine_iterations),
                                         _p(R, C.c_double), _p(t, C.c_double), _p(mask, C.c_uint8), _p(models, C.c_double),
                                         _p(info, C.c_int32))
    return {"mask": mask[:n].astype(bool), "R": R.reshape(3, 3), "t": t, "model": models[:12].reshape(3, 4),
            "lo_model": models[12:].reshape(3, 4), "score": int(info[0]), "iterations": int(info[1])}


# ------------------------------------------------------------------------------------------------
# guided matching (oracle/guided_oracle.c) -- groundwork, no product counterpart yet
# ------------------------------------------------------------------------------------------------
def epipolar_mask(b1, b2, R, t, threshold: float):
    """matching.compute_inliers_bearing_epipolar (matching.py:847-868): R = pose.get_R_cam_to_world(), t = pose.get_origin() of the
    second camera relative to the first.  -> (mask (n1, n2) bool, angle (n1, n2))."""
    b1 = np.ascontiguousarray(b1, np.float32).reshape(-1, 3)
    b2 = np.ascontiguousarray(b2, np.float32).reshape(-1, 3)
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    ang = np.zeros((len(b1), len(b2)))
    mask = np.zeros((len(b1), len(b2)), np.uint8)
    lib().oracle_epipolar_mask(_p(b1, C.c_float), len(b1), _p(b2, C.c_float), len(b2), _p(R, C.c_double), _p(t, C.c_double),
                               C.c_double(threshold), _p(ang, C.c_double), _p(mask, C.c_uint8))
    return mask.astype(bool), ang


def match_brute_force_masked(f1, f2, mask, ratio: float = 0.8, symmetric: bool = True) -> np.ndarray:
    """match_brute_force[_symmetric](f1, f2, config, maskij) (matching.py:723-777) -> (K, 2) sorted by (i, j)."""
    f1 = np.ascontiguousarray(f1, np.float32)
    f2 = np.ascontiguousarray(f2, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    assert mask.shape == (len(f1), len(f2))
    dim = f1.shape[1]
    if symmetric:
        cap = max(1, min(len(f1), len(f2)))
        out = np.empty((cap, 2), np.int32)
        n = lib().oracle_match_brute_force_symmetric_masked(_p(f1, C.c_float), len(f1), _p(f2, C.c_float), len(f2), dim, C.c_double(ratio),
                                                            _p(mask, C.c_uint8), _p(out, C.c_int), cap)
        return out[:n].copy()
    good = np.empty(max(len(f1), 1), np.int32)
    lib().oracle_match_brute_force_masked(_p(f1, C.c_float), len(f1), _p(f2, C.c_float), len(f2), dim, C.c_double(ratio), _p(mask, C.c_uint8),
                                          _p(good, C.c_int))
    good = good[: len(f1)]
    i = np.flatnonzero(good >= 0)
    return np.stack([i, good[i]], axis=1).astype(np.int32)


def match_words(f1, w1, f2, w2, ratio: float = 0.8, max_checks: int = 20) -> np.ndarray:
    """pyfeatures.match_using_words(f1, words1, f2, words2[:, 0], ratio, checks) (features/src/matching.cc:24-88): (m, 2) int32"""
    f1 = np.ascontiguousarray(f1, np.float32)
    f2 = np.ascontiguousarray(f2, np.float32)
    f1, f2 = f1.reshape(-1, 128), f2.reshape(-1, 128)
    w1 = np.ascontiguousarray(np.asarray(w1, np.int32).reshape(len(f1), -1)) if len(f1) else np.zeros((0, 1), np.int32)
    w2 = np.ascontiguousarray(np.asarray(w2, np.int32).reshape(-1))
    out = np.zeros((max(len(f1), 1), 2), np.int32)
    f = lib().oracle_match_words
    f.restype = C.c_int
    n = f(_p(f1, C.c_float), _p(w1, C.c_int32), len(f1), w1.shape[1] if len(f1) else 0, _p(f2, C.c_float), _p(w2, C.c_int32), len(f2),
          f1.shape[1] if f1.ndim == 2 else 0, C.c_float(ratio), int(max_checks), _p(out, C.c_int32))
    return out[:n].copy()


def match_words_symmetric(f1, w1, f2, w2, ratio: float = 0.8, max_checks: int = 20):
    """matching.match_words_symmetric (matching.py:659-680) as a sorted list of (i, j)"""
    w1 = np.asarray(w1, np.int32).reshape(len(f1), -1) if len(f1) else np.zeros((0, 1), np.int32)
    w2 = np.asarray(w2, np.int32).reshape(len(f2), -1) if len(f2) else np.zeros((0, 1), np.int32)
    ij = {(int(a), int(b)) for a, b in match_words(f1, w1, f2, w2[:, 0], ratio, max_checks)}
    ji = {(int(b), int(a)) for a, b in match_words(f2, w2, f1, w1[:, 0], ratio, max_checks)}
    return sorted(ij & ji)


def vlad_descriptor(features, centers) -> np.ndarray:
    features = np.ascontiguousarray(features, np.float32)
    centers = np.ascontiguousarray(centers, np.float32)
    out = np.zeros(centers.size, np.float32)
    lib().oracle_vlad_descriptor(_p(features, C.c_float), len(features), _p(centers, C.c_float), len(centers), centers.shape[1], _p(out, C.c_float))
    return out


def bow_distances(ref, others) -> np.ndarray:
    """np.fabs(h - h2).sum() for every row h2 of others, in numpy's summation order (pairs_selection.py:690-708)"""
    ref = np.ascontiguousarray(ref, np.float64)
    others = np.ascontiguousarray(others, np.float64).reshape(-1, len(ref))
    out = np.zeros(len(others), np.float64)
    lib().oracle_bow_distances(_p(ref, C.c_double), _p(others, C.c_double), len(others), len(ref), _p(out, C.c_double))
    return out


def vlad_distances(ref, others) -> np.ndarray:
    ref = np.ascontiguousarray(ref, np.float32)
    others = np.ascontiguousarray(others, np.float32).reshape(-1, len(ref))
    out = np.zeros(len(others), np.float64)
    lib().oracle_vlad_distances(_p(ref, C.c_float), _p(others, C.c_float), len(others), len(ref), _p(out, C.c_double))
    return out
