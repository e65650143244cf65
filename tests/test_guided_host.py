"""The product's guided-matching code (opensfm_amd/csrc/guided_wave.h), compiled for the host with a loop-based wave policy,
against the CPU oracle: explicit masks, epipolar masks, both directions, ties and starved queries."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from opensfm_amd import synthetic

HERE = os.path.dirname(os.path.abspath(__file__))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def build_host():
    src = os.path.join(HERE, "native", "guided_host.cpp")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "guided_host.so")
    deps = [src, os.path.join(HERE, "..", "opensfm_amd", "csrc", "guided_wave.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-o", so, src])
    return C.CDLL(so)


@pytest.fixture(scope="module")
def host():
    return build_host()


def host_match(host, f1, f2, mask=None, b1=None, b2=None, R=None, t=None, threshold=0.0, ratio=0.8, symmetric=True):
    f1, f2 = np.ascontiguousarray(f1, np.float32), np.ascontiguousarray(f2, np.float32)
    cap = max(len(f1), 1)
    out = np.zeros((cap, 2), np.int32)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
    else:
        b1, b2 = np.ascontiguousarray(b1, np.float32), np.ascontiguousarray(b2, np.float32)
        R, t = np.ascontiguousarray(R, np.float64), np.ascontiguousarray(t, np.float64)
    n = host.host_match_guided(_p(f1, C.c_float), len(f1), _p(f2, C.c_float), len(f2), _p(mask, C.c_uint8), _p(b1, C.c_float), _p(b2, C.c_float),
                               _p(R, C.c_double), _p(t, C.c_double), C.c_double(threshold), C.c_double(ratio), int(symmetric),
                               _p(out, C.c_int32), cap)
    return out[:n]


def _rodrigues(r):
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def guided_scene(rng, n, copies=2):
    """a repetitive facade: every descriptor appears `copies` times; bearings of a rigid two-view scene"""
    R = _rodrigues(rng.normal(0, 0.2, 3))
    o = rng.normal(0, 1, 3)
    X = np.c_[rng.uniform(-2, 2, n * copies), rng.uniform(-2, 2, n * copies), rng.uniform(4, 9, n * copies)]
    b1 = X / np.linalg.norm(X, axis=1, keepdims=True)
    Y = (X - o) @ R
    b2 = Y / np.linalg.norm(Y, axis=1, keepdims=True)
    base = rng.integers(0, 255, (n, 128))
    d1 = np.clip(np.concatenate([base] * copies) + rng.integers(-3, 4, (n * copies, 128)), 0, 255).astype(np.float32)
    d2 = np.clip(np.concatenate([base] * copies) + rng.integers(-3, 4, (n * copies, 128)), 0, 255).astype(np.float32)
    perm = rng.permutation(n * copies)
    return d1, d2[perm], b1.astype(np.float32), b2[perm].astype(np.float32), R, o, perm


def test_explicit_masks(host, oracle_lib):
    rng = np.random.default_rng(0)
    sc = synthetic.make_matching_scene(2, 300, seed=9, ragged=True)
    f1 = sc.desc[sc.offsets[0]: sc.offsets[1]].astype(np.float32)
    f2 = sc.desc[sc.offsets[1]: sc.offsets[2]].astype(np.float32)
    for density in (1.0, 0.5, 0.05, 0.004, 0.0):
        mask = (rng.random((len(f1), len(f2))) < density).astype(np.uint8)
        for sym in (True, False):
            for ratio in (0.8, 0.999):
                want = oracle_lib.match_brute_force_masked(f1, f2, mask, ratio, symmetric=sym)
                got = host_match(host, f1, f2, mask=mask, ratio=ratio, symmetric=sym)
                assert np.array_equal(got, want), (density, sym, ratio)
    assert len(oracle_lib.match_brute_force_masked(f1, f2, np.ones((len(f1), len(f2))), 0.8)) > 20


def test_epipolar_masks(host, oracle_lib):
    rng = np.random.default_rng(1)
    for n, thr in ((60, 0.005), (150, 0.02), (97, 0.3)):
        d1, d2, b1, b2, R, o, perm = guided_scene(rng, n)
        mask, _ = oracle_lib.epipolar_mask(b1, b2, R, o, thr)
        for sym in (True, False):
            want = oracle_lib.match_brute_force_masked(d1, d2, mask, 0.8, symmetric=sym)
            got = host_match(host, d1, d2, b1=b1, b2=b2, R=R, t=o, threshold=thr, ratio=0.8, symmetric=sym)
            assert np.array_equal(got, want), (n, thr, sym)
    # the narrow band resolves the repetition: feature i of image 1 <-> the feature of image 2 that shows the same point
    d1, d2, b1, b2, R, o, perm = guided_scene(rng, 100)
    got = host_match(host, d1, d2, b1=b1, b2=b2, R=R, t=o, threshold=0.005, ratio=0.8)
    assert len(got) > 120 and (perm[got[:, 1]] == got[:, 0]).mean() > 0.95


def test_ties_and_tiny_inputs(host, oracle_lib):
    """equal float distances keep the lowest train index (cv2's insertion), whichever lane saw them; 0 / 1 / 2 descriptors"""
    rng = np.random.default_rng(2)
    base = rng.integers(10, 245, (1, 128)).astype(np.float32)
    f2 = np.repeat(base, 200, axis=0)  # 200 identical train descriptors: d0 == d1 -> the ratio test fails for ratio < 1 ...
    f2[137] = np.clip(base + 1, 0, 255)
    f1 = np.repeat(base, 3, axis=0)
    f1[:, 0] += 2  # distance 2 to the 199 identical rows, sqrt(128) to row 137
    ones = np.ones((3, 200), np.uint8)
    for ratio in (0.8, 1.0, 1.5):  # ... and passes for ratio > 1 with the lowest index
        want = oracle_lib.match_brute_force_masked(f1, f2, ones, ratio, symmetric=False)
        got = host_match(host, f1, f2, mask=ones, ratio=ratio, symmetric=False)
        assert np.array_equal(got, want)
    assert len(want) == 3 and (want[:, 1] == 0).all()  # ratio 1.5: d0 == d1 == 2 passes, index 0 is the first of the ties
    m = np.ones((3, 200), np.uint8)
    m[:, :70] = 0  # now the lowest ALLOWED index wins, and it lives on another lane
    want = oracle_lib.match_brute_force_masked(f1, f2, m, 1.5, symmetric=False)
    assert np.array_equal(host_match(host, f1, f2, mask=m, ratio=1.5, symmetric=False), want) and (want[:, 1] == 70).all()
    for n1, n2 in ((0, 5), (5, 0), (1, 1), (1, 2), (2, 1), (2, 2)):
        a, b = rng.integers(0, 255, (n1, 128)).astype(np.float32), rng.integers(0, 255, (n2, 128)).astype(np.float32)
        mm = np.ones((n1, n2), np.uint8)
        want = oracle_lib.match_brute_force_masked(a, b, mm, 0.99, symmetric=True) if n1 and n2 else np.zeros((0, 2), np.int32)
        assert np.array_equal(host_match(host, a, b, mask=mm, ratio=0.99), want)


def composed_match_pairs_guided(product, guided_leaf, descs_of, pts_of):
    """what osfm_match_pairs_guided does per pair, composed on the host from (emulated) leaves: guided descriptor stage, gate, robust
    stage (product.robust_match over whatever leaves the caller patched in), gate"""
    def match_pairs_guided(store, pairs, bearings, relative_poses, config=None, robust=True, cameras=None, timings=None):
        assert config["matcher_type"] == "BRUTEFORCE" and config["symmetric_matching"] is True
        descs, pts = descs_of(store), pts_of(store)
        min_match = int(product._cfg(config, "robust_matching_min_match"))
        out = []
        for (a, b), rel in zip(pairs, relative_poses):
            m = np.zeros((0, 2), np.int32)
            if len(pts[a]) >= 2 and len(pts[b]) >= 2:
                m = np.asarray(guided_leaf(descs[a], descs[b], product._cfg(config, "lowes_ratio"), True, None, bearings[a], bearings[b],
                                           rel.get_R_cam_to_world(), rel.get_origin(), product._cfg(config, "guided_matching_threshold")), np.int32).reshape(-1, 2)
            if len(m) >= min_match and robust:
                if cameras is not None:
                    cam_a, cam_b = cameras[a], cameras[b]
                else:  # the batch takes the fundamental-matrix branch: any undistorted perspective camera selects it in robust_match
                    from types import SimpleNamespace
                    cam_a = cam_b = SimpleNamespace(projection_type="perspective", k1=0.0, k2=0.0, focal=1.0)
                m = np.asarray(product.robust_match(pts[a], pts[b], cam_a, cam_b, m, config), np.int32).reshape(-1, 2)
            out.append(m if len(m) >= min_match else np.zeros((0, 2), np.int32))
        counts = np.asarray([len(m) for m in out], np.int32)
        return counts, (np.concatenate(out) if len(out) else np.zeros((0, 2), np.int32))

    return match_pairs_guided


def test_gpu_guided_tests_logic_on_the_emulation(host, oracle_lib, monkeypatch):
    """The two guided-matching GPU tests (tests/test_gpu_zz_relpose.py) with the C-ABI calls redirected: osfm_match_guided -> this
    file's host emulation, bearings / relative pose -> the relpose host emulation.  Checks the Python glue (match_brute_force* with
    maskij, match_guided, the poses branch of match_images_with_pairs) and the expectations of the GPU tests without a GPU."""
    from opensfm_amd import matching

    import test_gpu_zz_relpose as gpu_tests
    import test_relpose_core_host as rp

    bearings, relpose_pairs = rp._emulated_calls(rp.build_host())

    def guided_leaf(f1, f2, ratio, symmetric, maskij=None, bearings1=None, bearings2=None, R=None, t=None, threshold=0.0, ctx=None):
        if maskij is not None:
            return host_match(host, f1, f2, mask=np.asarray(maskij) != 0, ratio=ratio, symmetric=symmetric)
        return host_match(host, f1, f2, b1=bearings1, b2=bearings2, R=R, t=t, threshold=threshold, ratio=ratio, symmetric=symmetric)

    class FakeStore:
        ctx = None

        def __init__(self, descs, pts, ctx=None):
            self.descs, self.pts = descs, pts
            self.counts = np.asarray([len(d) for d in descs], np.int32)
            self.n_images = len(descs)

        def close(self):
            pass

    match_pairs_guided = composed_match_pairs_guided(matching, guided_leaf, lambda st: st.descs, lambda st: st.pts)

    monkeypatch.setattr(matching, "match_pairs_guided", match_pairs_guided)
    monkeypatch.setattr(matching, "_match_guided_leaf", guided_leaf)
    monkeypatch.setattr(matching, "pixel_bearing_many", bearings)
    monkeypatch.setattr(matching, "relpose_pairs", relpose_pairs)
    monkeypatch.setattr(matching, "DescriptorStore", FakeStore)
    monkeypatch.setattr(matching, "default_context", lambda device=None: None)
    gpu_tests.test_masked_and_guided_leaf(oracle_lib)
    gpu_tests.test_guided_match_images_with_pairs(oracle_lib)
