"""opensfm_amd.compat on the device: the calls the reference makes through its pybind11 modules on the hot path, under those modules'
names, against the same oracles as the functions they wrap."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pyfeatures(oracle_lib, gpu_ctx):
    from opensfm_amd.compat import pyfeatures

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hahog_berlin01.npz"))
    pts, desc = pyfeatures.hahog(g["grey"].astype(np.float32) / 255, peak_threshold=1e-5, edge_threshold=10, target_num_features=1500)
    assert np.array_equal(pts[:, :3], g["points"][:, :3]) and np.abs(desc[:64] - g["desc_f32_head"]).max() <= 2e-6
    # match_using_words: the reference's own known-answer recipe (test_matching.py:50-68): a noisy copy matches i <-> i
    rng = np.random.default_rng(3)
    f1 = rng.uniform(0, 255, (300, 128)).astype(np.float32)
    f2 = f1 + rng.normal(0, 1.0, f1.shape).astype(np.float32)
    words = rng.integers(0, 40, (300, 5)).astype(np.int32)
    m = pyfeatures.match_using_words(f1, words, f2, words[:, 0], 0.99, 20)  # as matching.match_words calls it (matching.py:656)
    assert m.shape[1] == 2 and len(m) > 250 and np.array_equal(m[:, 0], m[:, 1])
    want = oracle_lib.match_words(f1, words, f2, words[:, 0].copy(), 0.99, 20)
    assert np.array_equal(np.asarray(m), np.asarray(want).reshape(-1, 2))
    # an image whose mask left no features (matching.match_words can get there): no matches, no exception
    assert pyfeatures.match_using_words(f1[:0], words[:0], f2, words[:, 0], 0.99, 20).shape == (0, 2)
    assert pyfeatures.match_using_words(f1, words, f2[:0], words[:0, 0], 0.99, 20).shape == (0, 2)
    centers = rng.normal(0, 1, (8, 128)).astype(np.float32)
    v = pyfeatures.compute_vlad_descriptor(f1[:50], centers)
    assert v.shape == (8 * 128,)
    with pytest.raises(RuntimeError):
        pyfeatures.compute_vlad_descriptor(f1[:5], np.zeros((0, 128), np.float32))
    d, names = pyfeatures.compute_vlad_distances({"a": v, "b": v + 1, "c": v * 0}, "a", ["b", "c", "missing"])
    assert names == ["b", "c"] and abs(d[0] - np.sqrt(len(v))) < 1e-3


def test_pyrobust(oracle_lib, gpu_ctx):
    from opensfm_amd import matching
    from opensfm_amd.compat import pyrobust

    rng = np.random.default_rng(8)
    X = np.c_[rng.uniform(-2, 2, 200), rng.uniform(-2, 2, 200), rng.uniform(4, 9, 200)]
    R = np.eye(3)
    t = np.array([0.5, 0.1, 0.0])
    b1 = X / np.linalg.norm(X, axis=1, keepdims=True)
    Y = X @ R.T - t
    b2 = Y / np.linalg.norm(Y, axis=1, keepdims=True)
    b2[:20] = rng.normal(0, 1, (20, 3))
    b2[:20] /= np.linalg.norm(b2[:20], axis=1, keepdims=True)
    params = pyrobust.RobustEstimatorParams()
    params.iterations = 1000
    res = pyrobust.ransac_relative_pose(b1, b2, 1 - np.cos(0.004), params, pyrobust.RansacType.RANSAC)
    direct, mask, _ = matching.relpose_pairs(b1, b2, [0, 200], 1 - np.cos(0.004), mode="ransac", iterations=1000, probability=0.99)
    assert np.array_equal(res.lo_model, direct[0]["lo_model"]) and res.inliers_indices == list(np.flatnonzero(mask))
    # against the ORACLE (robust_estimator.h's LO-RANSAC as oracle/relpose_oracle.c restates it, pinned to the reference's template
    # compiled in oracle/_ref): score, both models and the inlier list, bit for bit
    want = oracle_lib.ransac_relative_pose(b1, b2, 1 - np.cos(0.004), iterations=1000, probability=0.99)
    assert res.score == want["score"] and res.inliers_indices == [int(i) for i in want["inliers"]]
    assert np.array_equal(res.model, want["model"]) and np.array_equal(res.lo_model, want["lo_model"])
    assert len(res.inliers_indices) >= 170 and res.lo_model.shape == (3, 4)
    with pytest.raises(RuntimeError):
        pyrobust.ransac_relative_pose(b1, b2[:10], 1e-5, params, pyrobust.RANSAC)


def test_pybundle_and_pysfm_names(gpu_ctx):
    from opensfm_amd import bundle, opensfm_adapter
    from opensfm_amd.compat import pybundle, pysfm

    assert pybundle.BundleAdjuster is bundle.BundleAdjuster
    assert pysfm.BAHelpers.bundle is opensfm_adapter.bundle
