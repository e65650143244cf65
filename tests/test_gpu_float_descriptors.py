"""Float (non-integer) descriptors -- root-SIFT with feature_root (opensfm/features.py:292-298) -- through the matcher: the exact float
kernel against the oracle's restatement of cv2's float arithmetic, bit for bit."""
import numpy as np
import pytest

from opensfm_amd import matching, synthetic

pytestmark = pytest.mark.gpu


def root_features(desc_u8):
    """features.root_feature: L1-normalise, square root (then unit L2 norm)"""
    d = desc_u8.astype(np.float32)
    d /= np.maximum(d.sum(1, keepdims=True), 1e-7)
    return np.sqrt(d).astype(np.float32)


@pytest.mark.parametrize("n1,n2", [(300, 280), (700, 33), (2, 5), (1000, 1000), (1, 50)])
@pytest.mark.parametrize("flann", [False, True])
def test_leaf_matchers_on_root_descriptors(oracle_lib, gpu_ctx, n1, n2, flann):
    rng = np.random.default_rng(n1 * 7 + n2)
    base = rng.integers(0, 120, (max(n1, n2), 128))
    f1 = root_features(np.clip(base[:n1] + rng.integers(-3, 4, (n1, 128)), 0, 255))
    f2 = root_features(np.clip(base[:n2] + rng.integers(-3, 4, (n2, 128)), 0, 255))
    if n2 > 12:
        f2[11] = f2[4]  # exact duplicates: the lowest index wins
    cfg = {"lowes_ratio": 0.8, "matcher_type": "FLANN" if flann else "BRUTEFORCE"}
    if flann:
        got = matching.match_flann(matching.build_flann_index(f1, cfg), f2, cfg)
        want = [(int(a), int(b)) for a, b in oracle_lib.match_flann(f1, f2, 0.8)]
        assert [tuple(int(x) for x in g) for g in got] == want
    else:
        got = matching.match_brute_force(f1, f2, cfg)
        assert [tuple(int(x) for x in g) for g in got] == [tuple(int(x) for x in r) for r in oracle_lib.match_brute_force(f1, f2, 0.8)]
        sym = matching.match_brute_force_symmetric(f1, f2, cfg)
        assert sorted(sym) == sorted(tuple(int(x) for x in r) for r in oracle_lib.match_brute_force_symmetric(f1, f2, 0.8))
        if min(n1, n2) >= 280:
            assert len(sym) > 100


def test_full_pipeline_on_root_descriptors(oracle_lib, gpu_ctx):
    """match() for every pair of a float store: descriptor stage (float kernel), gates, F-RANSAC -- identical inlier sets"""
    sc = synthetic.make_matching_scene(10, 400, seed=6, ragged=True)
    desc = root_features(sc.desc)
    pairs = synthetic.all_pairs(10)
    store = matching.DescriptorStore.from_packed(desc, sc.pts, sc.offsets)
    counts, m = matching.match_pairs(store, pairs)
    want = oracle_lib.match_pairs(desc, sc.pts, sc.offsets, pairs, stage=1)
    got = matching.split_matches(counts, m)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert sum(len(w) > 0 for w in want) >= 5
    store.close()
