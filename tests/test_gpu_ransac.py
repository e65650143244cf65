"""GPU parity: batched F-RANSAC == CPU oracle (identical F bits, inlier mask and iteration count)."""
import numpy as np
import pytest

from opensfm_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize(
    "n,frac,seed", [(8, 1.1, 20), (9, 0.8, 21), (12, 0.7, 22), (14, 1.1, 23), (14, 0.5, 24),  # LMedS branch (n < 15)
                    (15, 0.9, 0), (20, 0.8, 1), (64, 0.7, 2), (65, 0.5, 3), (300, 0.6, 4), (1000, 0.3, 5), (2000, 0.45, 6),
                    (100, 0.0, 7), (4096, 0.5, 8), (8192, 0.55, 13),
                    (12000, 0.6, 14)]  # above the LDS point buffer: correspondences staged in HBM
)
def test_leaf_equals_oracle_bitwise(oracle_lib, gpu_ctx, n, frac, seed):
    from opensfm_amd import matching

    p1, p2, _ = synthetic.make_two_view(n, frac, seed)
    Fo, mo, ito = oracle_lib.find_fundamental_ransac(p1, p2, 0.004, 0.9999)
    Fg, mg = matching.find_fundamental_ransac(p1, p2, 0.004, 0.9999)
    assert (Fo is None) == (Fg is None)
    if Fo is not None:
        assert np.array_equal(Fo.view(np.uint64), Fg.view(np.uint64)), (Fo, Fg)
        assert np.array_equal(mo, mg.ravel().astype(bool))
    assert matching.find_fundamental_ransac.last_iters == ito


def test_robust_match_fundamental_contract(oracle_lib, gpu_ctx):
    """matching.py:780-802: < 8 matches -> empty; returns (F, matches[inliers])."""
    from opensfm_amd import matching

    p1, p2, inl = synthetic.make_two_view(400, 0.6, 17)
    P1 = np.concatenate([p1, np.full((400, 1), 0.004)], axis=1)
    P2 = np.concatenate([p2, np.full((400, 1), 0.004)], axis=1)
    matches = np.stack([np.arange(400), np.arange(400)], axis=1)
    cfg = {"robust_matching_threshold": 0.004}
    F, r = matching.robust_match_fundamental(P1, P2, matches, cfg)
    Fo, mo, _ = oracle_lib.find_fundamental_ransac(p1, p2, 0.004, 0.9999)
    assert np.array_equal(F, Fo) and np.array_equal(r, matches[mo])
    F0, r0 = matching.robust_match_fundamental(P1, P2, matches[:7], cfg)
    assert len(F0) == 0 and len(r0) == 0


def test_degenerate_points_no_model(oracle_lib, gpu_ctx):
    """All correspondences on one line: every subset is collinear -> getSubset fails -> F is None."""
    from opensfm_amd import matching

    t = np.linspace(-0.4, 0.4, 40)
    p1 = np.stack([t, 0.5 * t], axis=1)
    p2 = np.stack([t, -0.25 * t], axis=1)
    Fo, mo, _ = oracle_lib.find_fundamental_ransac(p1, p2)
    Fg, mg = matching.find_fundamental_ransac(p1, p2, 0.004)
    assert Fo is None and Fg is None and not mg.any()


def test_full_pipeline_equals_oracle(oracle_lib, gpu_ctx):
    """match() for every pair (descriptor stage, gates, F-RANSAC): identical inlier sets."""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(14, 500, seed=5, ragged=True)
    pairs = synthetic.all_pairs(14)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    counts, m = matching.match_pairs(store, pairs)
    want = oracle_lib.match_pairs(sc.desc.astype(np.float32), sc.pts, sc.offsets, pairs, stage=1)
    got = matching.split_matches(counts, m)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert sum(len(w) > 0 for w in want) >= 10 and sum(len(w) == 0 for w in want) >= 1


@pytest.mark.parametrize("n_images,n_features,px_noise,seed", [(10, 1500, 0.003, 9), (6, 4000, 1.0 / 2000.0, 3), (8, 2500, 0.002, 4)])
def test_pipeline_long_runs_and_large_pairs(oracle_lib, gpu_ctx, n_images, n_features, px_noise, seed):
    """Pairs the first kernel cannot finish: keypoint noise of the order of the threshold (low inlier ratio: hundreds of iterations
    in the long-run kernel), more than 512 correspondences (handed over from the start) and more than 1024 (read through from HBM)."""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(n_images, n_features, seed=seed, px_noise=px_noise, distractor_frac=0.1)
    pairs = synthetic.all_pairs(n_images)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    c0, _ = matching.match_pairs(store, pairs, robust=False)
    counts, m = matching.match_pairs(store, pairs)
    want = oracle_lib.match_pairs(sc.desc.astype(np.float32), sc.pts, sc.offsets, pairs, stage=1)
    got = matching.split_matches(counts, m)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert (c0 >= 20).sum() >= 5
    if n_features >= 4000:
        assert c0.max() > 1024
    store.close()
