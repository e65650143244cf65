"""CPU: the matching oracle against an independent numpy restatement and the reference's own
known-answer recipes (opensfm/test/test_matching.py:50-68, synthetic_generator.py:383-397)."""
import numpy as np
import pytest


def numpy_bf(f1, f2, ratio=0.8):
    """matching.py:723-756 in plain numpy (float32 distances, double ratio test, lowest-index ties)."""
    out = []
    if len(f2) < 2:
        return np.zeros((0, 2), np.int32)
    for i, a in enumerate(f1):
        d = np.sqrt(((f2 - a) ** 2).sum(axis=1, dtype=np.float32)).astype(np.float32)
        order = np.argsort(d, kind="stable")
        m, n = order[0], order[1]
        if float(d[m]) < ratio * float(d[n]):
            out.append((i, m))
    return np.asarray(out, np.int32).reshape(-1, 2)


def reference_descriptors(rng, n, dim=128, non_zeroes=5):
    # opensfm/synthetic_data/synthetic_generator.py:383-397
    d = np.zeros((n, dim))
    for k in range(n):
        for _ in range(non_zeroes):
            d[k, rng.integers(0, dim)] = rng.random() * 255
    return d.round().astype(np.float32)


@pytest.mark.parametrize("n1,n2,seed", [(50, 70, 0), (33, 2, 1), (1, 40, 2), (64, 64, 3)])
def test_oracle_one_way_equals_numpy(oracle_lib, n1, n2, seed):
    rng = np.random.default_rng(seed)
    f1 = reference_descriptors(rng, n1)
    f2 = reference_descriptors(rng, n2)
    f2[: min(n1, n2) // 2] = np.clip(f1[: min(n1, n2) // 2] + rng.integers(-3, 4, (min(n1, n2) // 2, 128)), 0, 255)
    got = oracle_lib.match_brute_force(f1, f2)
    assert np.array_equal(got, numpy_bf(f1, f2))


def test_oracle_symmetric_is_intersection(oracle_lib):
    rng = np.random.default_rng(5)
    f1 = reference_descriptors(rng, 80)
    f2 = np.clip(f1[rng.permutation(80)][:60] + rng.integers(-2, 3, (60, 128)), 0, 255).astype(np.float32)
    a = {tuple(x) for x in numpy_bf(f1, f2)}
    b = {(j, i) for i, j in numpy_bf(f2, f1)}
    want = np.asarray(sorted(a & b), np.int32).reshape(-1, 2)
    assert np.array_equal(oracle_lib.match_brute_force_symmetric(f1, f2), want)
    assert len(want) > 40


def test_noisy_copy_matches_identity(oracle_lib):
    """The reference's known-answer test for its native matcher (test_matching.py:50-68): a noisy
    copy of 1000 descriptors must match the identity permutation."""
    rng = np.random.default_rng(42)
    f1 = np.clip(np.rint(rng.random((1000, 128)) * 255), 0, 255).astype(np.float32)
    f2 = np.clip(f1 + np.rint(rng.normal(0, 2, f1.shape)), 0, 255).astype(np.float32)
    m = oracle_lib.match_brute_force_symmetric(f1, f2)
    assert len(m) == 1000 and np.array_equal(m[:, 0], m[:, 1])


def test_ties_pick_lowest_train_index(oracle_lib):
    f2 = np.zeros((6, 128), np.float32)
    f2[:, 0] = [10, 10, 50, 10, 200, 250]
    f1 = np.zeros((1, 128), np.float32)
    idx, d1, d2, _, _ = oracle_lib.knn2(f1, f2)
    assert idx[0] == 0 and d1[0] == 10 and d2[0] == 10
    # equal best/second -> ratio test fails (strict <)
    assert len(oracle_lib.match_brute_force(f1, f2)) == 0


def test_fewer_than_two_train_rows_gives_nothing(oracle_lib):
    f1 = np.ones((5, 128), np.float32)
    assert len(oracle_lib.match_brute_force(f1, f1[:1])) == 0
    assert len(oracle_lib.match_brute_force_symmetric(f1[:1], f1)) == 0


def test_flann_squared_ratio_variant(oracle_lib):
    # matching.py:695-696: d0 < float32(ratio**2) * d1 on squared distances
    rng = np.random.default_rng(7)
    f1 = reference_descriptors(rng, 40)
    f2 = reference_descriptors(rng, 50)
    got = oracle_lib.match_brute_force(f1, f2, squared=True)
    want = []
    r2 = np.float32(0.8**2)
    for i, a in enumerate(f1):
        s = ((f2 - a) ** 2).sum(axis=1, dtype=np.float32)
        o = np.argsort(s, kind="stable")
        if s[o[0]] < r2 * s[o[1]]:
            want.append((i, o[0]))
    assert np.array_equal(got, np.asarray(want, np.int32).reshape(-1, 2))


def numpy_flann(f1, f2, ratio=0.8):
    """matching.py:683-697 with an exact search, in plain numpy: index over f1, queries f2, squared float32 distances,
    `dists[:, 0] < squared_ratio * dists[:, 1]` evaluated by numpy itself (float32 array * python float)."""
    if len(f1) < 2:
        return np.zeros((0, 2), np.int32)
    results = np.zeros((len(f2), 2), np.int64)
    dists = np.zeros((len(f2), 2), np.float32)
    for j, b in enumerate(f2):
        d = ((f1 - b) ** 2).sum(axis=1, dtype=np.float32).astype(np.float32)
        order = np.argsort(d, kind="stable")
        results[j] = order[:2]
        dists[j] = d[order[:2]]
    squared_ratio = ratio**2
    good = dists[:, 0] < squared_ratio * dists[:, 1]
    return np.asarray(list(zip(results[good, 0], good.nonzero()[0])), np.int32).reshape(-1, 2)


@pytest.mark.parametrize("n1,n2,seed,ratio", [(60, 45, 0, 0.8), (45, 60, 1, 0.8), (2, 30, 2, 0.9), (80, 80, 3, 0.6), (30, 1, 4, 0.8)])
def test_oracle_flann_semantics_equal_numpy(oracle_lib, n1, n2, seed, ratio):
    """match_flann's squared float32 ratio test, query direction (second image against the index of the first) and result order."""
    rng = np.random.default_rng(seed)
    f1 = reference_descriptors(rng, n1)
    f2 = reference_descriptors(rng, n2)
    k = min(n1, n2) // 2
    f2[:k] = np.clip(f1[rng.permutation(n1)[:k]] + rng.integers(-3, 4, (k, 128)), 0, 255)
    assert np.array_equal(oracle_lib.match_flann(f1, f2, ratio), numpy_flann(f1, f2, ratio))
    # symmetric: intersection of both directions, canonically sorted
    a = {tuple(x) for x in numpy_flann(f1, f2, ratio)}
    b = {(j, i) for i, j in numpy_flann(f2, f1, ratio)}
    want = np.asarray(sorted(a & b), np.int32).reshape(-1, 2)
    assert np.array_equal(oracle_lib.match_brute_force_symmetric(f1, f2, ratio, squared=True), want)


def test_gemm_form_descriptor_stage_equals_the_direct_form(oracle_lib):
    """bench.py's second CPU figure (`cpu_baseline.gemm_form`): |a|^2 + |b|^2 - 2 a.b is exact on integer-valued levels, so the blocked
    product returns the direct form's symmetric matches pair for pair -- ragged sizes (not multiples of the 4 x 64 tile), ties included"""
    from opensfm_amd import synthetic

    sc = synthetic.make_matching_scene(5, 333, seed=3)
    d = sc.desc.astype(np.float32)
    d[sc.offsets[1] + 7] = d[sc.offsets[1] + 3]  # two identical rows: equal distances, the lower index must win in both forms
    pairs = synthetic.all_pairs(5)
    a = oracle_lib.match_pairs(d, sc.pts, sc.offsets, pairs, stage=0)
    b = oracle_lib.match_pairs_gemm(d, sc.offsets, pairs)
    assert sum(len(x) for x in a) > 100
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
