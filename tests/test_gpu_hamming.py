"""Binary descriptors (AKAZE MLDB, ORB) on the GPU: match_brute_force's BruteForce-Hamming branch (matching.py:737-740) through the
leaves and through a resident store with the gates and F-RANSAC, bit-exact against the oracle."""
import numpy as np
import pytest

from opensfm_amd import synthetic

pytestmark = pytest.mark.gpu


def _pair(rng, n1, n2, width, flip=0.03):
    a = rng.integers(0, 256, (n1, width)).astype(np.uint8)
    b = rng.integers(0, 256, (n2, width)).astype(np.uint8)
    k = min(n1, n2) // 2
    b[:k] = a[rng.permutation(n1)[:k]] ^ np.packbits(rng.random((k, width * 8)) < flip, axis=1)
    if k >= 4:
        b[k - 2: k] = b[:2]
    return a, b


@pytest.mark.parametrize("n1,n2,width,seed", [(2, 2, 32, 0), (300, 257, 61, 1), (2000, 1999, 32, 2), (70, 3000, 64, 3), (1, 50, 32, 4), (513, 40, 7, 5)])
def test_hamming_leaves_equal_oracle(oracle_lib, gpu_ctx, n1, n2, width, seed):
    from opensfm_amd import matching

    rng = np.random.default_rng(seed)
    a, b = _pair(rng, n1, n2, width)
    for ratio in (0.8, 1.0):
        cfg = {"lowes_ratio": ratio}
        assert matching.match_brute_force(a, b, cfg) == [tuple(int(v) for v in x) for x in oracle_lib.match_hamming(a, b, ratio)]
        assert matching.match_brute_force_symmetric(a, b, cfg) == [tuple(int(v) for v in x) for x in oracle_lib.match_hamming(a, b, ratio, symmetric=True)]


def test_hamming_store_pipeline(oracle_lib, gpu_ctx):
    """match_pairs over a binary store: Hamming descriptor stage, gate, fundamental-matrix RANSAC on the stored keypoints, gate"""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(5, 400, seed=33)  # keypoints + which scene point every feature shows come from the L2 scene
    rng = np.random.default_rng(2)
    # one bit string per scene point, a fresh one per distractor (point id -1), 2 % of the bits flipped per sighting
    n_pts = int(sc.point_ids.max()) + 1
    codes = rng.integers(0, 256, (n_pts, 61)).astype(np.uint8)
    desc = np.where((sc.point_ids >= 0)[:, None], codes[np.maximum(sc.point_ids, 0)], rng.integers(0, 256, (len(sc.point_ids), 61)).astype(np.uint8))
    desc = (desc ^ np.packbits(rng.random((len(desc), 61 * 8)) < 0.02, axis=1)).astype(np.uint8)
    pairs = synthetic.all_pairs(5)
    store = matching.DescriptorStore.from_packed(desc, sc.pts, sc.offsets, hamming=True)
    try:
        for robust in (False, True):
            counts, m = matching.match_pairs(store, pairs, robust=robust)
            got = matching.split_matches(counts, m)
            for (i, j), g in zip(pairs, got):
                di, dj = desc[sc.offsets[i]: sc.offsets[i + 1]], desc[sc.offsets[j]: sc.offsets[j + 1]]
                mm = oracle_lib.match_hamming(di, dj, 0.8, symmetric=True)
                want = np.zeros((0, 2), np.int32)
                if len(mm) >= 20:
                    want = mm
                    if robust:
                        pi, pj = sc.pts[sc.offsets[i]: sc.offsets[i + 1]], sc.pts[sc.offsets[j]: sc.offsets[j + 1]]
                        F, mask, _ = oracle_lib.find_fundamental_ransac(pi[mm[:, 0]], pj[mm[:, 1]], 0.004, 0.9999)
                        want = mm[mask] if F is not None and F[2, 2] != 0.0 and mask.sum() >= 20 else np.zeros((0, 2), np.int32)
                assert np.array_equal(g, want), (i, j, robust, len(g), len(want))
            assert counts.sum() > 200, counts
    finally:
        store.close()
    # matcher_type FLANN on the binary store (round 6; rounds 4-5 refused it): the exact search with FLANN's test, `d0 < lowes_ratio ** 2 * d1` in
    # doubles on the int distances (matching.py:695-696) -- the oracle's Hamming matcher with the squared ratio
    store = matching.DescriptorStore.from_packed(desc, sc.pts, sc.offsets, hamming=True)
    try:
        counts, m = matching.match_pairs(store, pairs, {"matcher_type": "FLANN", "lowes_ratio": 0.9, "symmetric_matching": True,
                                                        "robust_matching_min_match": 20, "robust_matching_threshold": 0.004}, robust=False)
        for (i, j), g in zip(pairs, matching.split_matches(counts, m)):
            di, dj = desc[sc.offsets[i]: sc.offsets[i + 1]], desc[sc.offsets[j]: sc.offsets[j + 1]]
            mm = oracle_lib.match_hamming(di, dj, 0.9 ** 2, symmetric=True)
            assert np.array_equal(g, mm if len(mm) >= 20 else np.zeros((0, 2), np.int32)), (i, j)
    finally:
        store.close()


def test_flann_semantics_on_bit_strings_leaves(oracle_lib, gpu_ctx):
    """match_flann / match_flann_symmetric on uint8 descriptors (the reference builds cv2's LSH index for them, features.py:660-667): the exact
    Hamming search with the squared-ratio test in doubles; one-way matching queries with the SECOND image and lists (index feature, query
    feature) in query order (matching.py:683-697)"""
    from opensfm_amd import matching

    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (300, 32)).astype(np.uint8)
    f1 = base.copy()
    f2 = (base[rng.permutation(300)[:260]] ^ np.packbits(rng.random((260, 256)) < 0.03, axis=1)).astype(np.uint8)
    cfg = {"lowes_ratio": 0.85}
    i1, i2 = matching.build_flann_index(f1, cfg), matching.build_flann_index(f2, cfg)
    assert i1.features.dtype == np.uint8
    got = np.asarray(matching.match_flann(i1, f2, cfg)).reshape(-1, 2)
    want = oracle_lib.match_hamming(f2, f1, 0.85 ** 2, symmetric=False)[:, ::-1]  # (query in f2, target in f1) -> (index feature, f2 feature)
    assert len(got) > 150 and np.array_equal(got, want)
    gs = np.asarray(matching.match_flann_symmetric(f1, i1, f2, i2, cfg)).reshape(-1, 2)
    assert len(gs) > 150 and np.array_equal(gs, oracle_lib.match_hamming(f1, f2, 0.85 ** 2, symmetric=True))
    # a lone query of the second image, and a train set of one row (knnSearch has no second neighbour: no match)
    assert np.array_equal(np.asarray(matching.match_flann(i1, f2[:1], cfg)).reshape(-1, 2), oracle_lib.match_hamming(f2[:1], f1, 0.85 ** 2)[:, ::-1])
    assert len(matching.match_flann(matching.build_flann_index(f1[:1], cfg), f2, cfg)) == 0
