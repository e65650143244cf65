"""matching_use_segmentation on the GPU: descriptors with the 129th column feature_loading.py:123-155 appends (35 x the segmentation
label) through the store, the leaves and the guided stage, against the oracle on the same 129-column float32 rows (oracle/match_oracle.c
l2sqr_f32: cv2's normL2Sqr_ adds a trailing element after the vector blocks, d += t * t).  Bit-exact: identical match lists."""
import numpy as np
import pytest

from opensfm_amd import synthetic

pytestmark = pytest.mark.gpu


def _with_labels(rng, desc, classes, mult=35.0):
    lab = rng.integers(0, classes, len(desc)).astype(np.float32)
    return np.concatenate([desc.astype(np.float32), (mult * lab)[:, None]], axis=1).astype(np.float32)


@pytest.mark.parametrize("n1,n2,classes,seed", [(300, 280, 4, 0), (1000, 777, 20, 1), (64, 2000, 256, 2), (33, 31, 2, 3)])
def test_leaf_with_segmentation_column(oracle_lib, gpu_ctx, n1, n2, classes, seed):
    """classes = 256: label differences up to 255, (35 * 255)^2 = 79.7 M is not a float32 integer -- the product must round where the
    oracle rounds (the product of the difference first, then the sum)"""
    from opensfm_amd import matching

    rng = np.random.default_rng(seed)
    f1 = synthetic._hahog_like(rng, n1)
    f2 = synthetic._hahog_like(rng, n2)
    k = min(n1, n2) // 2
    f2[:k] = np.clip(f1[rng.permutation(n1)[:k]] + np.rint(rng.normal(0, 3, (k, 128))), 0, 255)
    f2[k: k + k // 2] = f2[: k // 2]  # duplicates that only the label can tell apart
    a, b = _with_labels(rng, f1, classes), _with_labels(rng, f2, classes)
    for ratio in (0.8, 0.95):
        cfg = {"lowes_ratio": ratio}
        assert matching.match_brute_force_symmetric(a, b, cfg) == [tuple(int(v) for v in x) for x in oracle_lib.match_brute_force_symmetric(a, b, ratio)]
        assert matching.match_brute_force(a, b, cfg) == [tuple(int(v) for v in x) for x in oracle_lib.match_brute_force(a, b, ratio)]
    # the column changes the result
    assert matching.match_brute_force_symmetric(a, b, {"lowes_ratio": 0.8}) != matching.match_brute_force_symmetric(a[:, :128], b[:, :128], {"lowes_ratio": 0.8})


def test_store_with_segmentation_column_pipeline(oracle_lib, gpu_ctx):
    """match_pairs over a resident store whose descriptors carry the column: descriptor stage + gates + F-RANSAC, as the oracle's
    match_pairs on the 129-column rows; the same store without labels gives a different graph"""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(6, 500, seed=21)
    rng = np.random.default_rng(5)
    # labels drawn per feature: two thirds of the true matches sit across classes.  The column is 350 x the label here (the store takes
    # any float column): at the reference's 35 the penalty (1225) is far below this scene's second-nearest distances and changes nothing
    desc = _with_labels(rng, sc.desc, 3, mult=350.0)
    pairs = synthetic.all_pairs(6)
    store = matching.DescriptorStore.from_packed(desc, sc.pts, sc.offsets)
    try:
        for robust in (False, True):
            counts, m = matching.match_pairs(store, pairs, robust=robust)
            want = oracle_lib.match_pairs(desc, sc.pts, sc.offsets, pairs, stage=1 if robust else 0)
            got = matching.split_matches(counts, m)
            assert [len(g) for g in got] == [len(w) for w in want]
            assert all(np.array_equal(g, w) for g, w in zip(got, want))
        assert counts.sum() > 100
    finally:
        store.close()
    plain = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    try:
        c0, _ = matching.match_pairs(plain, pairs, robust=False)
        c1 = np.asarray([len(w) for w in oracle_lib.match_pairs(desc, sc.pts, sc.offsets, pairs, stage=0)])
        assert not np.array_equal(c0, c1)
    finally:
        plain.close()
    with pytest.raises(ValueError):
        matching.DescriptorStore([desc[:10], sc.desc[10:20]], [sc.pts[:10], sc.pts[10:20]])
    with pytest.raises(matching.OsfmError):  # root descriptors + labels: the reference raises too (feature_loading.py:126-133)
        matching.DescriptorStore([np.c_[np.sqrt(sc.desc[:10] / 255.0), np.ones(10)].astype(np.float32)], [sc.pts[:10]])


def test_guided_stage_with_segmentation_column(oracle_lib, gpu_ctx):
    """osfm_match_pairs_guided on a store with the column: the masked candidates' distances include the label term"""
    import test_guided_host as gh
    from opensfm_amd import matching

    rng = np.random.default_rng(12)
    descs, bears, pairs, rels, want = [], [], [], [], []
    for k, n in enumerate((150, 400)):
        d1, d2, b1, b2, R, o, perm = gh.guided_scene(rng, n // 2)
        lab1 = rng.integers(0, 3, len(d1))
        lab2 = np.where(rng.random(len(d2)) < 0.8, lab1[perm] if len(perm) == len(d2) else rng.integers(0, 3, len(d2)), rng.integers(0, 3, len(d2)))
        a = np.concatenate([d1.astype(np.float32), (35.0 * lab1)[:, None]], axis=1).astype(np.float32)
        b = np.concatenate([d2.astype(np.float32), (35.0 * lab2)[:, None]], axis=1).astype(np.float32)
        descs += [a, b]
        bears += [b1, b2]
        pairs.append((2 * k, 2 * k + 1))
        rels.append(np.concatenate([np.asarray(R).reshape(9), np.asarray(o).reshape(3)]))
        mask, _ = oracle_lib.epipolar_mask(b1, b2, R, o, 0.02)
        want.append(oracle_lib.match_brute_force_masked(a, b, mask, 0.8, symmetric=True))
    store = matching.DescriptorStore(descs, [np.zeros((len(d), 2)) for d in descs])
    try:
        counts, m = matching.match_pairs_guided(store, np.asarray(pairs, np.int32), bears, rels, {"guided_matching_threshold": 0.02, "lowes_ratio": 0.8},
                                                robust=False)
        got = matching.split_matches(counts, m)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        assert counts.sum() > 50
    finally:
        store.close()
