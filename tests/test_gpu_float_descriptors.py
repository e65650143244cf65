"""Float (non-integer) descriptors -- root-SIFT with feature_root (opensfm/features.py:292-298) -- through the matcher, bit for bit
against the oracle's restatement of cv2's float arithmetic: the fused int8 kernel in FQ mode (8-bit quantisation + rigorous bounds +
float evaluation of what they leave open, match.hip) and the exact float kernel that cross-checks it."""
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


def _run(store, pairs, flags=0, symmetric=True, ratio=0.8):
    """osfm_match_pairs without the robust stage; flags = 1: every pair on the exact float kernel"""
    import ctypes as C

    from opensfm_amd import _lib

    prm = matching.make_params({"lowes_ratio": ratio, "symmetric_matching": symmetric}, robust=False)
    prm.flags |= flags
    res = C.c_void_p()
    tm = _lib.MatchTimings()
    lib = _lib.load()
    pairs = np.ascontiguousarray(pairs, np.int32)
    _lib.check(lib.osfm_match_pairs(store.ctx.handle, store.handle, pairs.ctypes.data_as(C.POINTER(C.c_int32)), len(pairs),
                                    C.byref(prm), C.byref(res), C.byref(tm)))
    c, m = matching._fetch_result(lib, res)
    return c, m, tm


def _hard_descriptors(rng, n_images, n, transform):
    """every image sees the same n base descriptors under noise whose amplitude varies from feature to feature, so that the ratio
    d0 / d1 sweeps continuously through Lowe's threshold; plus exact duplicates and near-duplicates inside an image"""
    base = rng.integers(0, 90, (n, 128)).astype(np.float64)
    out = []
    for _ in range(n_images):
        amp = rng.uniform(0.0, 38.0, (n, 1))
        d = np.clip(base + rng.normal(0, 1, (n, 128)) * amp, 0, 255)
        d[rng.permutation(n)[: n // 50]] = d[rng.permutation(n)[: n // 50]]  # duplicates
        k = rng.permutation(n)[: n // 50]
        d[k] = np.clip(d[(k + 1) % n] + rng.normal(0, 0.02, (len(k), 128)), 0, 255)  # near-duplicates (below the quantisation step)
        out.append(transform(d[rng.permutation(n)]))
    return out


@pytest.mark.parametrize("kind", ["root", "signed", "wide"])
def test_fq_equals_oracle_and_exact_kernel_on_hard_ratios(oracle_lib, gpu_ctx, kind):
    """the three outcomes of the FQ bounds (surely fails / surely passes / float evaluation) all occur and the result is the
    oracle's, and the exact float kernel's, for every pair and both directions"""
    rng = np.random.default_rng(5)
    tf = {"root": lambda d: root_features(d),
          "signed": lambda d: ((d - 40.0) * 0.37).astype(np.float32),                      # negative values, arbitrary scale
          "wide": lambda d: (d * d * 1e-3 + rng.uniform(0, 1e-3, d.shape)).astype(np.float32)}[kind]  # skewed dynamic range
    n_images, n = 5, 700
    descs = _hard_descriptors(rng, n_images, n, tf)
    desc = np.concatenate(descs)
    offsets = np.arange(n_images + 1) * n
    pts = rng.uniform(-0.5, 0.5, (len(desc), 2))
    pairs = synthetic.all_pairs(n_images)
    store = matching.DescriptorStore.from_packed(desc, pts, offsets)
    for symmetric in (True, False):
        c1, m1, tm = _run(store, pairs, 0, symmetric)
        c2, m2, _ = _run(store, pairs, 1, symmetric)
        assert np.array_equal(c1, c2) and np.array_equal(m1, m2)
        assert tm.pairs_exact_path > 0  # some queries needed the float evaluation
        got = matching.split_matches(c1, m1)
        for (i, j), g in zip(pairs, got):
            if symmetric:
                want = sorted(tuple(int(x) for x in r) for r in oracle_lib.match_brute_force_symmetric(descs[i], descs[j], 0.8))
            else:
                want = [tuple(int(x) for x in r) for r in oracle_lib.match_brute_force(descs[i], descs[j], 0.8)]
            assert [tuple(int(x) for x in r) for r in g] == want
        assert 0.15 * n < np.mean(c1) < 0.9 * n
    store.close()


def test_fq_equals_exact_kernel_full_size(gpu_ctx):
    """BASELINE's per-image size (2000 x 128) on root descriptors: fused FQ kernel == exact float kernel for every pair"""
    sc = synthetic.make_matching_scene(16, 2000, seed=33)
    desc = root_features(sc.desc)
    pairs = synthetic.all_pairs(16)
    store = matching.DescriptorStore.from_packed(desc, sc.pts, sc.offsets)
    c1, m1, _ = _run(store, pairs, 0)
    c2, m2, _ = _run(store, pairs, 1)
    assert np.array_equal(c1, c2) and np.array_equal(m1, m2)
    assert c1.sum() > 3000
    store.close()


def test_unquantisable_store_stays_on_the_float_kernel(oracle_lib, gpu_ctx):
    """a value range below 1e-6 cannot carry the bounds (float32 squares underflow): such a store is matched by the exact float kernel"""
    rng = np.random.default_rng(9)
    f1 = (rng.integers(0, 50, (120, 128)) * 1e-9).astype(np.float32)
    f2 = (f1 + rng.integers(-2, 3, f1.shape) * 1e-9).astype(np.float32)[::-1].copy()
    got = matching.match_brute_force_symmetric(f1, f2, {"lowes_ratio": 0.8})
    assert sorted(got) == sorted(tuple(int(x) for x in r) for r in oracle_lib.match_brute_force_symmetric(f1, f2, 0.8))
    assert len(got) > 60
