"""GPU parity: fused MFMA matcher == CPU oracle, bit for bit, through the C ABI."""
import numpy as np
import pytest

from opensfm_amd import synthetic

pytestmark = pytest.mark.gpu


def _as_set(m):
    return {tuple(int(v) for v in x) for x in m}


def _rand_desc(rng, n, hi=256):
    return rng.integers(0, hi, (n, 128)).astype(np.float32)


@pytest.mark.parametrize(
    "n1,n2,seed",
    [(2, 2, 0), (3, 40, 1), (31, 33, 2), (32, 32, 3), (64, 257, 4), (300, 100, 5), (1000, 777, 6), (2000, 2000, 7),
     (129, 4096, 8), (4096, 300, 9), (5000, 6000, 10), (8192, 777, 11), (300, 8000, 12)],
)
def test_leaf_symmetric_equals_oracle(oracle_lib, gpu_ctx, n1, n2, seed):
    from opensfm_amd import matching

    rng = np.random.default_rng(seed)
    f1 = synthetic._hahog_like(rng, n1).astype(np.float32)
    f2 = synthetic._hahog_like(rng, n2).astype(np.float32)
    k = min(n1, n2) // 2
    f2[:k] = np.clip(f1[rng.permutation(n1)[:k]] + np.rint(rng.normal(0, 3, (k, 128))), 0, 255)
    cfg = {"lowes_ratio": 0.8}
    got = matching.match_brute_force_symmetric(f1, f2, cfg)
    want = oracle_lib.match_brute_force_symmetric(f1, f2)
    assert got == [tuple(int(v) for v in x) for x in want]
    got1 = matching.match_brute_force(f1, f2, cfg)
    want1 = oracle_lib.match_brute_force(f1, f2)
    assert got1 == [tuple(int(v) for v in x) for x in want1]
    if k >= 8:
        assert len(want) >= k // 2


def test_ties_and_duplicates(oracle_lib, gpu_ctx):
    """Duplicate train rows: lowest index wins and the ratio test (strict <) fails on equal distances."""
    from opensfm_amd import matching

    rng = np.random.default_rng(11)
    f1 = _rand_desc(rng, 100, 64)
    f2 = np.concatenate([f1[:50], f1[:50], _rand_desc(rng, 30, 64)])  # every query has two exact copies
    f2 = f2[rng.permutation(len(f2))]
    for ratio in (0.8, 1.0, 1.5):
        got = matching._match_leaf(f1, f2, ratio, False)
        want = oracle_lib.match_brute_force(f1, f2, ratio)
        assert np.array_equal(got, want)
        gots = matching._match_leaf(f1, f2, ratio, True)
        wants = oracle_lib.match_brute_force_symmetric(f1, f2, ratio)
        assert np.array_equal(gots, wants)


def test_extreme_range_takes_exact_path(oracle_lib, gpu_ctx):
    """Descriptors at 0/255: d^2 up to 128*255^2 > 2^22, where sqrtf() merges neighbouring integers;
    the flagged pairs must be re-run on the exact float-key kernel and still equal the oracle."""
    from opensfm_amd import matching

    rng = np.random.default_rng(12)
    f1 = rng.integers(0, 9, (200, 128)).astype(np.float32)
    f2 = (255 - rng.integers(0, 9, (200, 128))).astype(np.float32)  # every d^2 ~ 7.8e6 > 2^22
    store = matching.DescriptorStore([f1, f2], [np.zeros((200, 2)), np.zeros((200, 2))])
    from opensfm_amd._lib import MatchTimings

    tm = MatchTimings()
    counts, m = matching.match_pairs(store, np.array([[0, 1]], np.int32), {"lowes_ratio": 0.999}, robust=False, timings=tm)
    want = oracle_lib.match_brute_force_symmetric(f1, f2, 0.999)
    assert tm.pairs_exact_path == 1
    assert np.array_equal(m, want)


def test_descriptor_value_validation(gpu_ctx):
    """non-integer descriptors are matched in float (tests/test_gpu_float_descriptors.py); a constant float store has no value range
    to quantise and every distance 0: no match, as cv2 (0 < ratio * 0 is false); non-finite values are rejected; a uint8 array is a bit
    string matched by Hamming distance (matching.py:737-740; tests/test_gpu_hamming.py) -- up to 64 bytes wide, wider ones are refused"""
    from opensfm_amd import matching
    from opensfm_amd._lib import OsfmError

    f = np.full((10, 128), 0.5, np.float32)
    assert matching.match_brute_force_symmetric(f, f, {}) == []
    g = f.copy()
    g[3, 7] = np.nan
    with pytest.raises(OsfmError):
        matching.match_brute_force_symmetric(g, f, {})
    with pytest.raises(OsfmError):
        matching.match_brute_force(f.astype(np.uint8), f.astype(np.uint8), {})  # 128 bytes per descriptor


def test_batched_descriptor_stage_equals_oracle_ragged(oracle_lib, gpu_ctx):
    """All pairs of a ragged 10-image scene, descriptor stage only (ratio + mutual)."""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(10, 600, seed=21, ragged=True)
    pairs = synthetic.all_pairs(10)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    counts, m = matching.match_pairs(store, pairs, robust=False)
    want = oracle_lib.match_pairs(sc.desc.astype(np.float32), sc.pts, sc.offsets, pairs, stage=0)
    got = matching.split_matches(counts, m)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert sum(len(w) for w in want) > 500


def test_fused_equals_exact_kernel_full_size(gpu_ctx):
    """Size-independent cross-check at BASELINE's per-image size (2000 x 128): the MFMA kernel and
    the VALU float-key kernel must produce identical match lists for every pair."""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(24, 2000, seed=33)
    pairs = synthetic.all_pairs(24)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    c1, m1 = matching.match_pairs(store, pairs, robust=False)
    prm_cfg = {}
    import ctypes as C

    from opensfm_amd import _lib

    prm = matching.make_params(prm_cfg, robust=False)
    prm.flags = 1
    res = C.c_void_p()
    lib = _lib.load()
    _lib.check(lib.osfm_match_pairs(store.ctx.handle, store.handle, pairs.ctypes.data_as(C.POINTER(C.c_int32)), len(pairs),
                                    C.byref(prm), C.byref(res), None))
    n, tot = lib.osfm_result_num_pairs(res), lib.osfm_result_total_matches(res)
    c2 = np.zeros(n, np.int32)
    m2 = np.zeros((max(tot, 1), 2), np.int32)
    _lib.check(lib.osfm_result_fetch(res, c2.ctypes.data_as(C.POINTER(C.c_int32)), m2.ctypes.data_as(C.POINTER(C.c_int32))))
    lib.osfm_result_destroy(res)
    assert np.array_equal(c1, c2)
    assert np.array_equal(m1, m2[:tot])
    assert c1.sum() > 5000
    # mutual matches are a partial injection: each feature appears at most once per pair
    for g in matching.split_matches(c1, m1):
        assert len(set(g[:, 0])) == len(g) and len(set(g[:, 1])) == len(g)


def test_second_best_in_the_same_class_as_the_best(oracle_lib, gpu_ctx):
    """Rule-26 style forced branch for the v2 kernel: the true second-nearest neighbour sits in the
    winner's own class (same column modulo 32 / same lane's rows), so the class-level bound passes
    the ratio test while the exact second must make it FAIL (or pass, for the control half)."""
    from opensfm_amd import matching

    rng = np.random.default_rng(77)
    n1, n2 = 200, 420
    f1 = rng.integers(60, 196, (n1, 128)).astype(np.float32)
    f2 = rng.integers(0, 30, (n2, 128)).astype(np.float32)  # far background
    for i in range(96):
        a = (i * 3) % 32 + 32 * (i % 3)  # winner column
        b = a + 32 * (3 + i % 5)  # same class (a mod 32), different tile
        near = f1[i] + rng.integers(-2, 3, 128)
        f2[a] = np.clip(near, 0, 255)
        eps = 2 if i % 2 == 0 else 40  # even: second as close as the first (ratio fails); odd: clearly farther
        f2[b] = np.clip(f1[i] + rng.integers(-eps, eps + 1, 128), 0, 255)
    for ratio in (0.8, 0.95):
        for sym in (False, True):
            got = matching._match_leaf(f1, f2, ratio, sym)
            want = (oracle_lib.match_brute_force_symmetric if sym else oracle_lib.match_brute_force)(f1, f2, ratio)
            assert np.array_equal(got, want)
            got_t = matching._match_leaf(f2, f1, ratio, sym)  # transposed roles: column-direction lazy path
            want_t = (oracle_lib.match_brute_force_symmetric if sym else oracle_lib.match_brute_force)(f2, f1, ratio)
            assert np.array_equal(got_t, want_t)
    # sanity: the construction really produces both outcomes
    m = oracle_lib.match_brute_force(f1, f2, 0.8)
    assert 10 < len(m) < 96


def test_more_than_4096_features_full_pipeline(oracle_lib, gpu_ctx):
    """Images above 4096 features (OpenSfM's HAHOG default asks for >= 4000 per image): the v4 keys carry no tile
    index, the per-feature LDS state holds up to OSFM_MAX_FEATURES = 16000."""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(5, 5200, seed=19)
    pairs = synthetic.all_pairs(5)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    counts, m = matching.match_pairs(store, pairs)
    want = oracle_lib.match_pairs(sc.desc.astype(np.float32), sc.pts, sc.offsets, pairs)
    got = matching.split_matches(counts, m)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert counts.sum() > 1000


def test_panorama_sized_images_full_pipeline(oracle_lib, gpu_ctx):
    """feature_min_frames_panorama = 16000 (config.py:31): images at OSFM_MAX_FEATURES -- 160 KiB of LDS in the matcher, the RANSAC
    correspondences of the dense pair in HBM (more matches than its LDS point buffer holds) -- against the oracle"""
    from opensfm_amd import matching

    rng = np.random.default_rng(3)
    n = 16000
    base = rng.integers(0, 140, (n, 128))
    d0 = np.clip(base + rng.integers(-4, 5, base.shape), 0, 255).astype(np.uint8)
    d1 = np.clip(base + rng.integers(-4, 5, base.shape), 0, 255).astype(np.uint8)   # 16000 true matches with image 0
    d2 = np.clip(np.concatenate([base[:3000], rng.integers(0, 140, (9500, 128))]) + rng.integers(-4, 5, (12500, 128)), 0, 255).astype(np.uint8)
    p1, p2, _ = synthetic.make_two_view(n, 0.97, 5)
    pts = np.concatenate([p1, p2, np.concatenate([p1[:3000], rng.uniform(-0.4, 0.4, (9500, 2))])])  # image 2 = view 1 again: a proper pair with image 1
    desc = np.concatenate([d0, d1, d2])
    offsets = np.array([0, n, 2 * n, 2 * n + 12500])
    pairs = np.array([[0, 1], [0, 2], [1, 2]], np.int32)
    store = matching.DescriptorStore.from_packed(desc, pts, offsets)
    counts, m = matching.match_pairs(store, pairs)
    want = oracle_lib.match_pairs(desc.astype(np.float32), pts, offsets, pairs)
    got = matching.split_matches(counts, m)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert counts[0] > 12000 and counts[2] > 2000
    store.close()


def test_ties_and_duplicates_above_4096_features(oracle_lib, gpu_ctx):
    """Value-only keys (> 128 column tiles): with ratio > 1 equal distances pass the ratio test and cv2's
    lowest-index rule decides, across tiles AND across the column classes the kernel reduces over."""
    from opensfm_amd import matching

    rng = np.random.default_rng(21)
    f1 = _rand_desc(rng, 200, 64)
    f2 = np.concatenate([f1[:100], f1[:100], f1[50:150], _rand_desc(rng, 4300, 64)])
    f2 = f2[rng.permutation(len(f2))]
    for ratio in (0.8, 1.0, 1.5):
        for sym, fo in ((False, oracle_lib.match_brute_force), (True, oracle_lib.match_brute_force_symmetric)):
            assert np.array_equal(matching._match_leaf(f1, f2, ratio, sym), fo(f1, f2, ratio))
            assert np.array_equal(matching._match_leaf(f2, f1, ratio, sym), fo(f2, f1, ratio))


@pytest.mark.parametrize("n1,n2,seed,ratio", [(2, 2, 0, 0.8), (40, 3, 1, 0.8), (300, 100, 2, 0.8), (1000, 777, 3, 0.7), (2000, 2000, 4, 0.8),
                                              (4500, 300, 5, 0.9)])
def test_flann_semantics_equal_oracle(oracle_lib, gpu_ctx, n1, n2, seed, ratio):
    """matcher_type FLANN on the GPU = match_flann / match_flann_symmetric (matching.py:683-720) with an exact search: squared
    float32 ratio test, queries of the one-way variant are the SECOND image, matches listed in query order."""
    from opensfm_amd import matching

    rng = np.random.default_rng(seed)
    f1 = synthetic._hahog_like(rng, n1).astype(np.float32)
    f2 = synthetic._hahog_like(rng, n2).astype(np.float32)
    k = min(n1, n2) // 2
    f2[:k] = np.clip(f1[rng.permutation(n1)[:k]] + np.rint(rng.normal(0, 3, (k, 128))), 0, 255)
    cfg = {"lowes_ratio": ratio}
    i1, i2 = matching.build_flann_index(f1, cfg), matching.build_flann_index(f2, cfg)
    got = matching.match_flann(i1, f2, cfg)
    want = oracle_lib.match_flann(f1, f2, ratio)
    assert got == [tuple(int(v) for v in x) for x in want]
    gots = matching.match_flann_symmetric(f1, i1, f2, i2, cfg)
    wants = oracle_lib.match_brute_force_symmetric(f1, f2, ratio, squared=True)
    assert gots == [tuple(int(v) for v in x) for x in wants]
    if k >= 8:
        assert len(wants) >= k // 2


def test_flann_semantics_batched_pipeline(oracle_lib, gpu_ctx):
    """matcher_type FLANN through the batched entry point: squared-ratio descriptor stage + the same geometric stage."""
    from opensfm_amd import matching

    sc = synthetic.make_matching_scene(6, 500, seed=21)
    pairs = synthetic.all_pairs(6)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets, gpu_ctx)
    counts, m = matching.match_pairs(store, pairs, {"matcher_type": "FLANN"}, robust=False)
    got = matching.split_matches(counts, m)
    d = sc.desc.astype(np.float32)
    for (a, b), g in zip(pairs, got):
        want = oracle_lib.match_brute_force_symmetric(d[sc.offsets[a]:sc.offsets[a + 1]], d[sc.offsets[b]:sc.offsets[b + 1]], 0.8, squared=True)
        assert np.array_equal(g, want)
    store.close()


def test_two_threads_share_the_library(oracle_lib, gpu_ctx):
    """The reference calls the leaf functions from a joblib thread pool (context.py:47-67): concurrent leaf calls -- each thread
    on its own default context, and two threads sharing ONE context -- must give the single-threaded results."""
    import threading

    from opensfm_amd import _lib, matching

    rng = np.random.default_rng(77)
    jobs = []
    for k in range(6):
        f1 = synthetic._hahog_like(rng, 700 + 50 * k).astype(np.float32)
        f2 = synthetic._hahog_like(rng, 900 - 30 * k).astype(np.float32)
        f2[:300] = np.clip(f1[rng.permutation(len(f1))[:300]] + np.rint(rng.normal(0, 3, (300, 128))), 0, 255)
        jobs.append((f1, f2, [tuple(int(v) for v in x) for x in oracle_lib.match_brute_force_symmetric(f1, f2)]))
    for shared in (None, gpu_ctx):
        errors, seen_ctx = [], set()

        def work(tid):
            try:
                ctx = shared or _lib.default_context(0)
                seen_ctx.add(id(ctx))
                for rep in range(4):
                    for f1, f2, want in jobs[tid::2]:
                        got = matching._match_leaf(f1, f2, 0.8, True, ctx)
                        if [tuple(int(v) for v in x) for x in got] != want:
                            errors.append((tid, rep))
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert errors == []
        assert len(seen_ctx) == (1 if shared is not None else 2)  # per-thread default contexts


def test_match_arrays_outlive_the_store_and_the_context():
    """match_pairs hands out VIEWS of the buffers the call left in host memory (osfm_result_host_ptrs), and the result object lives as
    long as one of them does: nothing may depend on the store, the context or the order in which Python collects things"""
    import gc

    from opensfm_amd import _lib, matching

    sc = synthetic.make_matching_scene(8, 500, seed=5)
    pairs = synthetic.all_pairs(8)
    ctx = _lib.Context(0)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets, ctx)
    counts, m = matching.match_pairs(store, pairs)
    assert not counts.flags.owndata and not m.flags.owndata and m.shape == (int(counts.sum()), 2) and len(m) > 100
    c0, m0 = counts.copy(), m.copy()
    first = m[:5]  # a slice keeps the result alive on its own
    del store, counts, m
    ctx.close()
    del ctx
    gc.collect()
    junk = [np.random.default_rng(i).integers(0, 1 << 30, 1 << 18) for i in range(8)]  # churn the allocator
    assert np.array_equal(first, m0[:5])
    c1, m1 = matching.match_pairs(matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets), pairs)
    assert np.array_equal(c1, c0) and np.array_equal(m1, m0) and len(junk) == 8
