"""WORDS matcher, VLAD and the neighbour searches of pair preselection on the MI355X against the CPU oracle / scipy's kd-tree."""
import numpy as np
import pytest
from scipy import spatial

import words_cases as wc
from opensfm_amd import preselection, words

pytestmark = pytest.mark.gpu
CFG = {"lowes_ratio": 0.8, "bow_num_checks": 20}


def test_reference_match_using_words(gpu_ctx, oracle_lib):
    """opensfm/test/test_matching.py:50-68 through the drop-in leaves"""
    f, w = wc.example_features(1000, seed=0)
    m = words.match_words(f[0], w[0], f[1], w[1], CFG)
    assert len(m) == 1000 and (m[:, 0] == m[:, 1]).all()
    assert np.array_equal(m, oracle_lib.match_words(f[0], w[0], f[1], w[1][:, 0], 0.8, 20))
    s = words.match_words_symmetric(f[0], w[0], f[1], w[1], CFG)
    assert s == oracle_lib.match_words_symmetric(f[0], w[0], f[1], w[1], 0.8, 20)


@pytest.mark.parametrize("n1,n2,checks,ratio", [(300, 260, 20, 0.8), (513, 1, 5, 0.9), (1, 700, 1, 0.8), (2000, 2100, 20, 0.8), (128, 129, 1000, 1.0)])
def test_match_words_bit_identical(gpu_ctx, oracle_lib, n1, n2, checks, ratio):
    f, w = wc.random_features(n1, n2, seed=n1 + n2, n_words=40)
    cfg = {"lowes_ratio": ratio, "bow_num_checks": checks}
    got = words.match_words(f[0], w[0], f[1], w[1], cfg)
    assert np.array_equal(got, oracle_lib.match_words(f[0], w[0], f[1], w[1][:, 0], ratio, checks))
    assert words.match_words_symmetric(f[0], w[0], f[1], w[1], cfg) == oracle_lib.match_words_symmetric(f[0], w[0], f[1], w[1], ratio, checks)


def test_match_words_pairs_batch(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(3)
    sizes = [400, 0, 650, 333, 1, 512]
    feats, wds = [], []
    base = rng.random((700, 128)).astype(np.float32)
    for n in sizes:
        feats.append((base[:n] + rng.normal(0, 0.02, (n, 128))).astype(np.float32))
        wds.append(rng.integers(0, 30, (n, 4)).astype(np.int32))
    store = words.WordsStore(feats, wds)
    pairs = [(0, 2), (2, 0), (1, 3), (3, 1), (4, 5), (5, 5), (0, 3), (2, 5)]
    for symmetric in (True, False):
        got, ms = words.match_words_pairs(store, pairs, CFG, symmetric=symmetric)
        assert ms >= 0 and len(got) == len(pairs)
        for (a, b), m in zip(pairs, got):
            if symmetric:
                want = np.array(oracle_lib.match_words_symmetric(feats[a], wds[a], feats[b], wds[b], 0.8, 20), np.int32).reshape(-1, 2)
            else:
                want = oracle_lib.match_words(feats[a], wds[a], feats[b], wds[b][:, 0], 0.8, 20)
            assert np.array_equal(m, want), (a, b, symmetric)
    assert sum(len(m) for m in got) > 500
    empty, _ = words.match_words_pairs(store, np.zeros((0, 2), np.int32), CFG)
    assert empty == []
    with pytest.raises(Exception):
        words.match_words_pairs(store, [(0, 9)], CFG)
    store.close()


def test_vlad_descriptor_and_distances(gpu_ctx, oracle_lib):
    rng = np.random.default_rng(4)
    centers = rng.random((64, 128)).astype(np.float32)
    hist = {}
    for k, n in enumerate((0, 1, 700, 3000)):
        feats = rng.random((n, 128)).astype(np.float32)
        v = words.unnormalized_vlad(feats, centers)
        assert np.array_equal(v, oracle_lib.vlad_descriptor(feats, centers))  # same float sums in the same order
        if n > 1:
            hist["im%d" % k] = words.signed_square_root_normalize(v)
    hist["im9"] = hist["im2"] * np.float32(0.5)
    im, d, others = words.vlad_distances("im2", ["im9", "im3", "im2", "nope"], hist)
    assert im == "im2" and others == ["im3", "im9"]
    assert np.array_equal(d, oracle_lib.vlad_distances(hist["im2"], np.stack([hist[o] for o in others])))
    assert words.vlad_distances("nope", ["im2"], hist) == ("nope", [], [])
    assert words.unnormalized_vlad(rng.random((5, 64)).astype(np.float32), centers) is None
    h = preselection.vlad_histogram(rng.random((100, 128)).astype(np.float32), centers)
    assert abs(np.linalg.norm(h) - 1) < 1e-5


@pytest.mark.parametrize("n,m,k,radius", [(500, 40, 6, np.inf), (500, 40, 7, 120.0), (33, 33, 40, 80.0), (2000, 300, 1, np.inf), (10, 5, 3, 1e-9)])
def test_neighbour_searches_equal_the_kdtree(gpu_ctx, n, m, k, radius):
    rng = np.random.default_rng(n + m)
    pts = rng.uniform(-300, 300, (n, 3))
    pts[:, 2] = 0
    qry = np.r_[pts[: m // 2], rng.uniform(-300, 300, (m - m // 2, 3))]
    tree = spatial.cKDTree(pts)
    if k < n:
        dist, idx = preselection.knn_points(pts, qry, k, radius)
        wd, wi = tree.query(qry, k=k, distance_upper_bound=radius)
        wd, wi = wd.reshape(m, k), wi.reshape(m, k)
        assert np.array_equal(np.where(idx < 0, n, idx), wi)
        assert np.allclose(dist[np.isfinite(wd)], wd[np.isfinite(wd)], rtol=1e-12, atol=1e-12) and np.array_equal(np.isfinite(dist), np.isfinite(wd))
    hit = preselection.radius_points(pts, qry, min(radius, 1e300))
    want = np.zeros((m, n), bool)
    for q, js in enumerate(tree.query_ball_point(qry, min(radius, 1e300))):
        want[q, js] = True
    assert np.array_equal(hit, want)


def test_preselection_on_the_device_equals_the_kdtree_version(gpu_ctx, monkeypatch):
    import test_preselection as tp
    from opensfm_amd.geometry_types import TopocentricConverter

    exifs = tp.make_exifs(200, 11)
    images = sorted(exifs)
    reference = TopocentricConverter(45.0, 7.0, 0.0)
    device = [preselection.match_candidates_by_distance(images[:50], images, exifs, reference, nb, dist) for nb, dist in ((6, 0), (0, 100.0), (5, 150.0))]
    device_t = preselection.match_candidates_by_time(images[:50], images, exifs, 5)
    monkeypatch.setattr(preselection, "_neighbours", tp.kdtree_neighbours)
    host = [preselection.match_candidates_by_distance(images[:50], images, exifs, reference, nb, dist) for nb, dist in ((6, 0), (0, 100.0), (5, 150.0))]
    assert device == host and all(len(x) > 50 for x in device)
    assert device_t == preselection.match_candidates_by_time(images[:50], images, exifs, 5)


def test_match_images_with_pairs_words_matcher(gpu_ctx, oracle_lib):
    """matcher_type WORDS through match_images_with_pairs (matching.py:388-398, 563-634): masked features and words, match_words_symmetric,
    gate, fundamental-matrix RANSAC, gate, unfilter -- against the same chain built from the oracle's pieces"""
    from types import SimpleNamespace

    from opensfm_amd import matching, synthetic

    rng = np.random.default_rng(8)
    cam = SimpleNamespace(projection_type="perspective", k1=0.0, k2=0.0, focal=0.85)
    n = 500
    p1, p2, inl = synthetic.make_two_view(n, 0.8, 5)
    base = rng.random((n, 128)).astype(np.float32)
    vocabulary = rng.random((60, 128)).astype(np.float32)
    feats, wds, masks = {}, {}, {}
    for im, pts in (("a", p1), ("b", p2), ("c", p1[::-1].copy())):
        desc = (base if im != "c" else base[::-1]) + rng.normal(0, 0.01, (n, 128)).astype(np.float32)
        feats[im] = SimpleNamespace(points=np.c_[pts, np.ones((n, 2))], descriptors=desc.astype(np.float32))
        wds[im] = wc.assign_words(desc, vocabulary, 3)
        masks[im] = rng.random(n) > 0.05
    config = {"matcher_type": "WORDS", "robust_matching_min_match": 20, "robust_matching_threshold": 0.004, "lowes_ratio": 0.8,
              "symmetric_matching": True, "bow_num_checks": 20}
    data = SimpleNamespace(config=config, load_camera_models=lambda: {"cam": cam}, load_features=lambda im: feats[im],
                           load_features_mask=lambda im, pts: masks[im], load_words=lambda im: wds[im])
    exifs = {im: {"camera": "cam"} for im in feats}
    pairs = [("a", "b"), ("a", "c"), ("b", "c")]
    got = matching.match_images_with_pairs(data, {}, exifs, pairs)
    for ia, ib in pairs:
        fa, fb = feats[ia].descriptors[masks[ia]], feats[ib].descriptors[masks[ib]]
        m = np.array(oracle_lib.match_words_symmetric(fa, wds[ia][masks[ia]], fb, wds[ib][masks[ib]], 0.8, 20), np.int32).reshape(-1, 2)
        want = np.array([])
        if len(m) >= 20:
            pa, pb = feats[ia].points[masks[ia]][m[:, 0], :2], feats[ib].points[masks[ib]][m[:, 1], :2]
            F, mask, _ = oracle_lib.find_fundamental_ransac(pa, pb, 0.004, 0.9999)
            if F is not None and F[2, 2] != 0.0 and mask.sum() >= 20:
                want = matching.unfilter_matches(m[mask], masks[ia], masks[ib])
        assert np.array_equal(np.asarray(got[ia, ib]).reshape(-1, 2), np.asarray(want).reshape(-1, 2)), (ia, ib)
    assert len(got["a", "b"]) > 100


@pytest.mark.parametrize("n", [5, 8, 100, 129, 1000, 8192, 8193, 10000, 20001])
def test_bow_distances_equal_numpy(oracle_lib, gpu_ctx, n):
    """osfm_bow_distances = np.fabs(h - h2).sum() (pairs_selection.py:690-708) in numpy's own summation order: equal to numpy and to the
    oracle bit for bit, for lengths on both sides of the 8 / 128 / 8192 boundaries of the pairwise sum"""
    from opensfm_amd import words

    rng = np.random.default_rng(n)
    h = np.abs(rng.normal(size=(9, n)))
    h /= h.sum(1, keepdims=True)
    hist = {"im%d" % i: h[i] for i in range(9)}
    _, dist, other = words.bow_distances("im0", list(hist), hist)
    assert other == ["im%d" % i for i in range(1, 9)]
    want = [float(np.fabs(h[0] - h[i]).sum()) for i in range(1, 9)]
    assert dist == want
    assert np.array_equal(np.asarray(dist), oracle_lib.bow_distances(h[0], h[1:]))
