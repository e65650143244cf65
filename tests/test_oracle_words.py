"""The WORDS matcher / VLAD oracle (oracle/words_oracle.c) against the reference's own test of the call (opensfm/test/test_matching.py:24-68)
and against plain numpy restatements."""
import numpy as np

import words_cases as wc


def test_reference_match_using_words(oracle_lib):
    """test_matching.py:50-68: 1000 features and their slightly noisy copies must match i -> i (there with the shipped 10 000-word
    vocabulary and FLANN word assignment; here a random uchar-scale vocabulary with exact assignment, see words_cases.example_features)"""
    f, w = wc.example_features(1000, seed=0)
    m = oracle_lib.match_words(f[0], w[0], f[1], w[1][:, 0], 0.8, 20)
    assert len(m) == 1000 and (m[:, 0] == m[:, 1]).all()
    s = oracle_lib.match_words_symmetric(f[0], w[0], f[1], w[1], 0.8, 20)
    assert len(s) == 1000 and all(a == b for a, b in s)


def test_match_words_equals_a_python_restatement(oracle_lib):
    f, w = wc.random_features(300, 260, seed=1, n_words=40, words_per_feature=3)
    for checks in (1, 5, 20, 1000):
        got = oracle_lib.match_words(f[0], w[0], f[1], w[1][:, 0], 0.9, checks)
        want = wc.python_match_words(f[0], w[0], f[1], w[1][:, 0], 0.9, checks)
        assert np.array_equal(got, want)
        assert len(got) > 0


def test_vlad_descriptor_and_distances(oracle_lib):
    rng = np.random.default_rng(2)
    feats = rng.random((500, 128)).astype(np.float32)
    centers = rng.random((64, 128)).astype(np.float32)
    v = oracle_lib.vlad_descriptor(feats, centers)
    assign = ((feats[:, None, :] - centers[None]) ** 2).sum(2).argmin(1)
    want = np.zeros((64, 128))
    for i, c in enumerate(assign):
        want[c] += feats[i].astype(np.float64) - centers[c]
    assert np.allclose(v.reshape(64, 128), want, atol=1e-4)
    others = rng.random((7, 8192)).astype(np.float32)
    d = oracle_lib.vlad_distances(v, others)
    assert np.allclose(d, np.linalg.norm(others.astype(np.float64) - v, axis=1), rtol=1e-5)
