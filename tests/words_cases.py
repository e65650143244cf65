"""Inputs shared by the CPU (oracle) and GPU tests of the WORDS matcher."""
import numpy as np


def assign_words(f, vocabulary, k):
    d = ((f[:, None, :].astype(np.float64) - vocabulary[None].astype(np.float64)) ** 2).sum(2)
    return np.argsort(d, axis=1, kind="stable")[:, :k].astype(np.int32)


def example_features(nfeatures, seed, n_words=200, words_per_feature=5):
    """the inputs of test_matching.py:24-41: normal features normalised by the norm of the WHOLE matrix (entries ~ 3e-3) and their copies
    with noise / 500, worded against a vocabulary of uchar-scale descriptors as the shipped bow_hahog_root_uchar_* files hold them.  At
    that scale every feature gets the same closest words, so the matcher checks every feature of the other image (the check count
    is only tested after a word is exhausted) -- which is why the reference's test can demand all 1000 matches."""
    rng = np.random.default_rng(seed)
    f1 = rng.normal(size=(nfeatures, 128)).astype(np.float32)
    f1 /= np.linalg.norm(f1)
    f2 = f1 + rng.normal(size=f1.shape).astype(np.float32) / 500.0
    f2 /= np.linalg.norm(f2)
    vocabulary = rng.integers(0, 120, (n_words, 128)).astype(np.float32)
    return [f1, f2], [assign_words(f1, vocabulary, words_per_feature), assign_words(f2, vocabulary, words_per_feature)]


def random_features(n1, n2, seed, n_words=50, words_per_feature=5, duplicates=True):
    """unrelated features with few words (long candidate lists), exact duplicates (ties) and a word nobody carries"""
    rng = np.random.default_rng(seed)
    f1 = rng.random((n1, 128)).astype(np.float32)
    f2 = rng.random((n2, 128)).astype(np.float32)
    k = min(n1, n2) // 3
    f2[:k] = f1[:k] + rng.normal(0, 0.01, (k, 128)).astype(np.float32)
    if duplicates and n2 > 12:
        f2[10] = f2[3]  # two identical candidates: the lower index must win
        f2[11] = f2[3]
    w1 = rng.integers(0, n_words, (n1, words_per_feature)).astype(np.int32)
    w2 = rng.integers(0, n_words, (n2, words_per_feature)).astype(np.int32)
    w2[:k, 0] = w1[:k, 0]
    if duplicates and n2 > 12:
        w2[10, 0] = w2[11, 0] = w2[3, 0]
    w1[-1] = n_words + 5  # words without entries in the other image
    return [f1, f2], [w1, w2]


def python_match_words(f1, w1, f2, w2, ratio, max_checks):
    """features/src/matching.cc:24-72 in numpy float32 scalar arithmetic"""
    index = {}
    for i, w in enumerate(w2):
        index.setdefault(int(w), []).append(i)
    out = []
    ratio = np.float32(ratio)
    for i in range(len(f1)):
        best, second, bm, checks = np.float32(np.inf), np.float32(np.inf), -1, 0
        for word in w1[i]:
            for match in index.get(int(word), []):
                acc = np.float32(0)
                for a, b in zip(f1[i], f2[match]):
                    d = np.float32(a - b)
                    acc = np.float32(acc + np.float32(d * d))
                dist = np.sqrt(acc)
                if dist < best:
                    second, best, bm = best, dist, match
                elif dist < second:
                    second = dist
                checks += 1
            if checks >= max_checks:
                break
        if best < np.float32(ratio * second):
            out.append((i, bm))
    return np.array(out, np.int32).reshape(-1, 2)
