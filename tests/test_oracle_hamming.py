"""The Hamming branch of match_brute_force (matching.py:737-740: uint8 descriptors -> cv2 BruteForce-Hamming) as the oracle restates
it, against an independent numpy restatement: popcount of the xor, stable two smallest (lowest index first among equals), Lowe's test
on float32(int) in doubles, mutual check.  cv2 itself is not available: parity unpinned vs cv2 (oracle/match_oracle.c)."""
import numpy as np
import pytest


def _numpy_one_way(a, b, ratio):
    if len(b) < 2:
        return np.zeros((0, 2), np.int32)
    d = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    o = np.argsort(d, axis=1, kind="stable")
    rows = np.arange(len(a))
    d0, d1 = d[rows, o[:, 0]].astype(np.float32), d[rows, o[:, 1]].astype(np.float32)
    ok = d0.astype(np.float64) < ratio * d1.astype(np.float64)
    return np.stack([rows[ok], o[ok, 0]], 1).astype(np.int32)


@pytest.mark.parametrize("n1,n2,width,seed", [(60, 70, 32, 0), (200, 150, 61, 1), (5, 2, 64, 2), (40, 40, 7, 3), (3, 1, 32, 4)])
def test_hamming_oracle_equals_numpy(oracle_lib, n1, n2, width, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (n1, width)).astype(np.uint8)
    b = rng.integers(0, 256, (n2, width)).astype(np.uint8)
    k = min(n1, n2) // 2
    b[:k] = a[rng.permutation(n1)[:k]] ^ (rng.random((k, width)) < 0.03).astype(np.uint8)  # noisy copies
    if k >= 4:
        b[k - 2: k] = b[:2]  # exact duplicates in the train set: ties, the lowest index wins and the ratio test fails on equal distances
    for ratio in (0.8, 1.0):
        fwd = _numpy_one_way(a, b, ratio)
        assert np.array_equal(oracle_lib.match_hamming(a, b, ratio), fwd)
        bwd = {(int(i), int(j)) for j, i in _numpy_one_way(b, a, ratio)}
        want = np.array(sorted(set(map(tuple, fwd.tolist())) & bwd), np.int32).reshape(-1, 2)
        assert np.array_equal(oracle_lib.match_hamming(a, b, ratio, symmetric=True), want)
