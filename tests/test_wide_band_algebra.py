"""The wide band's factorisation of round 4 (ba.hip dbcr_* / dgj_*), restated in numpy: block cyclic reduction over dense clusters with the
level algebra of Solver::dbcr_factor, the blocked in-place Gauss-Jordan inverse of Solver::dbcr_invert_batch (panel steps: pivot block
inverse, R' = P R, A -= C R', A[:, J] = -C P, pivot rows = R' with the pivot block P) and the sweeps of dbcr_walk, against a dense solve.
No GPU: this pins the index logic (cluster counts that are not powers of two, a last panel narrower than the others, padding) the
kernels follow; the kernels themselves are checked against the oracle in tests/test_gpu_ba.py."""
import numpy as np
import pytest


def gj_inverse_blocked(A, T):
    """in place, panels of T columns (the last one narrower)"""
    A = A.copy()
    m = len(A)
    for j0 in range(0, m, T):
        w = min(T, m - j0)
        J = slice(j0, j0 + w)
        P = np.linalg.inv(A[J, J])
        R = A[J, :].copy()          # row panel (including the pivot block)
        C = A[:, J].copy()          # column panel, pivot rows zeroed
        C[J, :] = 0.0
        Rn = P @ R
        A -= C @ Rn                 # rows J unchanged (C's pivot rows are zero); column block J becomes 0 off the pivot rows
        A[:, J] = -C @ P            # pivot rows: 0
        A[J, :] = Rn                # pivot rows = P R ...
        A[J, J] = P                 # ... and the pivot block P
    return A


@pytest.mark.parametrize("m,T", [(6, 6), (150, 78), (612, 90), (96, 96), (102, 54)])
def test_blocked_gauss_jordan_is_the_inverse(m, T):
    rng = np.random.default_rng(m)
    B = rng.normal(size=(m, m))
    A = B @ B.T + m * np.eye(m)
    assert np.abs(gj_inverse_blocked(A, T) @ A - np.eye(m)).max() < 1e-9


def dbcr_solve(D, E, b, T):
    """D[i]: diagonal blocks, E[i] (i >= 1): coupling of cluster i to cluster i - 1; the level loop of dbcr_factor + the sweeps of dbcr_walk"""
    N = len(D)
    D = [d.copy() for d in D]
    cur = [None if e is None else e.copy() for e in E]
    G, H = {}, {}
    s = 1
    while s < N:
        ne, nR = (N - s - 1) // (2 * s) + 1, (N - 1) // (2 * s)
        ids = [(2 * k + 1) * s for k in range(ne)]
        assert all(i < N for i in ids) and sum(i + s < N for i in ids) == nR
        nxt = [None] * N
        for i in ids:
            D[i] = gj_inverse_blocked(D[i], T)  # D_i <- D_i^-1
            G[i] = D[i] @ cur[i]
            if i + s < N:
                H[i] = D[i] @ cur[i + s].T
        for i in ids:
            if i + s < N:
                D[i + s] = D[i + s] - cur[i + s] @ H[i]
            D[i - s] = D[i - s] - cur[i].T @ G[i]
            if i + s < N:
                nxt[i + s] = -cur[i + s] @ G[i]
        cur = nxt
        s *= 2
    D[0] = gj_inverse_blocked(D[0], T)
    m = len(D[0])
    x = [b[i * m:(i + 1) * m].copy() for i in range(N)]
    y = [None] * N
    levels = []
    s = 1
    while s < N:
        for k in range((N + 2 * s - 1) // (2 * s)):
            j = 2 * k * s
            if j - s >= 0:
                x[j] = x[j] - H[j - s].T @ x[j - s]
            if j + s < N:
                x[j] = x[j] - G[j + s].T @ x[j + s]
        levels.append(s)
        s *= 2
    y[0] = D[0] @ x[0]
    for s in reversed(levels):
        for k in range((N - s - 1) // (2 * s) + 1):
            i = (2 * k + 1) * s
            v = D[i] @ x[i] - G[i] @ y[i - s]
            if i + s < N:
                v = v - H[i] @ y[i + s]
            y[i] = v
    return np.concatenate(y)


@pytest.mark.parametrize("N,m,T", [(1, 12, 6), (2, 18, 12), (3, 12, 12), (5, 30, 18), (8, 12, 6), (13, 18, 12), (50, 12, 6)])
def test_dense_cluster_cyclic_reduction_solves_the_band(N, m, T):
    rng = np.random.default_rng(100 * N + m)
    n = N * m
    A = np.zeros((n, n))
    D, E = [], [None]
    for i in range(N):
        B = rng.normal(size=(m, m))
        D.append(B @ B.T + (2 + m) * np.eye(m))
        A[i * m:(i + 1) * m, i * m:(i + 1) * m] = D[-1]
        if i > 0:
            E.append(0.3 * rng.normal(size=(m, m)))
            A[i * m:(i + 1) * m, (i - 1) * m:i * m] = E[-1]
            A[(i - 1) * m:i * m, i * m:(i + 1) * m] = E[-1].T
    assert np.linalg.eigvalsh(A).min() > 0
    b = rng.normal(size=(n, 4))
    assert np.abs(dbcr_solve(D, E, b, T) - np.linalg.solve(A, b)).max() < 1e-10
