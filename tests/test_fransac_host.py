"""The product's fundamental-matrix RANSAC logic (opensfm_amd/csrc/fransac_core.h: the table-driven draws from the cv::RNG stream, the
7-point solver, the scoring, the sequential decision rules, the hand-over between the first kernel and the long-run kernel), compiled
for the host with loops in place of lanes (tests/native/fransac_host.cpp), against the CPU oracle -- F bits, mask and iteration count.
This pins everything of ransac.hip except the wave policy, the kernel wrappers and the in-place compaction."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from opensfm_amd import synthetic

HERE = os.path.dirname(os.path.abspath(__file__))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.fixture(scope="module")
def host():
    src = os.path.join(HERE, "native", "fransac_host.cpp")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "fransac_host.so")
    deps = [src, os.path.join(HERE, "..", "opensfm_amd", "csrc", "fransac_core.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wno-unknown-pragmas", "-o", so, src])
    return C.CDLL(so)


def run_host(host, p1, p2, thr=0.004, conf=0.9999, max_iters=1000, raw_cap=0, mode=0):
    p1 = np.ascontiguousarray(p1, np.float64)
    p2 = np.ascontiguousarray(p2, np.float64)
    n = len(p1)
    F = np.zeros(9)
    mask = np.zeros(max(n, 1), np.uint8)
    it, rounds, scored = C.c_int(0), C.c_int(0), C.c_longlong(0)
    r = host.fransac_host_run(_p(p1, C.c_double), _p(p2, C.c_double), n, C.c_double(thr), C.c_double(conf), max_iters, raw_cap, mode,
                              _p(F, C.c_double), _p(mask, C.c_uint8), C.byref(it), C.byref(scored), C.byref(rounds))
    assert r >= 0
    return (F.reshape(3, 3) if r == 1 else None), mask[:n].astype(bool), it.value, rounds.value


def check(host, oracle_lib, p1, p2, **kw):
    raw_cap = kw.pop("raw_cap", 0)
    Fo, mo, io = oracle_lib.find_fundamental_ransac(p1, p2, kw.get("thr", 0.004), kw.get("conf", 0.9999), kw.get("max_iters", 1000))
    for mode in (0, 1):  # the batched path (draw / solve / decide kernels + long runs) and the single-problem kernel
        Fh, mh, ih, rounds = run_host(host, p1, p2, raw_cap=raw_cap, mode=mode, **kw)
        assert (Fo is None) == (Fh is None)
        if Fo is not None:
            assert np.array_equal(Fo.view(np.uint64), Fh.view(np.uint64))
        assert np.array_equal(mo, mh) and io == ih
    return io, rounds


@pytest.mark.parametrize("n", [15, 16, 20, 33, 64, 100, 208, 513, 2000])
@pytest.mark.parametrize("inliers", [0.97, 0.6, 0.3])
def test_emulated_kernels_equal_oracle(host, oracle_lib, n, inliers):
    for seed in range(6):
        p1, p2, _ = synthetic.make_two_view(n, inlier_frac=inliers, seed=100 * n + seed)
        check(host, oracle_lib, p1, p2)


def test_long_runs_cross_many_rounds(host, oracle_lib):
    """pure outliers: the full 1000 iterations over many rounds of both schedules"""
    rng = np.random.default_rng(5)
    for n in (15, 40, 300):
        p1, p2 = rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-0.5, 0.5, (n, 2))
        it, rounds = check(host, oracle_lib, p1, p2)
        assert it >= 300 and rounds >= 8 and (n < 300 or (it == 1000 and rounds >= 18))  # rounds: of the single-problem schedule


def test_small_n_many_duplicate_draws_and_starved_table(host, oracle_lib):
    """n = 15..24: up to 40 % of the draws repeat an index; with a table of 20..60 raw values the rounds come out short and the
    sequential fallback (a subset that does not fit the table) is taken"""
    rng = np.random.default_rng(9)
    for n in (15, 16, 17, 19, 24):
        for raw_cap in (0, 60, 20, 8, 7):
            p1, p2 = rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-0.5, 0.5, (n, 2))
            check(host, oracle_lib, p1, p2, raw_cap=raw_cap, max_iters=150)


def test_collinear_rejections_and_the_attempt_limit(host, oracle_lib):
    """points on a coarse grid: most subsets hold a collinear triple and are rejected (getSubset draws again); all points on one line:
    every attempt is rejected, getSubset gives up after 10000 attempts and there is no model"""
    rng = np.random.default_rng(2)
    g = np.stack(np.meshgrid(np.arange(5), np.arange(5)), -1).reshape(-1, 2) / 10.0 - 0.2
    for seed in range(4):
        sel = rng.permutation(25)[:22]
        p1 = g[sel]
        p2 = g[rng.permutation(25)[:22]] + rng.normal(0, 1e-3, (22, 2)) * (seed > 1)
        check(host, oracle_lib, p1, p2, max_iters=60)
    line = np.c_[np.linspace(-0.4, 0.4, 30), np.linspace(-0.2, 0.2, 30)]
    p2 = rng.uniform(-0.5, 0.5, (30, 2))
    Fh, mh, ih, _ = run_host(host, line, p2)
    Fo, mo, io = oracle_lib.find_fundamental_ransac(line, p2)
    assert Fo is None and Fh is None and io == ih == 0 and not mh.any()


def test_confidence_and_threshold_variants(host, oracle_lib):
    p1, p2, _ = synthetic.make_two_view(150, inlier_frac=0.5, seed=77)
    for conf, thr, mi in ((0.99, 0.004, 1000), (0.9999, 0.001, 1000), (0.5, 0.01, 20), (0.9999, 0.004, 1), (2.0, -1.0, 100)):
        check(host, oracle_lib, p1, p2, thr=thr, conf=conf, max_iters=mi)
