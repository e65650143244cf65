"""Measurement of the two next-row kernels on one MI355X (not part of bench.py's headline line):
calibrated robust matching (osfm_relpose_pairs, MATCH mode) in pairs / s and guided matching (osfm_match_guided) in pairs / s,
each next to the CPU oracle on a bounded sample.

    python tools/relpose_bench.py [--pairs 2048] [--matches 300] [--outliers 0.4] [--features 2000]

Meant for `rocprofv3 --kernel-trace --stats -- python tools/relpose_bench.py`: the kernels are rp_walk_kernel / rp_solve5_kernel /
rp_solveN_kernel / rp_finish_kernel (relpose.hip, rounds) and guided_match_kernel.  Algorithmic work: ~150 fp64 operations per (model, correspondence) score evaluation,
~6 models per iteration; the fp64 vector peak is ~78.6 TFLOP/s."""
from __future__ import annotations

import argparse
import json
import time

import numpy as np

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opensfm_amd import matching  # noqa: E402
from opensfm_amd._lib import default_context  # noqa: E402

FP64_PEAK_TFLOPS = 78.6
SCORE_FLOPS = 150.0


def _rodrigues(r):
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def two_view_bearings(rng, n, outliers, noise=1e-3):
    R = _rodrigues(rng.normal(0, 0.3, 3))
    t = rng.normal(0, 1, 3)
    t /= np.linalg.norm(t)
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)]
    b1 = X + rng.normal(0, noise, X.shape)
    X2 = X @ R.T + t + rng.normal(0, noise, X.shape)
    bad = rng.random(n) < outliers
    X2[bad] = np.c_[rng.uniform(-2, 2, bad.sum()), rng.uniform(-2, 2, bad.sum()), rng.uniform(4, 9, bad.sum())]
    b1 /= np.linalg.norm(b1, axis=1, keepdims=True)
    return b1, X2 / np.linalg.norm(X2, axis=1, keepdims=True)


def run_relpose(ctx, pairs: int, matches: int, outliers: float, cpu_pairs: int = 32, seed: int = 1) -> dict:
    rng = np.random.default_rng(seed)
    parts = [two_view_bearings(rng, matches, outliers) for _ in range(pairs)]
    b1 = np.concatenate([p[0] for p in parts])
    b2 = np.concatenate([p[1] for p in parts])
    off = np.arange(pairs + 1, dtype=np.int64) * matches
    matching.relpose_pairs(b1[: 8 * matches], b2[: 8 * matches], off[:9], 0.004, "match", ctx=ctx)  # warm-up
    t0 = time.perf_counter()
    res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "match", ctx=ctx)
    wall = time.perf_counter() - t0
    iters = np.array([r["iterations"] for r in res])
    flops = SCORE_FLOPS * 6.0 * float(iters.sum()) * matches  # scoring only: the dominant term of the cost model
    out = {"kernel": "rp_walk + rp_solve5 + rp_solveN + rp_finish (rounds)", "pairs": pairs, "matches_per_pair": matches, "outlier_fraction": outliers,
           "kernel_ms": round(ms, 3), "pairs_per_s": round(pairs / (ms * 1e-3), 1), "pairs_per_s_incl_copies": round(pairs / wall, 1),
           "mean_ransac_iterations": round(float(iters.mean()), 1), "mean_inliers": round(float(np.mean([r["n_inliers"] for r in res])), 1),
           "scoring_tflops": round(flops / (ms * 1e-3) / 1e12, 4), "frac_of_fp64_peak": round(flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5)}
    if cpu_pairs:
        import oracle  # the checker, timed beside the kernel as bench.py's cpu_baseline does

        t0 = time.perf_counter()
        same = 0
        for p in range(min(cpu_pairs, pairs)):
            s = slice(off[p], off[p + 1])
            w = oracle.robust_match_calibrated_bearings(b1[s], b2[s], 0.004, 1000, 0.99, True, 10, 10)
            same += bool(np.array_equal(w["mask"], mask[s]))
        dt = time.perf_counter() - t0
        out["cpu_oracle_pairs_per_s"] = round(min(cpu_pairs, pairs) / dt, 2)
        out["cpu_sample"] = f"{min(cpu_pairs, pairs)} pairs, 1 thread"
        out["identical_inlier_sets_on_sample"] = f"{same}/{min(cpu_pairs, pairs)}"
    return out


def run_guided(ctx, features: int, repeats: int = 20, seed: int = 2) -> dict:
    rng = np.random.default_rng(seed)
    n = features
    R = _rodrigues(rng.normal(0, 0.2, 3))
    o = rng.normal(0, 1, 3)
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)]
    b1 = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    Y = (X - o) @ R
    b2 = (Y / np.linalg.norm(Y, axis=1, keepdims=True)).astype(np.float32)
    base = rng.integers(0, 255, (n // 2, 128))
    d1 = np.clip(np.concatenate([base, base]) + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.float32)
    d2 = np.clip(np.concatenate([base, base]) + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.float32)
    matching._match_guided_leaf(d1, d2, 0.8, True, None, b1, b2, R, o, 0.006, ctx)  # warm-up
    t0 = time.perf_counter()
    for _ in range(repeats):
        got = matching._match_guided_leaf(d1, d2, 0.8, True, None, b1, b2, R, o, 0.006, ctx)
    dt = (time.perf_counter() - t0) / repeats
    return {"kernel": "guided_match_kernel", "features_per_image": n, "seconds_per_pair_incl_copies": round(dt, 5),
            "pairs_per_s_incl_copies": round(1.0 / dt, 1), "matches": int(len(got)), "correct": round(float((got[:, 0] == got[:, 1]).mean()), 4)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8192)
    ap.add_argument("--matches", type=int, default=300)
    ap.add_argument("--outliers", type=float, default=0.4)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    ctx = default_context()
    print(json.dumps({"relpose": run_relpose(ctx, a.pairs, a.matches, a.outliers, 0 if a.no_cpu else 32),
                      "guided": run_guided(ctx, a.features)}))


if __name__ == "__main__":
    main()
