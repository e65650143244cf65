"""Second half of the headline metric: LM iterations / s of global BA at BASELINE.json configs[4]
(5 000 cameras / 500 000 points / 5 000 000 observations) on one MI355X, next to the CPU oracle.
A leg of bench.py (kept outside the product package: it times the oracle as the CPU baseline)."""
from __future__ import annotations

import time

import numpy as np

from opensfm_amd import bundle, synthetic

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MATVEC_BYTES_PER_OBS = 288.0  # stored point+pose blocks (144 B) read twice per Schur mat-vec


# per LM iteration the stored-Jacobian pass moves 0.88 GB at B5 (5e6 obs x (32 + 144) B) and every PCG mat-vec 1.44 GB
# (5e6 x 288 B): an LM iteration is bounded by (176 + 288 x matvecs) B per observation / HBM bandwidth
LM_BYTES_PER_OBS_FIXED = 176.0


def lm_iteration_line(g: dict, nobs: int) -> dict:
    ms = 1e3 * g["seconds_run"] / max(1, g["iterations"])
    mv = g["pcg_iterations"] / max(1, g["iterations"])
    bound_ms = (LM_BYTES_PER_OBS_FIXED + MATVEC_BYTES_PER_OBS * mv) * nobs / (HBM_PEAK_GBS * 1e9) * 1e3
    return {"ms": round(ms, 3), "matvecs_per_iteration": round(mv, 2),
            "hbm_bound_ms": round(bound_ms, 4), "roofline_frac": round(bound_ms / ms, 4) if ms > 0 else None,
            "roofline_note": "whole LM iteration against SURVEY.md 8(d)'s bound ((176 + 288 x matvecs) B per observation at 8 TB/s); the "
                             "'roofline' block above is the Schur mat-vec kernel pair alone",
            "note": "per LM iteration: Jacobian + gradients, band assembly, cyclic-reduction factor, camera border (one pass for all "
                    "columns), right-hand side, PCG (1-2 mat-vecs), back-substitution, candidate cost"}


def run_grid(ctx, rows: int = 50, cols: int = 100, points: int = 500000, track: int = 9, iters: int = 10, seed: int = 42) -> dict:
    """The same size on a NON-sequence topology: a rows x cols block survey, shots numbered line after line, every
    point seen from ~3 lines -- co-visibility half-width ~2 x cols in shot order.  The solver renumbers the shots by a sweep along the
    block's long side (~2 x rows) and factorises the exact band directly (block LDL^T, ba.hip wide_*); the truncated 15-shot band of
    round 2 needed ~1000 CG iterations per LM iteration here (0.126 LM-iters/s)."""
    pr = synthetic.make_ba_scene_grid(rows, cols, points, track, seed=seed)
    nobs = len(pr["obs_shot"])
    no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, **no_tol)
    inl = ~pr["is_outlier"]
    return {"workload": f"{rows} x {cols} grid of cameras / {points} pts / {nobs} obs (block survey: each point seen from 3 lines)",
            "value": round(g["iterations"] / g["seconds_run"], 3), "unit": "LM-iters/s", "lm_iterations": int(g["iterations"]),
            "pcg_iterations": int(g["pcg_iterations"]), "shot_bandwidth_input": int(g.get("shot_bandwidth_input", -1)),
            "shot_bandwidth": int(g.get("shot_bandwidth", -1)), "shots_reordered": int(g.get("shots_reordered", 0)),
            "preconditioner_bandwidth": int(g.get("preconditioner_bandwidth", -1)),
            "inlier_rmse_px": round(float(np.sqrt((g["reproj_err"][inl] ** 2).sum(1).mean()) * 2000.0), 4),
            "cost": [float(g["initial_cost"]), float(g["final_cost"])], "lm_iteration": lm_iteration_line(g, nobs),
            "solver_note": "exact band of half-width `preconditioner_bandwidth` shots factorised directly (one launch per 16-shot block "
                           "column, 16 pivot steps each: a latency chain, not flops); a solve = two sweeps of one small launch per block "
                           "column in push form; CG confirms in 1-2 iterations"}


def run(ctx, shots: int = 5000, points: int = 500000, track: int = 10, iters: int = 20, cpu_baseline: bool = True,
        cpu_iters: int = 20, seed: int = 42, grid: bool = True) -> dict:
    t0 = time.time()
    pr = synthetic.make_ba_scene(shots, points, track, seed=seed)
    t_gen = time.time() - t0
    nobs = len(pr["obs_shot"])
    no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)  # warm-up
    t0 = time.perf_counter()
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, **no_tol)
    wall = time.perf_counter() - t0
    inl = ~pr["is_outlier"]
    rmse_px = float(np.sqrt((g["reproj_err"][inl] ** 2).sum(1).mean()) * 2000.0)
    ms_mv = g["ms_per_matvec"]
    achieved = MATVEC_BYTES_PER_OBS * nobs / (ms_mv * 1e-3) / 1e9 if ms_mv else 0.0
    out = {
        "metric": "BA LM-iters/sec",
        "workload": f"{shots} cams / {points} pts / {nobs} obs, SoftLOne(1), shared perspective camera + priors, GPS priors",
        # LM iterations per second of the solve proper (what ceres::Solve covers: initial evaluation +
        # the LM loop); index build + H2D ("setup") and D2H ("teardown") are reported beside it, as the
        # reference reports wall_times.setup / run / teardown (ba_helpers.cc:749-753)
        "value": round(g["iterations"] / g["seconds_run"], 3),
        "unit": "LM-iters/s",
        "lm_iterations": int(g["iterations"]),
        "run_seconds": round(g["seconds_run"], 4),
        "setup_seconds": round(g["seconds_setup"], 4),
        "teardown_seconds": round(g["seconds_teardown"], 4),
        "whole_call_lm_iters_per_s": round(g["iterations"] / g["seconds_solver"], 3),
        "linear_solver_seconds": round(g["seconds_linear_solver"], 3),
        "call_seconds_incl_h2d": round(wall, 3),
        "pcg_iterations": int(g["pcg_iterations"]),
        "inlier_rmse_px": round(rmse_px, 4),
        "cost": [float(g["initial_cost"]), float(g["final_cost"])],
        "dtype": "f64",
        "roofline": {
            "bound": "hbm",
            "kernel": "schur mat-vec (schur_point_coop_kernel<0> + schur_shot_kernel)",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "avg_matvec_ms": round(ms_mv, 4) if ms_mv else None,
            "algorithmic_bytes_per_obs": MATVEC_BYTES_PER_OBS,
        },
        "scene_gen_s": round(t_gen, 2),
    }
    # HBM traffic of one mat-vec: PMC passes cannot run inside this process; the committed counters of `tools/prof_ba.py` at the same size
    # (tools/r03_final_profiles.sh -> profiles/r03_ba_pmc.json: FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE) are quoted
    import json
    import os

    pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_ba_pmc.json")
    if os.path.exists(pmc_file):
        try:
            pmc = json.load(open(pmc_file))
            k = nobs / float(pmc["units_per_launch"])
            out["roofline"]["traffic"] = {"hbm_read_bytes": pmc["fetch_bytes_corrected"] * k, "hbm_write_bytes": pmc["write_bytes"] * k,
                                          "algorithmic_bytes": MATVEC_BYTES_PER_OBS * nobs, "source": pmc["source"]}
        except (KeyError, ValueError):
            pass
    out["lm_iteration"] = lm_iteration_line(g, nobs)
    if grid:
        try:
            out["grid_topology"] = run_grid(ctx, points=points, seed=seed)
        except Exception as exc:  # noqa: BLE001  (a secondary workload must not take the line down)
            out["grid_topology"] = {"error": f"{type(exc).__name__}: {exc}"}
    if cpu_baseline:
        import oracle

        t0 = time.perf_counter()
        o = oracle.ba_solve(pr, max_iterations=cpu_iters, **no_tol)
        dt = time.perf_counter() - t0
        g2 = bundle.bundle_arrays(pr, {"bundle_max_iterations": cpu_iters}, ctx=ctx, **no_tol)
        rm_o = float(np.sqrt((o["reproj_err"] ** 2).sum(1).mean()) * 2000.0)
        rm_g = float(np.sqrt((g2["reproj_err"] ** 2).sum(1).mean()) * 2000.0)
        out["cpu_baseline"] = {
            "value": round(o["iterations"] / o["seconds_total"], 4),
            "unit": "LM-iters/s",
            "cores": oracle.num_threads(),
            "kind": "port",
            "kind_note": "port; OpenMP residuals / Jacobians AND Schur elimination (entry-owner partition), skyline Cholesky serial",
            "sample": f"all {cpu_iters} LM iterations of the same problem ({dt:.1f} s; exact Schur + skyline Cholesky on "
                      f"{oracle.num_threads()} threads)",
            "parity_iterations": int(cpu_iters),
            "rmse_px_diff_vs_gpu_same_iters": abs(rm_o - rm_g),
            "cost_history_max_rel_diff": float(np.max(np.abs(np.asarray(o["cost_history"]) - np.asarray(g2["cost_history"]))
                                                      / np.maximum(np.abs(np.asarray(o["cost_history"])), 1e-300))),
            "max_abs_diff": {"points": float(np.abs(o["points"] - g2["points"]).max()), "shot_pose": float(np.abs(o["shot_pose"] - g2["shot_pose"]).max())},
        }
    return out
