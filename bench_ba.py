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


def _pmc_traffic(name: str, kernels, nobs: int, alg_bytes_per_obs: float):
    """the committed PMC counters of a mat-vec pair scaled to this run's observations; None when the file is absent; an assertion when its
    kernel list is not the one the caller times (a stale file must not be quoted)"""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    if not os.path.exists(path):
        return None
    pmc = json.load(open(path))
    got = tuple(row[0] for row in pmc["per_kernel"])
    assert got == tuple(kernels), f"{path} holds counters of {got}, this run times {tuple(kernels)}: re-run tools/pmc_passes.sh"
    k = nobs / float(pmc["units_per_launch"])
    return {"hbm_read_bytes": pmc["fetch_bytes_corrected"] * k, "hbm_write_bytes": pmc["write_bytes"] * k,
            "algorithmic_bytes": alg_bytes_per_obs * nobs, "source": pmc["source"], "file": "profiles/" + name}


def run_grid(ctx, rows: int = 50, cols: int = 100, points: int = 500000, track: int = 9, iters: int = 10, seed: int = 42, cpu_iters: int = 0) -> dict:
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
    out = {"workload": f"{rows} x {cols} grid of cameras / {points} pts / {nobs} obs (block survey: each point seen from 3 lines)",
            "value": round(g["iterations"] / g["seconds_run"], 3), "unit": "LM-iters/s", "lm_iterations": int(g["iterations"]),
            "pcg_iterations": int(g["pcg_iterations"]), "shot_bandwidth_input": int(g.get("shot_bandwidth_input", -1)),
            "shot_bandwidth": int(g.get("shot_bandwidth", -1)), "shots_reordered": int(g.get("shots_reordered", 0)),
            "preconditioner_bandwidth": int(g.get("preconditioner_bandwidth", -1)),
            "inlier_rmse_px": round(float(np.sqrt((g["reproj_err"][inl] ** 2).sum(1).mean()) * 2000.0), 4),
            "cost": [float(g["initial_cost"]), float(g["final_cost"])], "lm_iteration": lm_iteration_line(g, nobs),
            "solver_note": "exact band of half-width `preconditioner_bandwidth` shots, block tridiagonal over dense clusters of that many "
                           "shots, factorised by cyclic reduction (log2(S / bw) levels; per level a blocked in-place Gauss-Jordan inverse -- pivot "
                           "panels in LDS -- and the hand-written fp64-MFMA batched GEMM, dgemm_mfma_kernel); a solve = one launch per level down and up; CG confirms in 1-2 iterations"}
    if cpu_iters > 0:
        out["cpu_baseline"] = _oracle_leg(pr, ctx, cpu_iters, "exact Schur elimination on all cores + skyline Cholesky (serial) of half-width 6 x bandwidth")
    return out


def _oracle_leg(pr: dict, ctx, iters: int, note: str) -> dict:
    """the CPU oracle on the first `iters` LM iterations of the same problem, with the GPU's trajectory beside it"""
    import oracle

    no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    t0 = time.perf_counter()
    o = oracle.ba_solve(pr, max_iterations=iters, **no_tol)
    dt = time.perf_counter() - t0
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, **no_tol)
    ch_o, ch_g = np.asarray(o["cost_history"]), np.asarray(g["cost_history"])
    value, factor = o["iterations"] / o["seconds_total"], "serial skyline factor"
    par = None
    try:  # the skyline Cholesky right-looking on up to 16 cores (same operations in the same order: identical bits); the faster one is the baseline
        oracle.ba_set_parallel(2)
        t0 = time.perf_counter()
        o2 = oracle.ba_solve(pr, max_iterations=iters, **no_tol)
        dt2 = time.perf_counter() - t0
        par = {"value": round(o2["iterations"] / o2["seconds_total"], 4), "seconds": round(dt2, 1),
               "identical_to_serial_factor": bool(np.asarray(o2["cost_history"]).shape == ch_o.shape and np.allclose(np.asarray(o2["cost_history"]), ch_o, rtol=1e-13, atol=0))}
        if par["value"] > value and par["identical_to_serial_factor"]:
            value, factor = par["value"], "right-looking skyline factor on up to 16 cores"
    except Exception as exc:  # noqa: BLE001
        par = {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        oracle.ba_set_parallel(1)
    return {"value": round(value, 4), "unit": "LM-iters/s", "cores": oracle.num_threads(), "kind": "port", "factor": factor,
            "serial_factor_value": round(o["iterations"] / o["seconds_total"], 4), "all_cores_factor": par,
            "sample": f"the first {iters} LM iterations of the same problem ({dt:.1f} s); {note}", "parity_iterations": int(iters),
            "cost_history_max_rel_diff": float(np.max(np.abs(ch_o - ch_g) / np.maximum(np.abs(ch_o), 1e-300))),
            "max_abs_diff": {"points": float(np.abs(o["points"] - g["points"]).max()), "shot_pose": float(np.abs(o["shot_pose"] - g["shot_pose"]).max())}}


def run_ragged(ctx, shots: int = 5000, points: int = 500000, track: int = 10, iters: int = 10, seed: int = 42, cpu_iters: int = 3) -> dict:
    """The same size with ragged tracks (VERDICT r3 missing #7): lengths 2 + Poisson, 15 % of the sightings missing -- no two tracks share
    a shot set, and the co-visibility half-width is the longest track (~25 shots), beyond the cluster-tridiagonal band.  The exact band
    goes through the wide solver: cyclic reduction over dense clusters of `preconditioner_bandwidth` shots (ba.hip dbcr_*)."""
    pr = synthetic.make_ba_scene(shots, points, track, seed=seed, ragged=True)
    nobs = len(pr["obs_shot"])
    no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, **no_tol)
    inl = ~pr["is_outlier"]
    out = {"workload": f"{shots} cams / {points} pts / {nobs} obs, ragged tracks (2 + Poisson lengths, 15 % of the sightings missing)",
           "value": round(g["iterations"] / g["seconds_run"], 3), "unit": "LM-iters/s", "lm_iterations": int(g["iterations"]),
           "pcg_iterations": int(g["pcg_iterations"]), "shot_bandwidth": int(g.get("shot_bandwidth", -1)),
           "preconditioner_bandwidth": int(g.get("preconditioner_bandwidth", -1)),
           "inlier_rmse_px": round(float(np.sqrt((g["reproj_err"][inl] ** 2).sum(1).mean()) * 2000.0), 4),
           "cost": [float(g["initial_cost"]), float(g["final_cost"])], "lm_iteration": lm_iteration_line(g, nobs)}
    if cpu_iters > 0:
        out["cpu_baseline"] = _oracle_leg(pr, ctx, cpu_iters, "exact Schur elimination on all cores + skyline Cholesky (serial)")
    return out


def run_general(ctx, shots: int = 5000, points: int = 500000, track: int = 10, iters: int = 10, seed: int = 42, cpu_size=(200, 12000, 8), cpu_iters: int = 6,
                at_size_iters: int = 2) -> dict:
    """What BAHelpers::Bundle builds on a CALIBRATED data set, at the headline size: a shared BROWN camera with its nine native intrinsics
    free (+ priors), position priors through a free per-camera similarity bias (bundle_compensate_gps_bias), 20 ground control points.
    `osfm_bundle_solve` = the streaming solver in its generic mode (csrc/ba_generic.inc): 16 border unknowns instead of 3, rows of
    2 x (10 + 9) components.  Round 4 sent this to a dense reduced system + rocSOLVER (7.2 GB and O(n^3) at this size; no bench line).
    The CPU leg: the general oracle (jets, points eliminated, dense Cholesky of the reduced system) on a smaller scene of the same
    make-up, with the GPU's trajectory on that scene beside it."""
    pr = synthetic.make_general_ba_scene(shots, points, track, model="brown", n_gcp=20, gps_bias=True, seed=seed)
    nobs = len(pr["obs_shot"])
    no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    bundle.bundle_general_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)
    g = bundle.bundle_general_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, verbose=bundle.BA_TIME_MATVEC, **no_tol)
    inl = ~pr["is_outlier"]
    out = {"workload": f"{shots} cams / {points} pts / {nobs} obs, shared BROWN camera (9 free intrinsics + priors), GPS priors through a free "
                       "similarity bias, 20 control points, SoftLOne(1)",
           "value": round(g["iterations"] / g["seconds_run"], 3), "unit": "LM-iters/s", "lm_iterations": int(g["iterations"]),
           "pcg_iterations": int(g["pcg_iterations"]), "border_unknowns": 16, "shot_bandwidth": int(g["shot_bandwidth"]),
           "preconditioner_bandwidth": int(g["preconditioner_bandwidth"]),
           "setup_seconds": round(g["seconds_setup"], 4), "run_seconds": round(g["seconds_run"], 4),
           "inlier_rmse_px": round(float(np.sqrt((g["reproj_err"][inl][:, :2] ** 2).sum(1).mean()) * 2000.0), 4),
           "cost": [float(g["initial_cost"]), float(g["final_cost"])], "ms_per_matvec": g["ms_per_matvec"],
           "bias": [float(x) for x in g["bias"][0]], "bias_true": [float(x) for x in pr["gt_bias"][0]],
           "lm_iteration_ms": round(1e3 * g["seconds_run"] / max(1, g["iterations"]), 3)}
    # the generic rows' mat-vec against the SAME figure as the specialised path: SURVEY 8(d)'s 288 B per observation and mat-vec.  (Rounds 4-5 priced the
    # generic rows at 432 B -- the two passes then read the 2 x 9 border slots of a stored row on top; since round 6 the COMPACT rows rebuild the slots from
    # (Xc, wt) and the passes move ~250 B per observation by the counters, so 432 would flatter the kernel.)
    alg = float(MATVEC_BYTES_PER_OBS)
    if g["ms_per_matvec"]:
        out["roofline"] = {"bound": "hbm", "kernel": "schur mat-vec, generic rows (gen_schur_point_kernel<2, 0> + gen_schur_shot_kernel<2, 9>)",
                           "achieved": round(alg * nobs / (g["ms_per_matvec"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(alg * nobs / (g["ms_per_matvec"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_obs": alg,
                           "traffic": _pmc_traffic("r06_ba_generic_pmc.json", ("gen_schur_point_kernelILi2ELi0E", "gen_schur_shot_kernel"), nobs, alg)}
    if cpu_iters > 0:
        import oracle

        ps = synthetic.make_general_ba_scene(cpu_size[0], cpu_size[1], cpu_size[2], model="brown", n_gcp=20, gps_bias=True, seed=seed)
        t0 = time.perf_counter()
        o = oracle.bundle_general(ps, max_iterations=cpu_iters, **no_tol)
        dt = time.perf_counter() - t0
        bundle.bundle_general_arrays(ps, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)  # first use of this problem shape (allocator blocks)
        gs = bundle.bundle_general_arrays(ps, {"bundle_max_iterations": cpu_iters}, ctx=ctx, **no_tol)
        ch_o, ch_g = np.asarray(o["cost_history"]), np.asarray(gs["cost_history"])
        rm = lambda e: float(np.sqrt((np.asarray(e)[:, :2] ** 2).sum(1).mean()) * 2000.0)  # noqa: E731
        out["cpu_baseline"] = {"value": round(o["iterations"] / dt, 4), "unit": "LM-iters/s", "cores": min(16, oracle.num_threads()), "kind": "port",
                               "kind_note": "the residuals' jets on up to 16 threads, the sums by one thread in observation order; elimination serial, Cholesky on 8",
                               "sample": f"{cpu_size[0]} cams / {cpu_size[1]} pts / {len(ps['obs_shot'])} obs of the same make-up, {cpu_iters} LM iterations "
                                         f"({dt:.1f} s): oracle/bundle_general_oracle.cc, jets + Schur elimination + dense Cholesky",
                               "gpu_lm_iters_per_s_same_scene": round(gs["iterations"] / gs["seconds_run"], 3),
                               "parity_iterations": int(cpu_iters),
                               "cost_history_max_rel_diff": float(np.max(np.abs(ch_o - ch_g) / np.maximum(np.abs(ch_o), 1e-300))),
                               "rmse_px_diff": abs(rm(o["reproj_err"]) - rm(gs["reproj_err"])),
                               "max_abs_diff": {k: float(np.abs(o[k] - gs[k]).max()) for k in ("cam_params", "rig_instance_pose", "points", "bias")}}
        if at_size_iters > 0:
            # the SAME problem as the line above, at configs[4] size, against the arrow-form oracle (envelope Cholesky of the 30 016 reduced
            # unknowns) over the first LM iterations: parity at the size the figure is quoted on, ~1 minute of host work
            t0 = time.perf_counter()
            oa = oracle.bundle_general(pr, max_iterations=at_size_iters, **no_tol)
            dta = time.perf_counter() - t0
            ga = bundle.bundle_general_arrays(pr, {"bundle_max_iterations": at_size_iters}, ctx=ctx, **no_tol)
            ca_o, ca_g = np.asarray(oa["cost_history"]), np.asarray(ga["cost_history"])
            out["cpu_baseline"]["at_size"] = {
                "sample": f"the first {at_size_iters} LM iterations of the line's own problem ({dta:.1f} s on {oracle.num_threads()} threads)",
                "value": round(oa["iterations"] / dta, 4), "unit": "LM-iters/s", "parity_iterations": int(at_size_iters),
                "cost_history_max_rel_diff": float(np.max(np.abs(ca_o - ca_g) / np.maximum(np.abs(ca_o), 1e-300))),
                "rmse_px_diff": abs(rm(oa["reproj_err"]) - rm(ga["reproj_err"])),
                "max_abs_diff": {k: float(np.abs(oa[k] - ga[k]).max()) for k in ("cam_params", "rig_instance_pose", "points", "bias")}}
    return out


def run_local(ctx, shots: int = 400, points: int = 40000, calls: int = 40, seed: int = 7, cpu_calls: int = 5) -> dict:
    """Local bundle adjustment as incremental reconstruction calls it (reconstruction.py:1512 -> BAHelpers::BundleLocal,
    ba_helpers.cc:117-311): once per added image, on the <= 30 interior shots around it + their boundary, cameras constant, 10 LM
    iterations.  `value` = solves per second of the solver calls (setup + run + teardown each); the neighbourhood extraction (host numpy,
    the reference's ShotNeighborhood) is timed beside it."""
    pr = synthetic.make_ba_scene(shots, points, 10, seed=seed)
    centres = np.linspace(20, shots - 21, calls).astype(int)
    t0 = time.perf_counter()
    subs = [bundle.local_problem(pr, int(c))[0] for c in centres]
    t_host = time.perf_counter() - t0
    no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    bundle.bundle_arrays(subs[0], {"bundle_max_iterations": 10}, ctx=ctx, **no_tol)
    t0 = time.perf_counter()
    reps = [bundle.bundle_arrays(sp, {"bundle_max_iterations": 10}, ctx=ctx, **no_tol) for sp in subs]
    dt = time.perf_counter() - t0
    s0 = subs[len(subs) // 2]
    out = {"workload": f"{calls} neighbourhoods of a {shots}-shot sequence: ~{len(s0['shot_pose'])} shots ({int((1 - s0['shot_fixed']).sum())} free) / "
                       f"{len(s0['points'])} pts / {len(s0['obs_shot'])} obs each, 10 LM iterations, cameras constant",
           "value": round(calls / dt, 2), "unit": "solves/s", "ms_per_solve": round(1e3 * dt / calls, 3),
           "ms_setup": round(1e3 * float(np.mean([r["seconds_setup"] for r in reps])), 3),
           "ms_run": round(1e3 * float(np.mean([r["seconds_run"] for r in reps])), 3),
           "ms_teardown": round(1e3 * float(np.mean([r["seconds_teardown"] for r in reps])), 3),
           "ms_neighbourhood_host": round(1e3 * t_host / calls, 3),
           "note": "latency-bound: 18 launches and one host round trip per LM iteration on a problem of 4e4 observations; the band (48 x 10 blocks of 6 x 6) is factorised and solved by one workgroup (sband_factor_kernel, ~85 us of the ~255 us an iteration takes)"}
    if cpu_calls > 0:
        import oracle

        nt = oracle.num_threads()
        oracle.set_num_threads(min(8, nt))  # a 4e4-observation problem does not feed more threads: all host cores are slower than one
        try:
            k = min(cpu_calls, calls)
            oracle.ba_solve(subs[0], max_iterations=10, **no_tol)
            t0 = time.perf_counter()
            outs = [oracle.ba_solve(subs[i], max_iterations=10, **no_tol) for i in range(k)]
            dto = time.perf_counter() - t0
        finally:
            oracle.set_num_threads(nt)
        rel = max(float(np.max(np.abs(np.asarray(o["cost_history"]) - np.asarray(r["cost_history"])) / np.abs(np.asarray(o["cost_history"]))))
                  for o, r in zip(outs, reps))
        out["cpu_baseline"] = {"value": round(k / dto, 3), "unit": "solves/s", "cores": min(8, nt), "kind": "port",
                               "sample": f"the first {k} of the same sub-problems, 10 LM iterations each ({dto:.2f} s)",
                               "cost_history_max_rel_diff": rel,
                               "max_abs_pose_diff": max(float(np.abs(o["shot_pose"] - r["shot_pose"]).max()) for o, r in zip(outs, reps))}
    return out


def run(ctx, shots: int = 5000, points: int = 500000, track: int = 10, iters: int = 20, cpu_baseline: bool = True,
        cpu_iters: int = 20, seed: int = 42, grid: bool = True) -> dict:
    t0 = time.time()
    pr = synthetic.make_ba_scene(shots, points, track, seed=seed)
    t_gen = time.time() - t0
    nobs = len(pr["obs_shot"])
    no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)  # warm-up
    t0 = time.perf_counter()
    g = bundle.bundle_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, verbose=bundle.BA_TIME_MATVEC, **no_tol)
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
            # round 6: what the two kernels are DESIGNED to move per observation -- pass A reads the compact row's Jp 48 + Jr 48 + (u, v, wt) 24 + sigma 8 +
            # two indices 8 and writes w 16; pass B recomputes its rows from xy 16 + sigma 8 + wt 8 + two indices 8 + the point 24 and gathers w 16 --
            # against the 288 B of SURVEY 8(d)'s stored-block mat-vec, which `achieved` keeps as its numerator (the contract's figure)
            "designed_bytes_per_obs": 232.0,
        },
        "scene_gen_s": round(t_gen, 2),
    }
    # HBM traffic of one mat-vec: PMC passes cannot run inside this process; the committed counters of `tools/prof_ba.py` at the same size
    # (tools/pmc_passes.sh -> profiles/r06_ba_pmc.json: FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE) are quoted -- and
    # refused when they were taken on other kernels than the pair this block times
    out["roofline"]["traffic"] = _pmc_traffic("r06_ba_pmc.json", ("23schur_point_coop_kernelILi0E", "17schur_shot_kernel"), nobs, MATVEC_BYTES_PER_OBS)
    out["lm_iteration"] = lm_iteration_line(g, nobs)
    if grid:
        for key, fn in (("grid_topology", lambda: run_grid(ctx, points=points, seed=seed, cpu_iters=2 if cpu_baseline else 0)),
                        ("ragged_topology", lambda: run_ragged(ctx, shots, points, track, seed=seed, cpu_iters=3 if cpu_baseline else 0)),
                        ("local_ba", lambda: run_local(ctx, cpu_calls=5 if cpu_baseline else 0)),
                        ("general", lambda: run_general(ctx, shots, points, track, seed=seed, cpu_iters=6 if cpu_baseline else 0))):
            try:
                out[key] = fn()
            except Exception as exc:  # noqa: BLE001  (a secondary workload must not take the line down)
                out[key] = {"error": f"{type(exc).__name__}: {exc}"}
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
