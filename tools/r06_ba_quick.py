"""Round-6 working script (GPU box): the BA lines of bench.py without their CPU legs, and the PCG-tolerance experiment at configs[2] size
against the oracle.  Output: one JSON document on stdout."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench_ba
from opensfm_amd import _lib, bundle, synthetic

ctx = _lib.default_context(0)
out = {}
what = sys.argv[1:] or ["lines", "tol"]
no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
if "lines" in what:
    r = bench_ba.run(ctx, cpu_baseline=False)
    out["lines"] = {"headline": {k: r[k] for k in ("value", "lm_iterations", "pcg_iterations", "run_seconds", "setup_seconds", "teardown_seconds", "lm_iteration", "roofline")}}
    for k in ("grid_topology", "ragged_topology", "local_ba", "general"):
        v = r.get(k, {})
        out["lines"][k] = {q: v.get(q) for q in ("value", "unit", "lm_iterations", "pcg_iterations", "lm_iteration", "ms_per_solve", "ms_run", "ms_setup", "ms_teardown", "error", "lm_iteration_ms", "roofline") if q in v}
if "tol" in what:
    import oracle

    oracle.build()
    pr = synthetic.make_ba_scene(500, 50000, 6, seed=42)
    o = oracle.ba_solve(pr, max_iterations=20, **no_tol)
    ch_o = np.asarray(o["cost_history"])
    rows = []
    for tol in (0.0, 1e-9, 1e-8, 1e-6, 1e-4):  # 0: no direct-solve rule, every solve iterates to pcg_tolerance = 1e-10 (rounds 1-5)
        bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)
        g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 20}, ctx=ctx, pcg_direct_tolerance=tol, **no_tol)
        ch_g = np.asarray(g["cost_history"])
        rows.append({"pcg_direct_tolerance": tol, "pcg_iterations": int(g["pcg_iterations"]), "lm_iterations": int(g["iterations"]),
                     "ms_per_lm_iteration": round(1e3 * g["seconds_run"] / g["iterations"], 4),
                     "cost_history_max_rel_diff": float(np.max(np.abs(ch_o - ch_g) / np.abs(ch_o))),
                     "rmse_px_diff": abs(float(np.sqrt((o["reproj_err"] ** 2).sum(1).mean()) * 2000.0) - float(np.sqrt((g["reproj_err"] ** 2).sum(1).mean()) * 2000.0)),
                     "max_abs_pose_diff": float(np.abs(o["shot_pose"] - g["shot_pose"]).max())})
    out["pcg_direct_tolerance_configs2"] = rows
    # the same at configs[4] size without the oracle: against the run at 1e-10
    pr = synthetic.make_ba_scene(5000, 500000, 10, seed=42)
    ref = None
    rows = []
    for tol in (0.0, 1e-8, 1e-6):
        bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)
        g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 20}, ctx=ctx, pcg_direct_tolerance=tol, **no_tol)
        ch = np.asarray(g["cost_history"])
        if ref is None:
            ref = ch
        rows.append({"pcg_direct_tolerance": tol, "pcg_iterations": int(g["pcg_iterations"]), "ms_per_lm_iteration": round(1e3 * g["seconds_run"] / g["iterations"], 4),
                     "cost_history_max_rel_diff_vs_no_rule": float(np.max(np.abs(ref - ch) / np.abs(ref)))})
    out["pcg_direct_tolerance_configs4"] = rows
print(json.dumps(out, indent=1))
