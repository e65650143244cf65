"""the GENERAL bundle adjustment (BROWN camera with nine free intrinsics, free GPS bias, 20 control points) at configs[4]
size -- 5 000 cams / 500 000 pts / 5 M obs -- against the arrow-form oracle (oracle/bundle_general_oracle.cc: jets, points eliminated, dense
Cholesky of the 30 016 reduced unknowns) over the first LM iterations.  Minutes of host work: run once per round, not inside bench.py.
usage: python tools/r06_general_parity_at_size.py [iterations=2] > profiles/r06_general_parity_at_size.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from opensfm_amd import _lib, bundle, synthetic

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
shots, points, track = (int(a) for a in (sys.argv[2:5] if len(sys.argv) > 4 else (5000, 500000, 10)))
oracle.build()
no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
pr = synthetic.make_general_ba_scene(shots, points, track, model="brown", n_gcp=20, gps_bias=True, seed=42)
ctx = _lib.default_context(0)
bundle.bundle_general_arrays(pr, {"bundle_max_iterations": 1}, ctx=ctx, **no_tol)
g = bundle.bundle_general_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, **no_tol)
t0 = time.perf_counter()
o = oracle.bundle_general(pr, max_iterations=iters, **no_tol)
dt = time.perf_counter() - t0
ch_o, ch_g = np.asarray(o["cost_history"]), np.asarray(g["cost_history"])
rm = lambda e: float(np.sqrt((np.asarray(e)[:, :2] ** 2).sum(1).mean()) * 2000.0)  # noqa: E731
print(json.dumps({
    "workload": f"{shots} cams / {points} pts / {len(pr['obs_shot'])} obs, shared BROWN camera (9 free intrinsics + priors), free GPS bias, 20 control points",
    "parity_iterations": iters, "oracle_seconds": round(dt, 1), "oracle_threads": oracle.num_threads(),
    "gpu_ms_per_lm_iteration": round(1e3 * g["seconds_run"] / max(1, g["iterations"]), 3),
    "cost_history_oracle": [float(x) for x in ch_o], "cost_history_gpu": [float(x) for x in ch_g],
    "cost_history_max_rel_diff": float(np.max(np.abs(ch_o - ch_g) / np.maximum(np.abs(ch_o), 1e-300))),
    "rmse_px_diff": abs(rm(o["reproj_err"]) - rm(g["reproj_err"])),
    "max_reprojection_diff_px": float(np.abs(np.asarray(o["reproj_err"])[:, :2] - np.asarray(g["reproj_err"])[:, :2]).max() * 2000.0),
    "max_abs_diff": {k: float(np.abs(o[k] - g[k]).max()) for k in ("cam_params", "rig_instance_pose", "points", "bias")},
}, indent=1))
