"""Driver for rocprofv3: global BA at BASELINE configs[4] size, a few LM iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opensfm_amd import bundle, synthetic
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
track = int(sys.argv[3]) if len(sys.argv) > 3 else 10
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
mode = sys.argv[5] if len(sys.argv) > 5 else ""
ragged = mode == "ragged"
no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
if mode.startswith("general"):  # a BROWN camera with free intrinsics, a free GPS bias, 20 control points: osfm_bundle_solve (generic rows)
    pr = synthetic.make_general_ba_scene(shots, pts, track, model="brown", n_gcp=20, gps_bias=True, seed=42, ragged=mode == "general_ragged")
    solve = bundle.bundle_general_arrays
else:
    pr = synthetic.make_ba_scene(shots, pts, track, seed=42, ragged=ragged)
    solve = bundle.bundle_arrays
if os.environ.get("PROF_WARM"):
    solve(pr, {"bundle_max_iterations": 1}, **no_tol)  # library initialisation, allocator caches
g = solve(pr, {"bundle_max_iterations": iters}, verbose=bundle.BA_TIME_MATVEC, **no_tol)
print(g["brief_report"]); print("setup", g["seconds_setup"], "run", g["seconds_run"], "teardown", g["seconds_teardown"]); print("solver s", g["seconds_solver"], "lin", g["seconds_linear_solver"], "ms/matvec", g["ms_per_matvec"], "bw", g["preconditioner_bandwidth"])
