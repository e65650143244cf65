"""Driver for rocprofv3: global BA at BASELINE configs[4] size, a few LM iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opensfm_amd import bundle, synthetic
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
track = int(sys.argv[3]) if len(sys.argv) > 3 else 10
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ragged = len(sys.argv) > 5 and sys.argv[5] == "ragged"
pr = synthetic.make_ba_scene(shots, pts, track, seed=42, ragged=ragged)
no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
if os.environ.get("PROF_WARM"):
    bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, **no_tol)  # library initialisation, allocator caches
g = bundle.bundle_arrays(pr, {"bundle_max_iterations": iters}, **no_tol)
print(g["brief_report"]); print("setup", g["seconds_setup"], "run", g["seconds_run"], "teardown", g["seconds_teardown"]); print("solver s", g["seconds_solver"], "lin", g["seconds_linear_solver"], "ms/matvec", g["ms_per_matvec"], "bw", g["preconditioner_bandwidth"])
