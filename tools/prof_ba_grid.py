"""Driver for rocprofv3: global BA at BASELINE configs[4] size on the block-survey (grid) topology, a few LM iterations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opensfm_amd import bundle, synthetic
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pts = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
pr = synthetic.make_ba_scene_grid(rows, cols, pts, 9, seed=42)
no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
if os.environ.get("PROF_WARM"):
    bundle.bundle_arrays(pr, {"bundle_max_iterations": 1}, **no_tol)  # library initialisation, allocator caches
g = bundle.bundle_arrays(pr, {"bundle_max_iterations": iters}, **no_tol)
print(g["brief_report"]); print("setup", g["seconds_setup"], "run", g["seconds_run"], "teardown", g["seconds_teardown"])
print("lin", g["seconds_linear_solver"], "pcg", g["pcg_iterations"], "bw", g["preconditioner_bandwidth"], "true bw", g.get("shot_bandwidth"), "input bw", g.get("shot_bandwidth_input"), "reordered", g.get("shots_reordered"))
print("cost", g["initial_cost"], g["final_cost"])
