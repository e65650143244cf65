"""Local bundle adjustment (BAHelpers::BundleLocal, ba_helpers.cc:117-311) as incremental reconstruction calls it: once per added image
on the <= 30-shot neighbourhood of that image, 10 LM iterations.  Times a run of such calls on a synthetic sequence; with `oracle` as the
last argument also the CPU oracle on the first few of the same sub-problems (parity + baseline)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from opensfm_amd import _lib, bundle, synthetic

shots = int(sys.argv[1]) if len(sys.argv) > 1 else 400
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 50
with_oracle = "oracle" in sys.argv[4:]
pr = synthetic.make_ba_scene(shots, pts, 10, seed=7)
ctx = _lib.Context()
centres = np.linspace(20, shots - 21, calls).astype(int)
t0 = time.perf_counter()
subs = [bundle.local_problem(pr, int(c)) for c in centres]
t_host = time.perf_counter() - t0
no_tol = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
bundle.bundle_arrays(subs[0][0], {"bundle_max_iterations": 10}, ctx=ctx, **no_tol)  # warm-up
t0 = time.perf_counter()
reps = [bundle.bundle_arrays(s[0], {"bundle_max_iterations": 10}, ctx=ctx, **no_tol) for s in subs]
dt = time.perf_counter() - t0
s0 = subs[0][0]
print("sub-problem 0: shots", len(s0["shot_pose"]), "free", int((1 - s0["shot_fixed"]).sum()), "points", len(s0["points"]), "obs", len(s0["obs_shot"]))
print("neighbourhood extraction (host numpy) ms per call", 1e3 * t_host / calls)
print("solves/s", calls / dt, "ms per solve", 1e3 * dt / calls, "iterations", [r["iterations"] for r in reps[:5]],
      "setup ms", 1e3 * np.mean([r["seconds_setup"] for r in reps]), "run ms", 1e3 * np.mean([r["seconds_run"] for r in reps]),
      "teardown ms", 1e3 * np.mean([r["seconds_teardown"] for r in reps]))
if with_oracle:
    import oracle

    k = min(5, calls)
    t0 = time.perf_counter()
    outs = [oracle.ba_solve(subs[i][0], max_iterations=10, **no_tol) for i in range(k)]
    dto = time.perf_counter() - t0
    print("oracle solves/s", k / dto, "threads", oracle.num_threads())
    for i in range(k):
        ch_o, ch_g = np.asarray(outs[i]["cost_history"]), np.asarray(reps[i]["cost_history"])
        n = min(len(ch_o), len(ch_g))
        print(" problem", i, "cost history max rel diff", float(np.max(np.abs(ch_o[:n] - ch_g[:n]) / np.abs(ch_o[:n]))),
              "pose diff", float(np.abs(outs[i]["shot_pose"] - reps[i]["shot_pose"]).max()))
