"""One-shot bring-up of relpose.hip on an MI355X (ctypes + numpy only, no torch import): writes gpurun_out/relpose_try.json
step by step so that a cut-off run still tells how far it got."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs("gpurun_out", exist_ok=True)
log = {"steps": []}


def step(name, **kw):
    log["steps"].append({"step": name, "t": round(time.time() - T0, 3), **kw})
    with open("gpurun_out/relpose_try.json", "w") as f:
        json.dump(log, f, indent=1)
    print(name, kw, flush=True)


T0 = time.time()
import oracle  # noqa: E402
from opensfm_amd import matching  # noqa: E402
from tests.test_gpu_zz_relpose import _batch  # noqa: E402

step("imports")
rng = np.random.default_rng(1)
b1, b2, off = _batch(rng, [30, 5, 200], [0.3, 0.0, 0.5])
res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "ransac", 37, 0.99, True, 10)
ok = []
for p in range(3):
    s = slice(off[p], off[p + 1])
    w = oracle.ransac_relative_pose(b1[s], b2[s], 0.004, 37, 0.99, True, 10)
    ok.append(bool((res[p]["score"], res[p]["iterations"]) == (w["score"], w["iterations"]) and np.array_equal(np.flatnonzero(mask[s]), w["inliers"])
                   and np.array_equal(res[p]["lo_model"].view(np.uint64), w["lo_model"].view(np.uint64))))
step("ransac37", ok=ok, ms=ms, got=[(r["score"], r["iterations"]) for r in res])
res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "match", 1000, 0.99, True, 10, 10)
ok = []
for p in range(3):
    s = slice(off[p], off[p + 1])
    w = oracle.robust_match_calibrated_bearings(b1[s], b2[s], 0.004, 1000, 0.99, True, 10, 10)
    ok.append(bool(np.array_equal(mask[s], w["mask"]) and (res[p]["score"], res[p]["iterations"]) == (w["score"], w["iterations"])))
step("match1000", ok=ok, ms=ms, got=[(r["score"], r["iterations"], r["n_inliers"]) for r in res])
sizes = [300] * 2048
b1, b2, off = _batch(rng, sizes, [0.4] * len(sizes))
res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "match", 1000, 0.99, True, 10, 10)
step("throughput", pairs=len(sizes), ms=ms, pairs_per_s=len(sizes) / (ms / 1e3), mean_iters=float(np.mean([r["iterations"] for r in res])),
     mean_inliers=float(np.mean([r["n_inliers"] for r in res])))
