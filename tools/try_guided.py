"""One-shot bring-up of guided.hip and of the generic bearing kernel on an MI355X (ctypes + numpy only)."""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.makedirs("gpurun_out", exist_ok=True)
log = {"steps": []}
T0 = time.time()


def step(name, **kw):
    log["steps"].append({"step": name, "t": round(time.time() - T0, 3), **kw})
    with open("gpurun_out/guided_try.json", "w") as f:
        json.dump(log, f, indent=1)
    print(name, kw, flush=True)


import oracle  # noqa: E402
from opensfm_amd import matching  # noqa: E402
import test_oracle_relpose as cams  # noqa: E402
import test_guided_host as gh  # noqa: E402

step("imports")
rng = np.random.default_rng(0)
d1, d2, b1, b2, R, o, perm = gh.guided_scene(rng, 150)
mask, _ = oracle.epipolar_mask(b1, b2, R, o, 0.02)
want = oracle.match_brute_force_masked(d1, d2, mask, 0.8, symmetric=True)
got = matching._match_guided_leaf(d1, d2, 0.8, True, mask)
step("explicit_mask", ok=bool(np.array_equal(got, want)), n=len(got), n_want=len(want))
got = matching._match_guided_leaf(d1, d2, 0.8, True, None, b1, b2, R, o, 0.02)
step("epipolar_mask", ok=bool(np.array_equal(got, want)), n=len(got))
got = matching._match_guided_leaf(d1, d2, 0.8, False, None, b1, b2, R, o, 0.02)
step("epipolar_one_way", ok=bool(np.array_equal(got, oracle.match_brute_force_masked(d1, d2, mask, 0.8, symmetric=False))), n=len(got))
errs = {}
for model, par in list(cams._BEARING_CAMERAS.items()) + [("spherical", [])]:
    ang, phi = rng.uniform(0, 0.9, 1000), rng.uniform(0, 2 * np.pi, 1000)
    X = np.c_[np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)]
    px = rng.uniform(-0.3, 0.3, (1000, 2)) if model == "spherical" else cams._forward(model, par, X)
    mid, names = matching._BEARING_MODELS[model]
    attrs = dict(zip(names, par))
    cam = SimpleNamespace(projection_type=model, principal_point=[attrs.pop("cx", 0.0), attrs.pop("cy", 0.0)], **attrs)
    errs[model] = float(np.abs(matching.pixel_bearing_many(cam, px) - oracle.pixel_bearings_generic(model, par, px)).max())
step("bearings", max_abs_diff=errs)
d1, d2, b1, b2, R, o, perm = gh.guided_scene(rng, 1000)
t = time.time()
got = matching._match_guided_leaf(d1, d2, 0.8, True, None, b1, b2, R, o, 0.006)
step("guided_2000x2000", seconds=round(time.time() - t, 4), n=len(got), correct=float((perm[got[:, 1]] == got[:, 0]).mean()))
