"""Round-6 diagnostic: the general BA with COMPACT rows against OSFM_BA_GEN_FULL_ROWS, difference per LM iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opensfm_amd import _lib, bundle, synthetic
NO_TOL = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
for model in ("brown", "perspective", "fisheye_opencv"):
    pr = synthetic.make_general_ba_scene(60, 1500, 6, model=model, n_gcp=5, gps_bias=True, seed=11)
    for iters in (1, 2, 5):
        res = []
        for full in (False, True):
            os.environ.pop("OSFM_BA_GEN_FULL_ROWS", None)
            if full:
                os.environ["OSFM_BA_GEN_FULL_ROWS"] = "1"
            ctx = _lib.Context()
            res.append(bundle.bundle_general_arrays(pr, {"bundle_max_iterations": iters}, ctx=ctx, **NO_TOL))
        a, b = res
        ch = np.asarray(a["cost_history"]) - np.asarray(b["cost_history"])
        print(model, iters, "cost diff", ch, "pcg", a["pcg_iterations"], b["pcg_iterations"],
              {k: float(np.abs(a[k] - b[k]).max()) for k in ("cam_params", "rig_instance_pose", "points", "bias")})
