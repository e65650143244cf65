#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_zz_relpose.py -x -q -m gpu -k "guided" > gpurun_out/guided_tests.log 2>&1
echo "exit $?" >> gpurun_out/guided_tests.log
tail -25 gpurun_out/guided_tests.log
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/guided_bench.log
import numpy as np, time
from opensfm_amd import matching, synthetic
from opensfm_amd._lib import MatchTimings
n_img = 300
sc = synthetic.make_matching_scene(n_img, 2000, seed=42)
pairs = synthetic.all_pairs(n_img)
near = pairs[(pairs[:, 1] - pairs[:, 0]) <= 16]
focal = 0.85
b = np.c_[sc.pts / focal, np.ones(len(sc.pts))]; b /= np.linalg.norm(b, axis=1, keepdims=True)
bears = [b[sc.offsets[i]:sc.offsets[i + 1]].astype(np.float32) for i in range(n_img)]
rels = []
for a, c in near:
    Ra, Rc, oa, oc = sc.cam_R[a], sc.cam_R[c], sc.cam_o[a], sc.cam_o[c]
    Rrel = Rc @ Ra.T                      # world-to-camera of c relative to a
    orel = Ra @ (oc - oa)
    rels.append(np.concatenate([Rrel.T.reshape(9), orel]))
store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
for robust in (False, True):
    for _ in range(2):
        tm = MatchTimings(); t0 = time.time()
        c, m = matching.match_pairs_guided(store, near, bears, rels, {"guided_matching_threshold": 0.006}, robust=robust, timings=tm)
        dt = time.time() - t0
    print("guided robust=%d: pairs %d descriptor-stage ms %.2f -> %.0f pairs/s; call %.1f ms -> %.0f pairs/s; matches/pair %.1f" % (robust, len(near), tm.ms_match_kernel, len(near) / tm.ms_match_kernel * 1e3, dt * 1e3, len(near) / dt, c.sum() / len(near)))
for robust in (False, True):
    for _ in range(2):
        tm = MatchTimings(); t0 = time.time()
        c, m = matching.match_pairs(store, near, robust=robust, timings=tm)
        dt = time.time() - t0
    print("unguided robust=%d: pairs %d descriptor-stage ms %.2f -> %.0f pairs/s; call %.1f ms -> %.0f pairs/s; matches/pair %.1f" % (robust, len(near), tm.ms_match_kernel, len(near) / tm.ms_match_kernel * 1e3, dt * 1e3, len(near) / dt, c.sum() / len(near)))
PY
