#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_float_descriptors.py tests/test_gpu_matching.py -x -q -m gpu > gpurun_out/float_tests.log 2>&1
echo "exit $?" >> gpurun_out/float_tests.log
tail -25 gpurun_out/float_tests.log
python - <<'PY'
import numpy as np, time
from opensfm_amd import matching, synthetic
from opensfm_amd._lib import MatchTimings
sc = synthetic.make_matching_scene(120, 2000, seed=42)
d = sc.desc.astype(np.float32); d /= np.maximum(d.sum(1, keepdims=True), 1e-7); d = np.sqrt(d).astype(np.float32)
pairs = synthetic.all_pairs(120)
near = pairs[(pairs[:, 1] - pairs[:, 0]) <= 16]
for name, desc in (("float (root) store", d), ("uint8 store", sc.desc)):
    store = matching.DescriptorStore.from_packed(desc, sc.pts, sc.offsets)
    for pl, nm in ((pairs, "all pairs"), (near, "neighbour pairs")):
        for _ in range(2):
            tm = MatchTimings(); c, m = matching.match_pairs(store, pl, robust=False, timings=tm)
        print(name, nm, len(pl), "match ms %.3f" % tm.ms_match_kernel, "pairs/s %.0f" % (len(pl) / tm.ms_match_kernel * 1e3), "matches", int(c.sum()), "pairs with float evaluation / exact path", tm.pairs_exact_path)
    store.close()
PY
