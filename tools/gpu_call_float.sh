#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_float_descriptors.py tests/test_gpu_matching.py -x -q -m gpu > gpurun_out/float_tests.log 2>&1
echo "exit $?" >> gpurun_out/float_tests.log
tail -25 gpurun_out/float_tests.log
python - <<'PY'
import numpy as np, time
from opensfm_amd import matching, synthetic
from opensfm_amd._lib import MatchTimings
sc = synthetic.make_matching_scene(40, 2000, seed=42)
d = sc.desc.astype(np.float32); d /= np.maximum(d.sum(1, keepdims=True), 1e-7); d = np.sqrt(d).astype(np.float32)
pairs = synthetic.all_pairs(40)
store = matching.DescriptorStore.from_packed(d, sc.pts, sc.offsets)
for _ in range(2):
    tm = MatchTimings(); c, m = matching.match_pairs(store, pairs, robust=False, timings=tm)
    print("float store: pairs", len(pairs), "match ms", tm.ms_match_kernel, "pairs/s", len(pairs) / tm.ms_match_kernel * 1e3, "matches", int(c.sum()))
PY
