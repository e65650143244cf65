"""Small driver for rocprofv3: one fused-matcher launch over all pairs of N images."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opensfm_amd import matching, synthetic
from opensfm_amd._lib import MatchTimings

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
robust = (sys.argv[2] == "1") if len(sys.argv) > 2 else False
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sc = synthetic.make_matching_scene(n, 2000, seed=42)
pairs = synthetic.all_pairs(n)
store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
for _ in range(reps):
    tm = MatchTimings()
    c, m = matching.match_pairs(store, pairs, robust=robust, timings=tm)
    print("pairs", len(pairs), "match ms", tm.ms_match_kernel, "ransac ms", tm.ms_ransac_kernel, "total", tm.ms_total,
          "pairs/s(kernel)", len(pairs) / tm.ms_match_kernel * 1e3, "matches", int(c.sum()))
