import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from opensfm_amd import matching, synthetic
from opensfm_amd._lib import MatchTimings
for nf in (64, 256, 512, 1024, 2000):
    sc = synthetic.make_matching_scene(200, nf, seed=42)
    pairs = synthetic.all_pairs(200)
    store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
    for _ in range(2):
        tm = MatchTimings()
        c, m = matching.match_pairs(store, pairs, robust=False, timings=tm)
    tiles = ((nf + 31) // 32) ** 2
    print("features", nf, "ms", round(tm.ms_match_kernel, 3), "us/pair x512:", round(tm.ms_match_kernel * 1e3 / len(pairs) * 512, 1), "tiles", tiles, "ns per tile-slot", round(tm.ms_match_kernel * 1e6 / len(pairs) * 512 / tiles, 1))
