"""Where the neighbour-list call spends its time: wall ms of matching.match_pairs on the 15 864-pair list for robust on / off, results to
the host or kept in HBM, and 1 .. 16 chunks (OSFM_MATCH_CHUNKS)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (HIP runtime first)

from opensfm_amd import matching, synthetic  # noqa: E402
from opensfm_amd._lib import MatchTimings  # noqa: E402

sys.path.insert(0, ROOT)
from bench import neighbour_pairs  # noqa: E402

sc = synthetic.make_matching_scene(1000, 2000, seed=42)
store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
pairs = neighbour_pairs(1000, 16)
matching.match_pairs(store, pairs[:512])
out = []
for chunks, on_b in ((1, 0), (2, 0), (4, 0), (8, 0), (16, 0), (4, 1)):
    os.environ["OSFM_MATCH_CHUNKS"] = str(chunks)
    os.environ.pop("OSFM_MATCH_RANSAC_STREAM_B", None)
    if on_b:
        os.environ["OSFM_MATCH_RANSAC_STREAM_B"] = "1"  # round-2 placement: the robust stage on stream B underneath the next matcher
    for robust in (True, False):
        for keep in (False, True):
            best = None
            for _ in range(4):
                tm = MatchTimings()
                t0 = time.perf_counter()
                r = matching.match_pairs(store, pairs, robust=robust, timings=tm, keep_device=keep)
                dt = 1e3 * (time.perf_counter() - t0)
                if keep:
                    r.close()
                if best is None or dt < best[0]:
                    best = (dt, tm.ms_total, tm.ms_match_kernel, tm.ms_ransac_kernel)
            out.append({"chunks": chunks, "ransac_stream": "B" if on_b else "A", "robust": robust, "keep_device": keep, "wall_ms": round(best[0], 3), "stream_ms": round(best[1], 3),
                        "match_ms": round(best[2], 3), "ransac_ms": round(best[3], 3)})
            print(out[-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_c5", "prof_neighbour.json"), "w"), indent=1)
