"""Where a workgroup of match_fused_kernel spends its time, on the neighbour list and on a slice of the exhaustive list.
Needs the instrumented build (tools/libosfm_dbg_phases.so: match.hip compiled with -DOSFM_DBG_PHASES, linked with the product's
other objects);  OSFM_MI355_LIB=tools/libosfm_dbg_phases.so python tools/match_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import neighbour_pairs  # noqa: E402
from opensfm_amd import _lib, matching, synthetic  # noqa: E402
from opensfm_amd._lib import MatchTimings, default_context  # noqa: E402

NAMES = ["A sweep", "A wait", "A merge+decide", "A class re-exam", "A queries re-examined",
         "B sweep", "B wait", "B merge+decide", "B class re-exam", "B queries re-examined",
         "pass A", "candidate list", "candidates + pass B", "emission", "whole workgroup", "workgroups"]


def phases(lib, reset=True):
    out = (C.c_ulonglong * 16)()
    assert lib.osfm_dbg_phases(out, int(reset)) == 0
    return np.array(out[:], np.float64)


def main():
    ctx = default_context(0)
    lib = _lib.load()
    scene = synthetic.make_matching_scene(1000, 2000, seed=0)
    store = matching.DescriptorStore.from_packed(scene.desc, scene.pts, scene.offsets, ctx)
    lists = {"neighbour (j - i <= 16)": neighbour_pairs(1000, 16), "exhaustive, first 60000": synthetic.all_pairs(1000)[:60000]}
    for name, pairs in lists.items():
        matching.match_pairs(store, pairs[:512], robust=False)
        phases(lib)
        tm = MatchTimings()
        matching.match_pairs(store, pairs, robust=False, timings=tm)
        ph = phases(lib)
        n = ph[15]
        print(f"== {name}: {len(pairs)} pairs, match kernel {tm.ms_match_kernel:.3f} ms, {int(n)} workgroups")
        for i, nm in enumerate(NAMES[:15]):
            if i in (4, 9):
                print(f"  {nm:26s} {ph[i] / n:9.1f} per workgroup")
            else:
                print(f"  {nm:26s} {ph[i] / n / 100.0:9.2f} us per workgroup   ({100.0 * ph[i] / max(ph[14], 1):5.1f} %)")


if __name__ == "__main__":
    main()
