"""Where a workgroup of HAHOG's orientation_kernel / descriptor_kernel spends its time (instrumented build: hahog.hip compiled with
-DOSFM_DBG_PHASES, linked with the product's other objects):  OSFM_MI355_LIB=tools/libosfm_dbg_phases.so python tools/hahog_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from opensfm_amd import _lib  # noqa: E402
from opensfm_amd._lib import default_context  # noqa: E402

NAMES = ["or: plan + taps", "or: sample patch", "or: smooth", "or: per-pixel terms", "or: ordered sum", "or: filter + peaks", "or: whole", "or: workgroups",
         "de: plan", "de: sample patch", "de: per-pixel terms", "de: ordered sum", "de: normalise + store", "de: whole", "de: workgroups"]


def main():
    ctx = default_context(0)
    lib = _lib.load()
    out = (C.c_ulonglong * 16)()
    bench.hahog_bench(ctx, False, reps=1)
    assert lib.osfm_dbg_hahog_phases(out, 1) == 0
    r = bench.hahog_bench(ctx, False, reps=3)
    assert lib.osfm_dbg_hahog_phases(out, 1) == 0
    ph = np.array(out[:], np.float64)
    print(r["ms_per_image"], "ms per image,", r["features"], "features")
    for base, n in ((0, ph[7]), (8, ph[14])):
        for i in range(base, base + (7 if base == 0 else 6)):
            print(f"  {NAMES[i]:24s} {ph[i] / n / 100.0:9.2f} us per workgroup")
        print(f"  workgroups {int(n)}")


if __name__ == "__main__":
    main()
