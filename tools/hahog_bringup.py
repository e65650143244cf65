"""Bring-up of the GPU HAHOG extractor against the compiled reference (oracle/_ref/libhahog_ref.so): final keypoints / descriptors,
and -- to localise a difference -- how many of the reference's detected features the GPU set contains."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from opensfm_amd import features  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "hahog_berlin01.npz"))
grey = g["grey"]
im = grey.astype(np.float32) / 255
for target in (1500, 200):
    t0 = time.time()
    pts, desc = features.hahog(im, 1e-5, 10.0, target)
    t1 = time.time()
    rp, rd = oracle.hahog_ref(im, 1e-5, 10.0, target)
    t2 = time.time()
    print(f"target {target}: gpu {len(pts)} features in {t1 - t0:.3f}s, reference {len(rp)} in {t2 - t1:.3f}s")
    n = min(len(pts), len(rp))
    if n:
        dxy = np.abs(pts[:n, :2] - rp[:n, :2]).max(axis=1)
        print("  rows with identical (x, y):", int((dxy == 0).sum()), "of", n, " first mismatch at", int(np.argmax(dxy > 0)) if (dxy > 0).any() else None)
        same = dxy == 0
        if same.any():
            print("  on those: max |size diff|", np.abs(pts[:n, 2] - rp[:n, 2])[same].max(), " max |angle diff|", np.abs(pts[:n, 3] - rp[:n, 3])[same].max(),
                  " max |desc diff|", np.abs(desc[:n] - rd[:n])[same].max(), " rows with identical desc", int((np.abs(desc[:n] - rd[:n])[same].max(axis=1) == 0).sum()))
        # set comparison irrespective of order
        a = {tuple(r) for r in np.round(pts[:, :3], 4).tolist()}
        b = {tuple(r) for r in np.round(rp[:, :3], 4).tolist()}
        print("  (x, y, size) sets: common", len(a & b), "gpu only", len(a - b), "reference only", len(b - a))
    if len(pts) and len(rp):
        print("  gpu first rows", pts[:3].tolist())
        print("  ref first rows", rp[:3].tolist())
    if n:
        size_same = pts[:n, 2] == rp[:n, 2]
        ang_same = pts[:n, 3] == rp[:n, 3]
        dd = np.abs(desc[:n] - rd[:n]).max(axis=1)
        print("  size identical rows", int(size_same.sum()), " angle identical rows", int(ang_same.sum()), " desc identical rows", int((dd == 0).sum()))
        print("  desc identical among size-identical rows", int(((dd == 0) & size_same).sum()), "of", int(size_same.sum()),
              "; among size-different rows", int(((dd == 0) & ~size_same).sum()), "of", int((~size_same).sum()))
        # first orientation of a feature vs later ones
        first = np.r_[True, (pts[1:n, 0] != pts[:n - 1, 0]) | (pts[1:n, 1] != pts[:n - 1, 1])]
        print("  desc identical among first orientations", int(((dd == 0) & first).sum()), "of", int(first.sum()))
        nz = np.count_nonzero(desc[:n] != rd[:n], axis=1)
        print("  differing entries per differing row: median", float(np.median(nz[nz > 0])) if (nz > 0).any() else 0, " max", int(nz.max()))
cfg = {"feature_root": True, "hahog_normalize_to_uchar": True, "hahog_peak_threshold": 1e-5, "hahog_edge_threshold": 10}
p8, d8 = features.extract_features_hahog(grey, cfg, 1500)
print("uint8 descriptors vs golden: rows", len(d8), "entries differing", int(np.count_nonzero(d8 != g["desc_u8"])), "max diff", float(np.abs(d8 - g["desc_u8"]).max()))
