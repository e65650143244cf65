"""Round-6 measurement: the host-image batch path of HAHOG by image type, flags and concurrency (which combination is slow, and is it the library or the wrapper)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opensfm_amd import features
from opensfm_amd._lib import default_context
ctx = default_context(0)
rng = np.random.default_rng(7)
rows, cols = 1536, 2048
yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
im = np.zeros((rows, cols), np.float32)
for _ in range(1500):
    cx, cy, sg = rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(1.5, 24)
    x0, x1, y0, y1 = int(max(0, cx - 4 * sg)), int(min(cols, cx + 4 * sg)), int(max(0, cy - 4 * sg)), int(min(rows, cy + 4 * sg))
    im[y0:y1, x0:x1] += rng.uniform(-1, 1) * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * sg * sg))
im += 0.05 * rng.standard_normal((rows, cols)).astype(np.float32)
im = np.ascontiguousarray((im - im.min()) / (im.max() - im.min()), np.float32)
im8 = np.ascontiguousarray(np.round(255 * im), np.uint8)
RU = features.HAHOG_ROOT | features.HAHOG_UCHAR
for rep in range(2):
    for name, img, fl in (("float,0", im, 0), ("float,root|uchar", im, RU), ("u8,0", im8, 0), ("u8,root|uchar", im8, RU)):
        for conc in (1, 4, 8):
            features.hahog_batch([img] * 8, 1e-5, 10.0, 10000, flags=fl, concurrency=conc, ctx=ctx)
            t0 = time.perf_counter()
            r = features.hahog_batch([img] * 32, 1e-5, 10.0, 10000, flags=fl, concurrency=conc, ctx=ctx)
            dt = time.perf_counter() - t0
            print("%-18s x%d: %7.1f images/s  (%d features)" % (name, conc, 32 / dt, len(r[0][0])), flush=True)
