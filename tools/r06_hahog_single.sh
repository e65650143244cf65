#!/bin/bash
# Round-6 measurement: HAHOG on the bench image, single-image calls only (the kernels alone on the chip): exact tests, kernel table, timeline, rates
OUT=/root/repo/gpurun_out/${1:-r06_hahog_single}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_hahog.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
cat > /tmp/hs.py <<'P'
import sys, time; sys.path.insert(0, '/root/repo')
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
n = int(sys.argv[1])
features.hahog(im, 1e-5, 10.0, 10000, ctx=ctx)
t0 = time.perf_counter()
for _ in range(n):
    pts, desc = features.hahog(im, 1e-5, 10.0, 10000, ctx=ctx)
ms = (time.perf_counter() - t0) / n * 1e3
import hashlib
print("single float image: %.3f ms = %.1f images/s, %d features, sha %s" % (ms, 1e3 / ms, len(pts), hashlib.sha1(pts.tobytes() + desc.tobytes()).hexdigest()[:16]))
if len(sys.argv) > 2:
    im8 = np.ascontiguousarray(np.round(255 * im), np.uint8)
    for conc in (8,):
        features.hahog_batch([im8] * 8, 1e-5, 10.0, 10000, flags=features.HAHOG_ROOT | features.HAHOG_UCHAR, concurrency=conc, ctx=ctx)
        t0 = time.perf_counter()
        features.hahog_batch([im8] * 32, 1e-5, 10.0, 10000, flags=features.HAHOG_ROOT | features.HAHOG_UCHAR, concurrency=conc, ctx=ctx)
        print("uint8 host batch x%d: %.1f images/s" % (conc, 32 / (time.perf_counter() - t0)))
P
python /tmp/hs.py 20 batch 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /tmp/hs.py 5 > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/hahog_kernels_by_grid.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --timeline _ 130 > $OUT/hahog_timeline.txt 2>&1
rm -rf $OUT/trace
grep "single float" $OUT/traced.txt
head -22 $OUT/hahog_kernels_by_grid.txt | cut -c1-150
