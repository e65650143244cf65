#!/bin/bash
# Round-6 measurement: HAHOG on the bench image, kernel table and launch timeline of a few single-image calls
OUT=/root/repo/gpurun_out/r06_hahog
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import bench
from opensfm_amd._lib import default_context
ctx = default_context(0)
bench.hahog_bench(ctx, False, reps=1)
print(bench.hahog_bench(ctx, False, reps=4))
" > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/hahog_kernels_by_grid.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --timeline _ 160 > $OUT/hahog_timeline.txt 2>&1
rm -rf $OUT/trace
tail -2 $OUT/traced.txt | cut -c1-300
head -25 $OUT/hahog_kernels_by_grid.txt | cut -c1-150
