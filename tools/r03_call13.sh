#!/bin/bash
# BA: full parity suite after the RhsSet refactor + padded LDS ring; grid and sequence timings
OUT=/root/repo/gpurun_out/r03_c13
mkdir -p $OUT
cd /root/repo
timeout 300 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_plain.txt 2>&1; tail -4 $OUT/grid_plain.txt
timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_plain.txt 2>&1; tail -3 $OUT/prof_ba_plain.txt
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_bundle_facade.py tests/test_gpu_berlin.py -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba_grid.py 50 100 500000 3 > $OUT/grid_trace.txt 2>&1
cd /root/repo
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB --by-kernel > $OUT/grid_kernels.txt 2>&1; head -8 $OUT/grid_kernels.txt | cut -c1-140
rm -rf $OUT/trace
