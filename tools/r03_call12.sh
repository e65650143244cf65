#!/bin/bash
# wide-band direct solver: parity tests, then the grid topology at configs[4] size (plain + kernel trace)
OUT=/root/repo/gpurun_out/r03_c12
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ba.py -q -x -k "wide_bandwidth or grid_topology or long_tracks or unordered or lm_trajectory or several_cameras or lund or banded_and" > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
timeout 300 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_plain.txt 2>&1; tail -4 $OUT/grid_plain.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba_grid.py 50 100 500000 3 > $OUT/grid_trace.txt 2>&1
cd /root/repo
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB --by-kernel > $OUT/grid_kernels.txt 2>&1; head -25 $OUT/grid_kernels.txt | cut -c1-140
rm -rf $OUT/trace
timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_plain.txt 2>&1; tail -3 $OUT/prof_ba_plain.txt
