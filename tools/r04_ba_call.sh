#!/bin/bash
# round 4: the window band assembly -- its GPU test, then 10 LM iterations of configs[4] under a kernel trace (one row per kernel and grid)
OUT=/root/repo/gpurun_out/${1:-r04_ba}
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_ba.py -m gpu -q -x -k "band_by_windows" > $OUT/pytest_band.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_band.log
timeout 200 python tools/prof_ba.py 5000 500000 10 20 > $OUT/run20.txt 2>&1; tail -3 $OUT/run20.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 10 > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/ba_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
head -30 $OUT/ba_kernels_by_grid.txt | cut -c1-150
