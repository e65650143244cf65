#!/bin/bash
# round 4, re-entry: kernel trace of 10 LM iterations of configs[4] on HEAD (one row per kernel and grid), the ragged scene, local BA calls
OUT=/root/repo/gpurun_out/r04_a
mkdir -p $OUT
cd /root/repo
timeout 120 python tools/prof_ba.py 5000 500000 10 20 > $OUT/run20.txt 2>&1; tail -3 $OUT/run20.txt
timeout 120 python tools/prof_ba.py 5000 500000 10 10 ragged > $OUT/ragged10.txt 2>&1; tail -3 $OUT/ragged10.txt
timeout 200 python tools/prof_local_ba.py 400 40000 40 oracle > $OUT/local_ba.txt 2>&1; tail -12 $OUT/local_ba.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 10 > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/ba_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace2 -- python /root/repo/tools/prof_local_ba.py 400 40000 10 > $OUT/traced_local.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace2 -name "*.db" | head -1) > $OUT/local_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace2
head -40 $OUT/ba_kernels_by_grid.txt | cut -c1-150
