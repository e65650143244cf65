#!/bin/bash
# One gpurun call for the BA work: parity tests, then the configs[4] run with and without the fused solve, then a kernel trace
OUT=/root/repo/gpurun_out/r02_ba
mkdir -p $OUT
cd /root/repo
if [ "$1" != "notest" ]; then timeout 600 python -m pytest tests/test_gpu_ba.py tests/test_gpu_bundle_facade.py -x -q -m gpu > $OUT/pytest_ba.log 2>&1; tail -5 $OUT/pytest_ba.log; fi
for it in 1 2 3; do timeout 300 python tools/prof_ba.py 5000 500000 10 20 2>&1 | tail -3; done | tee $OUT/run_fused.log

cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 20 > $OUT/prof.log 2>&1
python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1) > $OUT/ba_rocprof_stats.txt 2>&1
head -30 $OUT/ba_rocprof_stats.txt
rm -rf $OUT/trace
