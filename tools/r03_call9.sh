#!/bin/bash
# after: PCG test before the preconditioner, E blocks from the Jacobian kernel, cooperative point gradient, fork before the band assembly, level kernel tile in registers
OUT=/root/repo/gpurun_out/r03_c9
mkdir -p $OUT
cd /root/repo
timeout 120 tools/ubench_bcr > $OUT/ubench_bcr.txt 2>&1; cat $OUT/ubench_bcr.txt
timeout 300 python -m pytest tests/test_gpu_ba.py -q -x -k "lm_trajectory or lund_scale or banded_and or long_tracks or fixed_blocks or several_cameras" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_plain.txt 2>&1; tail -3 $OUT/prof_ba_plain.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 10 > $OUT/prof_ba_trace.txt 2>&1
cd /root/repo
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $OUT/ba_kernels.txt 2>&1; grep "bcr_" $OUT/ba_kernels.txt | head -40
rm -rf $OUT/trace
