#!/bin/bash
# BA after the one-launch-per-level cyclic reduction: parity tests, then a kernel trace of 10 LM iterations at configs[4]
OUT=/root/repo/gpurun_out/r03_c7
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_bundle_facade.py tests/test_gpu_berlin.py -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_plain.txt 2>&1; tail -3 $OUT/prof_ba_plain.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 10 > $OUT/prof_ba_trace.txt 2>&1
cd /root/repo
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $OUT/ba_kernels.txt 2>&1; head -45 $OUT/ba_kernels.txt
rm -rf $OUT/trace
