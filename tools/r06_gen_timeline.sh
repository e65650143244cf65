#!/bin/bash
# Round-6 measurement: the launch timeline (both streams) of the general BA at configs[4] size, last ~2 LM iterations
OUT=/root/repo/gpurun_out/r06_gentl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROF_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr -- python /root/repo/tools/prof_ba.py 5000 500000 10 6 general > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/tr -name "*.db" | head -1) --timeline _ 1500 > $OUT/general_timeline.txt 2>&1
PROF_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr2 -- python /root/repo/tools/prof_ba.py 5000 500000 10 6 > $OUT/traced2.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/tr2 -name "*.db" | head -1) --timeline _ 1200 > $OUT/headline_timeline.txt 2>&1
rm -rf $OUT/tr $OUT/tr2
tail -3 $OUT/traced.txt
