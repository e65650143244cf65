#!/bin/bash
# standalone kernel durations: everything on one stream (OSFM_BA_ONE_STREAM), 5 LM iterations of configs[4] under a kernel trace
OUT=/root/repo/gpurun_out/${1:-r04_ba_os}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OSFM_BA_ONE_STREAM=1 timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 5 ${2:-} > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/ba_kernels_one_stream.txt 2>&1
rm -rf $OUT/trace
tail -3 $OUT/traced.txt | cut -c1-200
head -${3:-32} $OUT/ba_kernels_one_stream.txt | cut -c1-150
