#!/bin/bash
OUT=/root/repo/gpurun_out/r04_e
mkdir -p $OUT
cd /root/repo
timeout 300 python tools/hahog_batch_bench.py > $OUT/hahog.json 2> $OUT/hahog.err; echo "bench rc $?"; tail -c 1800 $OUT/hahog.json; tail -3 $OUT/hahog.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/hahog_batch_trace.py 8 16 > $OUT/traced.txt 2>&1; grep "images/s" $OUT/traced.txt
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-kernel > $OUT/hahog_batch_kernels.txt 2>&1
rm -rf $OUT/trace
head -24 $OUT/hahog_batch_kernels.txt | cut -c1-150
