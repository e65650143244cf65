#!/bin/bash
# full GPU test suite, then the default bench under rocprofv3 kernel trace
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/final2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json
