#!/bin/bash
# full GPU suite
OUT=/root/repo/gpurun_out/r03_c18
mkdir -p $OUT
cd /root/repo
timeout 1700 python -m pytest tests -q -m gpu --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "exit $?" >> $OUT/pytest_gpu.txt; tail -30 $OUT/pytest_gpu.txt
