#!/bin/bash
OUT=/root/repo/gpurun_out/r03_c4
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_berlin.py -q -x > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
timeout 600 python tools/prof_neighbour.py > $OUT/prof_neighbour.log 2>&1; tail -22 $OUT/prof_neighbour.log
