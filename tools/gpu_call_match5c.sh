#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_matching.py -x -q -m gpu > gpurun_out/match5_tests.log 2>&1
echo "exit $?" >> gpurun_out/match5_tests.log
tail -8 gpurun_out/match5_tests.log
echo "== v5c"; timeout 300 python tools/prof_match.py 200 0 3 2>&1 | tail -1; timeout 300 python tools/prof_match.py 400 0 2 2>&1 | tail -1
echo "== v5c NORECHECK"; OSFM_MI355_LIB=tools/libosfm_NORECHECK.so timeout 300 python tools/prof_match.py 200 0 3 2>&1 | tail -1; OSFM_MI355_LIB=tools/libosfm_NORECHECK.so timeout 300 python tools/prof_match.py 400 0 2 2>&1 | tail -1
