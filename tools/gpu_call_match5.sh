#!/bin/bash
# GPU call: v5 matcher correctness (matcher + ransac suites: the F-RANSAC oracle changed too) and A/B speed against the v4 build
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_matching.py tests/test_gpu_ransac.py -x -q -m gpu > gpurun_out/match5_tests.log 2>&1
echo "exit $?" >> gpurun_out/match5_tests.log
tail -15 gpurun_out/match5_tests.log
echo "== v5"; timeout 300 python tools/prof_match.py 200 0 3 2>&1 | tail -3
echo "== v4"; OSFM_MI355_LIB=tools/libosfm_v4.so timeout 300 python tools/prof_match.py 200 0 3 2>&1 | tail -3
