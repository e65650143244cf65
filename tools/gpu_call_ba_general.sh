#!/bin/bash
OUT=/root/repo/gpurun_out/r02_ba_general
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_bundle_general.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log
