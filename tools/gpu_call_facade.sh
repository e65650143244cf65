#!/bin/bash
# GPU call: bundle facade / adapter tests + general BA parity
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bundle_facade.py tests/test_gpu_bundle_general.py tests/test_gpu_ba.py -x -q -m gpu > gpurun_out/facade_tests.log 2>&1
echo "exit $?" >> gpurun_out/facade_tests.log
tail -40 gpurun_out/facade_tests.log
