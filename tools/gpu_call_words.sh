#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_words.py -x -q -m gpu > gpurun_out/words_tests.log 2>&1
echo "exit $?" >> gpurun_out/words_tests.log
tail -25 gpurun_out/words_tests.log
