#!/bin/bash
OUT=/root/repo/gpurun_out/r03_c33
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > /dev/null 2>&1
timeout 185 python -m pytest tests/test_gpu_ba.py -m gpu -q -x > $OUT/pytest_ba.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_ba.log
