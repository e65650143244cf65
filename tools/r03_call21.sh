#!/bin/bash
OUT=/root/repo/gpurun_out/r03_c21
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_compat.py -q -x > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -5 $OUT/smoke.log
