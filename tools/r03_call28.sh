#!/bin/bash
# HAHOG: GPU tests, then the bench workload alone with a kernel trace
OUT=/root/repo/gpurun_out/r03_c28
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_hahog.py -q -x > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
timeout 600 python - > $OUT/hahog_bench.json 2> $OUT/hahog_bench.err <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from opensfm_amd._lib import default_context
print(json.dumps(bench.hahog_bench(default_context(0), True)))
PY
tail -3 $OUT/hahog_bench.err; cat $OUT/hahog_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import bench
from opensfm_amd._lib import default_context
bench.hahog_bench(default_context(0), False, reps=3)" > $OUT/trace.txt 2>&1
cd /root/repo
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB --by-kernel > $OUT/hahog_kernels.txt 2>&1; head -16 $OUT/hahog_kernels.txt | cut -c1-150
rm -rf $OUT/trace
