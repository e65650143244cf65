#!/bin/bash
# One gpurun call: HAHOG parity tests, the bench workload with the compiled reference beside it, and a kernel trace of the same
# function summarised per (kernel, grid).  Outputs under gpurun_out/hahog/ (scratch).
OUT=${OUT:-/root/repo/gpurun_out/hahog}
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_hahog.py -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 300 python - > $OUT/hahog_bench.json 2> $OUT/hahog_bench.err <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from opensfm_amd._lib import default_context
print(json.dumps(bench.hahog_bench(default_context(0), True)))
PY
tail -3 $OUT/hahog_bench.err; cat $OUT/hahog_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import bench
from opensfm_amd._lib import default_context
bench.hahog_bench(default_context(0), False, reps=3)" > $OUT/trace.txt 2>&1
cd /root/repo
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB --by-kernel > $OUT/hahog_kernels.txt 2>&1; head -8 $OUT/hahog_kernels.txt | cut -c1-150
rm -rf $OUT/trace
