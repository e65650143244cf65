#!/bin/bash
# round 4: HAHOG -- parity tests (compiled reference, batch, fused smoothing), images per second, kernel trace of the single-image call
OUT=/root/repo/gpurun_out/r04_d
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_hahog.py tests/test_gpu_berlin_e2e.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log | cut -c1-300
timeout 300 python tools/hahog_batch_bench.py --cpu > $OUT/hahog.json 2> $OUT/hahog.err; echo "bench rc $?"; tail -c 2600 $OUT/hahog.json; tail -3 $OUT/hahog.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python -c "
import sys; sys.path.insert(0, '/root/repo')
import bench
from opensfm_amd._lib import default_context
from opensfm_amd import features
features.hahog_batch = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('single-image trace'))
bench.hahog_bench(default_context(0), False, reps=3)" > $OUT/trace.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-kernel > $OUT/hahog_kernels.txt 2>&1; head -12 $OUT/hahog_kernels.txt | cut -c1-150
rm -rf $OUT/trace
