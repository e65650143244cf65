#!/bin/bash
OUT=/root/repo/gpurun_out/r03_c6
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ba.py -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 1200 python -c "
import json, sys
sys.path.insert(0, '.')
import bench_ba
from opensfm_amd._lib import default_context
print(json.dumps(bench_ba.run(default_context(0))))
" > $OUT/bench_ba.json 2> $OUT/bench_ba.err; tail -3 $OUT/bench_ba.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r03_c6/bench_ba.json'))
print({k:d[k] for k in ['value','lm_iterations','run_seconds','pcg_iterations','inlier_rmse_px']}); print(d['lm_iteration']); print(d.get('grid_topology')); print(d.get('cpu_baseline'))
PY
