#!/bin/bash
OUT=/root/repo/gpurun_out/r03_c5
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_berlin.py tests/test_gpu_ransac.py tests/test_gpu_matching.py tests/test_gpu_dist.py tests/test_gpu_zz_relpose.py -q -x > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
timeout 600 python tools/prof_neighbour.py > $OUT/prof_neighbour.log 2>&1; tail -26 $OUT/prof_neighbour.log
timeout 900 python bench.py --no-ba --no-tracks --no-calibrated --no-guided --no-float > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r03_c5/bench.json'))
print('value',d['value'],'stage',d['stage_ms_per_step'],'ransac',d['roofline_ransac'])
o=d.get('overlap_workload',{}); print({k:o.get(k) for k in ['value','match_kernel_ms','ransac_kernel_ms','call_ms','ransac_share_of_stream_time']}); print(o.get('cpu_baseline')); print(d.get('cpu_baseline'))
PY
