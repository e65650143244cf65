#!/bin/bash
# guided matching with the float32 pre-test: parity tests + the bench's guided workload
OUT=/root/repo/gpurun_out/r03_c20
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_zz_relpose.py tests/test_gpu_flow.py -q -x -k "guided or flow or device" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python bench.py --no-ba --no-tracks --no-overlap --no-calibrated --no-float --no-hahog --no-cpu-baseline --steps 1 --warmup 0 > $OUT/bench_guided.json 2> $OUT/bench_guided.err
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r03_c20/bench_guided.json') if l.startswith('{')][0])
print(json.dumps(d.get('guided'))[:1500])
PY
