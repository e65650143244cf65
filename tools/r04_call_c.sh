#!/bin/bash
# round 4: sliced band assembly (half-width 559), data/berlin end to end, the default bench line with the new BA sections
OUT=/root/repo/gpurun_out/r04_c
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_ba.py tests/test_gpu_berlin_e2e.py -m gpu -q -x -k "half_width or berlin" -s > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest.log | cut -c1-600
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 300 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r04_c/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
ba=d.get('ba',{})
for k in ('value','lm_iteration','grid_topology','ragged_topology','local_ba','cpu_baseline'):
    print(k, json.dumps(ba.get(k))[:900])
PY
