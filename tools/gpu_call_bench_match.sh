#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --no-ba --no-tracks --no-cpu-baseline --no-calibrated --steps 3 --warmup 1 > gpurun_out/bench_match.json 2> gpurun_out/bench_match.err
python - <<'PY'
import json
for l in open('gpurun_out/bench_match.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])); print(json.dumps(d.get('overlap_workload'))[:800]); print(json.dumps(d.get('stage_ms_per_step')))
PY
tail -3 gpurun_out/bench_match.err
