#!/bin/bash
mkdir -p gpurun_out/r03_c27
OSFM_MI355_LIB=/root/repo/tools/libosfm_dbg_phases.so timeout 300 python tools/match_phases.py > gpurun_out/r03_c27/phases.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_matching.py tests/test_gpu_float_descriptors.py -m gpu -x -q > gpurun_out/r03_c27/pytest.log 2>&1
tail -5 gpurun_out/r03_c27/pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-ba --no-tracks --no-cpu-baseline --no-calibrated --no-hahog --no-guided > gpurun_out/r03_c27/bench.json 2> gpurun_out/r03_c27/bench.err
tail -42 gpurun_out/r03_c27/phases.txt
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03_c27/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['overlap_workload']['value'], d['overlap_workload']['match_kernel_ms'], d['overlap_workload']['roofline']['frac'])
print(d.get('float_descriptors',{}).get('neighbour',{}).get('match_kernel_ms'), d.get('float_descriptors',{}).get('exhaustive',{}).get('match_kernel_ms'))
P
