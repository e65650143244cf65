#!/bin/bash
OUT=/root/repo/gpurun_out/r03_c32
mkdir -p $OUT
cd /root/repo
timeout 200 python bench.py --no-ba --no-tracks --no-cpu-baseline --no-calibrated --no-guided --no-hahog > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r03_c32/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'])
o=d['overlap_workload']; print({k:v for k,v in o.items() if not isinstance(v,dict)}, o['roofline'])
f=d['float_descriptors']; print(f['exhaustive']['value'], f['neighbour']['value'], f['neighbour']['call_ms'])
P
