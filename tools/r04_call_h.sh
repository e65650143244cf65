#!/bin/bash
OUT=/root/repo/gpurun_out/r04_h
mkdir -p $OUT
cd /root/repo
export PROF_WARM=1
for cs in 9 10; do OSFM_BA_CS=$cs timeout 120 python tools/prof_ba.py 5000 500000 10 20 > $OUT/seq_cs$cs.txt 2>&1; echo "cs $cs: $(tail -2 $OUT/seq_cs$cs.txt | head -1)"; done
timeout 300 python bench.py --no-ba --no-hahog --no-tracks --no-overlap --no-float --no-guided --no-cpu-baseline --steps 1 --warmup 0 --emulate-world 0 > $OUT/bench_calib.json 2> $OUT/bench_calib.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r04_h/bench_calib.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('calibrated',{}).get('match_end_to_end'))[:800])
PY
