#!/bin/bash
# round 4: local BA after the early PCG poll; timeline of the calibrated branch's rounds on the neighbour list
OUT=/root/repo/gpurun_out/r04_f
mkdir -p $OUT
cd /root/repo
timeout 200 python tools/prof_local_ba.py 400 40000 40 > $OUT/local_ba.txt 2>&1; tail -3 $OUT/local_ba.txt
timeout 200 python -m pytest tests/test_gpu_ba.py -m gpu -q -x -k "local or fixed or lm_trajectory" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/bench.py --no-ba --no-hahog --no-tracks --no-overlap --no-float --no-guided --no-cpu-baseline --steps 1 --warmup 0 --emulate-world 0 > $OUT/bench_calib.json 2> $OUT/bench_calib.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python /root/repo/tools/rocpd_summary.py $DB --timeline rp_,calib,bearings,gather_b,match_fused,compact 260 > $OUT/calib_timeline.txt 2>&1
rm -rf $OUT/trace
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r04_f/bench_calib.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('calibrated'))[:1500])
PY
tail -5 $OUT/calib_timeline.txt
