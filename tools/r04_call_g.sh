#!/bin/bash
# round 4: calibrated branch -- one chunk for the geometric stage, batched LDS reads in the refinement's ordered sums
OUT=/root/repo/gpurun_out/r04_g
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_zz_relpose.py tests/test_gpu_compat.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/bench.py --no-ba --no-hahog --no-tracks --no-overlap --no-float --no-guided --no-cpu-baseline --steps 1 --warmup 0 --emulate-world 0 > $OUT/bench_calib.json 2> $OUT/bench_calib.err
DB=$(find $OUT/trace -name "*.db" | head -1)
python /root/repo/tools/rocpd_summary.py $DB --timeline rp_,calib,bearings,match_fused,compact,Rounds 120 > $OUT/calib_timeline.txt 2>&1
python /root/repo/tools/rocpd_summary.py $DB > $OUT/calib_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r04_g/bench_calib.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('calibrated'))[:1500])
PY
grep -n "finish" $OUT/calib_kernels_by_grid.txt | cut -c1-150
