#!/bin/bash
# round 3, GPU call 2: the reorganised F-RANSAC (first-round wave kernel + long-run kernel): parity, then the bench's matching legs
OUT=/root/repo/gpurun_out/r03_c3
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ransac.py tests/test_gpu_matching.py tests/test_gpu_float_descriptors.py tests/test_gpu_zz_relpose.py -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python bench.py --no-ba --no-tracks --no-calibrated --no-guided > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r03_c3/bench.json'))
print('value',d['value'],'stage',d['stage_ms_per_step'],'ransac',d['roofline_ransac'])
o=d.get('overlap_workload',{}); print({k:o.get(k) for k in ['value','match_kernel_ms','ransac_kernel_ms','call_ms','ransac_share_of_stream_time']}); print(o.get('cpu_baseline')); print(d.get('cpu_baseline'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python /root/repo/bench.py --steps 2 --warmup 1 --no-ba --no-tracks --no-cpu-baseline --no-calibrated --no-float --no-guided > $OUT/trace_bench.json 2> $OUT/trace.err
python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1) > $OUT/rocprof_stats.txt 2>&1
rm -rf $OUT/trace; head -12 $OUT/rocprof_stats.txt
