#!/bin/bash
OUT=/root/repo/gpurun_out/r02_streams
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 3 --warmup 1 --no-ba --no-tracks --no-calibrated --no-float --no-guided --no-cpu-baseline"
for v in two one; do
  if [ $v = one ]; then export OSFM_MATCH_ONE_STREAM=1; else unset OSFM_MATCH_ONE_STREAM; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$v -o t -- $B > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace_$v/*.db $OUT/trace_$v/*/*.db 2>/dev/null | head -1) > $OUT/stats_$v.txt 2>&1
  rm -rf $OUT/trace_$v
  python - <<PY
import json
d = json.load(open("$OUT/bench_$v.json"))
o = d["overlap_workload"]
print("$v", "value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["stage_ms_per_step"], "| overlap", o["value"], o["match_kernel_ms"], o["ransac_kernel_ms"], o["call_ms"])
PY
  grep -E "match_fused|ransac_pairs|gather_matches" $OUT/stats_$v.txt
done
