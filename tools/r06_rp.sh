#!/bin/bash
# Round-6 working call for the calibrated branch: its GPU tests, the two calibrated bench legs, one kernel trace of them.
#   gpurun --timeout 900 -- 'bash tools/r06_rp.sh <tag> [notests] [notrace]'
TAG=${1:-rp1}
OUT=/root/repo/gpurun_out/r06_$TAG
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
BENCH="python /root/repo/bench.py --steps 1 --warmup 0 --no-ba --no-tracks --no-overlap --no-float --no-guided --no-hahog --no-cpu-baseline --emulate-world 0"
if [ "$2" != "notests" ]; then
  timeout 600 python -m pytest tests/test_gpu_zz_relpose.py -m gpu -q -x --durations=5 > $OUT/pytest_rp.txt 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest_rp.txt
fi
timeout 300 $BENCH > $OUT/bench_cal.json 2> $OUT/bench_cal.err; echo "bench rc $?"
python - <<P
import json
d=json.loads(open("$OUT/bench_cal.json").read().strip().splitlines()[-1])["calibrated"]
print(json.dumps(d.get("geometric_stage",d),indent=0)[:600]); print(json.dumps(d.get("match_end_to_end"),indent=0))
P
if [ "$3" != "notrace" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr -- $BENCH > $OUT/traced.txt 2>&1
  python /root/repo/tools/rocpd_summary.py $(find $OUT/tr -name "*.db" | head -1) > $OUT/cal_kernels_by_grid.txt 2>&1
  python /root/repo/tools/rocpd_summary.py $(find $OUT/tr -name "*.db" | head -1) --timeline rp_,match_fused,gather_bearings,compact_by,copyBuffer 160 > $OUT/cal_timeline.txt 2>&1
  rm -rf $OUT/tr
  python - <<P
import re,collections
tot=collections.defaultdict(lambda:[0,0.0])
for l in open("$OUT/cal_kernels_by_grid.txt").read().splitlines()[1:]:
    f=l.split()
    if len(f)<9: continue
    name=re.sub(r'.*N_1\d+','',f[0])[:24]
    tot[name][0]+=int(f[3]); tot[name][1]+=float(f[7])
for k,v in sorted(tot.items(), key=lambda x:-x[1][1])[:14]: print(k, v[0], round(v[1],2))
P
fi
