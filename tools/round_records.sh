#!/bin/bash
# The round's records in one GPU call: the default bench line, the headline command under `rocprofv3 --kernel-trace --stats`, one kernel
# trace (a row per kernel and grid) per BA workload, the whole GPU suite.  Outputs under gpurun_out/<tag>_records/ (scratch): copy what is
# quoted into profiles/ under the round's name (profiles/<tag>_README.md says which figure comes from which file).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round_records.sh r05'
TAG=${1:-r05}
OUT=/root/repo/gpurun_out/${TAG}_records
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 300 $OUT/${TAG}_bench.json; echo
timeout 600 python -m pytest tests -m gpu -q --durations=8 -s > $OUT/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/${TAG}_pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/headline -o headline -- python /root/repo/bench.py --headline-only --no-cpu-baseline > $OUT/headline.json 2> $OUT/headline.err
cp $(find $OUT/headline -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_headline_rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf $OUT/headline
trace() {  # name, command...
  local name=$1; shift
  PROF_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr_$name -- "$@" > $OUT/traced_$name.txt 2>&1
  python /root/repo/tools/rocpd_summary.py $(find $OUT/tr_$name -name "*.db" | head -1) > $OUT/${TAG}_${name}_kernels_by_grid.txt 2>&1
  rm -rf $OUT/tr_$name
}
trace ba python /root/repo/tools/prof_ba.py 5000 500000 10 10
trace ba_general python /root/repo/tools/prof_ba.py 5000 500000 10 10 general
trace ba_grid python /root/repo/tools/prof_ba_grid.py 50 100 500000 6
trace ba_ragged python /root/repo/tools/prof_ba.py 5000 500000 10 10 ragged
trace local_ba python /root/repo/tools/prof_local_ba.py
ls $OUT
