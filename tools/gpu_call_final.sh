#!/bin/bash
# Round-end validation in one gpurun call: full GPU suite, default bench.py, kernel trace of bench.py, PMC passes (matcher, BA mat-vec)
OUT=/root/repo/gpurun_out/r02_final
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; head -c 1500 $OUT/bench.json; echo; tail -4 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1) > $OUT/bench_rocprof_stats.txt 2>&1
head -12 $OUT/bench_rocprof_stats.txt
rm -rf $OUT/trace
B="python /root/repo/bench.py --steps 1 --warmup 0 --no-ba --no-tracks --no-cpu-baseline --no-overlap --no-calibrated --no-float --no-guided"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- $B > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- $B > $OUT/pmc_write.log 2>&1
python /root/repo/tools/pmc_to_json.py $(ls $OUT/pmc_fetch/*.db $OUT/pmc_fetch/*/*.db 2>/dev/null | head -1) $(ls $OUT/pmc_write/*.db $OUT/pmc_write/*/*.db 2>/dev/null | head -1) 124875 $OUT/r02_match_pmc.json | tail -8
rm -rf $OUT/pmc_fetch $OUT/pmc_write
bash /root/repo/tools/pmc_ba.sh > $OUT/pmc_ba.log 2>&1; tail -6 $OUT/pmc_ba.log
