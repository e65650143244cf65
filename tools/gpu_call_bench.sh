#!/bin/bash
# One gpurun call: full GPU suite, bench.py (default run), PMC passes for the HBM traffic of the matcher, kernel trace of bench.py
OUT=/root/repo/gpurun_out/r02_bench
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 1 --warmup 0 --no-ba --no-tracks --no-cpu-baseline --no-overlap --no-calibrated"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- $B > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- $B > $OUT/pmc_write.log 2>&1
python /root/repo/tools/pmc_to_json.py $(ls $OUT/pmc_fetch/*.db $OUT/pmc_fetch/*/*.db 2>/dev/null | head -1) $(ls $OUT/pmc_write/*.db $OUT/pmc_write/*/*.db 2>/dev/null | head -1) 124875 $OUT/r02_match_pmc.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1) > $OUT/bench_rocprof_stats.txt 2>&1
head -40 $OUT/bench_rocprof_stats.txt
timeout 900 python /root/repo/bench.py --full-parity --steps 1 --warmup 0 --no-ba --no-tracks --no-overlap --no-calibrated > $OUT/bench_full_parity.json 2> $OUT/bench_full_parity.err; tail -c 1200 $OUT/bench_full_parity.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/trace
