#!/bin/bash
# round-3 measurement: the default bench line, the same command under a kernel trace (one row per kernel AND grid), the PMC passes of the
# matcher (HBM bytes, SQ counters) and of the BA mat-vec (HBM bytes) -- each counter set in its own run, as the guide prescribes
OUT=/root/repo/gpurun_out/r03_final
mkdir -p $OUT
cd /root/repo
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json; tail -2 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/bench.py --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/bench_traced.err
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/bench_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
B="python /root/repo/bench.py --headline-only --no-cpu-baseline --steps 1 --warmup 0"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format rocpd -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format rocpd -d $OUT/write -- $B > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format rocpd -d $OUT/sq -- $B > $OUT/sq.log 2>&1
python /root/repo/tools/pmc_to_json.py $(find $OUT/fetch -name "*.db" | head -1) $(find $OUT/write -name "*.db" | head -1) 124875 $OUT/r03_match_pmc.json > $OUT/pmc_to_json.log 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/sq -name "*.db" | head -1) > $OUT/match_sq_counters.txt 2>&1
rm -rf $OUT/fetch $OUT/write $OUT/sq
BB="python /root/repo/tools/prof_ba.py 5000 500000 10 5"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format rocpd -d $OUT/bfetch -- $BB > $OUT/bfetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format rocpd -d $OUT/bwrite -- $BB > $OUT/bwrite.log 2>&1
python /root/repo/tools/pmc_to_json.py --ba $(find $OUT/bfetch -name "*.db" | head -1) $(find $OUT/bwrite -name "*.db" | head -1) 5000000 $OUT/r03_ba_pmc.json >> $OUT/pmc_to_json.log 2>&1
rm -rf $OUT/bfetch $OUT/bwrite
ls -la $OUT; cat $OUT/pmc_to_json.log | tail -5
