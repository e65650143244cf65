#!/bin/bash
# round 4, final tree: the default bench line and a kernel trace of the headline (match_fused_kernel's average duration by grid)
OUT=/root/repo/gpurun_out/r04_final3
mkdir -p $OUT
cd /root/repo
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 150 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --headline-only --no-cpu-baseline --emulate-world 0 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/headline_kernel_stats.csv
rm -rf $OUT/trace
head -8 $OUT/headline_kernel_stats.csv | cut -c1-200
