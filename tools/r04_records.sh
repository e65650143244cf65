#!/bin/bash
# round-4 records on the final matcher / RANSAC tree: configs[3] at N = 1 (every one of the 49 995 000 pairs), configs[1] with EVERY pair that
# has matches + 5 000 empties checked against the oracle, and a kernel trace of the headline (match_fused_kernel's avg duration by grid)
OUT=/root/repo/gpurun_out/r04_records
mkdir -p $OUT
cd /root/repo
timeout 500 python bench.py --gpus 1 --images 10000 --strong --steps 2 --warmup 1 --headline-only > $OUT/configs3_n1.json 2> $OUT/configs3.err; echo "configs3 rc $?"; tail -c 300 $OUT/configs3_n1.json
timeout 400 python bench.py --full-parity --headline-only --steps 2 > $OUT/bench_full_parity.json 2> $OUT/full_parity.err; echo "full-parity rc $?"; tail -c 600 $OUT/bench_full_parity.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/bench.py --headline-only --no-cpu-baseline --emulate-world 0 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/headline_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
head -12 $OUT/headline_kernels_by_grid.txt | cut -c1-160
