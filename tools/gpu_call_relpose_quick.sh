#!/bin/bash
# quick iteration: relpose GPU tests + benchmark at two batch sizes + kernel trace
OUT=/root/repo/gpurun_out/r02_relpose_q
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_zz_relpose.py -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
for P in 4096 16384 65536; do
  timeout 300 python tools/relpose_bench.py --pairs $P --matches 300 > $OUT/relpose_bench_$P.json 2> $OUT/relpose_bench_$P.err
  echo "pairs=$P: $(head -c 420 $OUT/relpose_bench_$P.json)"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/tools/relpose_bench.py --pairs 65536 --matches 300 --no-cpu > $OUT/relpose_prof.json 2> $OUT/relpose_prof.err
python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1) > $OUT/relpose_rocprof_stats.txt 2>&1
head -8 $OUT/relpose_rocprof_stats.txt
