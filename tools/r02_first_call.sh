#!/bin/bash
# Round-2 opener, one gpurun call (~4-6 GPU-minutes):  gpurun --timeout 600 -- 'bash tools/r02_first_call.sh'
#  1. the GPU tests of the three kernels that were changed after their last hardware run (relpose: sampler / ties / stop table;
#     guided; generic bearings) -- tests/test_gpu_zz_relpose.py
#  2. relpose_bench with the speculative batch schedule knob (OSFM_RELPOSE_BATCH0 = 64 (default), 16, 4, 1)
#  3. rocprofv3 kernel trace of relpose_bench (default schedule) -> per-kernel time of relpose_pairs_kernel / guided_match_kernel
OUT=/root/repo/gpurun_out/r02_first
mkdir -p $OUT
cd /root/repo
timeout 240 python -m pytest tests/test_gpu_zz_relpose.py -x -q -m gpu > $OUT/pytest_zz.log 2>&1; tail -3 $OUT/pytest_zz.log
for b in 64 16 4 1; do
  OSFM_RELPOSE_BATCH0=$b timeout 120 python /root/repo/tools/relpose_bench.py --pairs 2048 --matches 300 > $OUT/relpose_bench_b$b.json 2> $OUT/relpose_bench_b$b.err
  echo "batch0=$b: $(head -c 600 $OUT/relpose_bench_b$b.json)"
done
OSFM_RELPOSE_V2=1 timeout 120 python /root/repo/tools/relpose_bench.py --pairs 2048 --matches 300 > $OUT/relpose_bench_v2.json 2> $OUT/relpose_bench_v2.err
echo "v2 (cooperative): $(head -c 600 $OUT/relpose_bench_v2.json)"
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/tools/relpose_bench.py --pairs 2048 --matches 300 --no-cpu > $OUT/relpose_prof.json 2> $OUT/relpose_prof.err
python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace/*/*.db 2>/dev/null | head -1) > $OUT/relpose_rocprof_stats.txt 2>&1
head -20 $OUT/relpose_rocprof_stats.txt
