#!/bin/bash
# Opener for the next round: one gpurun call that re-establishes the baseline and collects the counters the open items need.
#   1. the GPU suite + smoke (everything of round 2 must still be green)
#   2. bench.py (default) -> gpurun_out/r03_first/bench.json
#   3. kernel trace of the BA run (per-kernel durations at configs[4]) and of the guided workload
#   4. PMC passes for the guided kernel (VALU / scalar-memory utilisation: is phase 1 issue-bound or latency-bound?)
OUT=/root/repo/gpurun_out/r03_first
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 600 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_ba -o t -- python /root/repo/tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba.log 2>&1
python /root/repo/tools/rocpd_summary.py $(ls $OUT/trace_ba/*.db $OUT/trace_ba/*/*.db 2>/dev/null | head -1) > $OUT/ba_rocprof_stats.txt 2>&1
rm -rf $OUT/trace_ba; head -25 $OUT/ba_rocprof_stats.txt
G="python /root/repo/bench.py --steps 1 --warmup 0 --no-ba --no-tracks --no-cpu-baseline --no-overlap --no-calibrated --no-float"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_guided -o g -- $G > $OUT/pmc_guided.log 2>&1
ls $OUT/pmc_guided | head
