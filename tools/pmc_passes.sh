#!/bin/bash
# HBM-traffic counters of the two dominant kernels, each counter in its own rocprofv3 run (the guide's recipe: --pmc with --kernel-trace only):
#   matcher   FETCH_SIZE / WRITE_SIZE of match_fused_kernel on `bench.py --headline-only --steps 1`      -> <tag>_match_pmc.json (+ SQ counters)
#   BA        FETCH_SIZE / WRITE_SIZE of the Schur mat-vec pair on tools/prof_ba.py at configs[4]         -> <tag>_ba_pmc.json
#   BA generic rows (a BROWN camera, free bias, control points: gen_schur_point<2,0> + gen_schur_shot)     -> <tag>_ba_generic_pmc.json
# usage (on the GPU box): bash tools/pmc_passes.sh r05 ; then copy gpurun_out/<tag>_pmc/*.json into profiles/
TAG=${1:-r05}
OUT=/root/repo/gpurun_out/${TAG}_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --headline-only --no-cpu-baseline --steps 1 --warmup 0"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format rocpd -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format rocpd -d $OUT/write -- $B > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format rocpd -d $OUT/sq -- $B > $OUT/sq.log 2>&1
python /root/repo/tools/pmc_to_json.py $(find $OUT/fetch -name "*.db" | head -1) $(find $OUT/write -name "*.db" | head -1) 124875 $OUT/${TAG}_match_pmc.json > $OUT/pmc_to_json.log 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/sq -name "*.db" | head -1) > $OUT/${TAG}_match_sq_counters.txt 2>&1
rm -rf $OUT/fetch $OUT/write $OUT/sq
BB="python /root/repo/tools/prof_ba.py 5000 500000 10 5"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format rocpd -d $OUT/bfetch -- $BB > $OUT/bfetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format rocpd -d $OUT/bwrite -- $BB > $OUT/bwrite.log 2>&1
python /root/repo/tools/pmc_to_json.py --ba $(find $OUT/bfetch -name "*.db" | head -1) $(find $OUT/bwrite -name "*.db" | head -1) 5000000 $OUT/${TAG}_ba_pmc.json >> $OUT/pmc_to_json.log 2>&1
rm -rf $OUT/bfetch $OUT/bwrite
BG="python /root/repo/tools/prof_ba.py 5000 500000 10 5 general"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format rocpd -d $OUT/gfetch -- $BG > $OUT/gfetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format rocpd -d $OUT/gwrite -- $BG > $OUT/gwrite.log 2>&1
python /root/repo/tools/pmc_to_json.py --ba-generic $(find $OUT/gfetch -name "*.db" | head -1) $(find $OUT/gwrite -name "*.db" | head -1) 5000000 $OUT/${TAG}_ba_generic_pmc.json >> $OUT/pmc_to_json.log 2>&1
rm -rf $OUT/gfetch $OUT/gwrite
ls -la $OUT; tail -5 $OUT/pmc_to_json.log
