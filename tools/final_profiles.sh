#!/bin/bash
# round-end measurement: full bench under rocprofv3 kernel trace, then separate PMC passes over the
# matching-only bench (1 step) for the HBM traffic of match_fused_kernel.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/bench.py > $OUT/bench.json 2> $OUT/bench.err
B="python /root/repo/bench.py --no-ba --no-cpu-baseline --no-tracks --steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace -d $OUT/sq -o sq -- $B > $OUT/sq.log 2>&1
tail -c 600 $OUT/bench.json
