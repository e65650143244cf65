#!/bin/bash
# HBM traffic of the BA mat-vec: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over tools/prof_ba.py at configs[4] -> profiles-ready JSON
OUT=/root/repo/gpurun_out/r02_ba
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/tools/prof_ba.py 5000 500000 10 5"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o f -- $B > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o w -- $B > $OUT/pmc_write.log 2>&1
python /root/repo/tools/pmc_to_json.py --ba $(ls $OUT/pmc_fetch/*.db $OUT/pmc_fetch/*/*.db 2>/dev/null | head -1) $(ls $OUT/pmc_write/*.db $OUT/pmc_write/*/*.db 2>/dev/null | head -1) 5000000 $OUT/r02_ba_pmc.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
