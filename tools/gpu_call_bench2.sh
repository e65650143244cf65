#!/bin/bash
OUT=/root/repo/gpurun_out/r02_bench2
mkdir -p $OUT
cd /root/repo
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json; tail -5 $OUT/bench.err
bash tools/pmc_ba.sh > $OUT/pmc_ba.log 2>&1; tail -5 $OUT/pmc_ba.log
