#!/bin/bash
OUT=/root/repo/gpurun_out/r03_c14
mkdir -p $OUT
cd /root/repo
timeout 600 python tools/hahog_bringup.py > $OUT/bringup.txt 2>&1; cat $OUT/bringup.txt | tail -40
