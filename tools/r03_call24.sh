#!/bin/bash
mkdir -p gpurun_out/r03_c24
OSFM_MI355_LIB=/root/repo/tools/libosfm_dbg_phases.so timeout 300 python tools/match_phases.py > gpurun_out/r03_c24/phases.txt 2>&1
tail -40 gpurun_out/r03_c24/phases.txt
