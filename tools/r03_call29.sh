#!/bin/bash
mkdir -p gpurun_out/r03_c29
OSFM_MI355_LIB=/root/repo/tools/libosfm_dbg_phases.so timeout 300 python tools/hahog_phases.py > gpurun_out/r03_c29/phases.txt 2>&1
tail -25 gpurun_out/r03_c29/phases.txt
