#!/bin/bash
# PMC passes over the v5 matcher (LIB = which build), 200-image probe
LIB=${1:-opensfm_amd/csrc/libosfm_mi355.so}
TAG=${2:-v5}
export OSFM_MI355_LIB=$PWD/$LIB
REPO=$PWD
cd /tmp && export TMPDIR=/tmp; mkdir -p $REPO/gpurun_out/prof
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d $REPO/gpurun_out/prof/${TAG}a -o a -- python $REPO/tools/prof_match.py 200 0 1 > $REPO/gpurun_out/prof/${TAG}a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace -d $REPO/gpurun_out/prof/${TAG}b -o b -- python $REPO/tools/prof_match.py 200 0 1 > $REPO/gpurun_out/prof/${TAG}b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace -d $REPO/gpurun_out/prof/${TAG}c -o c -- python $REPO/tools/prof_match.py 200 0 1 > $REPO/gpurun_out/prof/${TAG}c.log 2>&1
cd $REPO
for p in a b c; do python tools/rocpd_summary.py $(ls gpurun_out/prof/${TAG}$p/*/*.db | head -1) 2>&1 | grep -A12 "match_fused4" | head -14; done
