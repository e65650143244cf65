#!/bin/bash
# band assembly by groups of shots: parity subset, timing against the per-shot kernel, kernel trace
OUT=/root/repo/gpurun_out/r03_c19
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ba.py -q -x -k "lm_trajectory or lund or banded_and or fixed_blocks or several_cameras or unordered or up_vector" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_plain.txt 2>&1; tail -3 $OUT/prof_ba_plain.txt
OSFM_BA_BAND_PER_SHOT=1 timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_per_shot.txt 2>&1; tail -2 $OUT/prof_ba_per_shot.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 10 > $OUT/prof_ba_trace.txt 2>&1
cd /root/repo
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $OUT/ba_kernels.txt 2>&1; head -14 $OUT/ba_kernels.txt | cut -c1-150
rm -rf $OUT/trace
