#!/bin/bash
# Round-6 measurement: the general BA at configs[4] size, run time and one kernel trace
OUT=/root/repo/gpurun_out/r06_${1:-gt1}
mkdir -p $OUT
cd /root/repo
PROF_WARM=1 python tools/prof_ba.py 5000 500000 10 10 general 2>&1 | tail -2
PROF_WARM=1 python tools/prof_ba.py 5000 500000 10 10 general 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_bundle_general.py tests/test_gpu_bundle_facade.py -m gpu -q -x > $OUT/pytest_general.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_general.txt
cd /tmp && export TMPDIR=/tmp
PROF_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr -- python /root/repo/tools/prof_ba.py 5000 500000 10 10 general > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/tr -name "*.db" | head -1) > $OUT/general_kernels_by_grid.txt 2>&1
rm -rf $OUT/tr
grep -E "gen_prior|gen_border_shot|gen_border_point|gen_eval" $OUT/general_kernels_by_grid.txt | cut -c1-150
