#!/bin/bash
OUT=/root/repo/gpurun_out/r04_i
mkdir -p $OUT
cd /root/repo
export PROF_WARM=1
timeout 100 python tools/prof_ba.py 5000 500000 10 20 > $OUT/run20.txt 2>&1; tail -2 $OUT/run20.txt | head -1
timeout 500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_bundle_facade.py -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 10 > $OUT/traced.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/ba_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
grep "cam_reduce\|candidate_kernel\|prior_cost\|finish_reduce\|border_rhs\|border_dots\|pcg_init\|dot2" $OUT/ba_kernels_by_grid.txt | cut -c1-150
