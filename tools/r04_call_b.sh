#!/bin/bash
# round 4: cyclic reduction over dense clusters for the wide band -- its GPU tests, then ragged / grid scenes at configs[4] size with
# both factorisations (OSFM_BA_WIDE_LDLT = round 3's chain), and a kernel trace of the grid scene
OUT=/root/repo/gpurun_out/r04_b
mkdir -p $OUT
cd /root/repo
export PROF_WARM=1
timeout 400 python -m pytest tests/test_gpu_ba.py -m gpu -q -x -k "grid or ragged or dense or wide or two_free or unordered" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest.log
timeout 120 python tools/prof_ba.py 5000 500000 10 10 ragged > $OUT/ragged_dense.txt 2>&1; tail -3 $OUT/ragged_dense.txt
OSFM_BA_WIDE_LDLT=1 timeout 120 python tools/prof_ba.py 5000 500000 10 10 ragged > $OUT/ragged_ldlt.txt 2>&1; tail -2 $OUT/ragged_ldlt.txt
timeout 120 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_dense.txt 2>&1; tail -3 $OUT/grid_dense.txt
OSFM_BA_WIDE_LDLT=1 timeout 120 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_ldlt.txt 2>&1; tail -3 $OUT/grid_ldlt.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba_grid.py 50 100 500000 5 > $OUT/traced_grid.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/grid_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -- python /root/repo/tools/prof_ba.py 5000 500000 10 5 ragged > $OUT/traced_ragged.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/ragged_kernels_by_grid.txt 2>&1
rm -rf $OUT/trace
head -30 $OUT/grid_kernels_by_grid.txt | cut -c1-150
