#!/bin/bash
# Round-6 measurement: local bundle adjustment with the one-workgroup band factor against the cyclic reduction; BA tests; a kernel trace of the local run
TAG=${1:-sb1}
OUT=/root/repo/gpurun_out/r06_$TAG
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
python tools/prof_local_ba.py 400 40000 50 oracle > $OUT/local_sband.txt 2>&1; tail -8 $OUT/local_sband.txt
OSFM_BA_NO_SBAND=1 python tools/prof_local_ba.py 400 40000 50 > $OUT/local_bcr.txt 2>&1; tail -2 $OUT/local_bcr.txt
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_bundle_general.py tests/test_gpu_bundle_facade.py tests/test_gpu_compat.py -m gpu -q --durations=5 > $OUT/pytest_ba.txt 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest_ba.txt
cd /tmp && export TMPDIR=/tmp
PROF_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr -- python /root/repo/tools/prof_local_ba.py > $OUT/traced_local.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/tr -name "*.db" | head -1) > $OUT/local_ba_kernels_by_grid.txt 2>&1
python /root/repo/tools/rocpd_summary.py $(find $OUT/tr -name "*.db" | head -1) --timeline _ 400 > $OUT/local_timeline.txt 2>&1
rm -rf $OUT/tr
head -12 $OUT/local_ba_kernels_by_grid.txt
