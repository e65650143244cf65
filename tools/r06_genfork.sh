#!/bin/bash
# Round-6 measurement: the general BA at configs[4] size with the side stream forked before the assembly / after it / after the first level
OUT=/root/repo/gpurun_out/r06_${1:-gf1}
mkdir -p $OUT
cd /root/repo
for f in 2 1 0; do
  echo "OSFM_BA_FORK=$f"
  OSFM_BA_FORK=$f PROF_WARM=1 python tools/prof_ba.py 5000 500000 10 10 general 2>&1 | tail -2
done > $OUT/general_fork.txt 2>&1
cat $OUT/general_fork.txt
timeout 600 python -m pytest tests/test_gpu_bundle_general.py -m gpu -q -x > $OUT/pytest_general.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_general.txt
