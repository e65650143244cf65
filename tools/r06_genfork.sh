#!/bin/bash
# Round-6 measurement: the general BA at configs[4] size: border shot pass with five columns per launch (two launches) against all nine in one
OUT=/root/repo/gpurun_out/r06_${1:-gf1}
mkdir -p $OUT
cd /root/repo
for v in "" "OSFM_BA_BORDER_CH9=1"; do
  echo "variant: $v"
  env $v PROF_WARM=1 python tools/prof_ba.py 5000 500000 10 10 general 2>&1 | tail -2
  env $v PROF_WARM=1 python tools/prof_ba.py 5000 500000 10 10 general 2>&1 | tail -2
done > $OUT/general_variants.txt 2>&1
cat $OUT/general_variants.txt
PROF_WARM=1 python tools/prof_ba.py 5000 500000 10 10 2>&1 | tail -2
python tools/prof_local_ba.py 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_bundle_general.py tests/test_gpu_ba.py -m gpu -q -x > $OUT/pytest_general.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_general.txt
