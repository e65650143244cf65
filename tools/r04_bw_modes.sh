#!/bin/bash
OUT=/root/repo/gpurun_out/${1:-r04_bmmodes}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OSFM_BA_ONE_STREAM=1
for m in 0 1 2; do
  OSFM_BA_BM_MODE=$m timeout 200 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace$m -- python /root/repo/tools/prof_ba.py 5000 500000 10 3 > $OUT/traced$m.txt 2>&1
  python /root/repo/tools/rocpd_summary.py $(find $OUT/trace$m -name "*.db" | head -1) > $OUT/k$m.txt 2>&1
  rm -rf $OUT/trace$m
  echo "mode $m"; grep "band_mfma\|band_finish" $OUT/k$m.txt | cut -c1-150
done
