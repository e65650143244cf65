#!/bin/bash
# Round-6 measurement: the block-survey BA with everything on ONE stream, to separate a kernel's own duration from what the side stream's
# neighbours add to it (border_point_kernel<3>: 1.24 ms on the grid against 0.19 ms on the sequence in the two-stream traces)
OUT=/root/repo/gpurun_out/r06_grid1s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in two one; do
  if [ $mode = one ]; then export OSFM_BA_ONE_STREAM=1; fi
  PROF_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr_$mode -- python /root/repo/tools/prof_ba_grid.py 50 100 500000 6 > $OUT/traced_$mode.txt 2>&1
  python /root/repo/tools/rocpd_summary.py $(find $OUT/tr_$mode -name "*.db" | head -1) > $OUT/grid_${mode}_stream_kernels_by_grid.txt 2>&1
  python /root/repo/tools/rocpd_summary.py $(find $OUT/tr_$mode -name "*.db" | head -1) --timeline border_point,border_shot,dgemm,dgj_pivot,band_assemble,eval_kernel,dbcr_build 400 > $OUT/grid_${mode}_timeline.txt 2>&1
  rm -rf $OUT/tr_$mode
  grep -E "^setup|^lin" $OUT/traced_$mode.txt
  grep -E "border_point|border_shot|band_assemble|eval_kernel" $OUT/grid_${mode}_stream_kernels_by_grid.txt | cut -c1-150
done
