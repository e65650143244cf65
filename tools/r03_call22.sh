#!/bin/bash
# wide band: push-form walks (a launch per block column) against the single-workgroup walks
OUT=/root/repo/gpurun_out/r03_c22
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ba.py -q -x -k "wide_band or grid or two_free or long_tracks or unordered" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 300 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_push.txt 2>&1; tail -3 $OUT/grid_push.txt
OSFM_BA_WIDE_WALK_1WG=1 timeout 300 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_1wg.txt 2>&1; tail -3 $OUT/grid_1wg.txt
