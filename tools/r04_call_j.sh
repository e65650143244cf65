#!/bin/bash
# round 4: per-shot band assembly with accumulators over the shot's real partners -- wide-band tests, then ragged / grid timings
OUT=/root/repo/gpurun_out/r04_j
mkdir -p $OUT
cd /root/repo
export PROF_WARM=1
timeout 300 python -m pytest tests/test_gpu_ba.py -m gpu -q -x -n 4 -k "grid or ragged or dense or wide or two_free or half_width or long_tracks" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
timeout 120 python tools/prof_ba.py 5000 500000 10 10 ragged > $OUT/ragged.txt 2>&1; tail -2 $OUT/ragged.txt | head -1
OSFM_BA_BAND_FULL_ROWS=1 timeout 120 python tools/prof_ba.py 5000 500000 10 10 ragged > $OUT/ragged_full.txt 2>&1; tail -2 $OUT/ragged_full.txt | head -1
timeout 120 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid.txt 2>&1; tail -3 $OUT/grid.txt | head -1
OSFM_BA_BAND_FULL_ROWS=1 timeout 120 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_full.txt 2>&1; tail -3 $OUT/grid_full.txt | head -1
