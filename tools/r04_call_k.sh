#!/bin/bash
# round 4: look-ahead pivot chain in the wide band's blocked inversion -- wide-band tests, then ragged / grid timings with and without it
OUT=/root/repo/gpurun_out/r04_k
mkdir -p $OUT
cd /root/repo
export PROF_WARM=1
timeout 300 python -m pytest tests/test_gpu_ba.py -m gpu -q -x -n 4 -k "grid or ragged or dense or wide or two_free or half_width or long_tracks" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
timeout 120 python tools/prof_ba.py 5000 500000 10 10 ragged > $OUT/ragged.txt 2>&1; echo "ragged ahead: $(tail -2 $OUT/ragged.txt | head -1)"
OSFM_BA_NO_LOOKAHEAD=1 timeout 120 python tools/prof_ba.py 5000 500000 10 10 ragged > $OUT/ragged_no.txt 2>&1; echo "ragged plain: $(tail -2 $OUT/ragged_no.txt | head -1)"
timeout 120 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid.txt 2>&1; echo "grid ahead: $(tail -3 $OUT/grid.txt | head -1)"
OSFM_BA_NO_LOOKAHEAD=1 timeout 120 python tools/prof_ba_grid.py 50 100 500000 10 > $OUT/grid_no.txt 2>&1; echo "grid plain: $(tail -3 $OUT/grid_no.txt | head -1)"
