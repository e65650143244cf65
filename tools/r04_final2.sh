#!/bin/bash
# round 4, last records: the BA tests of the final tree that the previous run stopped before, then the default bench line
OUT=/root/repo/gpurun_out/r04_final2
mkdir -p $OUT
cd /root/repo
timeout 400 python -m pytest tests/test_gpu_ba.py -m gpu -q -n 4 -k "dense or ragged or half_width or band_by_windows or constant_cameras or other_models or unordered or fisheye or local_and_shot" > $OUT/pytest_ba_rest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_ba_rest.log
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 150 $OUT/bench.json; echo
