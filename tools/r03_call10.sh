#!/bin/bash
# BA after merging host round trips (status / border inverse on the device, candidate + its cost, cost + gradient norm) and the low-priority side stream
OUT=/root/repo/gpurun_out/r03_c10
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ba.py -q -x -k "not constant_cameras" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_plain.txt 2>&1; tail -3 $OUT/prof_ba_plain.txt
OSFM_BA_SIDE_PRIO_DEFAULT=1 timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_prio_default.txt 2>&1; tail -3 $OUT/prof_ba_prio_default.txt
OSFM_BA_ONE_STREAM=1 timeout 300 python tools/prof_ba.py 5000 500000 10 20 > $OUT/prof_ba_one_stream.txt 2>&1; tail -3 $OUT/prof_ba_one_stream.txt
