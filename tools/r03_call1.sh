#!/bin/bash
# round 3, GPU call 1: the device-resident exchange step on RCCL (1 rank; 2 ranks where the box has them) and configs[3] at N = 1
OUT=/root/repo/gpurun_out/r03_c1
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_matching.py -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 1500 python bench.py --gpus 1 --images 10000 --strong --steps 2 --warmup 0 --headline-only > $OUT/configs3_n1.json 2> $OUT/configs3_n1.err
tail -c 1500 $OUT/configs3_n1.json; tail -5 $OUT/configs3_n1.err
free -g | head -2 > $OUT/host.txt; nproc >> $OUT/host.txt
