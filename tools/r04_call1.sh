#!/bin/bash
# round 4, first GPU call: default bench line (with exchange_emulation) + the tests touched so far
OUT=/root/repo/gpurun_out/r04_c1
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 300 $OUT/bench.json; echo; tail -3 $OUT/bench.err
timeout 400 python -m pytest tests/test_gpu_hahog.py tests/test_gpu_compat.py tests/test_gpu_dist.py -m gpu -q -x -s > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -25 $OUT/pytest.log
