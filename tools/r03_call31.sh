#!/bin/bash
# final records of the round: the default bench line, then the whole GPU suite (4 workers, a file per worker)
OUT=/root/repo/gpurun_out/r03_c31
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
timeout 330 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 600 $OUT/bench.json
timeout 330 python -m pytest tests -m gpu -q -n 4 --dist loadfile --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/pytest.log
