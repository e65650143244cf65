#!/bin/bash
# round 4, final records of a tree: the default bench line, then the whole GPU suite (4 workers, a test file per worker).
OUT=${OUT:-/root/repo/gpurun_out/r04_final}
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 200 $OUT/bench.json; echo
timeout 800 python -m pytest tests -m gpu -q -n 4 --dist loadfile --durations=10 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -16 $OUT/pytest.log
