#!/bin/bash
# One gpurun call: the default bench line, then the whole GPU suite (4 workers, a test file per worker).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_suite_and_bench.sh'
# Outputs under gpurun_out/suite/ (scratch): copy bench.json and pytest.log into profiles/ under the round's name.
OUT=${OUT:-/root/repo/gpurun_out/suite}
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1   # one process builds the checker, not four at once
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -c 400 $OUT/bench.json; echo
timeout 450 python -m pytest tests -m gpu -q -n 4 --dist loadfile --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -14 $OUT/pytest.log
