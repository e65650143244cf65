#!/bin/bash
# round 4: the whole GPU suite on the final tree (4 workers, a test file per worker)
OUT=/root/repo/gpurun_out/r04_final4
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
timeout 640 python -m pytest tests -m gpu -q -n 4 --dist loadfile --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest.log
