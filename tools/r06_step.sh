#!/bin/bash
# Round-6 working call: BA-related GPU tests, the BA lines without CPU legs, optionally the PCG-tolerance experiment, one kernel trace.
#   gpurun --timeout 1500 -- 'bash tools/r06_step.sh <tag> [tests] [lines] [tol] [trace]'
TAG=${1:-s1}; shift
OUT=/root/repo/gpurun_out/r06_$TAG
mkdir -p $OUT
cd /root/repo
python -c "import oracle; oracle.build()" > $OUT/oracle_build.log 2>&1
for what in "$@"; do
  case $what in
    tests) timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_bundle_general.py tests/test_gpu_bundle_facade.py tests/test_gpu_golden_fisheye624.py tests/test_gpu_berlin.py tests/test_gpu_compat.py tests/test_gpu_flow.py -m gpu -q -x --durations=5 > $OUT/pytest_ba.txt 2>&1; echo "pytest rc $?"; tail -12 $OUT/pytest_ba.txt ;;
    alltests) timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -14 $OUT/pytest_gpu.txt ;;
    lines) timeout 900 python tools/r06_ba_quick.py lines > $OUT/ba_lines.json 2> $OUT/ba_lines.err; echo "lines rc $?"; cat $OUT/ba_lines.json | python -c "
import json,sys
d=json.load(sys.stdin)['lines']
for k,v in d.items(): print(k, v.get('value'), v.get('lm_iteration',{}).get('ms') if isinstance(v.get('lm_iteration'),dict) else '', v.get('pcg_iterations'), v.get('ms_per_solve',''), v.get('error',''))
"; tail -3 $OUT/ba_lines.err ;;
    tol) timeout 900 python tools/r06_ba_quick.py tol > $OUT/ba_tol.json 2> $OUT/ba_tol.err; echo "tol rc $?"; cat $OUT/ba_tol.json; tail -3 $OUT/ba_tol.err ;;
    trace*)
      cd /tmp && export TMPDIR=/tmp
      tr() { local name=$1; shift
        PROF_WARM=1 timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d $OUT/tr_$name -- "$@" > $OUT/traced_$name.txt 2>&1
        python /root/repo/tools/rocpd_summary.py $(find $OUT/tr_$name -name "*.db" | head -1) > $OUT/${name}_kernels_by_grid.txt 2>&1
        rm -rf $OUT/tr_$name; }
      tr ba python /root/repo/tools/prof_ba.py 5000 500000 10 10
      if [ "$what" = "traceall" ]; then
        tr ba_general python /root/repo/tools/prof_ba.py 5000 500000 10 10 general
        tr ba_grid python /root/repo/tools/prof_ba_grid.py 50 100 500000 6
        tr ba_ragged python /root/repo/tools/prof_ba.py 5000 500000 10 10 ragged
        tr local_ba python /root/repo/tools/prof_local_ba.py
      fi
      head -40 $OUT/ba_kernels_by_grid.txt | cut -c1-150
      cd /root/repo ;;
  esac
done
ls $OUT
