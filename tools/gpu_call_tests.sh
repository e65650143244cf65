#!/bin/bash
# usage: gpu_call_tests.sh <log name> <pytest args...>
mkdir -p gpurun_out
LOG=gpurun_out/$1.log; shift
timeout 1500 python -m pytest "$@" -x -q -m gpu > $LOG 2>&1
echo "exit $?" >> $LOG
tail -30 $LOG
