#!/bin/bash
# Round-6 records of the HAHOG row: single-image kernel table and timeline (tools/r06_hahog_single.sh), the phase ticks of the per-feature
# workgroups on the instrumented build (tools/build_phase_lib.sh first), the batch path by image type / flags / concurrency.
cd /root/repo
bash tools/r06_hahog_single.sh r06_hahog_final > gpurun_out/r06_hahog_final.log 2>&1
OSFM_MI355_LIB=/root/repo/tools/libosfm_dbg_phases.so python tools/hahog_phases.py > gpurun_out/r06_hahog_final/r06_hahog_phases.txt 2>&1
python tools/r06_hahog_batch_matrix.py > gpurun_out/r06_hahog_final/r06_hahog_batch_matrix.txt 2>&1
python - > gpurun_out/r06_hahog_final/r06_hahog_bench.json <<'P'
import json, sys
sys.path.insert(0, '/root/repo')
import bench
from opensfm_amd._lib import default_context
print(json.dumps(bench.hahog_bench(default_context(0), True)))
P
tail -4 gpurun_out/r06_hahog_final.log | cut -c1-160; tail -16 gpurun_out/r06_hahog_final/r06_hahog_phases.txt; tail -12 gpurun_out/r06_hahog_final/r06_hahog_batch_matrix.txt; cut -c1-1200 gpurun_out/r06_hahog_final/r06_hahog_bench.json
