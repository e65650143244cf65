#!/bin/bash
# One GPU call for the round's records: PMC passes (r06_*_pmc.json, SQ counters), then the default bench line, the GPU suite, the headline
# under rocprofv3 --stats and the five BA kernel traces (tools/round_records.sh).  The PMC JSONs are copied into profiles/ FIRST so that
# the bench line of the same call quotes them.
cd /root/repo
bash tools/pmc_passes.sh r06 > gpurun_out/r06_pmc.log 2>&1
cp gpurun_out/r06_pmc/r06_*_pmc.json gpurun_out/r06_pmc/r06_match_sq_counters.txt profiles/ 2>/dev/null
bash tools/round_records.sh r06
