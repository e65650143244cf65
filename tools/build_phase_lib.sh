#!/bin/bash
# Instrumented library for tools/match_phases.py and tools/hahog_phases.py: match.hip and hahog.hip compiled with -DOSFM_DBG_PHASES (100 MHz ticks of thread 0 of every
# workgroup, summed per phase), linked with the product's other objects.  Run opensfm_amd/csrc/build.sh first.
#   bash tools/build_phase_lib.sh && gpurun -- 'OSFM_MI355_LIB=/root/repo/tools/libosfm_dbg_phases.so python tools/match_phases.py'
set -e
cd "$(dirname "$0")/../opensfm_amd/csrc"
mkdir -p /tmp/osfm_dbg
for f in match hahog; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DOSFM_DBG_PHASES -c $f.hip -o /tmp/osfm_dbg/$f.o; done
OBJS=""
for s in api ransac ba tracks relpose calib guided words; do OBJS="$OBJS build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared /tmp/osfm_dbg/match.o /tmp/osfm_dbg/hahog.o $OBJS -o ../../tools/libosfm_dbg_phases.so
echo "built tools/libosfm_dbg_phases.so (git-ignored)"
