#!/bin/bash
# Builds libosfm_mi355.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS="api.hip match.hip ransac.hip ba.hip ba_general.hip tracks.hip relpose.hip calib.hip guided.hip words.hip"
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
  -Wall -Wno-unused-function $EXTRA_HIPCC_FLAGS $SRCS -o libosfm_mi355.so -L/opt/rocm/lib -lrocsolver -lrocblas -Wl,-rpath,/opt/rocm/lib
echo "built $(pwd)/libosfm_mi355.so"
