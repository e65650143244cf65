#!/bin/bash
# Builds libosfm_mi355.so for gfx950 (cross-compiles without a GPU).  One object per source, compiled in parallel and only when the
# source or a header is newer than the object (objects under build/, git-ignored and not shipped).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS="api.hip match.hip ransac.hip ba.hip tracks.hip relpose.hip calib.hip guided.hip words.hip hahog.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $EXTRA_HIPCC_FLAGS"
mkdir -p build
echo "$FLAGS" > build/.flags.new
if ! cmp -s build/.flags.new build/.flags; then rm -f build/*.o; mv build/.flags.new build/.flags; fi
newest_header=$(ls -t *.h *.inc ../../include/*.h | head -1)
pids=()
for s in $SRCS; do
  o=build/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$newest_header" -nt "$o" ]; then
    ( $HIPCC $FLAGS -c "$s" -o "$o.tmp" && mv "$o.tmp" "$o" ) &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
[ $rc -eq 0 ] || { echo "build failed"; exit 1; }
OBJS=""
for s in $SRCS; do OBJS="$OBJS build/${s%.hip}.o"; done
$HIPCC --offload-arch=gfx950 -fPIC -shared $OBJS -o libosfm_mi355.so
echo "built $(pwd)/libosfm_mi355.so"
