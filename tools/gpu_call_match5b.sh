#!/bin/bash
for v in "" NORECHECK NOEPI v4; do
  echo "== ${v:-v5}"
  if [ -z "$v" ]; then timeout 300 python tools/prof_match.py 200 0 3 2>&1 | tail -1; timeout 300 python tools/prof_match.py 400 0 2 2>&1 | tail -1
  else OSFM_MI355_LIB=tools/libosfm_$v.so timeout 300 python tools/prof_match.py 200 0 3 2>&1 | tail -1; OSFM_MI355_LIB=tools/libosfm_$v.so timeout 300 python tools/prof_match.py 400 0 2 2>&1 | tail -1; fi
done
