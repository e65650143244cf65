#!/bin/bash
OUT=/root/repo/gpurun_out/${1:-r04_ba_fork}
mkdir -p $OUT
cd /root/repo
for f in 0 1 2; do
  for rep in 1 2; do OSFM_BA_FORK=$f timeout 200 python tools/prof_ba.py 5000 500000 10 20 2>&1 | grep "^setup" | sed "s/^/fork $f: /"; done
done | tee $OUT/fork.txt
for rep in 1 2; do OSFM_BA_ONE_STREAM=1 timeout 200 python tools/prof_ba.py 5000 500000 10 20 2>&1 | grep "^setup" | sed "s/^/one stream: /"; done | tee -a $OUT/fork.txt
