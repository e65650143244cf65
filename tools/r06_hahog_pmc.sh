#!/bin/bash
# Round-6 measurement: SQ counters of the HAHOG kernels (single-image calls): what the fused smoothing and the per-feature kernels issue
OUT=/root/repo/gpurun_out/r06_hahog_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
bash /root/repo/tools/r06_hahog_single.sh r06_hahog_pmc_tmp > /dev/null 2>&1   # leaves /tmp/hs.py
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o pmc -- python /tmp/hs.py 3 > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'P' >> $OUT/r06_hahog_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60] + " grid " + r["Grid_Size"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
    print(k, {n: round(v) for n, v in d.items()})
P
  rm -rf $OUT/$tag
done
cat $OUT/r06_hahog_pmc.txt | cut -c1-260
