#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for B in 8 16 32; do
  export RKMH_PRE_BITS=$B
  OUT=$ROOT/gpurun_out/fetch_ab_$B
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o pmc -- python3 $ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $OUT.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/pmc_counter_collection.csv",recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_classify_tile" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("bits=$B FETCH_SIZE KB mean", sum(v)/len(v), "n", len(v))
PY
done
