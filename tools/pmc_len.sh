#!/bin/bash
# Usage (on the GPU box): BENCH_K=20 bash tools/pmc_len.sh <L> "<counters pass 1>" "<counters pass 2>" ...
# Times tools/bench_len.py and collects one --pmc pass per counter list for the classify kernel; prints per-read averages.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$1; shift
python3 $ROOT/tools/bench_len.py $L 2>/dev/null | tail -1
for ctr in "$@"; do
  rm -rf /tmp/pl_pmc
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pl_pmc -o pmc -- python3 $ROOT/tools/bench_len.py $L > /dev/null 2>&1)
  python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/pl_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "classify" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("  " + " ".join("%s=%.2f/read" % (k, sum(v) / len(v) / 1e6) for k, v in sorted(acc.items())))
PY
done
