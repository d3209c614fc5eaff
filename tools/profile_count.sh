#!/bin/bash
# Usage (on the GPU box): bash tools/profile_count.sh <tag>   -> gpurun_out/<tag>_M_*.{txt,csv}
# The -M count pass (1 M reads of 150 bp, k = 16) in both device forms at the reference's two table sizes (200 M slots: stream /
# classify, rkmh.cpp:739; 10 M: filter, :1187): timings, per-kernel rocprof stats, and HBM bytes per launch (FETCH_SIZE / WRITE_SIZE,
# each in its own --pmc pass).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 $ROOT/tools/bench_count_forms.py 16 > $OUT/${TAG}_M_times.txt 2>/dev/null
for SL in 200000000 10000000; do for F in 0 1; do
  rm -rf /tmp/pc; SLOTS=$SL FORMS=$F REPS=5 rocprofv3 --kernel-trace --stats -d /tmp/pc -o pc --output-format csv -- python3 $ROOT/tools/bench_count_forms.py 16 > /dev/null 2>&1
  cp /tmp/pc/*kernel_stats.csv $OUT/${TAG}_M_slots${SL}_form${F}_kernel_stats.csv 2>/dev/null || cp /tmp/pc/*/*kernel_stats.csv $OUT/${TAG}_M_slots${SL}_form${F}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc; SLOTS=$SL FORMS=$F REPS=2 timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pc -o pc --output-format csv -- python3 $ROOT/tools/bench_count_forms.py 16 > /dev/null 2>&1
    python3 - $SL $F $C >> $OUT/${TAG}_M_hbm.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/pc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if any(t in n for t in ("k_classify_tile", "k_slot", "k_bin", "k_count_bins", "fillBuffer")) and "k_enum" not in n:
            acc[n.split("(")[0][:60]].append(float(row["Counter_Value"]))
for n, v in sorted(acc.items()):
    print("slots=%s form=%s %-10s %-62s calls=%3d  mean per launch %.1f (counter units: see profiles/README.md)" % (sys.argv[1], sys.argv[2], sys.argv[3], n, len(v), sum(v) / len(v)))
PY
  done
done; done
