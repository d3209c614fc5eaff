#!/bin/bash
# Usage (on the GPU box): [BOUNDS=0] bash tools/profile_masked.sh <tag>   -> gpurun_out/<tag>_M2_*.{txt,csv}
# BOUNDS: the rk_set_min_num_bound values the masked pass is run under (default -1,0,1,6; one value for a PMC profile of that form)
# The masked classification of -M (pass 2: every window's keep bit from the bitmap of the depth table, rkmh.cpp:911-948) on 1 M reads of
# 150 bp, k = 16: time at three table sizes (the bitmap goes from 25 MB to 125 KB: what the random bit lookups cost), rocprof
# kernel stats, and the counters that say where the requests go -- each group in its own --pmc pass (kernel-trace only beside it).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 $ROOT/tools/bench_masked.py > $OUT/${TAG}_M2_times.txt 2>/dev/null
rm -rf /tmp/pm; SLOTS=200000000 REPS=10 rocprofv3 --kernel-trace --stats -d /tmp/pm -o pm --output-format csv -- python3 $ROOT/tools/bench_masked.py > /dev/null 2>&1
cp /tmp/pm/*kernel_stats.csv $OUT/${TAG}_M2_kernel_stats.csv 2>/dev/null || cp /tmp/pm/*/*kernel_stats.csv $OUT/${TAG}_M2_kernel_stats.csv
: > $OUT/${TAG}_M2_pmc.txt
for SL in 200000000 1000000; do
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pm; SLOTS=$SL REPS=3 timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm -o pm --output-format csv -- python3 $ROOT/tools/bench_masked.py > /dev/null 2>&1
  python3 - $SL >> $OUT/${TAG}_M2_pmc.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_classify_tile" in n or "k_classify_kmer" in n or "k_min_num_probe" in n:
            acc[(n.split("(")[0][-70:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (n, c), v in sorted(acc.items()):
    # the masked launches are the last ones of the run (the count pass and warm-ups come first); all launches of this kernel name are pass 2
    print("slots=%-10s %-72s %-22s calls=%3d mean per launch %.4g" % (sys.argv[1], n, c, len(v), sum(v) / len(v)))
PY
done; done
cat $OUT/${TAG}_M2_times.txt $OUT/${TAG}_M2_pmc.txt
