"""Summarises rocprofv3 output dirs produced by tools/profile.sh: per-kernel stats + per-dispatch PMC averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-70s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row.get("Name", "")[:70], row.get("Calls"), row.get("AverageNs"),
                                                                 row.get("TotalDurationNs"), row.get("Percentage")))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== pmc:", os.path.relpath(f, out))
        for k, cs in acc.items():
            if "classify" not in k and "hash" not in k and "sort" not in k:
                continue
            for c, v in cs.items():
                print("  %-60s %-28s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
