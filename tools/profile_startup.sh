#!/bin/bash
# Usage (GPU box): bash tools/profile_startup.sh -- when each kind of kernel first runs in a plain-FASTQ `bin/rkmh stream` (4 M reads): what the start of a run waits for
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, "$ROOT")
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 4000000, read_len=150, threads=16)
synth.write_fastq("/tmp/st.fq", qb, qo, synth.read_names(0, 4000000))
PY
gunzip -c tests/golden/data/all_pave_ref.fa.gz > /tmp/refs.fa
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pst
RKMH_SLOW_EXIT=1 RKMH_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/pst -o p --output-format csv -- $ROOT/bin/rkmh stream -r /tmp/refs.fa -f /tmp/st.fq -k 16 > /dev/null 2> /tmp/pst.err
grep "rkmh timing" /tmp/pst.err
python3 - <<'PY'
import csv, glob
k = []
for f in glob.glob("/tmp/pst/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): k.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
cp = []
for f in glob.glob("/tmp/pst/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0 = min([s for s, e, n in k] + [s for s, e in cp])
first = {}
for s, e, n in sorted(k):
    if n not in first: first[n] = (s, e)
for n, (s, e) in sorted(first.items(), key=lambda kv: kv[1][0]):
    print("%8.1f ms  first %-46s (%.2f ms)" % ((s - t0) / 1e6, n, (e - s) / 1e6))
print("last kernel ends at %.1f ms; first copy at 0" % ((max(e for s, e, n in k) - t0) / 1e6))
PY
