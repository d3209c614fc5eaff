#!/bin/bash
# Usage (GPU box): [W=3] [JOBKB=] bash tools/gz_timeline.sh [tag] -- kernel trace (begin / end of every dispatch) of bin/rkmh stream over four copies of
# /tmp/big.fq.gz (made by tools/gz_e2e.sh): which kernels of which jobs overlap.  Output: gpurun_out/<tag>_timeline.txt (one line per inflate / classify / index dispatch)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
TAG=${1:-r06}
[ -f /tmp/big.fq.gz ] || N=${N:-16000000} QUICK=1 bash tools/gz_e2e.sh tl_prep > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ptl
env RKMH_BGZF_DEVICE_WORKERS=${W:-3} ${JOBKB:+RKMH_BGZF_JOB_KB=$JOBKB} RKMH_SLOW_EXIT=1 rocprofv3 --kernel-trace -d /tmp/ptl -o p --output-format csv -- $ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -k 16 -f /tmp/big.fq.gz -f /tmp/big.fq.gz -f /tmp/big.fq.gz -f /tmp/big.fq.gz > /dev/null 2> /tmp/ptl.err
f=$(find /tmp/ptl -name "*kernel_trace.csv" | head -1)
python3 - $f > $ROOT/gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
keep = ("k_inflate_lanes", "k_inflate_place", "k_crc32", "k_classify_kmer", "k_fq_gather", "k_fq_records")
out = []
for r in rows:
    n = r["Kernel_Name"]
    tag = next((k for k in keep if k in n), None)
    if not tag: continue
    out.append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, tag, r.get("Queue_Id", "?")))
out.sort()
for s, e, tag, q in out:
    print("%9.3f ms .. %9.3f ms  (%7.3f)  %-18s queue %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, tag, q))
# busy time (union of intervals) against the wall of the trace
iv = sorted((s, e) for s, e, _, _ in out)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
print("# union of the listed kernels %.1f ms; first start .. last end %.1f ms; sum of durations %.1f ms" % (busy / 1e6, (iv[-1][1] - iv[0][0]) / 1e6 if iv else 0, sum(e - s for s, e in iv) / 1e6))
PY
tail -3 $ROOT/gpurun_out/${TAG}_timeline.txt
