#!/bin/bash
# Usage (GPU box): [N=8000000] bash tools/packed_profile.sh [tag] -- `stream -F` of eight packed files under rocprofv3 (kernel + memory-copy trace): what a block's time is made of
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
N=${N:-8000000}; TAG=${1:-r06}
[ -f /tmp/pk.rkp ] || { python3 tools/make_fastq.py /tmp/pk.fq $N; bin/rkmh pack -f /tmp/pk.fq -o /tmp/pk.rkp --no-quals 2>/dev/null; }
R="-r $ROOT/tests/golden/data/all_pave_ref.fa.gz -k 16"
F=""; for i in 1 2 3 4 5 6 7 8; do F="$F -F /tmp/pk.rkp"; done
cd /tmp && export TMPDIR=/tmp
export RKMH_SLOW_EXIT=1 RKMH_TIMING=1
rm -rf /tmp/ppk
timeout -s KILL 240 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/ppk -o p --output-format csv -- $ROOT/bin/rkmh stream $R $F > /dev/null 2> /tmp/ppk.err
grep "packed reads" /tmp/ppk.err
for f in $(find /tmp/ppk -name "*stats.csv"); do echo "== $f"; cut -c1-160 $f | head -8; done
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/ppk/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        k = r.get("Direction", "?")
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        b = by.setdefault(k, [0, 0, 0]); b[0] += 1; b[1] += d
    for k, (n, ns, _) in by.items(): print("copies %-28s n=%d total %.1f ms avg %.3f ms" % (k, n, ns / 1e6, ns / 1e6 / n))
    big = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:5]
    for r in big: print("  longest:", r.get("Direction"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms", {k: v for k, v in r.items() if "ize" in k or "ytes" in k})
PY
