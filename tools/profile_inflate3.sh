#!/bin/bash
# Usage (GPU box): bash tools/profile_inflate3.sh [file.fq.gz] -- kernel + memory-copy trace of bin/rkmh stream with RKMH_BGZF_DEVICE=1: how long the copies take, how much
# of the time the device computes or copies at all
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-/tmp/sw.fq.gz}
[ -f $F ] || (cd $ROOT; CFGS="65536 8" timeout 300 bash tools/gz_sweep.sh > /dev/null 2>&1)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pinf3
RKMH_BGZF_DEVICE=1 RKMH_SLOW_EXIT=1 RKMH_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/pinf3 -o p --output-format csv -- $ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -f $F -f $F -k 16 > /dev/null 2> /tmp/pinf3.err
grep "rkmh timing" /tmp/pinf3.err | tail -3
python3 - <<'PY'
import csv, glob
def union(v):
    u, cs, ce = 0, None, None
    for s, e in sorted(v):
        if ce is None or s > ce:
            if ce is not None: u += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    return u + (ce - cs if ce is not None else 0)
k = []
for f in glob.glob("/tmp/pinf3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): k.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:]))
cp = []
for f in glob.glob("/tmp/pinf3/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "?")), int(r.get("Bytes", r.get("Size", 0)) or 0)))
t0 = min([s for s, e, n in k] + [s for s, e, d, b in cp]); t1 = max([e for s, e, n in k] + [e for s, e, d, b in cp])
print("span %.1f ms; kernels busy (union) %.1f ms; copies busy (union) %.1f ms; either %.1f ms" % ((t1 - t0) / 1e6, union([(s, e) for s, e, n in k]) / 1e6,
      union([(s, e) for s, e, d, b in cp]) / 1e6, union([(s, e) for s, e, n in k] + [(s, e) for s, e, d, b in cp]) / 1e6))
by = {}
for s, e, d, b in cp:
    if b >= (1 << 20): by.setdefault(d, []).append((s, e, b))
for d, v in by.items():
    tot = sum(e - s for s, e, b in v); bytes_ = sum(b for s, e, b in v)
    print("copies >= 1 MB %-28s n=%4d  %.2f GB  mean %.2f ms  rate while copying %.1f GB/s  union %.1f ms" % (d, len(v), bytes_ / 1e9, tot / len(v) / 1e6, bytes_ / max(tot, 1), union([(s, e) for s, e, b in v]) / 1e6))
# idle gaps of the whole device (neither kernel nor copy) longer than 2 ms
ev = sorted([(s, e) for s, e, n in k] + [(s, e) for s, e, d, b in cp])
gaps, ce = [], None
for s, e in ev:
    if ce is not None and s - ce > 2e6: gaps.append(((ce - t0) / 1e6, (s - ce) / 1e6))
    ce = e if ce is None else max(ce, e)
print("idle gaps > 2 ms:", ["at %.0f: %.1f ms" % g for g in gaps][:30])
PY
