#!/bin/bash
# Usage (GPU box): [RKMH_RAW_BLOCK_KB=..] [RKMH_RAW_WORKERS=..] bash tools/profile_inflate2.sh <file.fq.gz>  -- kernel trace of bin/rkmh stream with device inflate:
# per-kernel durations and how many k_inflate_lanes launches overlap in time
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-/tmp/sw.fq.gz}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pinf2
RKMH_BGZF_DEVICE=1 RKMH_SLOW_EXIT=1 RKMH_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/pinf2 -o p --output-format csv -- $ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -f $F -k 16 > /dev/null 2> /tmp/pinf2.err
grep "rkmh timing" /tmp/pinf2.err | tail -4
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/pinf2/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = []
by = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-30:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    by.setdefault(n, []).append((s, e))
t0 = min(s for v in by.values() for s, e in v); t1 = max(e for v in by.values() for s, e in v)
print("kernel activity spans %.1f ms" % ((t1 - t0) / 1e6))
for n, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:8]:
    tot = sum(e - s for s, e in v)
    # union length
    u, cur_s, cur_e = 0, None, None
    for s, e in sorted(v):
        if cur_e is None or s > cur_e:
            if cur_e is not None: u += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: u += cur_e - cur_s
    print("%-32s launches %4d  mean %.3f ms  sum %.1f ms  union %.1f ms  (mean overlap %.2f)  first start %.1f ms" % (n, len(v), tot / len(v) / 1e6, tot / 1e6, u / 1e6, tot / max(u, 1), (min(s for s, e in v) - t0) / 1e6))
PY
