#!/bin/bash
# Usage (GPU box): [N=2000000] [GZL=6] bash tools/gunzip_profile.sh [tag] -- an ORDINARY gzip file of N reads through bin/rkmh stream under rocprofv3 (kernel
# trace + stats), one file and three in one run.  Output: gpurun_out/<tag>_gunzip_kernel_stats_{1,3}.csv.  (RKMH_SLOW_EXIT=1: the program must not fork under the profiler.)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
N=${N:-2000000}; GZL=${GZL:-6}; TAG=${1:-r06}
[ -f /tmp/one.fq.gz ] || python3 - <<PY
import os, sys, zlib, numpy as np
sys.path.insert(0, "$ROOT")
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
rb, ro = refs["bases"], refs["offsets"]
n, L = $N, 150
co = zlib.compressobj($GZL, zlib.DEFLATED, 31)
with open("/tmp/one.fq.gz", "wb") as fz, open("/tmp/one.fq", "wb") as fp:
    for lo in range(0, n, 500000):
        m = min(500000, n - lo)
        qb, _ = synth.generate_reads_fast(rb, ro, lo, lo + m, read_len=L, threads=16)
        rec = np.empty((m, 11 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 11 + L] = 10; rec[:, 10] = 10
        idx = np.arange(lo, lo + m, dtype=np.int64)
        for d in range(9):
            rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
        rec[:, 11:11 + L] = qb[: m * L].reshape(m, L)
        rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
        rec[:, 14 + L:14 + 2 * L] = np.random.default_rng(lo).integers(35, 75, size=(m, L), dtype=np.uint8); rec[:, 14 + 2 * L] = 10
        raw = rec.tobytes()
        fp.write(raw); fz.write(co.compress(raw))
    fz.write(co.flush())
PY
R="-r $ROOT/tests/golden/data/all_pave_ref.fa.gz -k 16"
cd /tmp && export TMPDIR=/tmp
export RKMH_SLOW_EXIT=1 RKMH_BGZF_TIMING=1
for nf in 1 3; do
  files=""; for i in $(seq $nf); do files="$files -f /tmp/one.fq.gz"; done
  rm -rf /tmp/pgu$nf
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d /tmp/pgu$nf -o p --output-format csv -- $ROOT/bin/rkmh stream $R $files > /dev/null 2> /tmp/pgu$nf.err
  grep "gzip device" /tmp/pgu$nf.err | head -4
  f=$(find /tmp/pgu$nf -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $ROOT/gpurun_out/${TAG}_gunzip_kernel_stats_$nf.csv && cut -c1-170 $f | head -14
done
