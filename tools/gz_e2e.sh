#!/bin/bash
# Usage (GPU box): [N=8000000] [N1=2000000] [GZL=6] [QUICK=1|SWEEP=1|DETAIL=1] bash tools/gz_e2e.sh [tag] -- ONE BGZF file of N reads (and the same text plain) through bin/rkmh stream: one file
# and four, plain text / device inflate (two root-table forms) / host inflate: wall, marginal reads/s of the three extra files, the
# [bgzf device] job lines, then a rocprofv3 kernel trace of the four-file device run.  Output: gpurun_out/<tag>_gz.txt, <tag>_gz_kernel_stats.csv
cd ${GRAFT_REPO_ROOT:-.}
N=${N:-8000000}
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_gz.txt
mkdir -p gpurun_out
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
rb, ro = refs["bases"], refs["offsets"]
n, L = $N, 150
with open("/tmp/big.fq.gz", "wb") as fb, open("/tmp/big.fq", "wb") as fp:
    for lo in range(0, n, 1000000):
        m = min(1000000, n - lo)
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
        fp.write(raw)
        img = synth.bgzf_compress(raw, level=1, threads=16)
        fb.write(img[:-28] if lo + m < n else img)
PY
ls -la /tmp/big.fq.gz /tmp/big.fq > $OUT
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
want=$(bin/rkmh stream $R -f /tmp/big.fq 2>/dev/null | sha256sum | cut -c1-16)
run() { # label file env...
  local label=$1 f=$2; shift 2
  local t1 t4
  for rep in 1 2; do
    rm -f /tmp/big.out; sleep 1; S=$(date +%s.%N); env "$@" RKMH_TIMING=1 timeout -s ABRT 300 bin/rkmh stream $R -f $f > /tmp/big.out 2>/tmp/big.err1 || { echo "$label failed" >> $OUT; tail -3 /tmp/big.err1 >> $OUT; return; }; E=$(date +%s.%N)
    t1=$(python3 -c "print($E - $S)")
  done
  got=$(sha256sum /tmp/big.out | cut -c1-16)
  for rep in 1 2; do
    rm -f /tmp/big.out; sleep 1; S=$(date +%s.%N); env "$@" RKMH_TIMING=1 RKMH_BGZF_TIMING=1 timeout -s ABRT 300 bin/rkmh stream $R -f $f -f $f -f $f -f $f > /tmp/big.out 2>/tmp/big.err4 || { echo "$label x4 failed" >> $OUT; tail -3 /tmp/big.err4 >> $OUT; return; }; E=$(date +%s.%N)
    t4=$(python3 -c "print($E - $S)")
  done
  python3 -c "print('%-44s 1 file %.3f s, 4 files %.3f s: marginal %.1f M reads/s; output %s' % ('$label', $t1, $t4, 3 * $N / ($t4 - $t1) / 1e6, 'identical to plain' if '$got' == '$want' else 'DIFFERS'))" >> $OUT
  echo "      one file:   $(grep -E "context|references|device front end  |main loop|since the program" /tmp/big.err1 | sed 's/\[rkmh timing\] //' | tr -s " " | tr "\n" ";")" >> $OUT
  echo "      four files: $(grep -E "context|references|device front end  |main loop|since the program" /tmp/big.err4 | sed 's/\[rkmh timing\] //' | tr -s " " | tr "\n" ";")" >> $OUT
  grep -E "device front end:" /tmp/big.err4 | tr -s " " | sed 's/^/      /' >> $OUT
  grep "bgzf device\|gzip device" /tmp/big.err4 | head -6 | sed 's/^/      /' >> $OUT
  [ -n "$DETAIL" ] && { echo "      --- the one-file run, every timing line:" >> $OUT; env "$@" RKMH_TIMING=1 RKMH_BGZF_TIMING=1 RKMH_INDEX_TIMING=1 bin/rkmh stream $R -f $f 2>&1 >/dev/null | grep -v "^\[rkmh index\]" | sed 's/^/        /' >> $OUT; }
}
run "plain text" /tmp/big.fq X=1
run "BGZF, device inflate (default: 3 workers)" /tmp/big.fq.gz X=1
if [ -n "$SWEEP" ]; then
  for w in 2 4 5 6; do run "BGZF, device inflate, $w workers" /tmp/big.fq.gz RKMH_BGZF_DEVICE_WORKERS=$w; done
  cat $OUT; exit 0
fi
# the first N1 reads as an ORDINARY gzip file (one deflate stream, zlib level GZL): inflated on the device (rk_gunzip.hip) / by zlib on the host
N1=${N1:-2000000}; [ $N1 -gt $N ] && N1=$N
GZL=${GZL:-6}
python3 - <<PY
import zlib
L = 11 + 150 + 3 + 150 + 1
with open("/tmp/big.fq", "rb") as f, open("/tmp/one.fq", "wb") as fo, open("/tmp/one.fq.gz", "wb") as fz:
    co = zlib.compressobj($GZL, zlib.DEFLATED, 31)
    left = $N1
    while left:
        m = min(left, 200000)
        raw = f.read(m * L)
        fo.write(raw); fz.write(co.compress(raw)); left -= m
    fz.write(co.flush())
PY
ls -la /tmp/one.fq /tmp/one.fq.gz >> $OUT
want=$(bin/rkmh stream $R -f /tmp/one.fq 2>/dev/null | sha256sum | cut -c1-16)
NALL=$N; N=$N1
run "plain text, the first $N1 reads" /tmp/one.fq X=1
run "ordinary gzip (level $GZL), device inflate" /tmp/one.fq.gz X=1
run "ordinary gzip (level $GZL), zlib on the host" /tmp/one.fq.gz RKMH_GZIP_DEVICE=0
N=$NALL
[ -n "$QUICK" ] && { cat $OUT; exit 0; }
run "BGZF, device inflate, 2 workers" /tmp/big.fq.gz RKMH_BGZF_DEVICE_WORKERS=2
run "BGZF, device inflate, 1 worker" /tmp/big.fq.gz RKMH_BGZF_DEVICE_WORKERS=1
run "BGZF, device inflate, 512 MB jobs" /tmp/big.fq.gz RKMH_BGZF_JOB_KB=524288
run "BGZF, host inflate" /tmp/big.fq.gz RKMH_BGZF_DEVICE=0
cat $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pgz && RKMH_SLOW_EXIT=1 RKMH_BGZF_DEVICE_WORKERS=1 rocprofv3 --kernel-trace --stats -d /tmp/pgz -o p --output-format csv -- $GRAFT_REPO_ROOT/bin/rkmh stream -r $GRAFT_REPO_ROOT/tests/golden/data/all_pave_ref.fa.gz -k 16 -f /tmp/big.fq.gz -f /tmp/big.fq.gz -f /tmp/big.fq.gz -f /tmp/big.fq.gz > /dev/null 2> /tmp/pgz.err
f=$(find /tmp/pgz -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_gz_kernel_stats.csv && cut -c1-150 $f | head -12
