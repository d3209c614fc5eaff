#!/bin/bash
# BASELINE config 3 as a whole command on ONE GPU: bin/rkmh stream -k 16 -s 1000 of N reads (default 100 M = 31.5 GB of FASTQ in /tmp) drawn
# from every bundled reference (266 sequences) against those references.  Prints wall times (cold page cache after writing = warm) and
# checks the line count.  Usage: bash tools/c3_e2e.sh [reads]
cd "$(dirname "$0")/.."
ROOT=$PWD
N=${1:-100000000}
D=$ROOT/tests/golden/data
REFS=""
for f in all_pave_ref zika.refs dengue new_refs hpv_16 zika yellow_fever hpv_16_allFasta; do gunzip -c $D/$f.fa.gz > /tmp/c3_$f.fa; REFS="$REFS -r /tmp/c3_$f.fa"; done
C3_PANEL=1 python3 tools/make_fastq.py /tmp/c3_reads.fq $N
ls -la /tmp/c3_reads.fq | awk '{print "fastq bytes", $5}'
for rep in 1 2 3; do
  sleep 2 # (the run before is still being taken apart by the kernel -- behind the command's return: a fresh context beside that is slower)
  rm -f /tmp/c3_out.tsv; : > /tmp/c3_out.tsv
  t0=$EPOCHREALTIME # (a shell variable: `date` would have to be forked, beside a process that is being taken apart)
  RKMH_TIMING=1 bin/rkmh stream $REFS -f /tmp/c3_reads.fq -k 16 -s 1000 > /tmp/c3_out.tsv 2> /tmp/c3_err.txt; rc=$?
  t1=$EPOCHREALTIME
  python3 -c "print('run $rep: rc=$rc wall %.2f s = %.1f M reads/s end to end' % ($t1 - $t0, $N / ($t1 - $t0) / 1e6))"
  python3 -c "
import re
m = re.search(r'loaded at epoch ([0-9.]+), leaving at epoch ([0-9.]+)', open('/tmp/c3_err.txt').read())
r = re.search(r'parent released at epoch ([0-9.]+)', open('/tmp/c3_err.txt').read())
if m: print('       of the wall clock: %.3f s before the program was loaded (fork + exec + the dynamic loader), %.3f s after its last line%s' % (float(m.group(1)) - $t0, $t1 - float(m.group(2)),
            ' (%.3f s until the parent was released, %.3f s for it to leave)' % (float(r.group(1)) - float(m.group(2)), $t1 - float(r.group(1))) if r else ''))"
  grep "rkmh timing" /tmp/c3_err.txt | head -8
done
echo "lines: $(wc -l < /tmp/c3_out.tsv)"
rm -f /tmp/c3_reads.fq /tmp/c3_out.tsv
