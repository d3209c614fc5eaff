#!/bin/bash
# Usage (GPU box): bash tools/fork_probe.sh -- does bin/rkmh release its parent when the output is complete?  (stream of 16 M reads, one and eight -r files)
cd ${GRAFT_REPO_ROOT:-.}
D=tests/golden/data
python3 tools/make_fastq.py /tmp/fp.fq 16000000 > /dev/null
REFS=""; for f in all_pave_ref zika.refs dengue new_refs hpv_16 zika yellow_fever hpv_16_allFasta; do gunzip -c $D/$f.fa.gz > /tmp/fp_$f.fa; REFS="$REFS -r /tmp/fp_$f.fa"; done
for refs in "-r $D/all_pave_ref.fa.gz" "$REFS"; do
  rm -f /tmp/fp.out
  t0=$EPOCHREALTIME
  RKMH_TIMING=1 bin/rkmh stream $refs -f /tmp/fp.fq -k 16 -s 1000 > /tmp/fp.out 2> /tmp/fp.err
  t1=$EPOCHREALTIME
  echo "refs: $(echo $refs | wc -w) words; wall $(python3 -c "print('%.3f' % ($t1 - $t0))") s; t1 = $t1"
  tail -4 /tmp/fp.err
done
env | grep -i "rocp\|hsa_\|preload" || echo "(no profiler-like variables in the environment)"
