#!/bin/bash
# Usage (GPU box): [N=4000000] [SINK=/dev/null] bash tools/packed_e2e.sh [tag] -- N reads as FASTQ text and packed (rkmh pack --no-quals): stream -f / -F, one file and eight, with stage timings
cd ${GRAFT_REPO_ROOT:-.}
N=${N:-4000000}; TAG=${1:-r06}; OUT=gpurun_out/${TAG}_packed.txt; mkdir -p gpurun_out; : > $OUT
python3 tools/make_fastq.py /tmp/pk.fq $N
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
bin/rkmh pack -f /tmp/pk.fq -o /tmp/pk.rkp --no-quals 2>> $OUT
want=$(bin/rkmh stream $R -f /tmp/pk.fq 2>/dev/null | sha256sum | cut -c1-16)
SINK=${SINK:-/tmp/pk.out} # SINK=/dev/null: the lines are formatted and dropped (what the input side can do); a file: what a user gets
echo "sink: $SINK" >> $OUT
t() { rm -f /tmp/pk.out; local S=$(date +%s.%N); "$@" > $SINK 2> /tmp/pk.err; local E=$(date +%s.%N); python3 -c "print('%.3f' % ($E - $S))"; }
for env in ${ENVS:-"X=1" "RKMH_PACKED_WORKERS=1" "RKMH_PACKED_WORKERS=2" "RKMH_OUT_WRITERS=1" "RKMH_OUT_WRITERS=8"}; do
  for rep in 1 2; do a=$(t env $env RKMH_TIMING=1 bin/rkmh stream $R -F /tmp/pk.rkp); done
  got=$(bin/rkmh stream $R -F /tmp/pk.rkp 2>/dev/null | sha256sum | cut -c1-16)
  for rep in 1 2; do b=$(t env $env RKMH_TIMING=1 bin/rkmh stream $R -F /tmp/pk.rkp -F /tmp/pk.rkp -F /tmp/pk.rkp -F /tmp/pk.rkp -F /tmp/pk.rkp -F /tmp/pk.rkp -F /tmp/pk.rkp -F /tmp/pk.rkp); done
  python3 -c "print('packed [%s]: 1 file %s s, 8 files %s s: marginal %.1f M reads/s; %s' % ('$env', '$a', '$b', 7 * $N / ($b - $a) / 1e6, 'identical to the text run' if '$got' == '$want' else 'DIFFERS'))" >> $OUT
  grep -E "page-locked|packed reads:|main loop|since the program" /tmp/pk.err | tr -s " " | sed 's/^/      /' | head -14 >> $OUT
done
for rep in 1 2; do a=$(t bin/rkmh stream $R -f /tmp/pk.fq); b=$(t bin/rkmh stream $R -f /tmp/pk.fq -f /tmp/pk.fq -f /tmp/pk.fq -f /tmp/pk.fq -f /tmp/pk.fq -f /tmp/pk.fq -f /tmp/pk.fq -f /tmp/pk.fq); done
python3 -c "print('text: 1 file $a s, 8 files $b s: marginal %.1f M reads/s' % (7 * $N / ($b - $a) / 1e6))" >> $OUT
cat $OUT
