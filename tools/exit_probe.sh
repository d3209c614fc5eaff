cd $GRAFT_REPO_ROOT
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, ".")
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 16000000, read_len=150, threads=16)
synth.write_fastq("/tmp/x.fq", qb, qo, synth.read_names(0, 16000000))
PY
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
probe() { # label, out, files...
  local label=$1 out=$2; shift 2
  [ "$out" != /dev/null ] && rm -f $out # (the shell would otherwise free the last run's gigabytes inside the timed region)
  S=$(date +%s.%N); RKMH_TIMING=1 "$@" > $out 2> /tmp/x.err; E=$(date +%s.%N)
  python3 -c "
import re
t=float(re.search(r'since the program was loaded\s+([0-9.]+)', open('/tmp/x.err').read()).group(1))
print('%-46s wall %.3f s, program %.3f s, outside the program %.3f s' % ('$label', $E-$S, t, $E-$S-t))"
}
probe "16 M reads -> file" /tmp/x.out bin/rkmh stream $R -f /tmp/x.fq
probe "16 M reads -> /dev/null" /dev/null bin/rkmh stream $R -f /tmp/x.fq
probe "64 M reads -> file" /tmp/x.out bin/rkmh stream $R -f /tmp/x.fq -f /tmp/x.fq -f /tmp/x.fq -f /tmp/x.fq
probe "64 M reads -> /dev/null" /dev/null bin/rkmh stream $R -f /tmp/x.fq -f /tmp/x.fq -f /tmp/x.fq -f /tmp/x.fq
probe "64 M reads -> file, 2 workers" /tmp/x.out env RKMH_RAW_WORKERS=2 bin/rkmh stream $R -f /tmp/x.fq -f /tmp/x.fq -f /tmp/x.fq -f /tmp/x.fq
probe "1000 reads (head) -> file" /tmp/x.out bash -c "head -4000 /tmp/x.fq > /tmp/x1.fq; exec bin/rkmh stream $R -f /tmp/x1.fq"
