cd $GRAFT_REPO_ROOT
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, ".")
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 4000000, read_len=150, threads=16)
synth.write_fastq("/tmp/w.fq", qb, qo, synth.read_names(0, 4000000))
open("/tmp/w.fq.gz", "wb").write(synth.bgzf_compress(open("/tmp/w.fq", "rb").read(), level=1, threads=16))
PY
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
for w in 0 1 0 1 0 1; do
  S=$(date +%s.%N); RKMH_WARM_UP=$w RKMH_TIMING=1 bin/rkmh stream $R -f /tmp/w.fq > /tmp/w.out 2> /tmp/w.err; E=$(date +%s.%N)
  python3 -c "print('plain 4 M reads, warm-up $w: wall %.3f s; %s' % ($E - $S, ' | '.join(l.split('timing] ')[1].strip() for l in open('/tmp/w.err') if 'timing' in l and 'front end:' not in l)))"
done
for w in 0 1 0 1; do
  S=$(date +%s.%N); RKMH_BGZF_DEVICE=1 RKMH_WARM_UP=$w RKMH_TIMING=1 bin/rkmh stream $R -f /tmp/w.fq.gz > /tmp/w.out 2> /tmp/w.err; E=$(date +%s.%N)
  python3 -c "print('BGZF device 4 M reads, warm-up $w: wall %.3f s; %s' % ($E - $S, ' | '.join(l.split('timing] ')[1].strip() for l in open('/tmp/w.err') if 'timing' in l and 'front end:' not in l)))"
done
