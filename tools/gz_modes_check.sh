#!/bin/bash
# Usage (GPU box): bash tools/gz_modes_check.sh -- 4 M reads as plain FASTQ and as BGZF: stream, stream -M 2, filter, filter -M 2 -N 3 print the same bytes whether the
# members are inflated on the device (the default: RKMH_BGZF_DEVICE=1) or by the workers (=0) (sha256 of stdout per command and route)
cd ${GRAFT_REPO_ROOT:-.}
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, ".")
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
n = 4000000
qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, read_len=150, threads=16)
synth.write_fastq("/tmp/m.fq", qb, qo, synth.read_names(0, n))
open("/tmp/m.fq.gz", "wb").write(synth.bgzf_compress(open("/tmp/m.fq", "rb").read(), level=1, threads=16))
PY
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16 -s 1000"
fail=0
for cmd in "stream" "stream -M 2" "filter" "filter -M 2 -N 3"; do
  want=$(timeout 120 bin/rkmh $cmd $R -f /tmp/m.fq 2>/dev/null | sha256sum | cut -c1-16)
  for mode in 0 1; do
    got=$(RKMH_BGZF_DEVICE=$mode timeout 120 bin/rkmh $cmd $R -f /tmp/m.fq.gz 2>/dev/null | sha256sum | cut -c1-16)
    [ "$got" = "$want" ] && r=same || { r=DIFFERENT; fail=1; }
    echo "rkmh $cmd: plain $want, BGZF with RKMH_BGZF_DEVICE=$mode $got $r"
  done
done
exit $fail
