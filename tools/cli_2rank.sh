#!/bin/bash
# 2-rank run of the torch.distributed CLI on ONE GPU (gloo) against the single-process C++ binary: same stdout expected.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 20001, threads=8)
synth.write_fastq("/tmp/r20k.fq", qb, qo, synth.read_names(0, 20001))
PY
bin/rkmh stream -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/r20k.fq -k 16 -s 1000 > /tmp/one.tsv 2>/dev/null
RKMH_ONE_DEVICE=1 RKMH_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 -m rkmh_amd.cli stream -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/r20k.fq -k 16 -s 1000 > /tmp/two.tsv 2>/tmp/two.err
RKMH_ONE_DEVICE=1 RKMH_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29542 -m rkmh_amd.cli stream -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/r20k.fq -k 16 -s 1000 > /tmp/three.tsv 2>/tmp/three.err
wc -l /tmp/one.tsv /tmp/two.tsv /tmp/three.tsv
grep -P "\t" /tmp/two.tsv > /tmp/two.f; grep -P "\t" /tmp/three.tsv > /tmp/three.f   # gloo prints its (interleaved) banner on stdout; result lines have tabs
cmp /tmp/one.tsv /tmp/two.f && cmp /tmp/one.tsv /tmp/three.f && echo "1-process C++ == 2-rank == 3-rank python CLI output"
# filter -M (counter all-reduce between the passes) and -I, 2 ranks
for FL in "-M 2 -N 3" "-I 2 -D 1"; do
  bin/rkmh filter -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/r20k.fq -k 16 -s 1000 $FL > /tmp/f1.out 2>/dev/null
  RKMH_ONE_DEVICE=1 RKMH_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 -m rkmh_amd.cli filter -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/r20k.fq -k 16 -s 1000 $FL > /tmp/f2.out 2>/tmp/f2.err
  grep -v -i "gloo\|rank" /tmp/f2.out | grep -v "^$" > /tmp/f2.f || true   # gloo's banner (two interleaved lines, one of them empty)
  wc -l /tmp/f1.out /tmp/f2.f | head -2
  cmp /tmp/f1.out /tmp/f2.f && echo "filter $FL: 1-process C++ == 2-rank python CLI output"
done
