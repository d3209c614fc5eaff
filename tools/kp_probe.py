import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "oracle")
import rkmh_amd, oracle
from rkmh_amd import api, synth
n = int(sys.argv[1])
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
rb, ro = refs["bases"], refs["offsets"]
qb, qo = synth.generate_reads_fast(rb, ro, 0, n)
ctx = rkmh_amd.Context(0)
ctx.set_references(rb, ro, [16], 1000)
got = ctx.classify(qb, qo)
sk, ln = ctx.get_reference_sketches()
want = oracle.classify_stream(qb, qo, [16], 1000, sk, ln, threads=8)
print(n, "rows equal:", bool((got == want).all()), "mismatches:", int((got != want).any(axis=1).sum()))
