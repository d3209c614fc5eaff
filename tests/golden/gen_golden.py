"""Generates tests/golden/classify_*.json with the CPU oracle (oracle/rk_oracle.c).

SELF-CONSISTENT goldens, conditional on the unpinned mkmh choices U1..U12: the reference
cannot be built here (mkmh submodule absent) and ships no expected outputs, so these vectors pin the
ORACLE's behaviour at default policy (and guard it and the HIP path against regressions); they are
regenerated after any policy flip.  Inputs are the reference's bundled data files (tests/golden/data/*.gz)
and the C2 generator (rkmh_amd/synth.py).  Run: python tests/golden/gen_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from rkmh_amd import synth  # noqa: E402  (numpy generator only; no GPU, no library)

DATA = os.path.join(HERE, "data")


def load(name):
    recs = oracle.kseq_parse_file(os.path.join(DATA, name))
    names = [r[0].decode() for r in recs]
    bases, offs = oracle.pack([r[1] for r in recs])
    return names, bases, offs


def run(tag, ref_file, reads, ks, S, extra=None, **kw):
    rn, rb, ro = load(ref_file)
    qn, qb, qo = reads
    sk, ln = oracle.sketch_refs(rb, ro, ks, S, threads=8)
    out = oracle.classify_stream(qb, qo, ks, S, sk, ln, threads=8, **kw)
    rows = [[qn[i], rn[out[i, 0]], int(out[i, 1]), int(out[i, 2]), int(out[i, 3])] for i in range(len(qn))]
    first = {}
    for i in (0, 1, 2):
        s = oracle.to_upper(bytes(qb[int(qo[i]): int(qo[i + 1])]))
        m = oracle.minhashes(oracle.calc_hashes(s, ks), S)
        first["read%d" % i] = [int(x) for x in m[:8]]
    first["ref0"] = [int(x) for x in sk[0, :8]]
    doc = {"note": "self-consistent oracle output, conditional on policy U1..U12 defaults (parity unpinned)",
           "ref_file": ref_file, "ks": list(ks), "sketch_size": S, "kwargs": {k: int(v) for k, v in kw.items()},
           "columns": ["read_name", "ref_name", "max_shared", "diff", "n_mins"], "rows": rows,
           "first_sketch_hashes": first, "ref_sketch_lens": [int(x) for x in ln]}
    if extra:
        doc.update(extra)
    json.dump(doc, open(os.path.join(HERE, "classify_%s.json" % tag), "w"))
    print(tag, len(rows), "rows; max_shared hist:", np.bincount(np.clip(out[:, 1], 0, 40))[:12])


# C1: rkmh classify -r data/hpv_16.fa -f data/minION25.fq -k 12 -s 1000
run("c1_hpv16_minion25", "hpv_16.fa.gz", load("minION25.fq.gz"), [12], 1000, {"reads_file": "minION25.fq.gz"})
# tie-heavy pair: 60 near-identical Zika references x wgsim reads
run("zika_z1", "zika.refs.fa.gz", load("z1.fq.gz"), [16], 1000, {"reads_file": "z1.fq.gz"})
# C2-mini: first 1000 reads of the C2 generator x all_pave_ref.fa
rn, rb, ro = load("all_pave_ref.fa.gz")
b, o = synth.generate_reads(rb, ro, 0, 1000)
run("c2mini_pave", "all_pave_ref.fa.gz", ([n.decode() for n in synth.read_names(0, 1000)], b, o), [16], 1000,
    {"reads": "rkmh_amd.synth.generate_reads(refs, 0, 1000)"})
# multi-k and the -M path on the same reads (small counter so that collisions are exercised)
run("c2mini_pave_k12_k16", "all_pave_ref.fa.gz", ([n.decode() for n in synth.read_names(0, 1000)], b, o), [12, 16], 1000,
    {"reads": "rkmh_amd.synth.generate_reads(refs, 0, 1000)"})
run("c2mini_pave_M2", "all_pave_ref.fa.gz", ([n.decode() for n in synth.read_names(0, 1000)], b, o), [16], 1000,
    {"reads": "rkmh_amd.synth.generate_reads(refs, 0, 1000)"}, min_kmer_occ=2, counter_slots=1000003)
