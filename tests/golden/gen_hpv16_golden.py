"""Generates tests/golden/hpv16_minion25.json with the CPU oracle's restatement of main_hpv16 (oracle/oracle.py::hpv16,
/root/reference/src/rkmh.cpp:2366-2723) on the reference's bundled files: data/all_pave_ref.fa (types), data/new_refs.fa
(HPV16 sublineages) and the 25 nanopore reads of data/minION25.fq, -k 16.

SELF-CONSISTENT golden (parity unpinned): besides U1..U12 it is conditional on U13/U14 (hash_set_intersection_size and
sort_by_similarity are absent from the reference snapshot; see oracle/oracle.py).  Run: python tests/golden/gen_hpv16_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

DATA = os.path.join(HERE, "data")
types = oracle.kseq_parse_file(os.path.join(DATA, "all_pave_ref.fa.gz"))
subs = oracle.kseq_parse_file(os.path.join(DATA, "new_refs.fa.gz"))
reads = oracle.kseq_parse_file(os.path.join(DATA, "minION25.fq.gz"))
lines, tst, err = oracle.hpv16([t[0] for t in types], [t[1] for t in types], [t[0] for t in subs], [t[1] for t in subs],
                               [r[0] for r in reads], [r[1] for r in reads], [16])
doc = {"note": "self-consistent oracle output, conditional on policies U1..U14 (parity unpinned)", "ks": [16],
       "sim_den": oracle.HPV16_SIM_DEN, "reads_file": "minION25.fq.gz", "stdout_lines": lines, "stderr_tables": err,
       "tst_sha256": hashlib.sha256(tst.encode()).hexdigest(), "tst_first_80": tst[:80]}
json.dump(doc, open(os.path.join(HERE, "hpv16_minion25.json"), "w"), indent=0)
print(len(lines), "lines;", err)
