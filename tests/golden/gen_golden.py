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


# ---- the command lines under two hashing-policy presets (--hash-policy default | mash; rk_policy_parse): the stdout of
# `rkmh classify` (C1), of `rkmh stream -N 2 -D 1` (the tie-heavy Zika panel) and the head + digest of `rkmh hash`, from the oracle
# with the same policy.  tests/test_gpu_parity.py::test_cli_stream_and_hash_output runs the binaries against these.
import hashlib  # noqa: E402

PRESETS = {"default": {}, "mash": {"fold": oracle.FOLD_H1, "drop_last_window": 0}}


def cli_golden(preset, fields):
    pol = oracle.default_policy(**fields)
    doc = {"note": "oracle output under --hash-policy %s (parity unpinned: self-consistent)" % preset, "preset": preset,
           "policy_fields": {k: int(v) for k, v in fields.items()}}
    for tag, ref_file, reads_file, ks, S, mm, md in (("classify_c1", "hpv_16.fa.gz", "minION25.fq.gz", [12], 1000, -1, 0),
                                                       ("stream_zika_N2_D1", "zika.refs.fa.gz", "z1.fq.gz", [16], 1000, 2, 1)):
        rn, rb, ro = load(ref_file)
        qn, qb, qo = load(reads_file)
        sk, ln = oracle.sketch_refs(rb, ro, ks, S, policy=pol, threads=8)
        out = oracle.classify_stream(qb, qo, ks, S, sk, ln, policy=pol, threads=8)
        doc[tag] = "".join(oracle.stream_line(rn[out[i, 0]], qn[i], out[i, 1], out[i, 2], out[i, 3], S, min_matches=mm, min_diff=md)
                           for i in range(len(qn)))
    rec = oracle.kseq_parse_file(os.path.join(DATA, "hpv_16.fa.gz"))[0]
    h = oracle.calc_hashes(oracle.to_upper(rec[1]), [12], pol)
    line = rec[0].decode() + "".join("\t%d" % v for v in h) + "\n"
    doc["hash_hpv16_k12"] = {"n_hashes": int(len(h)), "first": [int(v) for v in h[:6]], "sha256": hashlib.sha256(line.encode()).hexdigest()}
    json.dump(doc, open(os.path.join(HERE, "cli_%s.json" % preset), "w"))
    print("cli", preset, len(h), "hashes;", doc["classify_c1"].count("\n"), "+", doc["stream_zika_N2_D1"].count("\n"), "lines")


for preset, fields in PRESETS.items():
    cli_golden(preset, fields)
