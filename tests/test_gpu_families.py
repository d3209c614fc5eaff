"""Panels of near-identical genomes (BASELINE config 3's shape: 61 Zika genomes, 21 HPV16 variants beside unrelated references).
Most sketch hashes of a family are shared by most of its members, so the k-mer-space kernel stores their posting lists as
(base list, exceptions) and expands each touched base once per read (rk_api.hip build_kpost, rk_kmer.hip phase 2).  Ties are the
rule here -- identical genomes score the same and the FIRST reference must win (rkmh.cpp:878) -- and every row must equal the
oracle's, for every counter form (8-bit, 16-bit, sparse), with repeats inside reads, N runs, several k-mer sizes and -M."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pad(b):
    out = np.zeros(len(b) + 16, dtype=np.uint8)
    out[: len(b)] = b
    return out


def _family_panel(rng, nfam=(40, 22, 9), unrelated=25, glen=6000, rate=0.004):
    """families of mutated copies of one ancestor each (some members identical), interleaved with unrelated genomes"""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    refs = []
    for fi, n in enumerate(nfam):
        anc = rng.choice(acgt, size=glen)
        for m in range(n):
            g = anc.copy()
            if m % 5 != 0:                                   # every fifth member is an exact copy of the ancestor
                pos = rng.integers(0, glen, size=max(1, int(glen * rate * (1 + m % 4))))
                g[pos] = rng.choice(acgt, size=len(pos))
            refs.append(bytes(g))
        for _ in range(unrelated // len(nfam)):
            refs.append(bytes(rng.choice(acgt, size=glen)))
    order = rng.permutation(len(refs)) if False else np.arange(len(refs))
    return [refs[i] for i in order]


def _reads_from(rng, refs, n, L, with_repeats=True):
    seqs = []
    for i in range(n):
        g = refs[int(rng.integers(0, len(refs)))]
        p = int(rng.integers(0, len(g) - L))
        s = bytearray(g[p: p + L])
        r = rng.random()
        if r < 0.1:
            for q in rng.integers(0, L, size=3):
                s[int(q)] = ord("N")
        elif r < 0.2 and with_repeats and L >= 100:         # the same 40 bases twice: k-mers repeat inside the read (rank > 0)
            s[60:100] = s[10:50]
        elif r < 0.25:
            s = s[: int(rng.integers(5, L))]
        elif r < 0.3:
            s = bytearray(bytes(s).lower())
        seqs.append(bytes(s))
    return seqs


@pytest.mark.parametrize("ks,S,L,nbases", [([16], 1000, 150, None), ([16], 1000, 150, 0), ([16], 1000, 150, 1), ([12, 16], 600, 100, None),
                                           ([14], 500, 300, None), ([16], 2000, 250, None), ([15], 300, 150, None)])
def test_family_panels_equal_the_oracle(orc, ks, S, L, nbases):
    import rkmh_amd
    rng = np.random.default_rng(1234 + len(ks) + S + L)
    refs = _family_panel(rng)
    rb, ro = orc.pack(refs)
    rb = _pad(rb)
    seqs = _reads_from(rng, refs, 30000, L)
    qb, qo = orc.pack(seqs)
    qb = _pad(qb)
    T = min(16, os.cpu_count() or 1)
    if nbases is not None:
        os.environ["RKMH_KBASES"] = str(nbases)
    try:
        c = rkmh_amd.Context(0)
    finally:
        os.environ.pop("RKMH_KBASES", None)
    try:
        c.set_references(rb, ro, ks, S)
        assert c.kmer_form()[0]
        sk, ln = c.get_reference_sketches()
        wsk, wln = orc.sketch_refs(rb, ro, ks, S, threads=T)
        assert (sk == wsk).all() and (ln == wln).all()
        want = orc.classify_stream(qb, qo, ks, S, wsk, wln, threads=T)
        got = c.classify(qb, qo)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (len(bad), got[bad[:5]], want[bad[:5]])
        # -M on the same panel (bounded and exact)
        slots = 3000017
        mwant = orc.classify_stream(qb, qo, ks, S, wsk, wln, threads=T, min_kmer_occ=2, counter_slots=slots)
        cnt = rkmh_amd.Counter(c, slots=slots)
        c.count_batch(qb, qo, cnt)
        try:
            for bound in (0, -1):
                c.set_min_num_bound(bound)
                c.set_depth_filter(cnt, 2)
                got = c.classify(qb, qo)
                w = mwant.copy()
                if bound >= 0:
                    w[:, 3] = np.minimum(w[:, 3], bound)
                bad = np.nonzero((got != w).any(axis=1))[0]
                assert len(bad) == 0, (bound, len(bad), got[bad[:5]], w[bad[:5]])
        finally:
            c.set_depth_filter(None, 0)
            c.set_min_num_bound(-1)
            cnt.destroy()
    finally:
        c.close()


def test_family_panel_with_sparse_counters(orc):
    """more than 512 references: the per-read counters are a sparse map, which cannot subtract -- the plain form of every list is used"""
    import rkmh_amd
    rng = np.random.default_rng(77)
    refs = _family_panel(rng, nfam=(120, 60, 30), unrelated=390, glen=3000)
    assert len(refs) > 512
    rb, ro = orc.pack(refs)
    rb = _pad(rb)
    seqs = _reads_from(rng, refs, 20000, 150)
    qb, qo = orc.pack(seqs)
    qb = _pad(qb)
    T = min(16, os.cpu_count() or 1)
    c = rkmh_amd.Context(0)
    try:
        c.set_references(rb, ro, [16], 400)
        sk, ln = c.get_reference_sketches()
        want = orc.classify_stream(qb, qo, [16], 400, sk, ln, threads=T)
        got = c.classify(qb, qo)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert len(bad) == 0, (len(bad), got[bad[:5]], want[bad[:5]])
    finally:
        c.close()
