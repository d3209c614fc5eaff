"""k = 17 .. 20 on the k-mer-space kernel (wide k-mers: 64-bit packed, the km2 map and kkeys, rk_kmer.hip KT = 32).  The 4^k k-mer universe
is enumerated once per reference set (0.1 s at k = 17 ... seconds at k = 20: unasked only up to RKMH_KMER_ENUM_MAXK = 18, beyond that
from the cache file) -- every row must equal the oracle's, and the hash-space kernel's, for reads with N, ragged lengths, lower case,
repeats, families of near-identical references, -M, and a cache written by one context and loaded by the next."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pad(b):
    out = np.zeros(len(b) + 16, dtype=np.uint8)
    out[: len(b)] = b
    return out


def _reads(orc, rb, ro, n, seed, L=150):
    from rkmh_amd import synth
    qb, qo = synth.generate_reads_fast(rb, ro, 1000, 1000 + n, read_len=L)
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n):
        s = bytearray(qb[int(qo[i]): int(qo[i + 1])])
        r = rng.random()
        if r < 0.08:
            s[int(rng.integers(0, L))] = ord("N")
        elif r < 0.16:
            s = s[: int(rng.integers(1, L))]
        elif r < 0.2:
            s = bytearray(bytes(s).lower())
        elif r < 0.25:
            s[70:110] = s[10:50]
        seqs.append(bytes(s))
    b, o = orc.pack(seqs)
    return _pad(b), o


@pytest.mark.parametrize("k,S", [(17, 1000), (18, 1000), (18, 4000), (19, 600), (20, 2000)])
def test_wide_kmer_form_equals_the_oracle(orc, data_dir, tmp_path, k, S):
    import rkmh_amd
    recs = orc.kseq_parse_file(os.path.join(data_dir, "all_pave_ref.fa.gz"))
    zika = orc.kseq_parse_file(os.path.join(data_dir, "zika.refs.fa.gz"))[:25]
    rb, ro = orc.pack([r[1] for r in recs] + [r[1] for r in zika])
    rb = _pad(rb)
    qb, qo = _reads(orc, rb, ro, 30000, seed=k)
    T = min(16, os.cpu_count() or 1)
    cache = str(tmp_path / "wide.kmers")
    os.environ["RKMH_KMER_ENUM_MAXK"] = "20"
    try:
        want = None
        for rnd in range(2):
            c = rkmh_amd.Context(0)
            try:
                c.set_kmer_cache(cache)
                c.set_references(rb, ro, [k], S)
                active, found = c.kmer_form()
                assert active and found > 0, (k, S)
                assert c.kmer_cache_state() == (2 if rnd == 0 else 1)
                sk, ln = c.get_reference_sketches()
                if want is None:
                    wsk, wln = orc.sketch_refs(rb, ro, [k], S, threads=T)
                    assert (sk == wsk).all() and (ln == wln).all()
                    want = orc.classify_stream(qb, qo, [k], S, wsk, wln, threads=T)
                got = c.classify(qb, qo)
                bad = np.nonzero((got != want).any(axis=1))[0]
                assert len(bad) == 0, (k, S, rnd, len(bad), got[bad[:5]], want[bad[:5]])
                if rnd == 1:      # -M on the wide form: bounded (per-key mask tested in the kernel) and exact (hash-space kernel)
                    slots = 2000003
                    mwant = orc.classify_stream(qb, qo, [k], S, wsk, wln, threads=T, min_kmer_occ=2, counter_slots=slots)
                    for compact in (True, False):
                        c.set_min_num_bound(0)
                        cnt = rkmh_amd.Counter(c, slots=slots, compact=compact)
                        c.count_batch(qb, qo, cnt)
                        c.set_depth_filter(cnt, 2)
                        got = c.classify(qb, qo)
                        c.set_depth_filter(None, 0)
                        c.set_min_num_bound(-1)
                        cnt.destroy()
                        w = mwant.copy()
                        w[:, 3] = 0
                        bad = np.nonzero((got != w).any(axis=1))[0]
                        assert len(bad) == 0, (k, compact, len(bad), got[bad[:5]], w[bad[:5]])
            finally:
                c.close()
    finally:
        os.environ.pop("RKMH_KMER_ENUM_MAXK", None)


def test_k19_is_not_enumerated_unasked(orc, data_dir):
    """beyond RKMH_KMER_ENUM_MAXK (18) without a cache the hash-space kernel serves the run -- same rows"""
    import rkmh_amd
    recs = orc.kseq_parse_file(os.path.join(data_dir, "all_pave_ref.fa.gz"))[:40]
    rb, ro = orc.pack([r[1] for r in recs])
    rb = _pad(rb)
    qb, qo = _reads(orc, rb, ro, 5000, seed=3)
    c = rkmh_amd.Context(0)
    try:
        c.set_references(rb, ro, [19], 1000)
        assert not c.kmer_form()[0]
        sk, ln = c.get_reference_sketches()
        assert (c.classify(qb, qo) == orc.classify_stream(qb, qo, [19], 1000, sk, ln, threads=8)).all()
    finally:
        c.close()


def test_cli_keeps_the_wide_enumeration_beside_the_references(root, data_dir, tmp_path):
    """bin/rkmh at ONE k of 17 .. 20 without --kmer-cache: the first run enumerates (k = 19 here: beyond what is done unasked) and
    leaves <ref>.k19.s600.rkkc beside the reference file, the second run loads it; both use the k-mer kernel and print what
    --no-kmer-cache (the hash-space kernel) prints.  Another sketch size gets a file of its own."""
    import gzip
    import shutil
    import subprocess
    from rkmh_amd import api, synth
    ref = tmp_path / "panel.fa"
    with gzip.open(os.path.join(data_dir, "all_pave_ref.fa.gz"), "rb") as f, open(ref, "wb") as g:
        shutil.copyfileobj(f, g)
    refs = api.parse_files([str(ref)])
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 20000, read_len=150, threads=4)
    fq = tmp_path / "r.fq"
    synth.write_fastq(str(fq), qb, qo, synth.read_names(0, 20000))
    exe = os.path.join(root, "bin", "rkmh")
    env = dict(os.environ, RKMH_INDEX_TIMING="1")
    env.pop("RKMH_KMER_CACHE", None)
    for cmd in ("stream", "filter"):
        base = [exe, cmd, "-r", str(ref), "-f", str(fq), "-k", "19", "-s", "600" if cmd == "stream" else "700", "-N", "2"]
        cache = str(ref) + ".k19.s%s.rkkc" % ("600" if cmd == "stream" else "700")
        plain = subprocess.run(base + ["--no-kmer-cache"], capture_output=True, env=env)
        assert plain.returncode == 0 and not os.path.exists(cache) and b"k-mer enumeration" not in plain.stderr
        first = subprocess.run(base, capture_output=True, env=env)
        assert first.returncode == 0, first.stderr[-500:]
        assert os.path.getsize(cache) > 1000 and b"k-mer enumeration" in first.stderr
        second = subprocess.run(base, capture_output=True, env=env)
        assert second.returncode == 0 and b"k-mer lists from the cache" in second.stderr and b"k-mer enumeration" not in second.stderr
        assert first.stdout == plain.stdout and second.stdout == plain.stdout and len(plain.stdout) > 1000
