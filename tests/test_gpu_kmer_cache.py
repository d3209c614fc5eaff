"""rk_set_kmer_cache: the enumeration of the 4^k k-mer universe behind the k-mer-space kernel, kept in a file between runs.  The file
is used only when its tag -- a hash of every index key, k, fold and seed -- matches; anything else (other references, another fold,
another k, a truncated file) is ignored and overwritten.  Results never depend on the cache."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pad(b):
    out = np.zeros(len(b) + 16, dtype=np.uint8)
    out[: len(b)] = b
    return out


def test_cache_is_loaded_only_when_its_tag_matches(orc, data_dir, tmp_path):
    import rkmh_amd
    from rkmh_amd import synth
    recs = orc.kseq_parse_file(os.path.join(data_dir, "all_pave_ref.fa.gz"))
    rb, ro = orc.pack([r[1] for r in recs])
    rb = _pad(rb)
    qb, qo = synth.generate_reads_fast(rb, ro, 0, 20000)
    T = min(16, os.cpu_count() or 1)
    cache = str(tmp_path / "pave.kmers")

    def run(ks, refs_n=182, expect_state=None, **policy):
        c = rkmh_amd.Context(0, **policy)
        try:
            c.set_kmer_cache(cache)
            c.set_references(rb, ro[: refs_n + 1], ks, 1000)
            assert c.kmer_form()[0]
            st = c.kmer_cache_state()
            if expect_state is not None:
                assert st == expect_state, (ks, refs_n, policy, st)
            sk, ln = c.get_reference_sketches()
            pol = orc.default_policy(**policy)
            want = orc.classify_stream(qb, qo, ks, 1000, sk, ln, pol, threads=T)
            assert (c.classify(qb, qo) == want).all(), (ks, refs_n, policy)
            return st
        finally:
            c.close()

    run([16], expect_state=2)                     # no file yet: enumerated, written
    size = os.path.getsize(cache)
    assert size > 100000
    run([16], expect_state=1)                     # loaded
    run([16], expect_state=2, fold=1)             # another hash fold: the tag differs -> enumerated again, file replaced
    run([16], expect_state=1, fold=1)
    run([16], expect_state=2)                     # ... and back
    run([16], refs_n=60, expect_state=2)          # other references
    run([16], refs_n=60, expect_state=1)
    run([12, 16], expect_state=2)                 # two sizes: both lists in the file
    run([12, 16], expect_state=1)
    run([12], expect_state=2)                     # k = 12 alone: other sketches, other keys, another tag
    with open(cache, "r+b") as f:                 # a truncated file is ignored
        f.truncate(os.path.getsize(cache) // 2)
    run([12], expect_state=2)
    run([12], expect_state=1)
    # a place that cannot be written: the run enumerates and says so
    c = rkmh_amd.Context(0)
    try:
        c.set_kmer_cache("/proc/definitely/not/writable.kmers")
        c.set_references(rb, ro, [16], 1000)
        assert c.kmer_cache_state() == 3 and c.kmer_form()[0]
        c.set_kmer_cache(None)
        c.set_references(rb, ro, [16], 1000)
        assert c.kmer_cache_state() == 0
    finally:
        c.close()


def test_cli_kmer_cache(root, data_dir, tmp_path):
    from rkmh_amd import api, synth
    exe = os.path.join(root, "bin", "rkmh")
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    refs = api.parse_files([ref])
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 5000)
    fq = tmp_path / "r.fq"
    synth.write_fastq(str(fq), qb, qo, synth.read_names(0, 5000))
    base = [exe, "stream", "-r", ref, "-f", str(fq), "-k", "16"]
    want = subprocess.run(base, capture_output=True)
    assert want.returncode == 0
    cache = tmp_path / "c.kmers"
    for i in range(2):
        r = subprocess.run(base + ["--kmer-cache", str(cache)], capture_output=True)
        assert r.returncode == 0 and r.stdout == want.stdout
        assert cache.exists()
    r = subprocess.run(base + ["--devices", "0,0"], capture_output=True, env=dict(os.environ, RKMH_KMER_CACHE=str(cache)))
    assert r.returncode == 0 and r.stdout == want.stdout
    r = subprocess.run([exe, "filter", "-r", ref, "-f", str(fq), "-k", "16", "--kmer-cache", str(cache)], capture_output=True)
    assert r.returncode == 0 and r.stdout == subprocess.run([exe, "filter", "-r", ref, "-f", str(fq), "-k", "16"], capture_output=True).stdout


def test_sketch_writes_the_cache_that_stream_R_loads(root, data_dir, tmp_path):
    """`rkmh sketch --kmer-cache F` enumerates the k-mers behind the sketches it writes; `rkmh stream -R sketches.json --kmer-cache F`
    loads them (RKMH_INDEX_TIMING names the stage) and prints what a run without any cache prints; sketches of other references do
    not use the file (they enumerate, and overwrite it)."""
    from rkmh_amd import api, synth
    exe = os.path.join(root, "bin", "rkmh")
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    refs = api.parse_files([ref])
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 5000)
    fq = tmp_path / "r.fq"
    synth.write_fastq(str(fq), qb, qo, synth.read_names(0, 5000))
    want = subprocess.run([exe, "stream", "-r", ref, "-f", str(fq), "-k", "16"], capture_output=True)
    assert want.returncode == 0
    js, cache = tmp_path / "refs.json", tmp_path / "refs.kmers"
    r = subprocess.run([exe, "sketch", "-f", ref, "-k", "16", "-s", "1000", "-o", str(js), "--kmer-cache", str(cache)], capture_output=True)
    assert r.returncode == 0 and cache.exists() and cache.stat().st_size > 100000, r.stderr[-500:]
    env = dict(os.environ, RKMH_INDEX_TIMING="1")
    r = subprocess.run([exe, "stream", "-R", str(js), "-f", str(fq), "-k", "16", "--kmer-cache", str(cache)], capture_output=True, env=env)
    assert r.returncode == 0 and r.stdout == want.stdout
    assert b"k-mer lists from the cache" in r.stderr and b"k-mer enumeration" not in r.stderr, r.stderr[-800:]
    other = os.path.join(data_dir, "zika.refs.fa.gz")
    r = subprocess.run([exe, "stream", "-r", other, "-f", str(fq), "-k", "16", "--kmer-cache", str(cache)], capture_output=True, env=env)
    assert r.returncode == 0 and b"k-mer enumeration" in r.stderr and b"k-mer lists from the cache" not in r.stderr
