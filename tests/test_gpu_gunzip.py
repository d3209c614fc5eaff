"""An ordinary gzip file -- ONE deflate stream -- inflated on the device (rk_gunzip.hip): block headers found by a kernel, a lane per
chunk, the 32 KB in front of every chunk resolved in stream order.  What the slot receives, stretch after stretch, must be the
file's text, record for record: compared through filter's output (names, bases and qualities of every record) with the same
bytes given to an ordinary slot as plain text.  Stored, fixed-code and dynamic blocks, matches that reach (almost) 32 KB back
across chunk edges, many stretches per file, damage, a second member behind the first, and the command line."""
import ctypes as C
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fastq(rng, n, qual="random", name=b"read"):
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 400))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[0.3, 0.2, 0.2, 0.29, 0.01]))
        if qual == "random":
            q = bytes(rng.integers(33, 75, size=L, dtype=np.uint8))
        elif qual == "flat":
            q = b"I" * L
        else:
            q = bytes(rng.integers(33, 127, size=L, dtype=np.uint8))
        recs.append(b"@" + name + b"%d/%d comment\n" % (i, i % 7) + s + b"\n+\n" + q + b"\n")
    return b"".join(recs)


def _gzip_container(raw_deflate, text):
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + raw_deflate + (zlib.crc32(text) & 0xFFFFFFFF).to_bytes(4, "little")
            + (len(text) & 0xFFFFFFFF).to_bytes(4, "little"))


def _mixed_stream(pieces):
    """one deflate stream out of (text, level, strategy) pieces: every piece compressed on its own with the 32 KB in front of it as
    its dictionary and ended by a sync flush (an empty stored block), the last one finished -- stored, fixed and dynamic blocks in
    one stream, with matches that reach back across the joints"""
    out, done = [], b""
    for i, (text, level, strategy) in enumerate(pieces):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy, done[-32768:]) if done else zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        out.append(co.compress(text))
        out.append(co.flush(zlib.Z_FINISH if i == len(pieces) - 1 else zlib.Z_SYNC_FLUSH))
        done += text
    return b"".join(out), done


@pytest.fixture(scope="module")
def gctx(orc, data_dir):
    import rkmh_amd
    recs = orc.kseq_parse_file(os.path.join(data_dir, "hpv_16.fa.gz"))
    rb, ro = orc.pack([r[1] for r in recs])
    c = rkmh_amd.Context(0)
    c.set_references(np.concatenate([rb, np.zeros(16, np.uint8)]), ro, [16], 1000)
    yield c
    c.close()


def _through_device(gctx, path, text, slot_bytes, env):
    """every stretch of the file through a device-text slot; the records filter prints for each must equal those of the same bytes
    as plain text; returns the statuses"""
    from rkmh_amd import api
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    gz = api.Gzip.open(str(path))
    assert gz is not None and gz.first_byte() == ord("@")
    dev = api.FastqSlot(gctx, max_bytes=slot_bytes, device_text=True)
    plain = api.FastqSlot(gctx, max_bytes=slot_bytes)
    dev.set_filter_output(-1, -100)      # every read passes: the whole text comes back, as filter prints it
    statuses, at = [], 0
    try:
        ncalls = gz.plan(slot_bytes)
        for call in range(ncalls):
            st, n, off = dev.load_gzip(gz, call)
            statuses.append(st)
            if st != 0:
                break
            assert off == at, (call, off, at)
            if n == 0:
                continue
            want_text = text[off:off + n] if off + n <= len(text) else text[off:] + b"\n"
            assert len(want_text) == n
            res = dev.classify_raw(n)
            assert res.status == 0, (call, res.status)
            got = dev.filter_records(res, -1, -100)
            buf = plain.text_buffer()
            C.memmove(buf, want_text, n)
            res2 = plain.classify_raw(n)
            assert res2.status == 0 and res2.nrec == res.nrec
            assert got == plain.filter_records(res2, -1, -100), (call, off, n)
            at = off + (n if off + n <= len(text) else n - 1)
        if all(s == 0 for s in statuses):
            assert at == len(text)
    finally:
        dev.destroy(); plain.destroy(); gz.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return statuses


@pytest.mark.parametrize("level,qual,nrec", [(1, "random", 30000), (6, "random", 30000), (9, "flat", 30000), (6, "wide", 12000), (1, "flat", 12000)])
@pytest.mark.parametrize("chunk_kb,stretch_kb", [(32, 0), (4, 512), (1, 96)])
def test_device_gunzip_gives_the_text(gctx, tmp_path, level, qual, nrec, chunk_kb, stretch_kb):
    rng = np.random.default_rng(level * 31 + nrec % 13 + chunk_kb)
    text = _fastq(rng, nrec, qual)
    path = tmp_path / "t.fq.gz"
    path.write_bytes(gzip.compress(text, level))
    env = {"RKMH_GZIP_CHUNK_KB": str(chunk_kb)}
    if stretch_kb:
        env["RKMH_GZIP_STRETCH_KB"] = str(stretch_kb)
    st = _through_device(gctx, path, text, 32 << 20, env)
    assert st and all(s == 0 for s in st), st


@pytest.mark.parametrize("level,mem_level,wbits,strategy", [(6, 1, 15, 0), (6, 4, 15, 0), (9, 9, 15, 0), (6, 8, 9, 0), (1, 8, 12, 0), (6, 8, 15, zlib.Z_FILTERED),
                                                            (6, 2, 10, zlib.Z_RLE), (4, 9, 15, zlib.Z_HUFFMAN_ONLY), (3, 8, 15, zlib.Z_FIXED), (0, 8, 15, 0)])
def test_device_gunzip_zlib_parameters(gctx, tmp_path, level, mem_level, wbits, strategy):
    """deflate streams as zlib writes them with other parameters than gzip's defaults: tiny blocks (memLevel 1: 128 symbols each --
    hundreds of code tables per chunk), 32 K-symbol blocks (memLevel 9), small windows (wbits 9 .. 12), every strategy, stored only.
    Every stretch the device takes must be the text; a stream it hands over (status 1) must have been right up to there."""
    rng = np.random.default_rng(level * 7 + mem_level * 3 + wbits)
    text = _fastq(rng, 6000, "wide" if strategy == zlib.Z_HUFFMAN_ONLY else "random")
    co = zlib.compressobj(level, zlib.DEFLATED, -wbits, mem_level, strategy)
    raw = co.compress(text) + co.flush()
    path = tmp_path / "p.fq.gz"
    path.write_bytes(_gzip_container(raw, text))
    assert gzip.decompress(path.read_bytes()) == text
    st = _through_device(gctx, path, text, 16 << 20, {"RKMH_GZIP_CHUNK_KB": "4", "RKMH_GZIP_STRETCH_KB": "400"})
    assert st and all(s == 0 for s in st[:-1]) and st[-1] in (0, 1), st
    if st[-1] == 1:
        print("handed over:", (level, mem_level, wbits, strategy), st)
    # all but the all-stored and all-fixed streams (no dynamic header to start a chunk from: one long chunk per stretch) are the device's
    if strategy != zlib.Z_FIXED and level != 0:
        assert st[-1] == 0, st


# soak: RKMH_TEST_GZ_SEEDS=400 [RKMH_TEST_SEED_BASE=...]
@pytest.mark.parametrize("seed", range(int(os.environ.get("RKMH_TEST_SEED_BASE", "0")), int(os.environ.get("RKMH_TEST_SEED_BASE", "0")) + int(os.environ.get("RKMH_TEST_GZ_SEEDS", "8"))))
def test_device_gunzip_randomized(gctx, tmp_path, seed):
    """random FASTQ text (1 to 20 000 records, three kinds of quality strings) deflated with random zlib parameters, inflated on the
    device with random chunk and stretch sizes: every stretch is the text, record for record"""
    rng = np.random.default_rng(77000 + seed)
    nrec = int(rng.choice([1, 7, 60, 400, 3000, 9000, 20000]))
    text = _fastq(rng, nrec, str(rng.choice(["random", "flat", "wide"])), b"s%d_" % seed)
    level, mem_level, wbits = int(rng.integers(1, 10)), int(rng.integers(1, 10)), int(rng.integers(9, 16))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY]))
    co = zlib.compressobj(level, zlib.DEFLATED, -wbits, mem_level, strategy)
    raw = co.compress(text) + co.flush()
    path = tmp_path / "s.fq.gz"
    path.write_bytes(_gzip_container(raw, text))
    env = {"RKMH_GZIP_CHUNK_KB": str(int(rng.choice([1, 2, 4, 8, 32])))}
    if rng.integers(0, 2):
        env["RKMH_GZIP_STRETCH_KB"] = str(int(rng.integers(32, 3000)))
    st = _through_device(gctx, path, text, 16 << 20, env)
    assert st and all(x == 0 for x in st), (seed, level, mem_level, wbits, strategy, nrec, env, st)


def test_device_gunzip_tiny_and_odd_files(gctx, tmp_path):
    """three records; one record; a text without its last newline; a 70 KB record-free... no: records of 20 KB (long reads)"""
    rng = np.random.default_rng(2)
    for k, text in enumerate((_fastq(rng, 3), _fastq(rng, 1), _fastq(rng, 400)[:-1],
                              b"".join(b"@long%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=20000)) + b"\n+\n" + b"F" * 20000 + b"\n" for i in range(40)))):
        path = tmp_path / ("odd%d.fq.gz" % k)
        path.write_bytes(gzip.compress(text, 6))
        st = _through_device(gctx, path, text, 8 << 20, {"RKMH_GZIP_CHUNK_KB": "4"})
        assert st and all(s == 0 for s in st), (k, st)


def test_device_gunzip_stored_fixed_and_dynamic_blocks_with_far_matches(gctx, tmp_path):
    """stored (level 0), fixed-code (Z_FIXED) and dynamic blocks in one stream; the piece after every joint repeats records of the
    text ~31 KB back, so its first matches reach (almost) a whole window back -- across a chunk edge whenever a chunk begins there"""
    rng = np.random.default_rng(5)
    pieces, done = [], b""
    kinds = [(6, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
             (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_DEFAULT_STRATEGY)]
    for i, (level, strategy) in enumerate(kinds * 3):
        fresh = _fastq(rng, int(rng.integers(150, 600)), "random", b"p%d_" % i)
        if done:
            # whole records from ~31 KB back, again: matches at distances just below 32 KB
            back = done[-32300:]
            first = back.find(b"\n@") + 1
            fresh = back[first:first + 2500].rsplit(b"\n@", 1)[0] + b"\n" + fresh
            assert fresh.startswith(b"@")
        pieces.append((fresh, level, strategy))
        done += fresh
    raw, text = _mixed_stream(pieces)
    assert zlib.decompress(raw, -15) == text
    path = tmp_path / "mixed.fq.gz"
    path.write_bytes(_gzip_container(raw, text))
    for chunk_kb, stretch_kb in ((2, 0), (8, 200), (32, 0)):
        env = {"RKMH_GZIP_CHUNK_KB": str(chunk_kb)}
        if stretch_kb:
            env["RKMH_GZIP_STRETCH_KB"] = str(stretch_kb)
        st = _through_device(gctx, path, text, 32 << 20, env)
        assert st and all(s == 0 for s in st), (chunk_kb, stretch_kb, st)


def test_device_gunzip_hands_over_or_refuses(gctx, tmp_path):
    """a second member behind the first: the device route stops (status 1) with nothing of the unfinished stretch delivered;
    a damaged payload: status 1 (the host reader reports it) or an error -- never text"""
    from rkmh_amd import api
    rng = np.random.default_rng(8)
    a, b = _fastq(rng, 5000), _fastq(rng, 3000, name=b"second")
    two = tmp_path / "two.fq.gz"
    two.write_bytes(gzip.compress(a, 6) + gzip.compress(b, 6))
    st = _through_device(gctx, two, a + b, 32 << 20, {"RKMH_GZIP_CHUNK_KB": "8"})
    assert st[-1] == 1, st
    img = bytearray(gzip.compress(a, 6))
    img[len(img) // 2] ^= 0x10
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(img))
    try:
        st = _through_device(gctx, bad, a, 32 << 20, {"RKMH_GZIP_CHUNK_KB": "8"})
        assert st[-1] == 1, st
    except (RuntimeError, AssertionError) as e:
        assert "trailer" in str(e) or "damaged" in str(e) or "status" in str(e) or isinstance(e, AssertionError)


def test_cli_gzip_equals_plain(tmp_path, data_dir):
    """bin/rkmh stream / filter on reads.fq.gz (ordinary gzip) = on reads.fq, byte for byte: the default, many small stretches, the
    host route (RKMH_GZIP_DEVICE=0), two files in one run, -M, a two-member file (handed over mid-file) and a damaged one (refused)"""
    from rkmh_amd import api, synth
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "rkmh")
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    refs = api.parse_files([ref])
    n = 60000
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, read_len=150, threads=4)
    rng = np.random.default_rng(3)
    recs = []
    for i in range(n):
        sq = bytes(qb[int(qo[i]):int(qo[i + 1])])
        if i % 11 == 0:
            sq = sq[: int(rng.integers(16, 150))]
        recs.append(b"@read%07d comment\n" % i + sq + b"\n+\n" + bytes(rng.integers(33, 75, size=len(sq), dtype=np.uint8)) + b"\n")
    text = b"".join(recs)
    fq = tmp_path / "r.fq"
    fq.write_bytes(text)
    gz1 = tmp_path / "r.fq.gz"
    gz1.write_bytes(gzip.compress(text, 6))
    half = text.rfind(b"\n@", 0, len(text) // 2) + 1
    gz2 = tmp_path / "two.fq.gz"
    gz2.write_bytes(gzip.compress(text[:half], 1) + gzip.compress(text[half:], 9))
    base = ["-r", ref, "-k", "16", "-s", "1000"]

    def run(cmd, files, env=None, extra=()):
        r = subprocess.run([exe, cmd] + base + list(extra) + sum((["-f", str(f)] for f in files), []), capture_output=True,
                           env=dict(os.environ, RKMH_BGZF_TIMING="1", RKMH_TIMING="1", **(env or {})))
        assert r.returncode == 0, r.stderr[-600:]
        return r.stdout, r.stderr

    want, _ = run("stream", [fq])
    got, err = run("stream", [gz1])
    assert got == want and b"[gzip device]" in err and b"stops at byte" not in err, err[-800:]
    for env in ({"RKMH_GZIP_STRETCH_KB": "300", "RKMH_GZIP_CHUNK_KB": "4"}, {"RKMH_GZIP_DEVICE": "0"}):
        got, err = run("stream", [gz1], env)
        assert got == want, env
        assert (b"[gzip device]" in err) == ("RKMH_GZIP_DEVICE" not in env)
    got, _ = run("stream", [gz1, fq, gz1])
    assert got == want * 3
    got, err = run("stream", [gz2, gz1])             # the second member is the sequential reader's; the file behind it as well
    assert got == want * 2 and b"stops at byte" in err
    wantf, _ = run("filter", [fq])
    gotf, _ = run("filter", [gz1], {"RKMH_GZIP_STRETCH_KB": "700"})
    assert gotf == wantf and len(wantf) > 0
    wantf, _ = run("filter", [fq], extra=("-M", "2", "-N", "3"))
    gotf, _ = run("filter", [gz1, ], extra=("-M", "2", "-N", "3"))
    assert gotf == wantf
    wantm, _ = run("stream", [fq], extra=("-M", "2"))
    gotm, _ = run("stream", [gz1], extra=("-M", "2"))
    assert gotm == wantm
    img = bytearray(gz1.read_bytes())
    img[len(img) // 3] ^= 0x04
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(img))
    r = subprocess.run([exe, "stream"] + base + ["-f", str(bad)], capture_output=True)
    assert r.returncode != 0


def test_cli_gzip_references_through_the_device(tmp_path, data_dir):
    """-r genome.fa.gz (ordinary gzip): with RKMH_RAW_REFS=1 the references' text is inflated on the device straight into the FASTA
    loader (rk_fasta_load_put_gzip) -- same stdout as the host parser and as the uncompressed file, for the bundled panel and for a
    12 MB synthetic genome in 60-column lines inflated in several stretches, alone and beside a plain -r file."""
    from rkmh_amd import api, synth
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "rkmh")
    rng = np.random.default_rng(12)
    recs = []
    for i in range(6):
        sq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(1500000, 2500000))))
        recs.append(b">chr%d synthetic\n" % i + b"\n".join(sq[j:j + 60] for j in range(0, len(sq), 60)) + b"\n")
    fa = tmp_path / "g.fa"
    fa.write_bytes(b"".join(recs))
    fagz = tmp_path / "g.fa.gz"
    fagz.write_bytes(gzip.compress(fa.read_bytes(), 6))
    refs = api.parse_files([str(fa)])
    n = 20000
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, read_len=150, threads=4)
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"".join(b"@q%06d\n" % i + bytes(qb[int(qo[i]):int(qo[i + 1])]) + b"\n+\n" + b"I" * 150 + b"\n" for i in range(n)))
    panel_gz = os.path.join(data_dir, "hpv_16.fa.gz")

    def run(refs_, env):
        r = subprocess.run([exe, "stream", "-k", "16", "-s", "1000", "-f", str(fq)] + sum((["-r", str(x)] for x in refs_), []), capture_output=True,
                           env=dict(os.environ, RKMH_TIMING="1", **env))
        assert r.returncode == 0, r.stderr[-500:]
        return r.stdout, r.stderr

    want, _ = run([fa], {"RKMH_RAW_REFS": "0"})
    assert want.count(b"\n") == n
    fabz = tmp_path / "g.bgzf.fa.gz"
    fabz.write_bytes(synth.bgzf_compress(fa.read_bytes(), level=6))      # a bgzip'd genome: rk_fasta_load_put_bgzf
    for refs_, env in (([fagz], {"RKMH_RAW_REFS": "1"}), ([fagz], {"RKMH_RAW_REFS": "1", "RKMH_GZIP_STRETCH_KB": "900", "RKMH_GZIP_CHUNK_KB": "8"}), ([fa], {"RKMH_RAW_REFS": "1"}),
                       ([fabz], {"RKMH_RAW_REFS": "1"})):
        got, err = run(refs_, env)
        assert got == want, (refs_, env)
        assert b"references through the device" in err, err[-600:]
    for f_ in (fagz, fabz):
        got, _ = run([f_], {"RKMH_RAW_REFS": "0"})
        assert got == want
    want2, _ = run([fa, panel_gz], {"RKMH_RAW_REFS": "0"})
    got2, err2 = run([fagz, panel_gz], {"RKMH_RAW_REFS": "1"})      # (the small panel is gzip as well; its lower-case text may send all of them to the host parser: same lines either way)
    assert got2 == want2
