"""BGZF input (rk_bgzf_*, rk_parse.cpp): members located from their headers, jobs of consecutive members, every job inflated on its
own and cut to the whole FASTQ records that START in it -- the concatenation of all jobs' records is the original text, whatever
the job size, with libdeflate and with zlib; plain gzip and plain text are refused; a corrupt member is an error."""
import ctypes as C
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest


def _fastq(rng, n, lo=30, hi=400, tricky=True):
    recs = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L))
        q = bytearray(rng.integers(33, 127, size=L, dtype=np.uint8).tobytes())
        if tricky and i % 3 == 0:
            q[0] = ord("@")                       # a quality line that begins like a header
        if tricky and i % 7 == 0:
            q[0] = ord("+")
        recs.append(b"@read%d some comment\n" % i + s + b"\n+\n" + bytes(q) + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("nolibdeflate", [False, True])
def test_jobs_reassemble_the_text(tmp_path, nolibdeflate):
    if nolibdeflate:      # the zlib branch is chosen when the library is first used: a fresh interpreter
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k", "test_jobs_reassemble_the_text and False"],
                           env=dict(os.environ, RKMH_NO_LIBDEFLATE="1"), capture_output=True)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        return
    from rkmh_amd import api, synth
    rng = np.random.default_rng(5)
    text = _fastq(rng, 6000)
    path = tmp_path / "reads.fq.gz"
    path.write_bytes(synth.bgzf_compress(text, level=1, block=0xff00))
    z = api.Bgzf.open(str(path))
    assert z is not None and z.text_bytes == len(text) and z.first_byte() == ord("@")
    assert gzip.decompress(path.read_bytes()) == text        # a BGZF file is a valid multi-member gzip file
    cap = len(text) + 64
    dst = C.create_string_buffer(cap)
    for target in (1, 70000, 300000, 1 << 20, 1 << 30):
        first = z.plan(target)
        assert first[0] == 0 and first[-1] == z.members and all(a < b for a, b in zip(first, first[1:]))
        got, at = b"", 0
        for b0, b1 in zip(first, first[1:]):
            st, n, off = z.fastq_records(b0, b1, dst, cap)
            assert st == 0
            if n:
                assert off == at and dst.raw[:1] == b"@"
            got += dst.raw[:n]
            at += n
        assert got == text, target
    # tiny members: records span many of them
    path2 = tmp_path / "tiny.fq.gz"
    path2.write_bytes(synth.bgzf_compress(text[:200000], level=6, block=97))
    z2 = api.Bgzf.open(str(path2))
    first = z2.plan(1000)
    got = b""
    for b0, b1 in zip(first, first[1:]):
        st, n, off = z2.fastq_records(b0, b1, dst, cap)
        assert st == 0 and off == len(got) or n == 0
        got += dst.raw[:n]
    assert got == text[:200000]
    with pytest.raises(api.RkmhError):
        z2.fastq_records(0, z2.members, dst, 1000)            # the job's records do not fit
    z.close(); z2.close()


def test_what_is_not_bgzf_is_refused_and_corruption_is_an_error(tmp_path):
    from rkmh_amd import api, synth
    rng = np.random.default_rng(6)
    text = _fastq(rng, 500)
    plain = tmp_path / "a.fq"
    plain.write_bytes(text)
    gz = tmp_path / "a.fq.gz"
    gz.write_bytes(gzip.compress(text))
    assert api.Bgzf.open(str(plain)) is None and api.Bgzf.open(str(gz)) is None
    img = bytearray(synth.bgzf_compress(text, level=6, block=20000))
    ok = tmp_path / "ok.gz"
    ok.write_bytes(bytes(img))
    z = api.Bgzf.open(str(ok))
    dst = C.create_string_buffer(len(text) + 64)
    assert z.fastq_records(0, z.members, dst, len(text) + 64)[1] == len(text)
    z.close()
    img[len(img) // 2] ^= 0x55                                 # a flipped byte inside some member's deflate stream
    bad = tmp_path / "bad.gz"
    bad.write_bytes(bytes(img))
    zb = api.Bgzf.open(str(bad))
    if zb is not None:
        with pytest.raises(api.RkmhError):
            zb.fastq_records(0, zb.members, dst, len(text) + 64)
        zb.close()
    # FASTA text in BGZF: opens, but the front end's '@' check says no
    fa = tmp_path / "r.fa.gz"
    fa.write_bytes(synth.bgzf_compress(b">x\nACGT\n" * 100))
    zf = api.Bgzf.open(str(fa))
    assert zf.first_byte() == ord(">") and zf.fastq_records(0, zf.members, dst, len(text) + 64)[0] == 1
    zf.close()


def test_empty_member_inside_the_file_does_not_hide_the_previous_byte(tmp_path):
    """`cat a.fq.gz b.fq.gz`: the end-of-file marker of a.fq.gz is an EMPTY member in the middle of the file.  A job that begins
    right behind it must still see the last byte of the text in front (rk_bgzf_lead_member) -- here that text ends INSIDE a line
    whose continuation begins with '@', which a cutter that takes the empty member for a line start would call a record."""
    from rkmh_amd import api, synth
    rng = np.random.default_rng(8)
    text = _fastq(rng, 400, lo=60, hi=90, tricky=False)
    # cut inside a quality line, so that the second file begins with "@..." in mid-line, two lines above a '+' line
    recs = text.split(b"\n")
    lines = [recs[i] for i in range(len(recs))]
    k = 4 * 200 + 3                                            # a quality line
    q = bytearray(lines[k]); q[10] = ord("@"); lines[k] = bytes(q)
    text = b"\n".join(lines)
    at = sum(len(x) + 1 for x in lines[:k]) + 10               # the '@' inside the quality line
    a, b = text[:at], text[at:]
    img = synth.bgzf_compress(a, level=1, block=5000) + synth.bgzf_compress(b, level=1, block=5000)
    path = tmp_path / "cat.fq.gz"
    path.write_bytes(img)
    z = api.Bgzf.open(str(path))
    assert z is not None and z.text_bytes == len(text)
    cap = len(text) + 64
    dst = C.create_string_buffer(cap)
    na = (len(a) + 4999) // 5000                               # members of a's text; member na is the empty marker
    assert api.load_library().rk_bgzf_lead_member(z._h, na + 1) == na - 1
    for first in ([0, na + 1, z.members], [0, na, z.members], [0, na - 1, na + 1, na + 2, z.members]):
        got = b""
        for b0, b1 in zip(first, first[1:]):
            st, n, off = z.fastq_records(b0, b1, dst, cap)
            assert st == 0 and (n == 0 or off == len(got)), (first, b0, b1, off, len(got))
            got += dst.raw[:n]
        assert got == text, first
    z.close()


def test_corrupt_gzip_is_an_error_not_an_end_of_file(tmp_path):
    """A single-member .gz whose deflate stream is damaged (or cut off): gzread fails on the inflater thread; rk_parse_files and
    rk_reader_next report RK_ERR_IO instead of ending as if at a clean end of input."""
    from rkmh_amd import api
    rng = np.random.default_rng(9)
    text = _fastq(rng, 20000, tricky=False)
    img = bytearray(gzip.compress(text, 6))
    good = tmp_path / "good.fq.gz"
    good.write_bytes(bytes(img))
    assert api.parse_files([str(good)])["nseq"] == 20000
    img[len(img) // 2] ^= 0x5a
    img[len(img) // 2 + 1] ^= 0xff
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(img))
    with pytest.raises(api.RkmhError) as e:
        api.parse_files([str(bad)])
    assert "read failed" in str(e.value)
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(bytes(gzip.compress(text, 6))[: len(img) // 3])
    with pytest.raises(api.RkmhError):
        api.parse_files([str(cut)])
