"""Packed reads, the host side (rk_pack.cpp, `rkmh pack`): 2 bits per base + an exception list + names (+ qualities).  The file `rkmh pack`
writes decodes back to the reads the kseq grammar yields -- names, bases (acgt folded to upper case, as mkmh's to_upper does before anything
hashes; every other byte kept exactly) and quality strings -- for FASTQ and FASTA, several blocks, N runs and lower case."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest


def _blocks(d):
    magic, ver, flags, nreads, nbases, nblocks, doff = struct.unpack_from("<8sIIQQQQ", d, 0)
    assert magic == b"RKPK1\n\0\0" and ver == 1
    out = []
    for b in range(nblocks):
        out.append(struct.unpack_from("<QQQQQQIIQQII", d, doff + 80 * b))
    return flags, nreads, nbases, out


def _decode(lib, d):
    flags, nreads, nbases, blocks = _blocks(d)
    recs = []
    for (o_off, b_off, e_off, no_off, n_off, q_off, nrec, nexc, nb, nameb, maxlen, _pad) in blocks:
        offs = np.frombuffer(d, dtype=np.uint32, count=nrec + 1, offset=o_off)
        noffs = np.frombuffer(d, dtype=np.uint32, count=nrec + 1, offset=no_off)
        assert offs[0] == 0 and offs[-1] == nb and noffs[-1] == nameb and maxlen == int((offs[1:] - offs[:-1]).max())
        bases2, exc = d[b_off: b_off + (nb + 3) // 4 + 16], d[e_off: e_off + 8 * nexc]
        for r in range(nrec):
            n = int(offs[r + 1] - offs[r])
            out = C.create_string_buffer(max(n, 1))
            lib.rk_packed_decode(bases2, int(offs[r]), n, exc, nexc, out)
            q = d[q_off + offs[r]: q_off + offs[r + 1]] if q_off else None
            recs.append((d[n_off + noffs[r]: n_off + noffs[r + 1]], out.raw[:n], q))
    assert len(recs) == nreads
    return flags, recs


def _fold(seq):
    return bytes(c & 0xDF if (c & 0xDF) in b"ACGT" else c for c in seq)


@pytest.mark.parametrize("name,quals", [("z1.fq.gz", True), ("minION25.fq.gz", True), ("hpv_16_allFasta.fa.gz", False)])
def test_pack_round_trips_bundled_files(root, data_dir, orc, tmp_path, name, quals):
    from rkmh_amd import api
    out = tmp_path / "r.rkp"
    r = subprocess.run([os.path.join(root, "bin", "rkmh"), "pack", "-f", os.path.join(data_dir, name), "-o", str(out), "--block-reads", "1024"], capture_output=True)
    assert r.returncode == 0, r.stderr
    want = orc.kseq_parse_file(os.path.join(data_dir, name))
    flags, got = _decode(api.load_library(), out.read_bytes())
    assert (flags & 1) == (1 if quals else 0) and len(got) == len(want)
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[1] == _fold(w[1])
        if quals:
            assert g[2] == w[2]


def test_pack_keeps_every_byte_that_is_not_acgt(root, tmp_path):
    from rkmh_amd import api
    rng = np.random.default_rng(4)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNnRYKMSWBDHV*-.", np.uint8)
    recs = []
    for i in range(5000):
        L = int(rng.integers(1, 700))
        s = bytes(rng.choice(alphabet, size=L))
        recs.append((b"r%d" % i, s, bytes(rng.integers(33, 127, size=L, dtype=np.uint8))))
    fq = tmp_path / "x.fq"
    fq.write_bytes(b"".join(b"@" + n + b" comment\n" + s + b"\n+\n" + q + b"\n" for n, s, q in recs))
    exe = os.path.join(root, "bin", "rkmh")
    for extra, quals in (([], True), (["--no-quals"], False)):
        out = tmp_path / "x.rkp"
        r = subprocess.run([exe, "pack", "-f", str(fq), "-o", str(out), "--block-reads", "1500"] + extra, capture_output=True)
        assert r.returncode == 0, r.stderr
        flags, got = _decode(api.load_library(), out.read_bytes())
        assert (flags & 1) == (1 if quals else 0) and len(got) == len(recs)
        for g, w in zip(got, recs):
            assert g[0] == w[0] and g[1] == _fold(w[1]) and (g[2] == w[2] if quals else g[2] is None)
    assert len(_blocks(out.read_bytes())[3]) >= 1
