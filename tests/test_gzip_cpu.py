"""rk_gzip_* on the host (no GPU): the gzip container is parsed (RFC 1952: FEXTRA / FNAME / FCOMMENT / FHCRC), the text's first byte
comes from zlib, ISIZE gives the sizing hint, rk_gzip_plan cuts the compressed bytes into stretches of about a slot of text."""
import gzip
import io
import os
import zlib

import numpy as np


def _text(n):
    rng = np.random.default_rng(1)
    return b"".join(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=100)) + b"\n+\n" + b"I" * 100 + b"\n" for i in range(n))


def test_gzip_open_header_fields_first_byte_and_plan(tmp_path):
    from rkmh_amd import api
    text = _text(20000)
    plain = tmp_path / "a.fq.gz"
    plain.write_bytes(gzip.compress(text, 6))
    named = tmp_path / "b.fq.gz"
    with open(named, "wb") as f:
        with gzip.GzipFile(filename="some_name.fq", mode="wb", fileobj=f, compresslevel=1) as g:
            g.write(text)
    # every optional header field at once: FEXTRA, FNAME, FCOMMENT, FHCRC
    raw = zlib.compress(text, 6)[2:-4]
    hdr = b"\x1f\x8b\x08" + bytes([2 | 4 | 8 | 16]) + b"\0\0\0\0\0\x03" + (5).to_bytes(2, "little") + b"extra" + b"name\0" + b"a comment\0"
    hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
    full = tmp_path / "c.fq.gz"
    full.write_bytes(hdr + raw + (zlib.crc32(text) & 0xFFFFFFFF).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little"))
    assert gzip.decompress(full.read_bytes()) == text
    for p in (plain, named, full):
        gz = api.Gzip.open(str(p))
        assert gz is not None, p
        assert gz.first_byte() == ord("@")
        assert gz.text_bytes_hint == len(text)
        one = gz.plan(64 << 20)
        assert one == 1
        os.environ["RKMH_GZIP_STRETCH_KB"] = "64"
        try:
            many = gz.plan(64 << 20)
        finally:
            del os.environ["RKMH_GZIP_STRETCH_KB"]
        comp = os.path.getsize(p)
        assert (comp - 64) // 65536 <= many <= comp // 65536 + 1
        gz.close()
    notgz = tmp_path / "d.fq"
    notgz.write_bytes(text[:5000])
    assert api.Gzip.open(str(notgz)) is None
    assert api.Gzip.open(str(tmp_path / "missing.gz")) is None
    fa = tmp_path / "e.fa.gz"
    fa.write_bytes(gzip.compress(b">seq\nACGT\n"))
    gz = api.Gzip.open(str(fa))
    assert gz is not None and gz.first_byte() == ord(">")
    gz.close()
