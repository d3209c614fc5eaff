"""DEFLATE on the device (rk_inflate.hip) for BGZF members: for every job of a file, the text the GPU route leaves in the slot
(rk_fastq_slot_load_bgzf) must equal, byte for byte and offset for offset, what the host route inflates and cuts
(rk_bgzf_fastq_records) -- for stored, fixed and dynamic blocks (levels 0, 1, 6, 9), members from 200 bytes to 64 KB, long
self-overlapping matches, incompressible bytes, and jobs from one member to the whole file."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fastq(rng, n, qual="random"):
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 400))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[0.3, 0.2, 0.2, 0.29, 0.01]))
        if qual == "random":
            q = bytes(rng.integers(33, 75, size=L, dtype=np.uint8))
        elif qual == "flat":
            q = b"I" * L                              # long runs: matches that overlap their own output (distance 1)
        else:
            q = bytes(rng.integers(33, 127, size=L, dtype=np.uint8))
        recs.append(b"@read%d/%d comment\n" % (i, i % 7) + s + b"\n+\n" + q + b"\n")
    return b"".join(recs)


@pytest.fixture(scope="module")
def gctx(orc, data_dir):
    import rkmh_amd
    recs = orc.kseq_parse_file(os.path.join(data_dir, "hpv_16.fa.gz"))
    rb, ro = orc.pack([r[1] for r in recs])
    c = rkmh_amd.Context(0)
    c.set_references(np.concatenate([rb, np.zeros(16, np.uint8)]), ro, [16], 1000)
    yield c
    c.close()


@pytest.mark.parametrize("level,member,qual", [(1, 0xff00, "random"), (6, 0xff00, "random"), (9, 30000, "flat"), (0, 20000, "random"), (6, 200, "wide"),
                                               (1, 4096, "flat"), (6, 65280, "wide")])
def test_device_inflate_equals_host_inflate(gctx, tmp_path, level, member, qual):
    from rkmh_amd import api, synth
    rng = np.random.default_rng(level * 100 + member % 97)
    text = _fastq(rng, 9000 if member > 1000 else 1500, qual)
    path = tmp_path / "t.fq.gz"
    path.write_bytes(synth.bgzf_compress(text, level=level, block=member))
    z = api.Bgzf.open(str(path))
    assert z is not None and z.text_bytes == len(text)
    cap = 1 << 20
    slot = api.FastqSlot(gctx, max_bytes=cap)
    host = C.create_string_buffer(cap + 64)
    try:
        for target in (1, 150000, 600000):
            first = z.plan(target)
            if member == 200 and target == 1:
                first = first[:80]      # (a prefix of the one-member jobs is enough)
            got_all = b""
            for b0, b1 in zip(first, first[1:]):
                st, n, off = z.fastq_records(b0, b1, host, cap)
                assert st == 0
                dst, dn, doff = slot.load_bgzf(z, b0, b1)
                if member == 200 and dst == 1:          # records longer than the two members of lookahead: this job is the host's
                    got_all += host.raw[:n]
                    continue
                assert dst == 0, (level, member, b0, b1)
                assert (dn, doff) == (n, off) or n == 0, (b0, b1, dn, n, doff, off)
                res = slot.classify_raw(dn)       # (waits for the text's way back to the host; the device front end accepts it)
                assert res.status == 0
                assert bytes(slot.text_buffer()[:dn]) == host.raw[:n], (level, member, b0, b1)
                got_all += host.raw[:n]
            if len(first) == len(z.plan(target)):
                assert got_all == text
    finally:
        slot.destroy()
        z.close()


def _skewed_fastq(rng, n, alphabet):
    """records whose names and qualities draw from `alphabet` with geometric weights: the rare symbols get Huffman codes far longer
    than the decoder's root tables (9 bits literal/length, 7 bits distance)"""
    a = np.frombuffer(alphabet, np.uint8)
    w = 0.82 ** np.arange(len(a)); w /= w.sum()
    recs = []
    for i in range(n):
        L = int(rng.integers(20, 600))
        s = bytes(rng.choice(np.frombuffer(b"ACGTNacgtn", np.uint8), size=L, p=[0.28, 0.2, 0.2, 0.28, 0.01, 0.01, 0.005, 0.005, 0.005, 0.005]))
        q = bytes(rng.choice(a, size=L, p=w))
        nm = bytes(rng.choice(a, size=int(rng.integers(1, 60)), p=w)).replace(b"@", b"a")
        recs.append(b"@" + nm + b"\n" + s + b"\n+" + (nm if i % 5 == 0 else b"") + b"\n" + q + b"\n")
    return b"".join(recs)


@pytest.mark.parametrize("level,strategy,mem_level,member", [(6, 0, 8, 0xff00), (9, 0, 1, 0xff00), (6, 4, 8, 0xff00), (6, 3, 8, 50000), (6, 2, 8, 0xff00), (6, 1, 2, 30000),
                                                              (1, 0, 1, 0xff00), (9, 0, 9, 7000)])
def test_device_inflate_long_codes_many_blocks_every_strategy(gctx, tmp_path, level, strategy, mem_level, member):
    """zlib's strategies (0 default, 1 filtered, 2 Huffman only, 3 RLE: distance 1, 4 fixed codes) and memory levels (1: a deflate block
    every ~128 symbols' worth of buffer -- hundreds of block headers and code tables per member) on text with a skewed 90-symbol
    alphabet (code lengths up to 15): the device's text equals the host inflater's for every job."""
    import zlib
    from rkmh_amd import api, synth
    rng = np.random.default_rng(1000 * level + 10 * strategy + mem_level)
    alphabet = bytes(c for c in range(33, 127) if c != ord("@"))
    text = _skewed_fastq(rng, 2500, alphabet)
    path = tmp_path / "s.fq.gz"
    path.write_bytes(synth.bgzf_compress(text, level=level, block=member, strategy=strategy, mem_level=mem_level))
    z = api.Bgzf.open(str(path))
    assert z is not None and z.text_bytes == len(text)
    cap = 1 << 21
    slot = api.FastqSlot(gctx, max_bytes=cap)
    host = C.create_string_buffer(cap + 64)
    try:
        for target in (200000, 1500000):
            first = z.plan(target)
            got_all = b""
            on_device = 0
            for b0, b1 in zip(first, first[1:]):
                st, n, off = z.fastq_records(b0, b1, host, cap)
                assert st == 0
                dst, dn, doff = slot.load_bgzf(z, b0, b1)
                if dst == 0:
                    on_device += 1
                    assert (dn, doff) == (n, off) or n == 0, (b0, b1, dn, n, doff, off)
                    res = slot.classify_raw(dn)
                    assert bytes(slot.text_buffer()[:dn]) == host.raw[:n], (level, strategy, mem_level, b0, b1)
                got_all += host.raw[:n]
            assert got_all == text
            assert on_device >= len(first) - 2, (on_device, len(first))      # (the device inflated them: no silent hand-over of whole files)
    finally:
        slot.destroy()
        z.close()


def test_device_inflate_refuses_damaged_members(gctx, tmp_path):
    from rkmh_amd import api, synth
    rng = np.random.default_rng(3)
    text = _fastq(rng, 3000)
    img = bytearray(synth.bgzf_compress(text, level=6, block=40000))
    good = tmp_path / "good.gz"
    good.write_bytes(bytes(img))
    z = api.Bgzf.open(str(good))
    slot = api.FastqSlot(gctx, max_bytes=1 << 21)
    try:
        assert slot.load_bgzf(z, 0, z.members)[0] == 0
        slot.classify_raw(slot.load_bgzf(z, 0, z.members)[1])
        z.close()
        refused = 0
        for trial in range(12):
            bad = bytearray(img)
            pos = int(rng.integers(40, len(bad) - 60))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            p = tmp_path / ("bad%d.gz" % trial)
            p.write_bytes(bytes(bad))
            zb = api.Bgzf.open(str(p))
            if zb is None:
                continue
            st, n, _ = slot.load_bgzf(zb, 0, zb.members)
            if st == 0:                                 # accepted: then the text is the original's (k_crc32_members checked every member;
                slot.classify_raw(n)                    # a flip in a header field nobody reads leaves the members intact)
                assert bytes(slot.text_buffer()[:n]) == text, trial
            else:
                refused += 1
            zb.close()
        assert refused >= 8
    finally:
        slot.destroy()


def test_device_text_slot_brings_back_names_and_filtered_records_only(gctx, tmp_path):
    """RK_SLOT_DEVICE_TEXT: the job's text stays in HBM; finish() returns the rows and the NAMES packed on the device (stream's
    lines are written from them) or, after set_filter_output, the records filter prints -- byte-identical to what an ordinary slot,
    which brings the whole text back, gives the same formatters.  Jobs of one member to the whole file; a pack larger than the
    slot's page-locked buffer (filter that keeps everything)."""
    from rkmh_amd import api, synth
    rng = np.random.default_rng(11)
    text = _fastq(rng, 7000)
    path = tmp_path / "d.fq.gz"
    path.write_bytes(synth.bgzf_compress(text, level=1, block=0xff00))
    z = api.Bgzf.open(str(path))
    cap = 4 << 20
    plain, dev, devf = api.FastqSlot(gctx, max_bytes=cap), api.FastqSlot(gctx, max_bytes=cap, device_text=True), api.FastqSlot(gctx, max_bytes=cap, device_text=True)
    parts = api.LineParts([b"ref%d" % i for i in range(int(gctx._lib.rk_num_references(gctx._h)))], 1000, 2, 1)
    try:
        for mm, md in ((-1, -100), (3, 0)):         # (-1, -100): every read passes -- the packed records outgrow max_bytes / 8
            devf.set_filter_output(mm, md)
            for target in (1, 300000, 1 << 30):
                first = z.plan(target)[:40]
                for b0, b1 in zip(first, first[1:]):
                    st, n, off = plain.load_bgzf(z, b0, b1)
                    assert st == 0
                    res = plain.classify_raw(n)
                    want_lines, want_recs = (plain.stream_lines(parts, res), plain.filter_records(res, mm, md)) if res.nrec else (b"", b"")
                    for slot in (dev, devf):
                        st2, n2, off2 = slot.load_bgzf(z, b0, b1)
                        assert (st2, n2, off2) == (0, n, off)
                        r2 = slot.classify_raw(n2)
                        assert r2.status == 0 and r2.nrec == res.nrec
                        if r2.nrec == 0:
                            continue
                        if slot is dev:
                            assert slot.stream_lines(parts, r2) == want_lines, (target, b0, b1)
                        else:
                            assert slot.filter_records(r2, mm, md) == want_recs, (mm, md, target, b0, b1)
            assert len(want_recs) > 0 if mm < 0 else True
    finally:
        for sl in (plain, dev, devf):
            sl.destroy()
        z.close()
