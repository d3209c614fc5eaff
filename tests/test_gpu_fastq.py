"""The FASTQ front end on the device (rkmh_amd/csrc/rk_fastq.hip, rk_fastq_slot_* in include/rkmh_amd.h) against the literal kseq
grammar of the oracle (oracle.kseq_parse_bytes <- /root/reference/src/kseq.hpp:170-208): text that is strictly four lines per record
must come out record for record as kseq reads it (names, sequences) and classify to the oracle's rows; ANY other text must either
be reported irregular (status != 0: the caller's kseq-grammar scanner takes the block) or still come out exactly as kseq reads
it -- the device never guesses."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(ctx, orc, data_dir):
    import rkmh_amd
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    rb, ro = refs["bases"], refs["offsets"]
    ctx.set_references(rb, ro, [16], 1000)
    sk, ln = ctx.get_reference_sketches()
    slot = api.FastqSlot(ctx, max_bytes=1 << 24)
    yield ctx, slot, rb, ro, sk, ln
    slot.destroy()


def _fastq(reads, names=None, plus=None, quals=None):
    out = []
    for i, r in enumerate(reads):
        nm = names[i] if names else b"r%d" % i
        q = quals[i] if quals else b"I" * len(r)
        out.append(b"@" + nm + b"\n" + r + b"\n+" + (plus[i] if plus else b"") + b"\n" + q + b"\n")
    return b"".join(out)


def _check_against_kseq(orc, slot, text, sk, ln, must_be_regular=None):
    st, rows, names, seqs = slot.classify(text)
    want = orc.kseq_parse_bytes(text)
    if must_be_regular is True:
        assert st == 0, st
    if must_be_regular is False:
        assert st != 0
    if st != 0:
        return st
    assert len(names) == len(want), (len(names), len(want))
    for i, (nm, sq, q) in enumerate(want):
        assert names[i] == nm and seqs[i] == sq and slot.last_quals[i] == q, i
    if want:
        qb, qo = orc.pack([orc.to_upper(w[1]) for w in want])
        exp = orc.classify_stream(qb, qo, [16], 1000, sk, ln, threads=4)
        assert (rows == exp).all()
    return 0


def test_regular_text_equals_kseq_and_the_oracle_rows(setup, orc):
    from rkmh_amd import synth
    ctx, slot, rb, ro, sk, ln = setup
    qb, qo = synth.generate_reads_fast(rb, ro, 0, 20000, read_len=150, threads=4)
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(20000)]
    rng = np.random.default_rng(1)
    # names with descriptions, '+' lines that repeat the name, quality strings that begin with '@' and '+', lower case, reads of unequal length
    names, plus, quals = [], [], []
    for i, r in enumerate(reads):
        if i % 7 == 0:
            reads[i] = r = r.lower()
        if i % 11 == 0:
            reads[i] = r = r[: int(rng.integers(17, 150))]
        if i % 13 == 0:
            reads[i] = r = r[:5]                       # shorter than k: no windows
        names.append(b"read_%d/1" % i + (b" desc tab\there" if i % 3 == 0 else b"") + (b"\tx" if i % 5 == 0 else b""))
        plus.append(names[-1] if i % 4 == 0 else b"")
        q = bytearray(b"I" * len(r))
        if i % 6 == 0:
            q[0] = ord("@")
        if i % 9 == 0:
            q[0] = ord("+")
        if len(q) > 3 and i % 10 == 0:
            q[3] = 127
        quals.append(bytes(q))
    text = _fastq(reads, names, plus, quals)
    assert _check_against_kseq(orc, slot, text, sk, ln, must_be_regular=True) == 0
    # a block cut by rk_fastq_cut holds whole records only and the remainder starts at a record
    from rkmh_amd import api
    cut = api.fastq_cut(text[: len(text) // 2])
    assert cut > 0 and text[cut:cut + 1] == b"@" and text[cut - 1:cut] == b"\n"
    assert _check_against_kseq(orc, slot, text[:cut], sk, ln, must_be_regular=True) == 0
    assert _check_against_kseq(orc, slot, text[cut:], sk, ln, must_be_regular=True) == 0
    # long reads (several thousand bases: rows come back through the general path) keep their place
    long_reads = [bytes(rb[int(ro[j]):int(ro[j]) + 3000 + 500 * j]) for j in range(4)]
    mixed = reads[:50] + long_reads + reads[50:100]
    assert _check_against_kseq(orc, slot, _fastq(mixed), sk, ln, must_be_regular=True) == 0
    assert slot.classify(b"")[0] == 0


def test_two_slots_in_flight(setup, orc):
    """rk_fastq_slot_submit / _finish: two slots of one context with their blocks in flight together (what each worker of
    bin/rkmh stream does), finished in either order; finish without submit is an error, not a crash."""
    import rkmh_amd
    from rkmh_amd import api, synth
    ctx, slot, rb, ro, sk, ln = setup
    other = api.FastqSlot(ctx, max_bytes=1 << 22)
    try:
        qb, qo = synth.generate_reads_fast(rb, ro, 100000, 104000, read_len=150, threads=4)
        reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(4000)]
        t1, t2 = _fastq(reads[:2500], names=[b"a%d" % i for i in range(2500)]), _fastq(reads[2500:], names=[b"b%d" % i for i in range(1500)])
        want = {}
        for key, t in (("t1", t1), ("t2", t2)):
            st, rows, names, seqs = slot.classify(t)
            assert st == 0
            want[key] = (rows, names)
        for order in ((slot, other), (other, slot)):
            order[0].submit(t1)
            order[1].submit(t2)
            st2, rows2, names2, _ = order[1].finish()
            st1, rows1, names1, _ = order[0].finish()
            assert st1 == 0 and st2 == 0
            assert (rows1 == want["t1"][0]).all() and names1 == want["t1"][1]
            assert (rows2 == want["t2"][0]).all() and names2 == want["t2"][1]
        with pytest.raises(rkmh_amd.RkmhError):
            other.finish()
        other.submit(b"@x\nAC\nGT\n+\nIIII\n")   # irregular block through the split form
        assert other.finish()[0] != 0
    finally:
        other.destroy()


def test_irregular_text_is_refused_or_exact(setup, orc):
    ctx, slot, rb, ro, sk, ln = setup
    a, b = bytes(rb[100:250]), bytes(rb[1000:1150])
    ok = _fastq([a, b])
    cases = {
        "crlf": ok.replace(b"\n", b"\r\n"),
        "multi-line sequence": b"@r0\n" + a[:75] + b"\n" + a[75:] + b"\n+\n" + b"I" * 150 + b"\n",
        "blank line between records": _fastq([a]) + b"\n" + _fastq([b]),
        "leading blank line": b"\n" + ok,
        "no final newline": ok[:-1],
        "at-sign inside a sequence": _fastq([a[:70] + b"@" + a[71:], b]),
        "plus inside a sequence": _fastq([a[:70] + b"+" + a[71:], b]),
        "gt inside a sequence": _fastq([a[:70] + b">" + a[71:], b]),
        "space inside a sequence": _fastq([a[:70] + b" " + a[71:], b]),
        "short quality": b"@r0\n" + a + b"\n+\n" + b"I" * 149 + b"\n" + _fastq([b]),
        "long quality": b"@r0\n" + a + b"\n+\n" + b"I" * 151 + b"\n" + _fastq([b]),
        "quality byte 128": b"@r0\n" + a + b"\n+\n" + b"I" * 149 + b"\x80\n",
        "quality with a space": b"@r0\n" + a + b"\n+\n" + b"I" * 149 + b" \n",
        "empty sequence": b"@r0\n\n+\n\n" + _fastq([b]),
        "fasta": b">r0\n" + a + b"\n>r1\n" + b + b"\n",
        "missing plus line": b"@r0\n" + a + b"\n" + b"I" * 150 + b"\n" + _fastq([b]),
        "three lines": b"@r0\n" + a + b"\n+\n",
        "header only": b"@r0\n",
        "garbage": bytes(range(1, 255)) * 40,
        "only newlines": b"\n" * 1000,
        "very short records": b"@a\nA\n+\nI\n" * 3000,     # more lines than a block of that size is sized for -> refused, or exact
    }
    refused = 0
    for name, text in cases.items():
        st = _check_against_kseq(orc, slot, text, sk, ln)
        refused += st != 0
        if name in ("crlf", "multi-line sequence", "blank line between records", "no final newline", "at-sign inside a sequence",
                    "short quality", "long quality", "missing plus line", "three lines", "fasta", "empty sequence"):
            assert st != 0, name
    assert refused >= 11
    # and the slot still works
    assert _check_against_kseq(orc, slot, ok, sk, ln, must_be_regular=True) == 0


def test_mutated_text_never_parses_differently_from_kseq(setup, orc):
    """Fuzz: random byte edits of regular FASTQ text (RKMH_TEST_FUZZ raises the count).  Whatever the edit, the device either
    refuses the block or returns exactly the records kseq reads."""
    ctx, slot, rb, ro, sk, ln = setup
    rng = np.random.default_rng(int(os.environ.get("RKMH_TEST_SEED_BASE", "7")))
    reads = [bytes(rb[s:s + int(rng.integers(20, 160))]) for s in rng.integers(0, 100000, size=40)]
    base = _fastq(reads, names=[b"n%d d" % i for i in range(40)])
    pool = b"\n\n\n\r@+> \tACGTNacgtI!~\x7f\x80\x00"
    n = int(os.environ.get("RKMH_TEST_FUZZ", "400"))
    accepted = 0
    for it in range(n):
        t = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            op = int(rng.integers(0, 3))
            pos = int(rng.integers(0, len(t)))
            if op == 0:
                t[pos] = pool[int(rng.integers(0, len(pool)))]
            elif op == 1:
                del t[pos]
            else:
                t.insert(pos, pool[int(rng.integers(0, len(pool)))])
        accepted += _check_against_kseq(orc, slot, bytes(t), sk, ln) == 0
    assert 0 < accepted < n      # both outcomes occur: harmless edits are accepted, structural ones refused


def _cli(root, args, env=None, stdin=None):
    r = subprocess.run([os.path.join(root, "bin", "rkmh")] + args, capture_output=True, env=dict(os.environ, **(env or {})), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


def test_cli_device_front_end_equals_the_scanner(root, data_dir, tmp_path, orc):
    """bin/rkmh stream on uncompressed FASTQ files: the device front end (default) and the kseq-grammar scanner (RKMH_RAW=0) print
    the same bytes -- one block, many small blocks (cuts at record starts, a worker pool far larger than the block count and far
    smaller), a last line without newline, several -f files, two contexts on one GPU; and a file that stops being four lines per
    record half way through (multi-line sequences, then CRLF) is handed to the scanner at that block, the output still the same."""
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    rb, ro = refs["bases"], refs["offsets"]
    n = 30000
    qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=150, threads=4)
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(n)]
    rng = np.random.default_rng(3)
    for i in range(0, n, 17):
        reads[i] = reads[i][: int(rng.integers(16, 150))]
    text = _fastq(reads, names=[b"read%07d some description" % i for i in range(n)])
    fq = tmp_path / "reads.fq"
    fq.write_bytes(text)
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    base = ["stream", "-r", ref, "-k", "16", "-s", "1000"]
    want = _cli(root, base + ["-f", str(fq)], env={"RKMH_RAW": "0"})
    assert want.count(b"\n") == n
    for env in ({}, {"RKMH_RAW_BLOCK_KB": "64"}, {"RKMH_RAW_BLOCK_KB": "256", "RKMH_RAW_WORKERS": "2"}, {"RKMH_RAW_BLOCK_KB": "40", "RKMH_RAW_WORKERS": "24"}):
        err = subprocess.run([os.path.join(root, "bin", "rkmh")] + base + ["-f", str(fq)], capture_output=True, env=dict(os.environ, RKMH_TIMING="1", **env))
        assert err.stdout == want, env
        assert b"device front end: " in err.stderr and b" %d records" % n in err.stderr, err.stderr[-600:]   # the device really parsed all of it
    # no newline after the last quality string
    fq2 = tmp_path / "nonl.fq"
    fq2.write_bytes(text[:-1])
    assert _cli(root, base + ["-f", str(fq2)], env={"RKMH_RAW_BLOCK_KB": "512"}) == want
    # several files, several contexts
    got = _cli(root, base + ["-f", str(fq), "-f", str(fq2), "--devices", "0,0"], env={"RKMH_RAW_BLOCK_KB": "300"})
    assert got == want + want
    # a gzip file beside a plain one: the scanner takes the first, the device the second
    import gzip
    gz = tmp_path / "reads.fq.gz"
    gz.write_bytes(gzip.compress(text, 1))
    assert _cli(root, base + ["-f", str(gz), "-f", str(fq)]) == want + want
    # irregular from the middle on: records with their sequence on two lines, then CRLF line ends
    half = _fastq(reads[: n // 2], names=[b"read%07d some description" % i for i in range(n // 2)])
    odd = b"".join(b"@m%d\n" % i + r[:70] + b"\n" + r[70:] + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads[n // 2: n // 2 + 500]))
    crlf = _fastq(reads[n // 2 + 500: n // 2 + 900]).replace(b"\n", b"\r\n")
    tail = _fastq(reads[n // 2 + 900: n // 2 + 2000], names=[b"t%d" % i for i in range(1100)])
    mixed = tmp_path / "mixed.fq"
    mixed.write_bytes(half + odd + crlf + tail)
    want_m = _cli(root, base + ["-f", str(mixed)], env={"RKMH_RAW": "0"})
    assert want_m.count(b"\n") == n // 2 + 500 + 400 + 1100
    for env in ({}, {"RKMH_RAW_BLOCK_KB": "128"}, {"RKMH_RAW_BLOCK_KB": "64", "RKMH_RAW_WORKERS": "3"}):
        r = subprocess.run([os.path.join(root, "bin", "rkmh")] + base + ["-f", str(mixed)], capture_output=True, env=dict(os.environ, RKMH_TIMING="1", **env))
        assert r.returncode == 0 and r.stdout == want_m, env
        assert b"the scanner reads on from there" in r.stderr
    # standard output a regular FILE: the workers write their blocks side by side at their final offsets (no writer thread);
    # what the scanner writes after a hand-over continues behind them
    for src, want_x, env in ((fq, want, {"RKMH_RAW_BLOCK_KB": "96"}), (mixed, want_m, {"RKMH_RAW_BLOCK_KB": "128"}), (fq, want, {})):
        outp = tmp_path / "out.tsv"
        with open(outp, "wb") as f:
            f.write(b"# header written before the run\n")
            f.flush()
            r = subprocess.run([os.path.join(root, "bin", "rkmh")] + base + ["-f", str(src), "-f", str(fq2)], stdout=f, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        assert outp.read_bytes() == b"# header written before the run\n" + want_x + want
    # the oracle's kseq grammar agrees with both on the mixed file (names in order)
    names = [x[0] for x in orc.kseq_parse_bytes(half + odd + crlf + tail)]
    assert [l.split(b"\t")[1] for l in want_m.splitlines()] == names


def test_slot_count_equals_count_batch(setup, orc):
    """Pass 1 of -M on raw text (rk_fastq_slot_count) fills the depth table exactly as rk_count_batch does from the parsed reads
    -- short reads, reads longer than the fused kernel's limit (the tile hasher takes the block), several blocks into one table --
    and an irregular block leaves the table untouched."""
    import torch
    from rkmh_amd import api, synth
    ctx, slot, rb, ro, sk, ln = setup
    slots = 1 << 20
    rng = np.random.default_rng(11)
    qb, qo = synth.generate_reads_fast(rb, ro, 5, 6000, read_len=150, threads=4)
    nr = len(qo) - 1
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(nr)]
    for i in range(0, nr, 13):
        reads[i] = reads[i][: int(rng.integers(1, 150))]
    long_reads = [bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(n), p=[.24, .24, .24, .24, .04])) for n in (5000, 150, 70000, 16)]
    for name, blocks in (("short", [reads[:3000], reads[3000:]]), ("with long reads", [reads[:500] + long_reads, reads[500:900]])):
        ta_, tb_ = (torch.zeros(slots, dtype=torch.int32, device="cuda") for _ in range(2))
        torch.cuda.synchronize()
        a, b = api.Counter(ctx, slots, device_ptr=ta_.data_ptr()), api.Counter(ctx, slots, device_ptr=tb_.data_ptr())
        snap = lambda k, t: (k.get(0), t.cpu().numpy())[1]     # (rk_counter_get settles the passes into the table first)
        every = []
        for blk in blocks:
            st, nrec = slot.count(_fastq(blk), a)
            assert (st, nrec) == (0, len(blk)), name
            every += blk
        pb, po = orc.pack([orc.to_upper(r) for r in every])
        ctx.count_batch(pb, po, b)
        ta, tb = snap(a, ta_), snap(b, tb_)
        assert ta.sum() > 0 and (ta == tb).all(), name
        # an irregular block: reported, nothing counted
        st, nrec = slot.count(_fastq(reads[:50]).replace(b"\n", b"\r\n"), a)
        assert st != 0 and (snap(a, ta_) == ta).all()


def test_cli_minus_M_and_filter_through_the_device_front_end(root, data_dir, tmp_path):
    """bin/rkmh stream -M and bin/rkmh filter (with and without -M) on uncompressed FASTQ files: both passes through the device front
    end print the same bytes as the parse-everything path (RKMH_RAW=0) -- many small blocks, two files, two contexts; a file that is
    not four lines per record makes -M fall back as a whole (pass 1 cleared) and plain filter hand over at the block."""
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    rb, ro = refs["bases"], refs["offsets"]
    qb, qo = synth.generate_reads_fast(rb, ro, 1, 20000, read_len=150, threads=4)
    n = len(qo) - 1
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(n)]
    rng = np.random.default_rng(5)
    for i in range(0, n, 19):
        reads[i] = reads[i][: int(rng.integers(16, 150))]
    for i in range(0, n, 23):
        reads[i] = reads[i].lower()           # filter prints them upper-cased (rkmh.cpp:280)
    quals = [bytes(rng.integers(33, 74, size=len(r)).astype(np.uint8)) for r in reads]
    text = _fastq(reads, names=[b"read%07d extra words" % i for i in range(n)], quals=quals)
    fq = tmp_path / "reads.fq"
    fq.write_bytes(text)
    fq2 = tmp_path / "second.fq"
    fq2.write_bytes(_fastq(reads[:5000], names=[b"again%d" % i for i in range(5000)], quals=quals[:5000])[:-1])
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    opts = ["-r", ref, "-k", "16", "-s", "1000", "-f", str(fq), "-f", str(fq2)]
    exe = os.path.join(root, "bin", "rkmh")
    cases = (("stream", ["-M", "2"]), ("stream", ["-M", "3", "-N", "4"]), ("filter", ["-N", "6"]), ("filter", ["-M", "2", "-N", "6"]),
             ("filter", ["-M", "2", "-N", "2", "-D", "1"]))
    for cmd, extra in cases:
        want = _cli(root, [cmd] + opts + extra, env={"RKMH_RAW": "0"})
        assert len(want) > 1000, (cmd, extra)
        for env, more in (({}, []), ({"RKMH_RAW_BLOCK_KB": "96", "RKMH_RAW_WORKERS": "5"}, []), ({"RKMH_RAW_BLOCK_KB": "200"}, ["--devices", "0,0"])):
            r = subprocess.run([exe, cmd] + opts + extra + more, capture_output=True, env=dict(os.environ, RKMH_TIMING="1", **env))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            assert r.stdout == want, (cmd, extra, env)
            assert b"device front end: " in r.stderr, r.stderr[-600:]
            if "-M" in extra:
                assert b" %d records" % (2 * (n + 5000)) in r.stderr, r.stderr[-600:]   # both passes on the device
    # filter's records really are >name / SEQ / + / QUAL of reads in the file
    got = _cli(root, ["filter"] + opts + ["-N", "6"]).split(b"\n")
    assert got[0].startswith(b">read") and got[2] == b"+" and got[1] == got[1].upper() and len(got[3]) == len(got[1])
    # irregular text half way through the first file
    odd = b"".join(b"@m%d\n" % i + r[:70] + b"\n" + r[70:] + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads[:300]) if len(r) > 80)
    mixed = tmp_path / "mixed.fq"
    mixed.write_bytes(text[: text.index(b"@read0012000")] + odd + _fastq(reads[100:2000]))
    optm = ["-r", ref, "-k", "16", "-s", "1000", "-f", str(mixed), "-f", str(fq2)]
    for cmd, extra, note in (("stream", ["-M", "2"], b"the host scanner reads the run"), ("filter", ["-M", "2", "-N", "6"], b"the host scanner reads the run"),
                             ("filter", ["-N", "6"], b"the scanner reads on from there")):
        want = _cli(root, [cmd] + optm + extra, env={"RKMH_RAW": "0"})
        r = subprocess.run([exe, cmd] + optm + extra, capture_output=True, env=dict(os.environ, RKMH_TIMING="1", RKMH_RAW_BLOCK_KB="128"))
        assert r.returncode == 0 and r.stdout == want, (cmd, extra)
        assert note in r.stderr, r.stderr[-800:]
    # filter -i: the depth table of the file pass serves the STDIN pass, whichever front end filled it
    stdin = _fastq(reads[:200])
    a = subprocess.run([exe, "filter"] + opts + ["-M", "2", "-N", "6", "-i"], input=stdin, capture_output=True, env=dict(os.environ, RKMH_RAW="0"))
    b = subprocess.run([exe, "filter"] + opts + ["-M", "2", "-N", "6", "-i"], input=stdin, capture_output=True)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and b"Sample: " in b.stdout


def test_ranks_read_their_own_byte_range(root, data_dir, tmp_path):
    """python -m rkmh_amd.cli under torch.distributed.run (two and three ranks on GPU 0, gloo): every rank parses only ITS byte
    range of an uncompressed FASTQ file (rk_reader_open_range) and formats its own lines; rank 0 prints the gathered text, which
    equals the single-process binary's -- for stream, stream -M 2, filter, two -f files, and for input that cannot be split by
    bytes (gzip; text that is not four lines per record), where every rank falls back to parsing everything."""
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    rb, ro = refs["bases"], refs["offsets"]
    n = 6000
    qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=150, threads=4)
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(n)]
    rng = np.random.default_rng(5)
    for i in range(0, n, 9):
        reads[i] = reads[i][: int(rng.integers(16, 150))]
    text = _fastq(reads, names=[b"q%d x" % i for i in range(n)], quals=[bytes(rng.integers(33, 74, size=len(r), dtype=np.uint8)) for r in reads])
    fq = tmp_path / "r.fq"
    fq.write_bytes(text)
    import gzip
    gz = tmp_path / "r.fq.gz"
    gz.write_bytes(gzip.compress(text, 1))
    odd = tmp_path / "odd.fq"      # the second half has its sequences on two lines
    odd.write_bytes(_fastq(reads[:3000]) + b"".join(b"@m%d\n" % i + r[:40] + b"\n" + r[40:] + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads[3000:])))
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    exe = os.path.join(root, "bin", "rkmh")
    port = [29600]

    def ranks(world, args, extra_env=None):
        port[0] += 1
        env = dict(os.environ, RKMH_ONE_DEVICE="1", RKMH_DIST_BACKEND="gloo", **(extra_env or {}))
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                            "--master-port", str(port[0]), "-m", "rkmh_amd.cli"] + args, capture_output=True, cwd=root, env=env, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        last_err[0] = r.stderr
        return r.stdout

    last_err = [b""]

    def one(args):
        r = subprocess.run([exe] + args, capture_output=True, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return r.stdout

    st = ["stream", "-r", ref, "-k", "16", "-s", "1000"]
    want = one(st + ["-f", str(fq)])
    assert want.count(b"\n") == n
    assert ranks(2, st + ["-f", str(fq)]) == want
    assert ranks(3, st + ["-f", str(fq), "-f", str(fq)]) == want + want
    assert ranks(2, st + ["-f", str(gz)]) == want                                   # cannot be split by bytes: whole parse on every rank
    assert ranks(2, st + ["-f", str(fq)], {"RKMH_CLI_WHOLE_PARSE": "1"}) == want
    assert ranks(3, st + ["-f", str(odd)]) == one(st + ["-f", str(odd)])            # not four lines per record: every rank notices and falls back
    assert ranks(2, st + ["-f", str(fq), "-M", "2"]) == one(st + ["-f", str(fq), "-M", "2"])
    fl = ["filter", "-r", ref, "-k", "16", "-s", "1000", "-N", "3"]
    wf = one(fl + ["-f", str(fq)])
    assert wf.startswith(b">q") and 0 < len(wf) < len(text) + n      # some reads pass, some do not
    assert ranks(2, fl + ["-f", str(fq)]) == wf
    # since round 4 the ranks do not parse at all: their byte ranges go through the device front end in raw blocks (cli._device_ingest);
    # the parsing path per byte range (RKMH_RAW=0) and the whole-parse path stay covered above and here
    assert ranks(2, st + ["-f", str(fq)], {"RKMH_TIMING": "1"}) == want
    assert last_err[0].count(b"device front end: ") == 2 and b"refused" not in last_err[0]
    assert ranks(2, st + ["-f", str(fq)], {"RKMH_RAW": "0", "RKMH_TIMING": "1"}) == want
    assert b"device front end: " not in last_err[0]
    assert ranks(3, st + ["-f", str(fq), "-f", str(fq)], {"RKMH_RAW_BLOCK_KB": "64", "RKMH_RAW_WORKERS": "3", "RKMH_TIMING": "1"}) == want + want
    assert b"device front end: " in last_err[0]
    assert ranks(2, fl + ["-f", str(fq), "-M", "2"], {"RKMH_RAW_BLOCK_KB": "100"}) == one(fl + ["-f", str(fq), "-M", "2"])
    assert ranks(2, st + ["-f", str(fq), "-M", "3", "-N", "4"], {"RKMH_RAW_BLOCK_KB": "300"}) == one(st + ["-f", str(fq), "-M", "3", "-N", "4"])
    assert ranks(3, st + ["-f", str(odd)], {"RKMH_TIMING": "1"}) == one(st + ["-f", str(odd)])
    assert b"refused" in last_err[0]
    # BGZF: every rank inflates ITS share of the members through the device front end (no rank parses, none inflates the whole file)
    bg = tmp_path / "r.bgzf.fq.gz"
    bg.write_bytes(synth.bgzf_compress(text, level=1, block=30000))
    assert ranks(2, st + ["-f", str(bg)], {"RKMH_TIMING": "1", "RKMH_RAW_BLOCK_KB": "256"}) == want
    assert last_err[0].count(b"device front end: ") == 2 and b"refused" not in last_err[0]
    assert ranks(3, st + ["-f", str(bg), "-f", str(fq), "-M", "2"], {"RKMH_RAW_BLOCK_KB": "128"}) == one(st + ["-f", str(fq), "-f", str(fq), "-M", "2"])
    assert ranks(2, fl + ["-f", str(bg)], {"RKMH_RAW_BLOCK_KB": "512"}) == wf
    assert ranks(2, st + ["-f", str(bg)], {"RKMH_BGZF": "0"}) == want                # left to zlib: whole parse on every rank
    assert ranks(2, st + ["-f", str(bg), "-M", "2"], {"RKMH_BGZF_DEVICE": "1", "RKMH_RAW_BLOCK_KB": "200"}) == one(st + ["-f", str(fq), "-M", "2"])   # inflated on the device (rkmh_amd.cli: opt-in, block-sized jobs)
    # rank 0 alone reads the references (here forced through the device: rk_fasta_load_*), the others get names and sketches
    plain_ref = tmp_path / "pave.fa"
    plain_ref.write_bytes(gzip.open(ref).read())
    st_plain = ["stream", "-r", str(plain_ref), "-k", "16", "-s", "1000"]
    assert ranks(2, st_plain + ["-f", str(fq)], {"RKMH_RAW_REFS": "1", "RKMH_TIMING": "1", "RKMH_RAW_BLOCK_KB": "256"}) == want
    assert last_err[0].count(b"references through the device: 182 sequences") == 1
    assert ranks(2, ["filter", "-r", str(plain_ref), "-k", "16", "-s", "1000", "-N", "3", "-I", "2", "-f", str(fq)], {"RKMH_RAW_REFS": "1"}) == one(fl + ["-I", "2", "-f", str(fq)])
    r1 = subprocess.run([sys.executable, "-m", "rkmh_amd.cli"] + st + ["-f", str(fq)], capture_output=True, cwd=root, env=dict(os.environ, RKMH_TIMING="1"), timeout=900)
    assert r1.returncode == 0 and r1.stdout == want and b"device front end: " in r1.stderr      # one process, no launcher
    # one process with -M (its blocks stream out in order while later ones are on the device), into a pipe and into a regular file
    # that already holds a line; two ranks into a regular file (every rank writes its blocks at their final offsets)
    wm = one(st + ["-f", str(fq), "-f", str(fq), "-M", "2"])
    r2 = subprocess.run([sys.executable, "-m", "rkmh_amd.cli"] + st + ["-f", str(fq), "-f", str(fq), "-M", "2"], capture_output=True, cwd=root,
                        env=dict(os.environ, RKMH_RAW_BLOCK_KB="128"), timeout=900)
    assert r2.returncode == 0 and r2.stdout == wm
    for world, args, expect in ((1, st + ["-f", str(fq), "-M", "2"], one(st + ["-f", str(fq), "-M", "2"])), (2, st + ["-f", str(fq), "-f", str(fq)], want + want),
                                (3, fl + ["-f", str(fq)], wf),
                                (1, st + ["-f", str(fq), "-f", str(odd)], want + one(st + ["-f", str(odd)]))):   # refused half way: cut back, parsed
        outp = tmp_path / "cli_out.txt"
        with open(outp, "wb") as f:
            f.write(b"# header\n")
            f.flush()
            port[0] += 1
            cmd = [sys.executable, "-m", "rkmh_amd.cli"] if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                                                                            "--master-addr", "127.0.0.1", "--master-port", str(port[0]), "-m", "rkmh_amd.cli"]
            r3 = subprocess.run(cmd + args, stdout=f, stderr=subprocess.PIPE, cwd=root,
                                env=dict(os.environ, RKMH_ONE_DEVICE="1", RKMH_DIST_BACKEND="gloo", RKMH_RAW_BLOCK_KB="200"), timeout=900)
            assert r3.returncode == 0, r3.stderr.decode()[-2000:]
            f.write(b"# trailer\n")
        assert outp.read_bytes() == b"# header\n" + expect + b"# trailer\n", (world, args)


def test_cli_bgzf_input_goes_through_the_device_front_end(root, data_dir, tmp_path, orc):
    """BGZF (bgzip) FASTQ: the front end's workers inflate the members of their jobs themselves (rk_bgzf_*) -- stdout is byte-identical
    to the plain-text run for stream, stream -M 2 (both passes), filter and filter -M 2, for member sizes from 300 bytes (records span
    many members) to the full 64 KB, jobs of 40 KB to 16 MB, a last line without newline, and a file that turns irregular half way
    (handed to the zlib scanner at that job, by its offset in the TEXT).  RKMH_BGZF=0 leaves the file to the scanner: same bytes."""
    import gzip
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    rb, ro = refs["bases"], refs["offsets"]
    n = 40000
    qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=150, threads=4)
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(n)]
    rng = np.random.default_rng(8)
    for i in range(0, n, 13):
        reads[i] = reads[i][: int(rng.integers(16, 150))]
    text = _fastq(reads, names=[b"read%07d comment" % i for i in range(n)])
    fq = tmp_path / "reads.fq"
    fq.write_bytes(text)
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    exe = os.path.join(root, "bin", "rkmh")
    for cmd, flags in (("stream", []), ("stream", ["-M", "2"]), ("filter", []), ("filter", ["-M", "2", "-N", "3"])):
        base = [cmd, "-r", ref, "-k", "16", "-s", "1000"] + flags
        want = _cli(root, base + ["-f", str(fq)])
        assert len(want) > 1000
        # (the default: the members inflated on the device, rk_inflate.hip, a third of the file per job, the text never on the host;
        # RKMH_BGZF_DEVICE=0: by the workers, libdeflate / zlib -- same bytes)
        for member, level, env in ((0xff00, 1, {"RKMH_BGZF_DEVICE": "0"}), (300, 6, {"RKMH_BGZF_DEVICE": "0", "RKMH_RAW_BLOCK_KB": "40"}),
                                   (20000, 6, {"RKMH_BGZF_DEVICE": "0", "RKMH_RAW_BLOCK_KB": "128", "RKMH_RAW_WORKERS": "3"}),
                                   (0xff00, 1, {}), (20000, 6, {"RKMH_BGZF_JOB_KB": "300"}),
                                   # (several output pieces per job, formatted by the helper threads; one device worker; many small jobs)
                                   (0xff00, 1, {"RKMH_BGZF_JOB_KB": "2048", "RKMH_BGZF_PIECES": "5"}),
                                   (20000, 6, {"RKMH_BGZF_JOB_KB": "200", "RKMH_BGZF_DEVICE_WORKERS": "1", "RKMH_BGZF_PIECES": "2"})):
            if cmd == "filter" and member == 300:
                continue
            gz = tmp_path / ("reads_%d.fq.gz" % member)
            if not gz.exists():
                gz.write_bytes(synth.bgzf_compress(text, level=level, block=member))
            r = subprocess.run([exe] + base + ["-f", str(gz)], capture_output=True, env=dict(os.environ, RKMH_TIMING="1", **env))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            assert r.stdout == want, (cmd, flags, member)
            assert b"device front end: " in r.stderr and (b" %d records" % (n * (2 if "-M" in flags else 1))) in r.stderr, r.stderr[-800:]
        assert _cli(root, base + ["-f", str(tmp_path / "reads_65280.fq.gz")], env={"RKMH_BGZF": "0"}) == want
    base = ["stream", "-r", ref, "-k", "16", "-s", "1000"]
    want = _cli(root, base + ["-f", str(fq)])
    # no final newline; a BGZF file beside a plain one and a plain gzip one
    nonl = tmp_path / "nonl.fq.gz"
    nonl.write_bytes(synth.bgzf_compress(text[:-1], level=1))
    plain_gz = tmp_path / "plain.fq.gz"
    plain_gz.write_bytes(gzip.compress(text, 1))
    assert _cli(root, base + ["-f", str(nonl), "-f", str(fq), "-f", str(plain_gz)], env={"RKMH_RAW_BLOCK_KB": "700"}) == want * 3
    # several files, runs of jobs taken by the device's workers while the others inflate (the run leaves the queue in one step)
    for mode, extra in ((None, {}), (None, {"RKMH_BGZF_JOB_KB": "500", "RKMH_BGZF_PIECES": "4"}), (None, {"RKMH_BGZF_JOB_KB": "150", "RKMH_BGZF_DEVICE_WORKERS": "2"}),
                        (None, {"RKMH_BGZF_REGISTER": "0"})):      # (the last one: the compressed bytes uploaded from the unpinned mapping)
        for rep in range(2):
            env = dict({"RKMH_RAW_BLOCK_KB": "150"}, **extra)
            if mode is not None:
                env["RKMH_BGZF_DEVICE"] = mode
            r = subprocess.run([exe] + base + ["-f", str(nonl), "-f", str(tmp_path / "reads_65280.fq.gz"), "-f", str(nonl), "-f", str(fq)], capture_output=True,
                               env=dict({k: v for k, v in os.environ.items() if k != "RKMH_BGZF_DEVICE"}, RKMH_BGZF_TIMING="1", **env))
            assert r.returncode == 0 and r.stdout == want * 4, (mode, extra, r.stderr[-600:])
            assert b"[bgzf device]" in r.stderr, (mode, extra)      # (jobs did go through rk_inflate.hip)
    # irregular from the middle on (sequences on two lines): the scanner takes over at that job's first record
    half = _fastq(reads[: n // 2], names=[b"read%07d comment" % i for i in range(n // 2)])
    odd = b"".join(b"@m%d\n" % i + r[:70] + b"\n" + r[70:] + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads[n // 2: n // 2 + 800]) if len(r) > 80)
    mixed = tmp_path / "mixed.fq"
    mixed.write_bytes(half + odd + half)
    mixed_gz = tmp_path / "mixed.fq.gz"
    mixed_gz.write_bytes(synth.bgzf_compress(half + odd + half, level=1))
    want_mixed = _cli(root, base + ["-f", str(mixed)], env={"RKMH_RAW": "0"})
    r = subprocess.run([exe] + base + ["-f", str(mixed_gz)], capture_output=True, env=dict(os.environ, RKMH_TIMING="1", RKMH_RAW_BLOCK_KB="256"))
    assert r.returncode == 0 and r.stdout == want_mixed
    assert b"not four lines per record" in r.stderr
    for env in ({"RKMH_BGZF_DEVICE": "0"}, {"RKMH_BGZF_JOB_KB": "256", "RKMH_BGZF_PIECES": "3"}):     # (the hand-over happens at the first block number of a job)
        r = subprocess.run([exe] + base + ["-f", str(mixed_gz)], capture_output=True, env=dict(os.environ, RKMH_TIMING="1", RKMH_RAW_BLOCK_KB="256", **env))
        assert r.returncode == 0 and r.stdout == want_mixed, env
        assert b"not four lines per record" in r.stderr


def test_cli_reads_from_a_registered_file_mapping(root, data_dir, tmp_path):
    """RKMH_RAW_MMAP=1: the FASTQ file is mapped, the mapping page-locked and every block uploaded from the page cache where it lies
    (rk_fastq_slot_set_source) -- same bytes out as with the workers' pread copies, for stream, stream -M 2, filter, several files,
    small blocks and a file without its final newline (whose last block is copied)."""
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    n = 30000
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, read_len=150, threads=4)
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(n)]
    text = _fastq(reads, names=[b"mm%07d x" % i for i in range(n)])
    fq = tmp_path / "r.fq"
    fq.write_bytes(text)
    nonl = tmp_path / "nonl.fq"
    nonl.write_bytes(text[:-1])
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    for cmd, flags in (("stream", []), ("stream", ["-M", "2"]), ("filter", ["-N", "3"])):
        base = [cmd, "-r", ref, "-k", "16", "-s", "1000"] + flags
        want = _cli(root, base + ["-f", str(fq)])
        assert len(want) > 1000
        for env in ({"RKMH_RAW_MMAP": "1"}, {"RKMH_RAW_MMAP": "1", "RKMH_RAW_BLOCK_KB": "96", "RKMH_RAW_WORKERS": "5"}):
            assert _cli(root, base + ["-f", str(fq)], env=env) == want, (cmd, flags, env)
        two = base + ["-f", str(nonl), "-f", str(fq)]      # (-M counts over both files: compare the same command with and without the mapping)
        assert _cli(root, two, env={"RKMH_RAW_MMAP": "1", "RKMH_RAW_BLOCK_KB": "512"}) == _cli(root, two, env={"RKMH_RAW_BLOCK_KB": "512"})


def test_cli_bgzf_payload_damage_is_refused_on_both_routes(root, data_dir, tmp_path):
    """A member whose payload was changed WITHOUT changing its length or the four-line shape of the text (a base of a stored block
    replaced by another base: every structural check still passes) fails its CRC-32: the device route (k_crc32_members) hands the
    job to the host inflater, which reports the member; the host route (RKMH_BGZF_DEVICE=0) reports it itself.  Either way the run
    ends with an error instead of classifying the changed read -- what gzread does for the reference (src/rkmh.cpp:238-263)."""
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    n = 6000
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, read_len=150, threads=4)
    reads = [bytes(qb[int(qo[i]):int(qo[i + 1])]) for i in range(n)]
    text = _fastq(reads, names=[b"d%06d" % i for i in range(n)])
    ref = os.path.join(data_dir, "all_pave_ref.fa.gz")
    exe = os.path.join(root, "bin", "rkmh")
    base = ["stream", "-r", ref, "-k", "16", "-s", "1000"]
    for level in (0, 1):
        img = bytearray(synth.bgzf_compress(text, level=level, block=0xff00))
        good = tmp_path / ("good%d.fq.gz" % level)
        good.write_bytes(bytes(img))
        want = _cli(root, base + ["-f", str(good)])
        assert want.count(b"\n") == n
        if level == 0:      # stored blocks: the text lies in the file as it is -- change one base of a read in the second member
            at = bytes(img).find(reads[300][20:60])
            assert at > 70000
            img[at + 7] = ord("A") if img[at + 7] != ord("A") else ord("C")
        else:               # deflated: flip one bit of the CRC-32 in the SECOND member's footer (the text is intact, the member is not)
            import struct
            first = struct.unpack_from("<H", img, 16)[0] + 1
            second = struct.unpack_from("<H", img, first + 16)[0] + 1
            img[first + second - 8] ^= 0x10
        bad = tmp_path / ("bad%d.fq.gz" % level)
        bad.write_bytes(bytes(img))
        for env in ({}, {"RKMH_BGZF_DEVICE": "0"}, {"RKMH_BGZF_JOB_KB": "200"}):
            r = subprocess.run([exe] + base + ["-f", str(bad)], capture_output=True, env=dict(os.environ, **env))
            assert r.returncode != 0, (level, env)
            assert b"corrupt BGZF member" in r.stderr, (level, env, r.stderr[-400:])
            assert r.stdout != want
