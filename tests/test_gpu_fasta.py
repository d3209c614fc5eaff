"""Reference FASTA text stripped on the device (rkmh_amd/csrc/rk_fasta.hip, rk_fasta_load_* in include/rkmh_amd.h) against the
host parser (rk_parse_files, itself tested against the oracle's literal kseq grammar in tests/test_parser_parallel.py): regular
text must give the same names, the same sequence boundaries and bit-identical reference sketches; anything that is not plain
line-structured FASTA must be refused (status != 0: the host parser takes the files) -- the device never guesses."""
import gzip
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(api, ctx, slot, text, step):
    ld = api.FastaLoad(ctx, len(text))
    for off in range(0, len(text), step):
        ld.put(slot, off, text[off: off + step])
    return ld


def _wrap(seq, width):
    return b"\n".join(seq[i: i + width] for i in range(0, len(seq), width)) + b"\n"


@pytest.fixture(scope="module")
def slot(ctx):
    from rkmh_amd import api
    s = api.FastqSlot(ctx, max_bytes=1 << 22)
    yield s
    s.destroy()


def test_regular_fasta_equals_the_host_parser(ctx, slot, data_dir, tmp_path):
    from rkmh_amd import api
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGTacgtN", np.uint8)
    rnd = lambda n: bytes(acgt[rng.integers(0, len(acgt), size=n)])
    panel = b"".join(gzip.open(os.path.join(data_dir, f)).read() + b"\n" for f in ("all_pave_ref.fa.gz", "zika.fa.gz", "hpv_16.fa.gz"))
    synth = (b"\n\n>one description words\ttab\n" + _wrap(rnd(300000), 60) +            # leading blank lines, a description
             b">single_line\n" + rnd(70000) + b"\n" +                                       # one line across many 4 KB chunks
             b">empty_record\n>after_empty x\n" + _wrap(rnd(5000), 80) + b"\n\n" +          # a record without bases, blank lines
             b">" + b"n" * 300 + b" long name\n" + _wrap(rnd(12345), 61) +
             b">last\n" + rnd(4097) + b"\n")
    tiny = b">a\nACGTACGTACGTACGTACGTAC\n"
    for name, text, step in (("panel", panel, 1 << 22), ("panel in odd blocks", panel, 70001), ("synthetic", synth, 4096), ("synthetic odd", synth, 999),
                             ("tiny", tiny, 7)):
        fa = tmp_path / "t.fa"
        fa.write_bytes(text)
        want = api.parse_files([str(fa)])
        ld = _load(api, ctx, slot, text, step)
        st, names, offs = ld.finish()
        assert st == 0, (name, st)
        assert len(names) == want["nseq"], name
        assert names == [bytes(n) for n in want["names"]], name
        assert (offs == np.asarray(want["offsets"], dtype=np.uint64)).all(), name
        assert bytes(ld.bases()).upper() == bytes(want["bases"][: int(offs[-1])]).upper(), name
        for ks, s, kw in (([16], 1000, {}), ([12, 20], 500, {}), ([16], 1000, {"max_samples": 3})):
            ld.set_references(ks, s, max_samples=kw.get("max_samples", -1), counter_slots=10000000 if kw else 0)
            got_sk, got_ln = ctx.get_reference_sketches()
            ctx.set_references(want["bases"], want["offsets"], ks, s, max_samples=kw.get("max_samples"), counter_slots=10000000 if kw else 0)
            exp_sk, exp_ln = ctx.get_reference_sketches()
            assert (got_ln == exp_ln).all() and (got_sk == exp_sk).all(), (name, ks, s, kw)
        ld.destroy()


def test_irregular_fasta_is_refused(ctx, slot):
    from rkmh_amd import api
    good = b">a\nACGTACGT\nACGT\n>b\nGGGG\n"
    cases = {
        "carriage returns": good.replace(b"\n", b"\r\n"),
        "'@' line": b">a\nACGT\n@CGT\nAAAA\n",
        "'+' inside": b">a\nAC+GT\n",
        "'>' inside a sequence line": b">a\nAC>GT\n",
        "bases before the first header": b"ACGT\n>a\nACGT\n",
        "space in a sequence line": b">a\nAC GT\n",
        "byte 200": b">a\nAC\xc8GT\n",
        "no record": b"\n\n\n",
    }
    assert _load(api, ctx, slot, good, 5).finish()[0] == 0
    for name, text in cases.items():
        for step in (3, 1 << 20):
            st, names, offs = _load(api, ctx, slot, text, step).finish()
            assert st != 0, name


def test_mutated_fasta_never_parses_differently_from_kseq(ctx, slot, orc):
    """Fuzz: random byte edits of regular FASTA text (RKMH_TEST_FUZZ raises the count).  Whatever the edit, the device either refuses
    the text or returns exactly the records the oracle's literal kseq grammar reads (names, sequences)."""
    from rkmh_amd import api
    rng = np.random.default_rng(int(os.environ.get("RKMH_TEST_SEED_BASE", "11")))
    acgt = np.frombuffer(b"ACGTacgtN", np.uint8)
    rnd = lambda n: bytes(acgt[rng.integers(0, len(acgt), size=n)])
    base = b"".join(b">s%d some words\n" % i + _wrap(rnd(int(rng.integers(1, 400))), int(rng.integers(20, 90))) for i in range(12))
    pool = b"\n\n\n\r@+>> \tACGTNacgt!~\x7f\x80\x00"
    n = int(os.environ.get("RKMH_TEST_FUZZ", "300"))
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
        text = bytes(t)
        if not text.endswith(b"\n"):
            text += b"\n"          # (the callers of rk_fasta_load_put end every file with a newline)
        ld = _load(api, ctx, slot, text, int(rng.integers(1, 5000)))
        st, names, offs = ld.finish()
        if st == 0:
            accepted += 1
            want = orc.kseq_parse_bytes(text)
            assert names == [w[0] for w in want], (it, text[:200])
            got = bytes(ld.bases())
            assert [got[int(offs[i]): int(offs[i + 1])] for i in range(len(names))] == [w[1] for w in want], it
        ld.destroy()
    assert 0 < accepted < n


def _cli(root, args, env=None):
    r = subprocess.run([os.path.join(root, "bin", "rkmh")] + args, capture_output=True, env=dict(os.environ, RKMH_TIMING="1", **(env or {})), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout, r.stderr


def test_cli_references_through_the_device(root, data_dir, tmp_path):
    """bin/rkmh stream / filter with the -r files taken by the device (RKMH_RAW_REFS=1; by default only genome-sized references are)
    print what the host-parsed run prints -- several files, -I, small blocks -- and irregular references fall back to the host."""
    refs = []
    for f in ("all_pave_ref.fa.gz", "zika.fa.gz"):
        p = tmp_path / f[:-3]
        p.write_bytes(gzip.open(os.path.join(data_dir, f)).read())
        refs += ["-r", str(p)]
    fq = tmp_path / "reads.fq"
    fq.write_bytes(gzip.open(os.path.join(data_dir, "z1.fq.gz")).read() if os.path.exists(os.path.join(data_dir, "z1.fq.gz")) else open(os.path.join(data_dir, "z1.fq"), "rb").read())
    for cmd, extra in (("stream", []), ("stream", ["-I", "2"]), ("filter", ["-N", "3"]), ("filter", ["-N", "3", "-I", "2", "-M", "1"])):
        args = [cmd] + refs + ["-f", str(fq), "-k", "16", "-s", "1000"] + extra
        want, _ = _cli(root, args, env={"RKMH_RAW_REFS": "0"})
        assert len(want) > 100, (cmd, extra)
        for env, more in (({"RKMH_RAW_REFS": "1"}, []), ({"RKMH_RAW_REFS": "1", "RKMH_RAW_BLOCK_KB": "64", "RKMH_RAW_WORKERS": "3"}, []),
                          ({"RKMH_RAW_REFS": "1", "RKMH_RAW_BLOCK_KB": "128"}, ["--devices", "0,0"])):      # two contexts: the text goes to the first
            got, err = _cli(root, args + more, env=env)
            assert got == want, (cmd, extra, env, more)
            assert b"references through the device: " in err, err[-800:]
    # CRLF references: refused by the device, parsed on the host, same output
    crlf = tmp_path / "crlf.fa"
    crlf.write_bytes(gzip.open(os.path.join(data_dir, "zika.fa.gz")).read().replace(b"\n", b"\r\n"))
    args = ["stream", "-r", str(crlf), "-f", str(fq), "-k", "16", "-s", "1000"]
    want, _ = _cli(root, args, env={"RKMH_RAW_REFS": "0"})
    got, err = _cli(root, args, env={"RKMH_RAW_REFS": "1"})
    assert got == want and b"the host parser reads them" in err
