"""CPU-only: the C-ABI library loads and exports every symbol include/rkmh_amd.h declares; the product
refuses to run without a GPU (no CPU fallback); host-side helpers (parser, formatter, generator) work."""
import os
import re

import numpy as np
import pytest

import rkmh_amd
from rkmh_amd import api, synth


def _declared(root):
    txt = open(os.path.join(root, "include", "rkmh_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(root):
    lib = rkmh_amd.load_library()
    names = _declared(root)
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    # and the Python binding covers all of them
    assert set(names) <= set(api._SIGS), set(names) - set(api._SIGS)


def test_product_does_not_import_the_oracle(root):
    for dp, _, fs in os.walk(os.path.join(root, "rkmh_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "rk_oracle" not in src and "import oracle" not in src and "librkoracle" not in src, f


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rkmh_amd.RkmhError, match="no CPU fallback"):
        rkmh_amd.Context(0)


def test_format_stream_line_matches_reference_format():
    f = api.format_stream_line
    assert f(b"R", b"q", 5, 2, 100, 1000) == b"R\tq\t5\t1000\t\t\n"
    assert f(b"R", b"q", 5, 0, 100, 1000) == b"R\tq\t5\t1000\t\tFAIL:DIFF\n"
    assert f(b"R", b"q", 5, 2, 100, 1000, 200, 0) == b"R\tq\t5\t1000FAIL:DEPTH\tFAIL:MATCHES\t\n"


@pytest.mark.parametrize("name", ["all_pave_ref.fa.gz", "minION25.fq.gz", "z1.fq.gz", "zika.refs.fa.gz", "hpv_16.fa.gz"])
def test_parser_matches_kseq_grammar_on_bundled_files(orc, data_dir, name):
    want = orc.kseq_parse_file(os.path.join(data_dir, name))
    got = api.parse_files([os.path.join(data_dir, name)])
    assert got["nseq"] == len(want)
    assert got["names"] == [w[0] for w in want]
    for i, w in enumerate(want):
        assert bytes(got["bases"][int(got["offsets"][i]): int(got["offsets"][i + 1])]) == w[1]
    if want[0][2] is not None:
        assert got["quals"] == [w[2] for w in want]
    else:
        assert got["quals"] is None


EDGE = [
    b"", b"\n\n", b">a\nACGT\n", b">a desc here\nAC\nGT\n\n>b\n\nTT\n", b"@r1\nACGT\n+\nIIII\n@r2\nGG\n+r2\n@@\n",
    b"junk before\n>a\nAC GT\tAA\n", b">a\nACGT", b"@r\nACGT\n+\nII\n@s\nAA\n+\nII\n",  # truncated quality ends the file
    b">x\nAC>y\nGG\n", b"@r\nAC\n+\nII", b">", b">a", b"@r\nACGT\n+", b">a\n>b\nAC\n",
    b"@r1 c\nAC\nGT\n+\nII\nII\n@r2\nA\n+\nI\n", b">a\r\nACGT\r\n>b\r\nGG\r\n",
]


@pytest.mark.parametrize("i", range(len(EDGE)))
def test_parser_edge_cases(orc, tmp_path, i):
    p = tmp_path / "x.fa"
    p.write_bytes(EDGE[i])
    want = orc.kseq_parse_bytes(EDGE[i])
    got = api.parse_files([str(p)])
    assert got["names"] == [w[0] for w in want]
    seqs = [bytes(got["bases"][int(got["offsets"][j]): int(got["offsets"][j + 1])]) for j in range(got["nseq"])]
    assert seqs == [w[1] for w in want]


def test_streaming_reader_equals_whole_file(data_dir):
    whole = api.parse_files([os.path.join(data_dir, "z1.fq.gz")])
    rd = api.Reader(os.path.join(data_dir, "z1.fq.gz"))
    names, total = [], 0
    while True:
        b = rd.next_batch(max_records=137)
        if b is None:
            break
        assert b["nseq"] <= 137
        names += b["names"]
        total += int(b["offsets"][-1])
    assert names == whole["names"] and total == int(whole["offsets"][-1])


def test_multi_file_concatenation(data_dir):
    a = api.parse_files([os.path.join(data_dir, "hpv_16.fa.gz"), os.path.join(data_dir, "zika.fa.gz")])
    assert a["nseq"] == 2  # rkmh.cpp:244-261: files concatenated in argument order


def test_synth_generator_deterministic_and_sharded(data_dir):
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    b, o = synth.generate_reads(refs["bases"], refs["offsets"], 0, 3000)
    b2, _ = synth.generate_reads(refs["bases"], refs["offsets"], 1000, 2000)
    assert (b2[:150000] == b[150000:300000]).all()
    bf, of = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, 3000, threads=3)
    assert (bf == b).all() and (of == o).all()
    r = b[:450000].reshape(3000, 150)
    assert set(np.unique(r)) <= set(b"ACGTNWY")
    assert 0 < (r == ord("N")).any(axis=1).sum() < 30


def test_block_formatters_match_the_line_formatter():
    """rk_fastq_stream_lines / rk_fastq_filter_records (rk_format.cpp: the output of a whole classified block written from the raw
    text) against rk_format_stream_line per read and a literal restatement of filter's rule (rkmh.cpp:1292-1300 with the decision of
    equiv.hpp:324-353) -- host code only, on a hand-made result structure."""
    import ctypes as C
    lib = rkmh_amd.load_library()
    rng = np.random.default_rng(9)
    ref_names = [b"ref%d|with|bars" % i for i in range(7)] + [b"a-very-long-reference-name-" + b"x" * 70]
    recs, text = [], bytearray()
    n = 500
    out4 = np.zeros((n, 4), np.int32)
    name_off, name_len, seq_off, seq_len, qual_off = (np.zeros(n, np.uint32) for _ in range(5))
    for i in range(n):
        nm = b"read_%d/%d" % (i, i % 3) + (b"Z" * int(rng.integers(0, 40)))
        ln = int(rng.integers(1, 200))
        sq = bytes(rng.choice(np.frombuffer(b"ACGTacgtNn", np.uint8), size=ln))
        ql = bytes(rng.integers(33, 127, size=ln, dtype=np.uint8))
        text += b"@"
        name_off[i], name_len[i] = len(text), len(nm)
        text += nm + b" description\n"
        seq_off[i], seq_len[i] = len(text), ln
        text += sq + b"\n+\n"
        qual_off[i] = len(text)
        text += ql + b"\n"
        out4[i] = (int(rng.integers(0, len(ref_names))), int(rng.integers(-1, 30)), int(rng.integers(-3, 12)), int(rng.integers(0, 140)))
        recs.append((nm, sq, ql))
    text += b"\0" * 64      # the slot's buffer has this slack: names are copied in 16-byte steps
    res = api.FastqResult()
    res.status, res.nrec = 0, n
    res.out4 = out4.ctypes.data_as(C.POINTER(C.c_int32))
    for fld, arr in (("name_off", name_off), ("name_len", name_len), ("seq_off", seq_off), ("seq_len", seq_len), ("qual_off", qual_off)):
        setattr(res, fld, arr.ctypes.data_as(C.POINTER(C.c_uint32)))
    tbuf = (C.c_char * len(text)).from_buffer(text)
    for sketch, mm, md in ((1000, -1, 0), (2000, 5, 2), (37, 100, -5)):
        parts = api.LineParts(ref_names, sketch, mm, md)
        cap = int(lib.rk_fastq_stream_lines_bound(parts._h, C.byref(res)))
        dst = C.create_string_buffer(cap)
        got = lib.rk_fastq_stream_lines(parts._h, C.byref(res), tbuf, dst, cap)
        want = b"".join(api.format_stream_line(ref_names[int(r[0])], recs[i][0], int(r[1]), int(r[2]), int(r[3]), sketch, mm, md) for i, r in enumerate(out4))
        assert got == len(want) and dst.raw[:got] == want, (sketch, mm, md)
        assert lib.rk_fastq_stream_lines(parts._h, C.byref(res), tbuf, dst, cap - 1) < 0            # a buffer below the bound is refused
        parts.destroy()
        want_f = bytearray()
        for i, r in enumerate(out4):
            shared, diff_ok = (0, 0 > md) if r[1] <= 0 else (int(r[1]), (int(r[2]) - (1 if r[0] == 0 else 0)) > md)
            # rkmh.cpp:1292: read_min_lens <= 0 -- which implies shared == 0 (a shared hash is a min); the formatter tests the
            # conjunction, the same predicate on real rows, so that rows with min_num clamped to 0 (rk_set_min_num_bound) format right
            if (r[3] <= 0 and shared <= 0) or shared < mm or not diff_ok:
                continue
            up = bytes((c - 32) if c > 91 else c for c in recs[i][1])
            want_f += b">" + recs[i][0] + b"\n" + up + b"\n+\n" + recs[i][2] + b"\n"
        cap = int(lib.rk_fastq_filter_records_bound(C.byref(res)))
        dst = C.create_string_buffer(cap)
        got = lib.rk_fastq_filter_records(C.byref(res), tbuf, mm, md, dst, cap)
        assert got == len(want_f) and dst.raw[:got] == bytes(want_f), (mm, md)
    bad = api.FastqResult()
    bad.status = 4
    assert lib.rk_fastq_filter_records(C.byref(bad), tbuf, 0, 0, dst, cap) < 0                       # a refused block has nothing to print


def test_cut_before_finds_record_starts(tmp_path):
    """cli._cut_before (how ranks and blocks turn byte ranges into ranges of whole records): every cut is a record start, cuts are
    monotone, never past the position asked for and never far before it -- on records whose quality strings begin with '@' and '+'."""
    from rkmh_amd import cli
    rng = np.random.default_rng(4)
    recs, starts, pos = [], [], 0
    for i in range(3000):
        ln = int(rng.integers(1, 300))
        q = bytearray(rng.integers(33, 127, size=ln, dtype=np.uint8).tobytes())
        if i % 3 == 0:
            q[0] = ord("@")
        if i % 5 == 0:
            q[0] = ord("+")
        r = b"@r%d extra\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=ln)) + b"\n+\n" + bytes(q) + b"\n"
        starts.append(pos)
        pos += len(r)
        recs.append(r)
    p = tmp_path / "r.fq"
    p.write_bytes(b"".join(recs))
    size, sset = pos, set(starts)
    fd = os.open(str(p), os.O_RDONLY)
    try:
        assert cli._cut_before(fd, size, 0) == 0 and cli._cut_before(fd, size, size) == size and cli._cut_before(fd, size, size + 5) == size
        last = 0
        for q in sorted(int(x) for x in rng.integers(1, size, size=400)):
            c = cli._cut_before(fd, size, q)
            assert c in sset and c <= q and c >= last, (q, c)
            assert q - c < 3 * 640, (q, c)      # (a record start counts once its '+' line is in sight: the cut may lie a record or two back)
            last = c
    finally:
        os.close(fd)


def test_hash_policy_text_round_trips():
    """rk_policy_parse / rk_policy_describe (host code: no GPU needed): presets, single keys, order of application, refusals."""
    from rkmh_amd import api
    d = api.describe_policy
    assert d(api.parse_policy(None)) == "fold=swap32,windows=len-k,zero=count,mask=lt,freqmax=incl,seed=42"
    assert d(api.parse_policy("mash")) == "fold=h1,windows=len-k+1,zero=count,mask=lt,freqmax=incl,seed=42"
    assert d(api.parse_policy("mash,default")) == d(api.parse_policy("default"))
    assert d(api.parse_policy(" fold=w2w1 , seed=0x10,zero=skip,mask=le,freqmax=excl")) == "fold=w2w1,windows=len-k,zero=skip,mask=le,freqmax=excl,seed=16"
    p = api.parse_policy("windows=len-k+1", base=api.parse_policy("fold=h1"))
    assert d(p) == d(api.parse_policy("mash")) and d(api.parse_policy(d(p))) == d(p)
    lib = api.load_library()
    import ctypes as C
    assert lib.rk_policy_same_hashes(C.byref(p), C.byref(api.parse_policy("mash,mask=le"))) == 1
    assert lib.rk_policy_same_hashes(C.byref(p), C.byref(api.parse_policy("default"))) == 0
    for bad in ("mesh", "fold", "fold=", "fold=H1", "windows=len", "seed=", "seed=4294967296", "k=3"):
        with pytest.raises(api.RkmhError):
            api.parse_policy(bad)
