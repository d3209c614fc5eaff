"""-M with a bounded min_num (rk_set_min_num_bound; src/rkmh.cpp:916-917, :938): the mask is applied per index key, no window
outside the index is looked up in the depth map, and row field 3 is min(num_mins, bound).  Every row must equal the oracle's
two-pass result with its fourth column clamped -- on the k-mer-space kernel (k 8..16, one and several sizes), on the hash-space
kernel (k = 20, 21), through reroutes (reads with more windows than the sketch keeps), for tables small enough that most keys
are dropped, and for both comparison policies."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pad(b):
    out = np.zeros(len(b) + 16, dtype=np.uint8)
    out[: len(b)] = b
    return out


@pytest.fixture(scope="module")
def pave(orc, data_dir):
    recs = orc.kseq_parse_file(os.path.join(data_dir, "all_pave_ref.fa.gz"))
    rb, ro = orc.pack([r[1] for r in recs])
    return [r[0] for r in recs], _pad(rb), ro


def _clamped(want, bound):
    w = want.copy()
    if bound >= 0:
        w[:, 3] = np.minimum(w[:, 3], bound)
    return w


def _reads(rb, ro, lo, n, seed, with_n=True, ragged=True):
    """synthetic reads; some with N, some shorter, a few shorter than k"""
    from rkmh_amd import synth
    qb, qo = synth.generate_reads_fast(rb, ro, lo, lo + n)
    qb = qb.copy()
    rng = np.random.default_rng(seed)
    if with_n:
        for i in rng.integers(0, n, size=n // 20):
            qb[int(qo[i]) + int(rng.integers(0, 150))] = ord("N")
        for i in rng.integers(0, n, size=n // 50):   # lower case: to_upper applies
            s = int(qo[i])
            qb[s: s + 40] = np.frombuffer(bytes(qb[s: s + 40]).lower(), dtype=np.uint8)
    if ragged:
        seqs = []
        for i in range(n):
            s = bytes(qb[int(qo[i]): int(qo[i + 1])])
            r = rng.random()
            if r < 0.05:
                s = s[: int(rng.integers(0, 30))]
            elif r < 0.25:
                s = s[: int(rng.integers(30, 150))]
            seqs.append(s)
        import oracle
        qb, qo = oracle.pack(seqs)
    return _pad(qb), qo


@pytest.mark.parametrize("ks,slots,min_occ", [([16], 200000000, 2), ([16], 1000003, 3), ([16], 4099, 60), ([12, 16], 1000003, 2),
                                              ([15], 65537, 6), ([20], 1000003, 2), ([21], 4099, 40), ([10], 99991, 12)])
def test_bounded_rows_equal_the_clamped_oracle(orc, pave, ks, slots, min_occ):
    import rkmh_amd
    _, rb, ro = pave
    T = min(16, os.cpu_count() or 1)
    n = 20000
    qb, qo = _reads(rb, ro, 7000, n, seed=len(ks) * 1000 + ks[0] + slots % 97)
    c = rkmh_amd.Context(0)
    try:
        c.set_references(rb, ro, ks, 1000)
        sk, ln = c.get_reference_sketches()
        want = orc.classify_stream(qb, qo, ks, 1000, sk, ln, threads=T, min_kmer_occ=min_occ, counter_slots=slots)
        assert (want[:, 1] > 0).any()
        cnt = rkmh_amd.Counter(c, slots=slots)
        c.count_batch(qb, qo, cnt)
        try:
            # the bound is set before the filter, then changed while the filter is set, then back to exact
            c.set_min_num_bound(0)
            c.set_depth_filter(cnt, min_occ)
            for bound in (0, 1, 4, 200, -1, 2):
                c.set_min_num_bound(bound)
                got = c.classify(qb, qo)
                w = _clamped(want, bound)
                bad = np.nonzero((got != w).any(axis=1))[0]
                assert len(bad) == 0, (ks, slots, min_occ, bound, len(bad), got[bad[:4]], w[bad[:4]])
        finally:
            c.set_depth_filter(None, 0)
            c.set_min_num_bound(-1)
            cnt.destroy()
        # without a filter the bound means nothing
        c.set_min_num_bound(0)
        plain = orc.classify_stream(qb, qo, ks, 1000, sk, ln, threads=T)
        assert (c.classify(qb, qo) == plain).all()
    finally:
        c.close()


def test_bounded_mode_uses_the_kmer_space_kernel_and_reroutes(orc, pave):
    """k = 16 with a bounded filter must stay on the k-mer-space form (the point of the bound); reads longer than the sketch
    (reroute through the general path, which masks by slot) still come back exact with the clamp applied."""
    import rkmh_amd
    from rkmh_amd import synth
    _, rb, ro = pave
    T = min(16, os.cpu_count() or 1)
    S = 100
    rng = np.random.default_rng(5)
    short_b, short_o = synth.generate_reads_fast(rb, ro, 100, 100 + 3000)
    seqs = [bytes(short_b[int(short_o[i]): int(short_o[i + 1])]) for i in range(3000)]
    for j in range(0, 3000, 7):   # every seventh read: 400 bases = 384 windows > S -> bottom-S selection matters
        r = int(rng.integers(0, len(ro) - 1))
        st = int(ro[r]) + int(rng.integers(0, int(ro[r + 1] - ro[r]) - 400))
        seqs[j] = bytes(rb[st: st + 400])
    qb, qo = orc.pack(seqs)
    qb = _pad(qb)
    c = rkmh_amd.Context(0)
    try:
        c.set_references(rb, ro, [16], S)
        assert c.kmer_form()[0] == 1
        sk, ln = c.get_reference_sketches()
        slots = 300007
        want = orc.classify_stream(qb, qo, [16], S, sk, ln, threads=T, min_kmer_occ=2, counter_slots=slots)
        cnt = rkmh_amd.Counter(c, slots=slots)
        c.count_batch(qb, qo, cnt)
        c.set_depth_filter(cnt, 2)
        try:
            for bound in (0, 3, 150):
                c.set_min_num_bound(bound)
                got = c.classify(qb, qo)
                w = _clamped(want, bound)
                bad = np.nonzero((got != w).any(axis=1))[0]
                assert len(bad) == 0, (bound, len(bad), got[bad[:4]], w[bad[:4]])
        finally:
            c.set_depth_filter(None, 0)
            cnt.destroy()
    finally:
        c.close()


@pytest.mark.parametrize("strict", [1, 0])
def test_bounded_mode_honours_the_mask_comparison_policy(orc, pave, strict):
    import rkmh_amd
    _, rb, ro = pave
    T = min(16, os.cpu_count() or 1)
    qb, qo = _reads(rb, ro, 0, 8000, seed=99, ragged=False)
    pol = orc.default_policy(mask_strict_less=strict)
    c = rkmh_amd.Context(0, mask_strict_less=strict)
    try:
        c.set_references(rb, ro, [16], 1000)
        sk, ln = c.get_reference_sketches()
        want = orc.classify_stream(qb, qo, [16], 1000, sk, ln, pol, threads=T, min_kmer_occ=2, counter_slots=500009)
        cnt = rkmh_amd.Counter(c, slots=500009)
        c.count_batch(qb, qo, cnt)
        c.set_min_num_bound(3)
        c.set_depth_filter(cnt, 2)
        try:
            got = c.classify(qb, qo)
        finally:
            c.set_depth_filter(None, 0)
            cnt.destroy()
        assert (got == _clamped(want, 3)).all()
    finally:
        c.close()


def test_new_references_under_a_bounded_filter(orc, pave):
    """rk_set_references while a bounded filter is set: the per-key mask follows the new key ids."""
    import rkmh_amd
    _, rb, ro = pave
    T = min(16, os.cpu_count() or 1)
    qb, qo = _reads(rb, ro, 500, 5000, seed=3, ragged=False)
    c = rkmh_amd.Context(0)
    try:
        c.set_references(rb, ro, [16], 1000)
        cnt = rkmh_amd.Counter(c, slots=700001)
        c.count_batch(qb, qo, cnt)
        c.set_min_num_bound(1)
        c.set_depth_filter(cnt, 2)
        try:
            for nref, k in ((60, 16), (182, 14)):
                c.set_references(rb, ro[: nref + 1], [k], 1000)
                sk, ln = c.get_reference_sketches()
                # the table was counted with k = 16; any table is a table: the oracle counts with the classify k, so build the
                # expectation from the library's own count with that k instead
                cnt2 = rkmh_amd.Counter(c, slots=700001)
                c.count_batch(qb, qo, cnt2)
                c.set_depth_filter(cnt2, 2)
                want = orc.classify_stream(qb, qo, [k], 1000, sk, ln, threads=T, min_kmer_occ=2, counter_slots=700001)
                got = c.classify(qb, qo)
                assert (got == _clamped(want, 1)).all(), (nref, k)
                c.set_depth_filter(cnt, 2)
                cnt2.destroy()
        finally:
            c.set_depth_filter(None, 0)
            cnt.destroy()
    finally:
        c.close()


def test_clis_print_the_same_bytes_in_bounded_and_exact_form(orc, root, data_dir, tmp_path):
    """bin/rkmh and python -m rkmh_amd.cli set the bound from -N (stream: num_mins <= -N, rkmh.cpp:938; filter: read_min_lens <= 0,
    :1292).  stdout must be byte-identical to the oracle's lines and to the exact form (RKMH_EXACT_MIN_NUM=1) for -M 2 with and
    without -N / -D, including reads that match nothing, reads with few valid windows (N runs) and reads shorter than k -- the
    cases in which FAIL:DEPTH can fire."""
    import subprocess
    import sys
    from rkmh_amd import synth, api
    exe = os.path.join(root, "bin", "rkmh")
    refs = orc.kseq_parse_file(os.path.join(data_dir, "all_pave_ref.fa.gz"))[:80]
    ref_fa = tmp_path / "refs.fa"
    ref_fa.write_bytes(b"".join(b">" + r[0] + b"\n" + r[1] + b"\n" for r in refs))
    R = api.parse_files([str(ref_fa)])
    n = 3000
    qb, qo = synth.generate_reads(R["bases"], R["offsets"], 0, n)
    names = synth.read_names(0, n)
    rng = np.random.default_rng(11)
    seqs = [bytearray(qb[int(qo[i]): int(qo[i + 1])]) for i in range(n)]
    for i in range(0, n, 5):        # random reads: every k-mer occurs once in the run -> masked by -M 2 -> num_mins small or 0
        seqs[i] = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=150).tobytes())
    for i in range(1, n, 11):       # a run of N leaves only a few valid windows
        keep = int(rng.integers(16, 24))
        seqs[i][keep:] = b"N" * (150 - keep)
    for i in range(2, n, 97):       # shorter than k
        seqs[i] = seqs[i][: int(rng.integers(1, 16))]
    seqs = [bytes(s) for s in seqs]
    quals = [b"I" * len(s) for s in seqs]
    fq = tmp_path / "reads.fq"
    fq.write_bytes(b"".join(b"@" + names[i] + b"\n" + seqs[i] + b"\n+\n" + quals[i] + b"\n" for i in range(n)))
    rb, ro = orc.pack([r[1] for r in refs])
    pb, po = orc.pack(seqs)
    sk, ln = orc.sketch_refs(rb, ro, [16], 1000, threads=4)

    def run(cmd, flags, exact, python_cli=False, full_map=False):
        env = dict(os.environ)
        env["RKMH_EXACT_MIN_NUM"] = "1" if exact else "0"
        env["RKMH_FULL_DEPTH_MAP"] = "1" if full_map else "0"
        argv = ([sys.executable, "-m", "rkmh_amd.cli"] if python_cli else [exe]) + [cmd, "-r", str(ref_fa), "-f", str(fq), "-k", "16"] + flags
        r = subprocess.run(argv, capture_output=True, cwd=root, env=env)
        assert r.returncode == 0, r.stderr
        return r.stdout

    # stream: 200 M slots (rkmh.cpp:739)
    rows = orc.classify_stream(pb, po, [16], 1000, sk, ln, threads=8, min_kmer_occ=2, counter_slots=200000000)
    assert (rows[:, 3] <= 5).any() and (rows[:, 3] > 5).any()
    for flags, kw in ((["-M", "2"], {}), (["-M", "2", "-N", "5"], dict(min_matches=5)), (["-M", "2", "-N", "0", "-D", "3"], dict(min_matches=0, min_diff=3)),
                      (["-M", "2", "-N", "40"], dict(min_matches=40))):
        want = "".join(orc.stream_line(refs[rows[i, 0]][0].decode(), names[i].decode(), rows[i, 1], rows[i, 2], rows[i, 3], 1000, **kw)
                       for i in range(n)).encode()
        if "-N" in flags:
            assert b"FAIL:DEPTH" in want
        for exact in (False, True):
            assert run("stream", flags, exact) == want, (flags, exact)
        assert run("stream", flags, False, python_cli=True) == want, ("python cli", flags)
        assert run("stream", flags, False, full_map=True) == want, ("full depth table", flags)
    # filter: 10 M slots (:1187); -D -1 lets reads through that share nothing, so `read_min_lens <= 0` alone decides for them
    frows = orc.classify_stream(pb, po, [16], 1000, sk, ln, threads=8, min_kmer_occ=2, counter_slots=10000000)
    for flags, kw in ((["-M", "2"], {}), (["-M", "2", "-N", "3"], dict(min_matches=3)), (["-M", "2", "-D", "-1"], dict(min_diff=-1))):
        want = b""
        for i in range(n):
            if orc.filter_decision(frows[i], kw.get("min_matches", -1), kw.get("min_diff", 0))[3]:
                want += orc.filter_record(names[i], orc.to_upper(seqs[i]), quals[i])
        assert want
        for exact in (False, True):
            assert run("filter", flags, exact) == want, (flags, exact)
        assert run("filter", flags, False, python_cli=True) == want, ("python cli", flags)
        assert run("filter", flags, False, full_map=True) == want, ("full depth table", flags)


@pytest.mark.parametrize("ks,slots,min_occ", [([16], 200000000, 2), ([16], 4099, 60), ([12, 16], 1000003, 2), ([20], 10000000, 2), ([21], 65537, 9)])
def test_compact_depth_map_equals_the_full_table(orc, pave, ks, slots, min_occ):
    """rk_counter_create_compact counts only the slots index keys map to -- exactly: the entry of every sampled key equals the full
    table's slot, two half-batches summed with rk_counter_add equal one pass, and the rows classified under it equal the oracle's
    two-pass rows (min_num clamped to bound 0)."""
    import rkmh_amd
    _, rb, ro = pave
    T = min(16, os.cpu_count() or 1)
    n = 20000
    qb, qo = _reads(rb, ro, 3000, n, seed=17 + slots % 13)
    c = rkmh_amd.Context(0)
    try:
        c.set_references(rb, ro, ks, 1000)
        sk, ln = c.get_reference_sketches()
        want = _clamped(orc.classify_stream(qb, qo, ks, 1000, sk, ln, threads=T, min_kmer_occ=min_occ, counter_slots=slots), 0)
        full = rkmh_amd.Counter(c, slots=slots)
        c.count_batch(qb, qo, full)
        comp = rkmh_amd.Counter(c, slots=slots, compact=True)
        assert comp.compact and comp.entries == rkmh_amd.Counter.compact_entries(c, slots) <= len(np.unique(sk[sk != 0]))
        c.count_batch(qb, qo, comp)
        keys = np.unique(sk[sk != 0])
        rng = np.random.default_rng(1)
        for key in rng.choice(keys, size=40, replace=False):
            assert comp.get(int(key)) == full.get(int(key)), int(key)
        # two halves on two maps, summed
        h1 = rkmh_amd.Counter(c, slots=slots, compact=True)
        h2 = rkmh_amd.Counter(c, slots=slots, compact=True)
        half = n // 2
        c.count_batch(qb, qo[: half + 1], h1)
        ob = qo[half:] - qo[half]
        c.count_batch(_pad(qb[int(qo[half]): int(qo[-1])]), ob, h2)
        h1.add(h2)
        for key in rng.choice(keys, size=20, replace=False):
            assert h1.get(int(key)) == full.get(int(key)), int(key)
        with pytest.raises(rkmh_amd.api.RkmhError):
            c.set_depth_filter(comp, min_occ)           # exact min_num needs the full table
        c.set_min_num_bound(0)
        try:
            for cnt in (comp, h1, full):
                c.set_depth_filter(cnt, min_occ)
                got = c.classify(qb, qo)
                bad = np.nonzero((got != want).any(axis=1))[0]
                assert len(bad) == 0, (ks, slots, cnt.compact, len(bad), got[bad[:4]], want[bad[:4]])
            c.set_depth_filter(comp, min_occ)
            with pytest.raises(rkmh_amd.api.RkmhError):
                c.set_min_num_bound(3)
        finally:
            c.set_depth_filter(None, 0)
            c.set_min_num_bound(-1)
        with pytest.raises(rkmh_amd.api.RkmhError):
            comp.save("/tmp/never.bin")
        for k in (full, comp, h1, h2):
            k.destroy()
    finally:
        c.close()


def test_compact_depth_map_refuses_what_it_cannot_answer(orc, pave):
    import rkmh_amd
    from rkmh_amd import api
    _, rb, ro = pave
    c = rkmh_amd.Context(0)
    try:
        with pytest.raises(api.RkmhError):
            rkmh_amd.Counter(c, slots=1000003, compact=True)        # no references yet
        c.set_references(rb, ro, [16], 100)
        comp = rkmh_amd.Counter(c, slots=1000003, compact=True)
        short = _pad(rb[:3000].copy())
        c.count_batch(short, np.arange(0, 3001, 100, dtype=np.uint64), comp)   # 100-base reads: 84 windows <= 100
        with pytest.raises(api.NeedFullDepthMap):                   # 150-base reads: 134 windows > s = 100
            c.count_batch(short, np.arange(0, 3001, 150, dtype=np.uint64), comp)
        with pytest.raises(api.NeedFullDepthMap):                   # long reads
            c.count_batch(_pad(rb[:8000].copy()), np.array([0, 4000, 8000], dtype=np.uint64), comp)
        c.set_references(rb, ro[:50], [16], 100)                    # another index: the map is stale
        with pytest.raises(api.RkmhError):
            c.count_batch(short, np.arange(0, 3001, 100, dtype=np.uint64), comp)
        c.set_min_num_bound(0)
        with pytest.raises(api.RkmhError):
            c.set_depth_filter(comp, 2)
        c.set_min_num_bound(-1)
        comp.destroy()
    finally:
        c.close()


def test_clis_fall_back_to_full_depth_tables_for_reads_beyond_the_sketch(orc, root, data_dir, tmp_path):
    """stream -M 2 / filter -M 2 start with compact depth maps; a read with more hashes than the sketch keeps (here 1200 bases
    against s = 1000, in the LAST block of the file so that pass 1 is well under way) makes the count pass answer
    RK_ERR_NEED_FULL and the CLIs repeat it with the reference's full tables.  stdout equals the oracle's, through the device
    front end and through the parse-everything path (RKMH_RAW=0), for both CLIs."""
    import subprocess
    import sys
    from rkmh_amd import synth, api
    exe = os.path.join(root, "bin", "rkmh")
    refs = orc.kseq_parse_file(os.path.join(data_dir, "all_pave_ref.fa.gz"))[:40]
    ref_fa = tmp_path / "refs.fa"
    ref_fa.write_bytes(b"".join(b">" + r[0] + b"\n" + r[1] + b"\n" for r in refs))
    R = api.parse_files([str(ref_fa)])
    n = 2000
    qb, qo = synth.generate_reads(R["bases"], R["offsets"], 0, n)
    names = [b"r%d" % i for i in range(n)]
    seqs = [bytes(qb[int(qo[i]): int(qo[i + 1])]) for i in range(n)]
    for i in (n - 3, n - 40):
        seqs[i] = refs[i % 40][1][100:1300]
    quals = [b"F" * len(s) for s in seqs]
    fq = tmp_path / "reads.fq"
    fq.write_bytes(b"".join(b"@" + names[i] + b"\n" + seqs[i] + b"\n+\n" + quals[i] + b"\n" for i in range(n)))
    rb, ro = orc.pack([r[1] for r in refs])
    pb, po = orc.pack(seqs)
    sk, ln = orc.sketch_refs(rb, ro, [16], 1000, threads=4)
    rows = orc.classify_stream(pb, po, [16], 1000, sk, ln, threads=8, min_kmer_occ=2, counter_slots=200000000)
    want_stream = "".join(orc.stream_line(refs[rows[i, 0]][0].decode(), names[i].decode(), rows[i, 1], rows[i, 2], rows[i, 3], 1000)
                          for i in range(n)).encode()
    frows = orc.classify_stream(pb, po, [16], 1000, sk, ln, threads=8, min_kmer_occ=2, counter_slots=10000000)
    want_filter = b"".join(orc.filter_record(names[i], orc.to_upper(seqs[i]), quals[i]) for i in range(n) if orc.filter_decision(frows[i])[3])
    for cmd, want in (("stream", want_stream), ("filter", want_filter)):
        for raw in ("1", "0"):
            for python_cli in (False, True):
                env = dict(os.environ, RKMH_RAW=raw, RKMH_RAW_BLOCK_KB="64", RKMH_TIMING="1")
                argv = ([sys.executable, "-m", "rkmh_amd.cli"] if python_cli else [exe]) + [cmd, "-r", str(ref_fa), "-f", str(fq), "-k", "16", "-M", "2"]
                r = subprocess.run(argv, capture_output=True, cwd=root, env=env)
                assert r.returncode == 0, r.stderr
                assert r.stdout == want, (cmd, raw, python_cli)
                if raw == "1" and not python_cli:
                    assert b"pass 1 restarts with full depth tables" in r.stderr, r.stderr
