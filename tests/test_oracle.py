"""CPU-only: pins the oracle (oracle/rk_oracle.c) -- murmur3 against independent known-answer vectors,
the mkmh restatements against straightforward Python definitions, and the committed golden files."""
import json
import os

import numpy as np
import pytest

from helpers import golden, rand_dna


def test_murmur3_known_answers(orc, golden_dir):
    vec = json.load(open(os.path.join(golden_dir, "murmur3_kat.json")))["vectors"]
    assert len(vec) > 50
    for v in vec:
        assert orc.murmur3_x64_128(bytes.fromhex(v["key_hex"]), v["seed"]) == (v["h1"], v["h2"])
    # the two vectors quoted in SURVEY.md section 8c
    assert orc.murmur3_x64_128(b"hello", 0) == (0xcbd8a7b341bd9b02, 0x5b1e906a48ae1d19)
    assert orc.murmur3_x64_128(b"ACGTACGTACGTACGT", 42) == (4706917051267373191, 12844982669895470291)


def _rc(s):
    return bytes({65: 84, 84: 65, 67: 71, 71: 67}.get(c, c) for c in reversed(s))


def _fold(h1, h2, fold):
    w = [h1 & 0xffffffff, h1 >> 32, h2 & 0xffffffff, h2 >> 32]
    return {0: (w[0] << 32) | w[1], 1: h1, 2: (w[2] << 32) | w[1]}[fold]


@pytest.mark.parametrize("fold", [0, 1, 2])
@pytest.mark.parametrize("drop", [0, 1])
def test_calc_hashes_definition(orc, fold, drop):
    rng = np.random.default_rng(7)
    pol = orc.default_policy(fold=fold, drop_last_window=drop)
    for k in (5, 12, 16, 20, 31, 33):
        s = rand_dna(rng, 90, b"ACGTACGTACGTN")
        h = orc.calc_hashes(s, [k], pol)
        n = len(s) - k + (0 if drop else 1)
        assert len(h) == n
        for i in range(n):
            w = s[i:i + k]
            if any(c not in b"ACGT" for c in w):
                assert h[i] == 0
            else:
                f = _fold(*orc.murmur3_x64_128(w, 42), fold)
                r = _fold(*orc.murmur3_x64_128(_rc(w), 42), fold)
                assert int(h[i]) == min(f, r)
                assert orc.calc_hash(w, pol) == min(f, r)


def test_multi_k_concatenates(orc):
    s = b"ACGTTGCAAGGCTTAACCGGTTAAGGCC"
    a, b = orc.calc_hashes(s, [4]), orc.calc_hashes(s, [7])
    assert (orc.calc_hashes(s, [4, 7]) == np.concatenate([a, b])).all()


def test_short_sequence_is_empty(orc):
    assert len(orc.calc_hashes(b"ACGT", [16])) == 0
    assert len(orc.calc_hashes(b"", [16])) == 0
    assert len(orc.calc_hashes(b"ACGTACGTACGTACGT", [16])) == 0  # len-k windows (policy U3)
    assert len(orc.calc_hashes(b"ACGTACGTACGTACGT", [16], orc.default_policy(drop_last_window=0))) == 1


def test_to_upper_quirk(orc):
    assert orc.to_upper(b"acgtnACGTN") == b"ACGTNACGTN"
    assert orc.to_upper(b"[\\]^_`{|}~") == b"[<=>?@[\\]^"  # every char > 91 gets -32 (mkmh quirk)
    assert orc.to_upper(bytes([200, 255])) == bytes([200, 255])


def test_minhashes_definition(orc):
    rng = np.random.default_rng(3)
    h = rng.integers(1, 50, size=200, dtype=np.uint64)
    h[::7] = 0
    for S in (1, 10, 150, 500):
        want = np.sort(h[h != 0])[:S]
        assert (orc.minhashes(h, S) == want).all()
    assert len(orc.minhashes(np.zeros(5, np.uint64), 10)) == 0
    assert len(orc.minhashes(np.zeros(0, np.uint64), 10)) == 0


def test_materialised_intersection(orc):
    """A5f, the 7-argument hash_intersection of filter (equiv.hpp:308,340,364): (array, start, length) x 2, sketch size."""
    f = orc.hash_intersection
    assert list(f([9, 1, 2, 3], 1, 3, [2, 3, 4, 7], 0, 3, 10)) == [2, 3]
    assert list(f([5, 5, 5], 0, 3, [5, 5], 0, 2, 10)) == [5, 5]
    assert list(f([5, 5, 5], 0, 3, [5, 5], 0, 2, 1)) == [5]            # at most sketch_size matches
    assert list(f([0, 0, 5], 0, 3, [0, 5], 0, 2, 10)) == [5]
    assert list(f([1, 2], 0, 0, [1, 2], 0, 2, 10)) == []
    rng = np.random.default_rng(6)
    for _ in range(20):
        a = np.sort(rng.integers(1, 40, size=60, dtype=np.uint64))
        b = np.sort(rng.integers(1, 40, size=80, dtype=np.uint64))
        a0, b0 = int(rng.integers(0, 20)), int(rng.integers(0, 20))
        al, bl = int(rng.integers(0, 40)), int(rng.integers(0, 60))
        got = f(a, a0, al, b, b0, bl, 1000)
        sa, sb = a[a0:a0 + al], b[b0:b0 + bl]
        want = [v for v in np.unique(sa) for _ in range(min((sa == v).sum(), (sb == v).sum()))]
        assert list(got) == want
        assert len(got) == orc.hash_intersection_size(sa, sb)


def test_intersection_is_multiset_merge(orc):
    f = orc.hash_intersection_size
    assert f([1, 2, 3], [2, 3, 4]) == 2
    assert f([5, 5], [5]) == 1
    assert f([5], [5, 5]) == 1
    assert f([5, 5, 5], [5, 5]) == 2
    assert f([0, 0, 5], [0, 5]) == 1  # leading zeros skipped
    assert f([], [1]) == 0
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = np.sort(rng.integers(1, 40, size=60, dtype=np.uint64))
        b = np.sort(rng.integers(1, 40, size=80, dtype=np.uint64))
        want = sum(min((a == v).sum(), (b == v).sum()) for v in np.unique(a))
        assert f(a, b) == want


def test_argmax_diff_rules(orc):
    f = orc.argmax_diff
    assert f([0, 0, 0]) == (0, 0, 1)          # all zero: first ref, diff = 0 - (-1)
    assert f([3, 5, 5, 1]) == (1, 5, 2)       # first max wins; later equal score does not change diff
    assert f([5, 3, 5]) == (0, 5, 6)
    assert f([1, 2, 3]) == (2, 3, 1)
    assert f([2, 2]) == (0, 2, 3)


def test_stream_line_format(orc):
    assert orc.stream_line("R", "q", 5, 2, 100, 1000) == "R\tq\t5\t1000\t\t\n"
    assert orc.stream_line("R", "q", 5, 0, 100, 1000) == "R\tq\t5\t1000\t\tFAIL:DIFF\n"
    assert orc.stream_line("R", "q", 5, 2, 100, 1000, min_matches=200) == "R\tq\t5\t1000FAIL:DEPTH\tFAIL:MATCHES\t\n"


def test_counter_and_depth_path(orc):
    # -M path equals "mask then classify" computed by hand
    rng = np.random.default_rng(11)
    refs = [rand_dna(rng, 400) for _ in range(3)]
    reads = [refs[i % 3][10 * i:10 * i + 80] for i in range(12)] + [rand_dna(rng, 80)]
    rb, ro = orc.pack(refs)
    qb, qo = orc.pack(reads)
    sk, ln = orc.sketch_refs(rb, ro, [11], 50)
    out = orc.classify_stream(qb, qo, [11], 50, sk, ln, min_kmer_occ=2, counter_slots=10007)
    counts = {}
    allh = [orc.calc_hashes(r, [11]) for r in reads]
    for h in allh:
        for v in h:
            counts[int(v) % 10007] = counts.get(int(v) % 10007, 0) + 1
    for i, h in enumerate(allh):
        m = np.array([v if counts[int(v) % 10007] >= 2 else 0 for v in h], dtype=np.uint64)
        mins = orc.minhashes(m, 50)
        shared = [orc.hash_intersection_size(mins, sk[j, :ln[j]]) for j in range(3)]
        assert tuple(out[i]) == orc.argmax_diff(shared) + (len(mins),)


@pytest.mark.parametrize("tag", ["c1_hpv16_minion25", "zika_z1", "c2mini_pave", "c2mini_pave_k12_k16", "c2mini_pave_M2"])
def test_golden_files_match_oracle(orc, golden_dir, data_dir, tag):
    """The committed goldens are what the oracle produces today (regression pin)."""
    g = golden(golden_dir, tag)
    refs = orc.kseq_parse_file(os.path.join(data_dir, g["ref_file"]))
    if "reads_file" in g:
        reads = orc.kseq_parse_file(os.path.join(data_dir, g["reads_file"]))
        qn = [r[0].decode() for r in reads]
        qb, qo = orc.pack([r[1] for r in reads])
    else:
        from rkmh_amd import synth
        rb0, ro0 = orc.pack([r[1] for r in refs])
        qb, qo = synth.generate_reads(rb0, ro0, 0, 1000)
        qn = [n.decode() for n in synth.read_names(0, 1000)]
    rb, ro = orc.pack([r[1] for r in refs])
    rn = [r[0].decode() for r in refs]
    sk, ln = orc.sketch_refs(rb, ro, g["ks"], g["sketch_size"], threads=4)
    assert [int(x) for x in ln] == g["ref_sketch_lens"]
    assert [int(x) for x in sk[0, :8]] == g["first_sketch_hashes"]["ref0"]
    out = orc.classify_stream(qb, qo, g["ks"], g["sketch_size"], sk, ln, threads=4, **g["kwargs"])
    rows = [[qn[i], rn[out[i, 0]], int(out[i, 1]), int(out[i, 2]), int(out[i, 3])] for i in range(len(qn))]
    assert rows == g["rows"]


def test_c1_expectation(golden_dir):
    """C1 gives 25 lines, all naming HPV16."""
    g = golden(golden_dir, "c1_hpv16_minion25")
    assert len(g["rows"]) == 25
    assert all("HPV16" in r[1] for r in g["rows"])


def test_hash_set_intersection_size_policy_u13(orc):
    """U13 (mkmh function absent from the reference; call site rkmh.cpp:2673): distinct non-zero values present in both arrays."""
    a = np.array([0, 0, 3, 3, 5, 9, 9, 9, 12], dtype=np.uint64)
    b = np.array([0, 3, 4, 9, 9, 13], dtype=np.uint64)
    assert orc.hash_set_intersection_size(a, b) == 2            # {3, 9}: duplicates count once, 0 never
    assert orc.hash_intersection_size(a[2:], b[1:]) == 3        # the multiset merge of the stream path counts 9 twice
    assert orc.hash_set_intersection_size(a, np.zeros(0, np.uint64)) == 0


def test_sort_by_similarity_policy_u14(orc):
    """U14 (absent; call sites rkmh.cpp:2688, :2700): similarity = intersection / list size, descending, ties in list order."""
    h = np.array([1, 2, 3, 4, 5, 6], dtype=np.uint64)
    refs = [np.array([1, 2, 9, 10], dtype=np.uint64), np.array([3, 4, 5, 11, 12, 13, 14, 15], dtype=np.uint64),
            np.array([6, 20], dtype=np.uint64), np.zeros(0, np.uint64), np.array([1, 30], dtype=np.uint64)]
    names, sims, inter = orc.sort_by_similarity(h, ["a", "b", "c", "d", "e"], refs, "ref")
    assert names == ["a", "c", "e", "b", "d"] and inter == [2, 1, 1, 3, 0] and sims == [0.5, 0.5, 0.5, 0.375, 0.0]
    names, sims, inter = orc.sort_by_similarity(h, ["a", "b", "c", "d", "e"], refs, "read")
    assert names == ["b", "a", "c", "e", "d"] and inter == [3, 2, 1, 1, 0]


def test_hpv16_golden_matches_oracle(orc, golden_dir, data_dir):
    """oracle.hpv16 (restatement of rkmh.cpp:2366-2723) on the reference's bundled files == the committed golden."""
    import hashlib
    g = json.load(open(os.path.join(golden_dir, "hpv16_minion25.json")))
    types = orc.kseq_parse_file(os.path.join(data_dir, "all_pave_ref.fa.gz"))
    subs = orc.kseq_parse_file(os.path.join(data_dir, "new_refs.fa.gz"))
    reads = orc.kseq_parse_file(os.path.join(data_dir, g["reads_file"]))
    lines, tst, err = orc.hpv16([t[0] for t in types], [t[1] for t in types], [t[0] for t in subs], [t[1] for t in subs],
                                [r[0] for r in reads], [r[1] for r in reads], g["ks"], sim_den=g["sim_den"])
    assert lines == g["stdout_lines"] and err == g["stderr_tables"]
    assert hashlib.sha256(tst.encode()).hexdigest() == g["tst_sha256"] and tst[:80] == g["tst_first_80"]
    # every one of the reference's own HPV16 nanopore reads names the HPV16 type reference
    assert all("HPV16" in l.split("\t")[1] for l in lines)


def test_pin_tooling_reports_exactly_the_policy_of_the_probed_library(tmp_path):
    """tools/pin_compare.py, which decides U1-U12 the day mkmh's sources exist (tools/pin_from_mkmh.sh), exercised on stand-in
    probe files written from the oracle under KNOWN policies: exit 0 and no flips for the shipped defaults, exit 1 and exactly the
    differing constants otherwise -- so that the first real run is not the script's debut."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen, cmp_ = os.path.join(root, "tools", "pin_probes_from_oracle.py"), os.path.join(root, "tools", "pin_compare.py")

    def run(policy):
        f = str(tmp_path / "probes.json")
        subprocess.check_call([sys.executable, gen, f] + ["%s=%d" % kv for kv in policy.items()])
        r = subprocess.run([sys.executable, cmp_, f], capture_output=True)
        flips = {}
        for line in r.stdout.decode().splitlines():
            if line.startswith("  ") and " -> " in line:
                k, v = line.strip().split(" -> ")
                flips[k] = int(v)
        return r.returncode, flips, r.stdout.decode()

    rc, flips, out = run({})
    assert rc == 0 and flips == {} and "reproduce every probe" in out
    for policy in ({"fold": 1, "drop_last_window": 0, "counter_counts_zero": 0, "mask_strict_less": 0}, {"fold": 2}, {"drop_last_window": 0},
                   {"counter_counts_zero": 0}, {"mask_strict_less": 0}, {"fold": 1, "mask_strict_less": 0}):
        rc, flips, out = run(policy)
        assert rc == 1 and flips == policy, (policy, out)
        assert "AMBIGUOUS" not in out and "NO candidate" not in out


@pytest.mark.parametrize("preset", ["default", "mash"])
def test_cli_preset_goldens_match_oracle(orc, golden_dir, data_dir, preset):
    """tests/golden/cli_<preset>.json (what the binaries must print under --hash-policy <preset>) against the oracle run with the
    policy fields the file names; the two presets really differ (another fold and one more window per sequence)."""
    import hashlib
    g = json.load(open(os.path.join(golden_dir, "cli_%s.json" % preset)))
    pol = orc.default_policy(**g["policy_fields"])
    recs = orc.kseq_parse_file(os.path.join(data_dir, "hpv_16.fa.gz"))
    reads = orc.kseq_parse_file(os.path.join(data_dir, "minION25.fq.gz"))
    rb, ro = orc.pack([r[1] for r in recs])
    qb, qo = orc.pack([r[1] for r in reads])
    sk, ln = orc.sketch_refs(rb, ro, [12], 1000, policy=pol)
    out = orc.classify_stream(qb, qo, [12], 1000, sk, ln, policy=pol)
    want = "".join(orc.stream_line(recs[out[i, 0]][0].decode(), reads[i][0].decode(), out[i, 1], out[i, 2], out[i, 3], 1000) for i in range(len(reads)))
    assert want == g["classify_c1"]
    h = orc.calc_hashes(orc.to_upper(recs[0][1]), [12], pol)
    line = recs[0][0].decode() + "".join("\t%d" % v for v in h) + "\n"
    assert hashlib.sha256(line.encode()).hexdigest() == g["hash_hpv16_k12"]["sha256"] and len(h) == g["hash_hpv16_k12"]["n_hashes"]
    other = json.load(open(os.path.join(golden_dir, "cli_%s.json" % ("mash" if preset == "default" else "default"))))
    assert other["hash_hpv16_k12"]["sha256"] != g["hash_hpv16_k12"]["sha256"]
    assert abs(other["hash_hpv16_k12"]["n_hashes"] - g["hash_hpv16_k12"]["n_hashes"]) == 1
