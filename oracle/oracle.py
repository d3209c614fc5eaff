"""ctypes front-end of the CPU ORACLE (oracle/rk_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package rkmh_amd/.
PARITY UNPINNED: see oracle/rk_oracle.h (mkmh submodule absent; policy U1..U12 switchable).

Also holds a small pure-Python restatement of the kseq record grammar
(/root/reference/src/kseq.hpp:170-208) and of the stream/classify TSV line
(/root/reference/src/rkmh.cpp:887-893) used to check the product's parser / CLI.
"""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FOLD_SWAP32, FOLD_H1, FOLD_W2W1 = 0, 1, 2


class Policy(C.Structure):
    _fields_ = [("fold", C.c_int32), ("drop_last_window", C.c_int32),
                ("counter_counts_zero", C.c_int32), ("mask_strict_less", C.c_int32),
                ("freq_max_inclusive", C.c_int32), ("seed", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "librkoracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.rko_calc_hash.restype = C.c_uint64
        _LIB.rko_counter_new.restype = C.c_void_p
        _LIB.rko_counter_get.restype = C.c_int32
        _LIB.rko_max_threads.restype = C.c_int
    return _LIB


def default_policy(**kw):
    p = Policy()
    lib().rko_default_policy(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def murmur3_x64_128(key: bytes, seed: int):
    out = (C.c_uint64 * 2)()
    lib().rko_murmur3_x64_128(key, len(key), C.c_uint32(seed), out)
    return int(out[0]), int(out[1])


def to_upper(s: bytes) -> bytes:
    b = C.create_string_buffer(s, len(s))
    lib().rko_to_upper(b, len(s))
    return b.raw


def calc_hash(kmer: bytes, policy=None) -> int:
    policy = policy or default_policy()
    return int(lib().rko_calc_hash(kmer, len(kmer), C.byref(policy)))


def calc_hashes(seq: bytes, ks, policy=None) -> np.ndarray:
    """calc_hashes on an ALREADY upper-cased sequence (as the reference call sites pass it)."""
    policy = policy or default_policy()
    ks = np.asarray(ks, dtype=np.int32)
    out = C.POINTER(C.c_uint64)()
    n = C.c_int()
    lib().rko_calc_hashes(seq, len(seq), _p(ks, C.c_int), len(ks), C.byref(out), C.byref(n), C.byref(policy))
    r = np.ctypeslib.as_array(out, shape=(max(n.value, 0),)).copy() if n.value > 0 else np.zeros(0, np.uint64)
    lib().rko_free(out)
    return r.astype(np.uint64)


def minhashes(h: np.ndarray, S: int) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.uint64).copy()
    out = C.POINTER(C.c_uint64)()
    m = C.c_int()
    lib().rko_minhashes(_p(h, C.c_uint64), len(h), S, C.byref(out), C.byref(m))
    r = np.ctypeslib.as_array(out, shape=(max(m.value, 1),))[: m.value].copy()
    lib().rko_free(out)
    return r.astype(np.uint64)


def hash_intersection_size(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = C.c_int()
    lib().rko_hash_intersection_size(_p(a, C.c_uint64), len(a), _p(b, C.c_uint64), len(b), C.byref(out))
    return out.value


def hash_intersection(a, a_start, a_len, b, b_start, b_len, S) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out, n = C.POINTER(C.c_uint64)(), C.c_int()
    lib().rko_hash_intersection(_p(a, C.c_uint64), a_start, a_len, _p(b, C.c_uint64), b_start, b_len, S, C.byref(out), C.byref(n))
    r = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[: n.value].copy()
    lib().rko_free(out)
    return r.astype(np.uint64)


def argmax_diff(shared):
    s = np.ascontiguousarray(shared, dtype=np.int32)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().rko_argmax_diff(_p(s, C.c_int), len(s), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def pack(seqs):
    """list[bytes] -> (uint8 bases, uint64 offsets[n+1])"""
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return bases, offs


def sketch_refs(bases, offsets, ks, S, policy=None, threads=1, max_samples=None, counter_slots=200000000, distinct=False):
    policy = policy or default_policy()
    ks = np.asarray(ks, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    nref = len(offsets) - 1
    sk = np.zeros((nref, S), dtype=np.uint64)
    ln = np.zeros(nref, dtype=np.int32)
    if max_samples is None:
        lib().rko_sketch_refs(_p(bases, C.c_char), _p(offsets, C.c_uint64), nref, _p(ks, C.c_int), len(ks), S,
                              _p(sk, C.c_uint64), _p(ln, C.c_int32), C.byref(policy), threads)
    else:
        lib().rko_sketch_refs_maxsamples(_p(bases, C.c_char), _p(offsets, C.c_uint64), nref, _p(ks, C.c_int),
                                         len(ks), S, int(max_samples), C.c_uint64(counter_slots), 1 if distinct else 0,
                                         _p(sk, C.c_uint64), _p(ln, C.c_int32), C.byref(policy), threads)
    return sk, ln


def classify_stream(bases, offsets, ks, S, ref_sketches, ref_lens, policy=None, threads=1,
                    min_kmer_occ=None, counter_slots=200000000):
    """The literal loop rkmh.cpp:845-898 (or :904-948 when min_kmer_occ is given).
    Returns int32 [nreads,4] = (max_id, max_shared, diff, min_num)."""
    policy = policy or default_policy()
    ks = np.asarray(ks, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    ref_sketches = np.ascontiguousarray(ref_sketches, dtype=np.uint64)
    ref_lens = np.ascontiguousarray(ref_lens, dtype=np.int32)
    n = len(offsets) - 1
    out = np.zeros((n, 4), dtype=np.int32)
    if min_kmer_occ is None:
        lib().rko_classify_stream(_p(bases, C.c_char), _p(offsets, C.c_uint64), C.c_int64(n),
                                  _p(ks, C.c_int), len(ks), S, _p(ref_sketches, C.c_uint64),
                                  _p(ref_lens, C.c_int32), len(ref_lens), _p(out, C.c_int32),
                                  C.byref(policy), threads)
    else:
        lib().rko_classify_stream_depth(_p(bases, C.c_char), _p(offsets, C.c_uint64), C.c_int64(n),
                                        _p(ks, C.c_int), len(ks), S, _p(ref_sketches, C.c_uint64),
                                        _p(ref_lens, C.c_int32), len(ref_lens), int(min_kmer_occ),
                                        C.c_uint64(counter_slots), _p(out, C.c_int32), C.byref(policy), threads)
    return out


def max_threads():
    return lib().rko_max_threads()


# ---------------------------------------------------------------------------------------------
# kseq record grammar, restated from /root/reference/src/kseq.hpp:170-208 (byte-at-a-time).
# Returns list of (name, seq, qual_or_None).  A truncated-quality record (-2) ENDS parsing, as
# the reference's loop condition `kseq_read(seq) >= 0` (src/rkmh.cpp:251) does.
# ---------------------------------------------------------------------------------------------
def kseq_parse_bytes(data: bytes):
    recs = []
    pos, n = 0, len(data)
    last_char = 0

    def getc():
        nonlocal pos
        if pos >= n:
            return -1
        c = data[pos]
        pos += 1
        return c

    SPACE = b" \t\n\v\f\r"
    while True:
        if last_char == 0:
            while True:
                c = getc()
                if c == -1 or c == ord(">") or c == ord("@"):
                    break
            if c == -1:
                break
            last_char = c
        # name: up to first whitespace (ks_getuntil delimiter 0 = isspace), kseq.hpp:181
        if pos >= n:
            break
        start = pos
        while pos < n and data[pos] not in SPACE:
            pos += 1
        name = data[start:pos]
        c = data[pos] if pos < n else -1
        if pos < n:
            pos += 1
        if c != ord("\n") and c != -1:
            while pos < n and data[pos] != ord("\n"):  # comment, kseq.hpp:182
                pos += 1
            if pos < n:
                pos += 1
        seq = bytearray()
        while True:
            c = getc()
            if c == -1 or c == ord(">") or c == ord("+") or c == ord("@"):
                break
            if 33 <= c <= 126:  # isgraph, kseq.hpp:184
                seq.append(c)
        if c == ord(">") or c == ord("@"):
            last_char = c
        if c != ord("+"):
            recs.append((name, bytes(seq), None))
            if c == -1:
                break
            continue
        while True:  # skip rest of '+' line
            c = getc()
            if c == -1 or c == ord("\n"):
                break
        if c == -1:
            break  # -2: truncated
        qual = bytearray()
        while True:
            c = getc()
            if c == -1 or not (len(qual) < len(seq)):
                break
            if 33 <= c <= 127:
                qual.append(c)
        last_char = 0
        if len(qual) != len(seq):
            break  # -2 ends the loop
        recs.append((name, bytes(seq), bytes(qual)))
    return recs


def kseq_parse_file(path):
    with open(path, "rb") as f:
        head = f.read(2)
    opener = gzip.open if head == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        return kseq_parse_bytes(f.read())


def stream_line(ref_name, read_name, max_shared, diff, min_num, sketch_size, min_matches=-1, min_diff=0):
    """One stdout line of stream/classify: /root/reference/src/rkmh.cpp:887-893."""
    diff_filter = diff > min_diff
    depth_filter = min_num <= min_matches
    match_filter = max_shared < min_matches
    return "%s\t%s\t%d\t%d%s\t%s\t%s\n" % (
        ref_name, read_name, max_shared, sketch_size, "FAIL:DEPTH" if depth_filter else "",
        "FAIL:MATCHES" if match_filter else "", "" if diff_filter else "FAIL:DIFF")


# ---------------------------------------------------------------------------------------------
# main_filter (/root/reference/src/rkmh.cpp:996-1424) on top of the stream rows.  The sketches are the same
# (mask by counter, bottom-S); only the decision differs: classify_and_count_diff_filter
# (/root/reference/src/equiv.hpp:324-353) starts from max_shared = prev_best = 0 with an empty sample name.
# ---------------------------------------------------------------------------------------------
def filter_decision(row, min_matches=-1, min_diff=0):
    """row = (max_id, max_shared, diff, min_num) of the stream loop -> (ref_index or None, shared, diff_ok, passes)."""
    max_id, max_shared, diff, nmins = (int(x) for x in row)
    if max_shared <= 0:
        ref, shared, d = None, 0, 0
    else:
        ref, shared = max_id, max_shared
        d = diff - (1 if max_id == 0 else 0)   # stream's scan starts at -1, filter's at 0
    diff_ok = d > min_diff
    depth_filter = nmins <= 0
    match_filter = shared < min_matches
    return ref, shared, diff_ok, (not depth_filter) and (not match_filter) and diff_ok


def filter_record(name: bytes, seq_upper: bytes, qual: bytes) -> bytes:
    """stdout of a passing read, rkmh.cpp:1299-1302"""
    return b">" + name + b"\n" + seq_upper + b"\n+\n" + qual + b"\n"


def filter_stdin_line(name, ref_name, shared, union, nmins, diff_ok, min_matches=-1):
    """-i mode, rkmh.cpp:1397-1399"""
    return "Sample: %s\tResult: %s\t%d\t%d\t%s\t%s\t%s\n" % (
        name, ref_name, shared, union, "FAIL:DEPTH" if nmins <= 0 else "",
        "FAIL:MATCHES" if shared < min_matches else "", "" if diff_ok else "FAIL:DIFF")


# ---------------------------------------------------------------------------------------------
# main_call (/root/reference/src/rkmh.cpp:1455-1904), single-threaded semantics, restated literally
# (pure Python: small inputs only).  Returns the VCF text the reference prints with default flags.
# ---------------------------------------------------------------------------------------------
CALL_HEADER = ("##fileformat=VCF4.2\n##source=rkmh\n##reference=%s\n"
               "##INFO=<ID=KD,Number=1,Type=Integer,Description=\"Number of times call for specific kmer appears\">\n"
               "##INFO=<ID=MD,Number=1,Type=Integer,Description=\"Maximum depth found for the rescue kmer.\">\n"
               "##INFO=<ID=RD,Number=1,Type=Integer,Description=\"Average depth in region\">"
               "##INFO=<ID=OD,Number=1,Type=Integer,Description=\"Depth of original kmer at site before modification.\">\n")
_SNP_ALTS = {ord("A"): b"CTG", ord("C"): b"TGA", ord("T"): b"CGA", ord("G"): b"ACT"}   # rkmh.cpp:1634-1637


def call_rows(ref_names, ref_seqs, read_seqs, k, window_len=100, policy=None):
    """ref_seqs / read_seqs: raw bytes (upper-cased here as rkmh.cpp:1609,1614 do). -> sorted list of VCF row strings."""
    from collections import Counter
    policy = policy or default_policy()
    depth_map = Counter()
    for r in read_seqs:                                            # rkmh.cpp:1613-1622
        for h in calc_hashes(to_upper(r), [k], policy):
            depth_map[int(h)] += 1
    cc, cmax, cavg, corig = {}, {}, {}, {}
    d_window = []                                                  # thread-private, NOT reset between refs (:1769)
    for name, raw in zip(ref_names, ref_seqs):
        seq = to_upper(raw)
        hashes = calc_hashes(seq, [k], policy)
        for j in range(len(hashes)):
            depth = depth_map.get(int(hashes[j]), 0)
            d_window.append(depth)
            if len(d_window) > window_len:
                d_window.pop(0)
            avg_d = int(float(sum(d_window)) / float(len(d_window)))   # double -> int, :1791
            if depth < 0.5 * avg_d:                                # :1801
                ref = bytearray(seq[j:j + k])
                d_alt = seq[j - 1:j + k] if j > 0 else b""
                for alt_pos in range(k):                           # SNPs, :1807-1840
                    orig = ref[alt_pos]
                    for x in _SNP_ALTS.get(orig, b""):
                        alt = bytearray(ref)
                        alt[alt_pos] = x
                        alt_depth = depth_map.get(calc_hash(bytes(alt), policy), 0)
                        if (alt_depth >= 0.1 * avg_d) & (alt_depth > depth):
                            key = "%s\t%d\t.\t%s\t%s" % (name, j + alt_pos + 1, chr(orig), chr(x))
                            cc[key] = cc.get(key, 0) + 1
                            cavg[key] = max(avg_d, cavg.get(key, 0))
                            corig[key] = max(corig.get(key, 0), depth)
                            cmax[key] = max(cmax.get(key, 0), alt_depth)
                if j > 0:                                          # deletions, :1847-1865
                    for alt_pos in range(1, len(d_alt)):
                        orig = d_alt[alt_pos]
                        mod = d_alt[:alt_pos] + d_alt[alt_pos + 1:]
                        alt_depth = depth_map.get(calc_hash(mod, policy), 0)
                        if alt_depth > 0.9 * avg_d:
                            key = "%s\t%d\t.\t%s\t-" % (name, j + alt_pos + 1, chr(orig))
                            cc[key] = cc.get(key, 0) + 1
                            cavg[key] = max(cavg.get(key, 0), avg_d)
                            corig[key] = max(corig.get(key, 0), depth)
                            cmax[key] = max(cmax.get(key, 0), alt_depth)
    return ["%s\t99\tPASS\tKC=%d;MD=%d;RD=%d;OD=%d\n" % (key, cc[key], cmax[key], cavg[key], corig[key])
            for key in sorted(cc, key=lambda s: s.encode())]     # std::map<string,...>: byte-wise order


# ---- hpv16 (main_hpv16, /root/reference/src/rkmh.cpp:2366-2723) --------------------------------------------------------
# Two more mkmh functions absent from /root/reference appear on this path; what they do is restated from their call sites
# and isolated as named policies (UNVERIFIED, switchable -- DESIGN.md section 0):
#   U13 hash_set_intersection_size(alpha, alen, beta, blen, shared)   call rkmh.cpp:2673
#       = number of DISTINCT NON-ZERO values present in both ascending arrays ("set" as opposed to the multiset merge of
#         hash_intersection_size; 0 is the invalid-k-mer sentinel everywhere else in the reference, rkmh.cpp:1218,1260).
#   U14 sort_by_similarity(alpha, len, names, n, ref_hashes, ref_lens, out_names, out_sims, out_intersections)   :2688, :2700
#       = per reference i: inter_i = U13(alpha, ref_i); sim_i = inter_i / (double) DEN; all three output vectors ordered by
#         sim descending, ties in reference order (stable).  DEN: HPV16_SIM_DEN = "ref" (size of the reference list; a list of
#         size 0 gives sim 0) or "read" (len, the read's hash count).
HPV16_SIM_DEN = "ref"


def hash_set_intersection_size(a, b) -> int:
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    return int(len(np.intersect1d(a[a != 0], b[b != 0])))  # intersect1d works on the unique values of both


def sort_by_similarity(h, names, ref_hashes, sim_den=None):
    den = sim_den or HPV16_SIM_DEN
    inter = [hash_set_intersection_size(h, r) for r in ref_hashes]
    sims = []
    for x, r in zip(inter, ref_hashes):
        d = len(r) if den == "ref" else len(h)
        sims.append(float(x) / float(d) if d > 0 else 0.0)
    order = sorted(range(len(names)), key=lambda i: -sims[i])   # Python's sort is stable: ties keep reference order
    return [names[i] for i in order], [sims[i] for i in order], [inter[i] for i in order]


def _specific_sets(keys, hashes):
    """rkmh.cpp:2560-2596 / :2614-2645: union of the hashes of every sequence of a (sub)lineage as a std::set, then the
    chain of std::set_difference against every OTHER (sub)lineage in map order.  Returns (names in map order, sorted arrays)."""
    groups = {}
    for k, h in zip(keys, hashes):
        groups.setdefault(k, set()).update(int(x) for x in h)
    names = sorted(groups)            # std::map<char,...> / std::map<string,...> iterate in key order
    out = []
    for x in names:
        xdiff = set(groups[x])
        for y in names:
            if y != x:
                xdiff -= groups[y]
        out.append(np.array(sorted(xdiff), dtype=np.uint64))
    return names, out


def hpv16(type_names, type_seqs, sub_names, sub_seqs, read_names, read_seqs, ks, policy=None, min_kmer_occ=None,
          counter_slots=800000000, sim_den=None):
    """Returns (stdout lines, text of lineage_specific_hashes.<k>.tst, stderr table lines).  Sequences are upper-cased here as
    parse_fastas does (rkmh.cpp:227); names are the first header token."""
    policy = policy or default_policy()
    k0 = [int(ks[0])]
    type_hashes = [np.sort(calc_hashes(to_upper(s), k0, policy)) for s in type_seqs]     # :2546 + the in-place sort of minhashes :2547
    sub_hashes = [calc_hashes(to_upper(s), k0, policy) for s in sub_seqs]                 # :2553
    lin_names, lin_hashes = _specific_sets([chr(n[0]) for n in sub_names], sub_hashes)    # key[0], :2561
    sublin_names, sublin_hashes = _specific_sets([n[:2].decode() for n in sub_names], sub_hashes)  # substr(0,2), :2615
    tst = "".join("%s\t%s\n" % (n, "".join("%d\t" % int(x) for x in h)) for n, h in zip(lin_names, lin_hashes))   # :2598-2611
    err = ["Lineage specific kmer table created:"] + ["\t%s\t%d" % (n, len(h)) for n, h in zip(lin_names, lin_hashes)]
    err += ["Sublineage specific kmer table created:"] + ["\t%s\t%d" % (n, len(h)) for n, h in zip(sublin_names, sublin_hashes)]
    reads_h = [calc_hashes(to_upper(s), ks, policy) for s in read_seqs]                   # :2661 (every -k)
    if min_kmer_occ is not None:                                                          # :2524-2528 + :2663
        table = {}
        for h in reads_h:
            for x in h:
                if policy.counter_counts_zero or x != 0:
                    s = int(x) % counter_slots
                    table[s] = table.get(s, 0) + 1
        masked = []
        for h in reads_h:
            h = h.copy()
            for i, x in enumerate(h):
                c = table.get(int(x) % counter_slots, 0)
                if (c < min_kmer_occ) if policy.mask_strict_less else (c <= min_kmer_occ):
                    h[i] = 0
            masked.append(h)
        reads_h = masked
    lines = []
    for name, h in zip(read_names, reads_h):
        h = np.sort(h)                                                                    # :2666
        max_shared, max_id = -1, 0
        for j, th in enumerate(type_hashes):                                              # :2669-2679
            shared = hash_set_intersection_size(h, th)
            if shared > max_shared:
                max_shared, max_id = shared, j
        ln, ls, li = sort_by_similarity(h, lin_names, lin_hashes, sim_den)
        sn, ss, si = sort_by_similarity(h, sublin_names, sublin_hashes, sim_den)
        st = "%s\t%s\t%d/%d\t" % (name.decode(), type_names[max_id].decode(), max_shared, len(h))
        st += "".join("%s:%s;" % (a, "%g" % b) for a, b in zip(ln, ls)) + "\t"            # ostream << double = %g, precision 6
        st += "".join("%s:%s;" % (a, "%g" % b) for a, b in zip(sn, ss)) + "\t"
        st += "".join("%d;" % x for x in li) + "\t"
        st += "".join("%d;" % x for x in si) + "\n"
        lines.append(st)
    return lines, tst, err
