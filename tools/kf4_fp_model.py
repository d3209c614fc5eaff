#!/usr/bin/env python3
"""Offline (CPU, numpy) model of the k-mer-space group filter kf4 (rkmh_amd/csrc/rk_device.hpp): candidates per read that a
bit-selection scheme passes to the exact map, at the shipped density, on C2-like reads.  The found k-mers are taken from the
references themselves (every window whose canonical hash is in its reference panel's sketches), which is what the device
enumeration finds up to a handful of chance collisions.  Used to vet a cheaper bit test BEFORE spending GPU time on it:
    python3 tools/kf4_fp_model.py [reads]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402

K = 16
CODE = np.zeros(256, np.uint64); CODE[ord("A")] = 0; CODE[ord("C")] = 1; CODE[ord("T")] = 2; CODE[ord("G")] = 3
M32 = np.uint64(0xFFFFFFFF)


def pack_windows(seq):
    """packed 2-bit k-mers of every window of an upper-case ACGT byte array (base i of the window in bits [2i, 2i+2)); -1 where invalid"""
    a = np.frombuffer(seq, np.uint8)
    c = CODE[a]
    valid = np.isin(a, np.frombuffer(b"ACGT", np.uint8))
    n = len(a) - K + 1
    x = np.zeros(n, np.uint64)
    ok = np.ones(n, bool)
    for i in range(K):
        x |= c[i:i + n] << np.uint64(2 * i)
        ok &= valid[i:i + n]
    return x, ok


def revcomp(x):
    r = np.zeros_like(x)
    for i in range(K):
        r |= (((x >> np.uint64(2 * i)) & np.uint64(3)) ^ np.uint64(2)) << np.uint64(2 * (K - 1 - i))
    return r


def sector(core, n):
    return (((core * np.uint64(0x85EBCA6B)) & M32) * np.uint64(n)) >> np.uint64(32)


def bits_scheme(name, x):
    C = np.uint64(0x9E3779B1)
    lo = (x * C) & M32
    hi = (x * C) >> np.uint64(32)
    one = np.uint64(1)
    f = lambda v, s: one << ((v >> np.uint64(s)) & np.uint64(31))  # noqa: E731
    if name == "shipped (lo 27,22,17)":
        return f(lo, 27) | f(lo, 22) | f(lo, 17)
    if name == "hi bytes 3,2,1":
        return f(hi, 24) | f(hi, 16) | f(hi, 8)
    if name == "lo bytes 3,2,1":
        return f(lo, 24) | f(lo, 16) | f(lo, 8)
    if name == "lo^lo>>16 bytes 3,2,1":
        g = lo ^ (lo >> np.uint64(16))
        return f(g, 24) | f(g, 16) | f(g, 8)
    if name == "lo>>3 bytes 3,2,1":
        g = lo >> np.uint64(3)
        return f(g, 24) | f(g, 16) | f(g, 8)
    if name == "lo rot: 27, 19, 11":
        return f(lo, 27) | f(lo, 19) | f(lo, 11)
    if name == "hi bytes 2,1,0":
        return f(hi, 16) | f(hi, 8) | f(hi, 0)
    if name == "mid (prod>>16) bytes 3,2,1":
        g = ((x * C) >> np.uint64(16)) & M32
        return f(g, 24) | f(g, 16) | f(g, 8)
    if name == "lo byte 3 + hi bytes 0,1":
        return f(lo, 24) | f(hi, 0) | f(hi, 8)
    if name == "lo byte 3 + hi bytes 0,2":  # = RK_KF4_MID (shipped from round 4)
        return f(lo, 24) | f(hi, 0) | f(hi, 16)
    if name == "lo bytes 2,3 + hi byte 0":
        return f(lo, 24) | f(lo, 16) | f(hi, 0)
    raise KeyError(name)


def main():
    nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    data = os.path.join(ROOT, "tests", "golden", "data")
    refs = orc.kseq_parse_file(os.path.join(data, "all_pave_ref.fa.gz"))
    seqs = [orc.to_upper(r[1]) for r in refs]
    rb, ro = orc.pack(seqs)
    sk, ln = orc.sketch_refs(rb, ro, [K], 1000, threads=8)
    keys = np.unique(np.concatenate([sk[i, :ln[i]] for i in range(len(refs))]))
    found = []
    for s in seqs:
        h = orc.calc_hashes(s, [K])
        x, ok = pack_windows(s)
        x, ok = x[:len(h)], ok[:len(h)]
        hit = ok & np.isin(h, keys)
        found.append(x[hit])
    found = np.unique(np.concatenate(found))
    both = np.unique(np.concatenate([found, revcomp(found)]))
    canon = np.unique(np.minimum(found, revcomp(found)))
    nsect = int(8 * len(canon) / 13)
    print("keys %d, found canonical k-mers %d, oriented %d, sectors %d" % (len(keys), len(canon), len(both), nsect))
    from rkmh_amd import synth
    qb, qo = synth.generate_reads(rb, ro, 0, nreads, read_len=150)
    L = 150
    R = np.frombuffer(qb[: nreads * L].tobytes(), np.uint8).reshape(nreads, L)
    cm = np.uint64((1 << (2 * (K - 3))) - 1)
    schemes = ["shipped (lo 27,22,17)", "hi bytes 3,2,1", "lo bytes 3,2,1", "lo^lo>>16 bytes 3,2,1", "lo>>3 bytes 3,2,1", "lo rot: 27, 19, 11", "hi bytes 2,1,0",
               "mid (prod>>16) bytes 3,2,1", "lo byte 3 + hi bytes 0,1", "lo byte 3 + hi bytes 0,2", "lo bytes 2,3 + hi byte 0"]
    # windows of all reads (drop_last_window: L - K windows), invalid ones dropped from the count
    nw = L - K
    X = np.zeros((nreads, nw), np.uint64)
    OK = np.ones((nreads, nw), bool)
    Cd = CODE[R]
    V = np.isin(R, np.frombuffer(b"ACGT", np.uint8))
    for i in range(K):
        X |= Cd[:, i:i + nw] << np.uint64(2 * i)
        OK &= V[:, i:i + nw]
    pos = np.arange(nw)
    g, j = pos // 4, pos % 4
    # the group's core = the 13-mer at read position 4 g + 3 (taken from the window that starts there, or from window 4 g shifted)
    W0 = np.zeros((nreads, nw), np.uint64)          # super-window start k-mer of the group (window 4 g)
    W0[:, :] = X[:, np.minimum(4 * g, nw - 1)]
    # bases 3..15 of window 4g are bits 6..31; for the last partial group use the true bases (same as the kernel reading past)
    core = (W0 >> np.uint64(6)) & cm
    # k-mer bits beyond the read for the last group's core do not matter for this estimate
    true_hit = np.isin(X, both) & OK
    print("true hits per read: %.2f" % (true_hit.sum() / nreads))
    for name in schemes:
        f4 = np.zeros((nsect, 4), np.uint64)
        for jj in range(4):
            c_ = (both >> np.uint64(2 * (3 - jj))) & cm
            np.bitwise_or.at(f4[:, jj], sector(c_, nsect).astype(np.int64), bits_scheme(name, both))
        b = bits_scheme(name, X)
        fw = f4[sector(core, nsect).astype(np.int64), j[None, :].repeat(nreads, 0)]
        cand = ((b & fw) == b) & OK
        assert not (true_hit & ~cand).any(), name
        print("%-28s candidates per read %.2f (false %.2f)" % (name, cand.sum() / nreads, (cand & ~true_hit).sum() / nreads))


def paired():
    """The paired layout: lanes 2i and 2i+1 (groups 2i, 2i+1 = eight consecutive windows) read the two halves of one 32-byte sector
    chosen by the 9-mer the eight windows share; a key is entered under 8 (group parity x alignment) x 2 orientations.  Prints the
    candidates per read by entries per sector and bits per entry."""
    nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    data = os.path.join(ROOT, "tests", "golden", "data")
    refs = orc.kseq_parse_file(os.path.join(data, "all_pave_ref.fa.gz"))
    seqs = [orc.to_upper(r[1]) for r in refs]
    rb, ro = orc.pack(seqs)
    sk, ln = orc.sketch_refs(rb, ro, [K], 1000, threads=8)
    keys = np.unique(np.concatenate([sk[i, :ln[i]] for i in range(len(refs))]))
    found = []
    for s_ in seqs:
        h = orc.calc_hashes(s_, [K])
        x, ok = pack_windows(s_)
        x, ok = x[:len(h)], ok[:len(h)]
        found.append(x[ok & np.isin(h, keys)])
    found = np.unique(np.concatenate(found))
    both = np.unique(np.concatenate([found, revcomp(found)]))
    from rkmh_amd import synth
    qb, qo = synth.generate_reads(rb, ro, 0, nreads, read_len=150)
    L = 150
    R = np.frombuffer(qb[: nreads * L].tobytes(), np.uint8).reshape(nreads, L)
    nw = L - K
    X = np.zeros((nreads, nw), np.uint64)
    OK = np.ones((nreads, nw), bool)
    Cd = CODE[R]
    V = np.isin(R, np.frombuffer(b"ACGT", np.uint8))
    for i in range(K):
        X |= Cd[:, i:i + nw] << np.uint64(2 * i)
        OK &= V[:, i:i + nw]
    true_hit = np.isin(X, both) & OK
    pos = np.arange(nw)
    slot = pos % 8                      # 4 a + j
    m18 = np.uint64((1 << 18) - 1)
    C = np.uint64(0x9E3779B1)
    one = np.uint64(1)

    def bits(x, nb):
        lo = (x * C) & M32
        hi = (x * C) >> np.uint64(32)
        f = lambda v, s_: one << ((v >> np.uint64(s_)) & np.uint64(31))  # noqa: E731
        b = f(lo, 24) | f(hi, 0) | f(hi, 16)
        if nb >= 4:
            b |= f(hi, 8)
        return b
    # the 9-mer of window x under slot s = its bases (7 - s) .. (15 - s)
    core_q = (X >> (np.uint64(2) * (np.uint64(7) - slot[None, :].astype(np.uint64)))) & m18
    print("true hits per read %.2f; oriented keys %d" % (true_hit.sum() / nreads, len(both)))
    m26 = np.uint64((1 << 26) - 1)
    # the 13-mer core of the window's own group (parity a = slot // 4, alignment j = slot % 4): bases (3 - j) .. (15 - j)
    jq = (slot % 4)[None, :].astype(np.uint64)
    core13_q = (X >> (np.uint64(2) * (np.uint64(3) - jq))) & m26
    for SUB in (1, 2, 4):          # 16-byte sub-sectors per lane inside the pair's line (line = 32 * SUB bytes)
        for nb in (3, 4):
            bq = bits(X, nb)
            bk = bits(both, nb)
            for mb in (2.0, 2.4, 3.0):
                nline = int(mb * 1048576 / (32 * SUB))
                f = np.zeros((nline, 2, SUB, 4), np.uint64)
                for s_ in range(8):
                    a_, j_ = s_ // 4, s_ % 4
                    c9 = (both >> np.uint64(2 * (7 - s_))) & m18
                    c13 = (both >> np.uint64(2 * (3 - j_))) & m26
                    sub = ((c13 * np.uint64(0xC2B2AE35)) & M32) >> np.uint64(30) if SUB == 4 else (((c13 * np.uint64(0xC2B2AE35)) & M32) >> np.uint64(31) if SUB == 2 else np.zeros_like(c13))
                    np.bitwise_or.at(f, (sector(c9, nline).astype(np.int64), a_, sub.astype(np.int64), j_), bk)
                subq = ((core13_q * np.uint64(0xC2B2AE35)) & M32) >> np.uint64(30) if SUB == 4 else (((core13_q * np.uint64(0xC2B2AE35)) & M32) >> np.uint64(31) if SUB == 2 else np.zeros_like(core13_q))
                fw = f[sector(core_q, nline).astype(np.int64), (slot // 4)[None, :].repeat(nreads, 0), subq.astype(np.int64), (slot % 4)[None, :].repeat(nreads, 0)]
                cand = ((bq & fw) == bq) & OK
                assert not (true_hit & ~cand).any()
                print("paired, line %3d B, %d bits, %.1f MB: candidates per read %.2f (false %.2f)"
                      % (32 * SUB, nb, mb, cand.sum() / nreads, (cand & ~true_hit).sum() / nreads))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "paired":
        paired()
    else:
        main()
