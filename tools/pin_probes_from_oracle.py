#!/usr/bin/env python3
"""Writes the probe file of oracle/pin/pin_driver.cpp (same keys, same inputs) FROM THE ORACLE under a given policy, e.g.
    python3 tools/pin_probes_from_oracle.py /tmp/probes.json fold=1 drop_last_window=0
It stands in for a real mkmh build in the self-test of tools/pin_compare.py (tests/test_oracle.py): a stand-in "mkmh" whose
choices are known must make pin_compare report exactly those choices.  Test infrastructure only; it pins nothing."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle as orc  # noqa: E402

KMERS = ["ACGTACGTACGTACGT", "AAAAAAAAAAAAAAAA", "TTTTTTTTTTTTTTTT", "ACGTTGCATGCAACGA", "GATTACAGATTACAGA", "ACGTACGTACGT",
         "ACGTACGTACGTACGTACGT", "TGCATGCATGCATGCATGCATGCATGCATGC"]
SEQ = b"ACGTTGCATGCAACGATTACAGGANCTTGACCTAGGATCCAacgtTTGACA"
CSEQ = b"ACGTTGCATGCAACGATTACAGGANCTTGACCTAGGATCCA"
SLOTS = 1000003


def probes(pol):
    p = {"calc_hash": {k: str(int(orc.calc_hash(k.encode(), pol))) for k in KMERS}}
    h16 = orc.calc_hashes(orc.to_upper(SEQ), [16], pol)
    h2 = orc.calc_hashes(orc.to_upper(SEQ), [12, 16], pol)
    p.update(seq=SEQ.decode(), seq_len=len(SEQ), n_k16=len(h16), hashes_k16=[str(int(x)) for x in h16],
             n_k12_k16=len(h2), hashes_k12_k16=[str(int(x)) for x in h2],
             n_len15_k16=len(orc.calc_hashes(b"ACGTACGTACGTACG", [16], pol)), n_len16_k16=len(orc.calc_hashes(b"ACGTTGCATGCAACGA", [16], pol)))
    p["to_upper"] = list(orc.to_upper(bytes(range(1, 128))))
    arr = np.array([9, 0, 5, 5, 3, 0, 7, 5], dtype=np.uint64)
    for S, key in ((4, "minhashes_S4"), (100, "minhashes_S100")):
        m = orc.minhashes(arr.copy(), S)
        p[key] = [str(int(x)) for x in m]
        p[key + "_n"] = len(m)
    p["intersection_00555_8__0559"] = int(orc.hash_intersection_size(np.array([0, 0, 5, 5, 5, 8], dtype=np.uint64), np.array([0, 5, 5, 9], dtype=np.uint64)))
    p["intersection_555__55"] = int(orc.hash_intersection_size(np.array([5, 5, 5], dtype=np.uint64), np.array([5, 5], dtype=np.uint64)))
    # the counted calc_hashes + mask_by_frequency, as the driver calls them (HASHTCounter of 1000003 slots: slot = hash % slots)
    h = [int(x) for x in orc.calc_hashes(orc.to_upper(CSEQ), [16], pol)]
    table = {}
    for x in h:
        if x != 0 or pol.counter_counts_zero:
            table[x % SLOTS] = table.get(x % SLOTS, 0) + 1
    get = lambda x: table.get(x % SLOTS, 0)  # noqa: E731

    def kept(min_occ):
        return sum(1 for x in h if x != 0 and not (get(x) < min_occ if pol.mask_strict_less else get(x) <= min_occ))
    p.update(counter_n=len(h), counter_get_0=get(0), counter_get_first=get(h[0]) if h else -1, mask_min1_kept=kept(1), mask_min2_kept=kept(2))
    return p


def main():
    out = sys.argv[1]
    kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
    with open(out, "w") as f:
        json.dump(probes(orc.default_policy(**kw)), f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
