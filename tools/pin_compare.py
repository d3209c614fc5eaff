#!/usr/bin/env python3
"""Compares the probes of a REAL mkmh build (oracle/_ref/pin_probes.json, written by oracle/pin/pin_driver.cpp through
tools/pin_from_mkmh.sh) with the oracle under every candidate policy, and says which of the assumptions U1-U12 of SURVEY.md
section 8c / DESIGN.md section 0 hold.  Exit status 0 = the shipped defaults reproduce every probe (parity can be declared
pinned: commit the probes as tests/golden/mkmh_probes.json); 1 = some default must flip (the list is printed; --apply rewrites
nothing automatically that it cannot verify: it regenerates the goldens only when a single consistent policy was found).
A real mkmh cannot be probed in the build container (its sources are absent there); the script itself is exercised by
tests/test_oracle.py on probe files that tools/pin_probes_from_oracle.py writes from the oracle under known policies."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle as orc  # noqa: E402


def main():
    probes = json.load(open(sys.argv[1]))
    apply = "--apply" in sys.argv[2:]
    report, flips = [], {}

    # U1/U2 (fold, canonical = min over folded values) and U3 (window count): search the policy grid
    seq = probes["seq"].encode()
    want16 = [int(x) for x in probes["hashes_k16"]]
    matches = []
    for fold in (0, 1, 2):
        for drop in (1, 0):
            pol = orc.default_policy(fold=fold, drop_last_window=drop)
            got = [int(x) for x in orc.calc_hashes(orc.to_upper(seq), [16], pol)]
            if got == want16:
                matches.append((fold, drop))
    found = matches[0] if matches else None
    if len(matches) > 1:
        report.append("AMBIGUOUS: %d (fold, window rule) candidates reproduce calc_hashes: %s -- the probe sequence differs too little" % (len(matches), matches))
    d = orc.default_policy()
    if found is None:
        report.append("U1/U2/U3/U4: NO candidate (fold x window rule) reproduces calc_hashes -- read mkmh.cpp's calc_hashes and murmur call")
    else:
        report.append("U1 fold = %d, U3 drop_last_window = %d reproduce calc_hashes(k=16) incl. the N window and lower case (U2, U4 hold)" % found)
        if found[0] != d.fold:
            flips["fold"] = found[0]
        if found[1] != d.drop_last_window:
            flips["drop_last_window"] = found[1]
    pol = orc.default_policy(**flips)
    for kmer, val in probes["calc_hash"].items():
        if int(orc.calc_hash(kmer.encode(), pol)) != int(val):
            report.append("calc_hash(%s) differs under the policy found above: %s" % (kmer, val))
    if [int(x) for x in orc.calc_hashes(orc.to_upper(seq), [12, 16], pol)] != [int(x) for x in probes["hashes_k12_k16"]]:
        report.append("U5: several k are NOT concatenated per k in the order given (hashes_k12_k16 differs)")
    for key, ln in (("n_len15_k16", 15), ("n_len16_k16", 16)):
        if probes[key] != max(0, ln - 16 + (0 if pol.drop_last_window else 1)):
            report.append("U3: %s = %d (a sequence of %d bases): the short-sequence rule differs" % (key, probes[key], ln))
    up = bytes(range(1, 128))
    if list(orc.to_upper(up)) != probes["to_upper"]:
        bad = [i + 1 for i, (a, b) in enumerate(zip(orc.to_upper(up), probes["to_upper"])) if a != b]
        report.append("to_upper differs for bytes %s" % bad[:20])
    arr = np.array([9, 0, 5, 5, 3, 0, 7, 5], dtype=np.uint64)
    for S, key in ((4, "minhashes_S4"), (100, "minhashes_S100")):
        if [int(x) for x in orc.minhashes(arr.copy(), S)] != [int(x) for x in probes[key]]:
            report.append("U6: minhashes(S=%d) = %s, oracle %s" % (S, probes[key], list(orc.minhashes(arr.copy(), S))))
    a = np.array([0, 0, 5, 5, 5, 8], dtype=np.uint64); b = np.array([0, 5, 5, 9], dtype=np.uint64)
    if orc.hash_intersection_size(a, b) != probes["intersection_00555_8__0559"]:
        report.append("U7: hash_intersection_size([0,0,5,5,5,8],[0,5,5,9]) = %d, oracle %d" % (probes["intersection_00555_8__0559"], orc.hash_intersection_size(a, b)))
    if orc.hash_intersection_size(np.array([5, 5, 5], dtype=np.uint64), np.array([5, 5], dtype=np.uint64)) != probes["intersection_555__55"]:
        report.append("U7: multiset rule differs on [5,5,5] x [5,5]: %d" % probes["intersection_555__55"])
    # U12: does the counted calc_hashes increment the 0 sentinel?  the probe sequence holds one N => 16 zero hashes
    zero_counted = probes["counter_get_0"] > 0
    if bool(d.counter_counts_zero) != zero_counted:
        flips["counter_counts_zero"] = int(zero_counted)
    report.append("U12: the counted calc_hashes %s the 0 sentinel (get(0) = %d)" % ("counts" if zero_counted else "does not count", probes["counter_get_0"]))
    # U9: counts are 1: threshold 1 keeps everything under '<', nothing non-zero under '<='
    strict = probes["mask_min1_kept"] > 0
    if bool(d.mask_strict_less) != strict:
        flips["mask_strict_less"] = int(strict)
    report.append("U9: mask_by_frequency zeroes a hash when count %s min_occ" % ("<" if strict else "<="))
    print("\n".join(report))
    if not flips and not any("differs" in r or "NO candidate" in r or "AMBIGUOUS" in r for r in report):
        print("\nRESULT: the shipped policy defaults reproduce every probe.  Commit oracle/_ref/pin_probes.json as tests/golden/mkmh_probes.json,"
              "\nadd the oracle-vs-probes test, and change 'parity unpinned' to 'pinned by mkmh @ <commit>' in DESIGN.md section 0.")
        return 0
    print("\nRESULT: policy defaults that must flip (oracle/rk_oracle.c rko_default_policy, rkmh_amd/csrc/rk_api.hip rk_policy defaults):")
    for k, v in flips.items():
        print("  %s -> %d" % (k, v))
    if apply and found is not None:
        print("--apply: regenerate the goldens after editing the two defaults above:\n  python3 tests/golden/gen_golden.py && python3 tests/golden/gen_hpv16_golden.py && python3 -m pytest tests -q -m 'not gpu'")
    return 1


if __name__ == "__main__":
    sys.exit(main())
