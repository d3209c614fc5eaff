"""Writes a synthetic FASTQ (the C2 read generator; fixed-width records "@r%09d", 150 bp, quality 'I'): python tools/make_fastq.py <out.fq> [reads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rkmh_amd import api, synth
out, n, L = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4000000, 150
files = ["all_pave_ref"] if not os.environ.get("C3_PANEL") else ["all_pave_ref", "zika.refs", "dengue", "new_refs", "hpv_16", "zika", "yellow_fever", "hpv_16_allFasta"]
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/%s.fa.gz" % f) for f in files])
rb, ro = refs["bases"], refs["offsets"]
with open(out, "wb") as f:
    for lo in range(0, n, 1000000):
        m = min(1000000, n - lo)
        qb, _ = synth.generate_reads_fast(rb, ro, lo, lo + m, read_len=L, threads=16)
        rec = np.empty((m, 11 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 11 + L] = 10; rec[:, 10] = 10
        idx = np.arange(lo, lo + m, dtype=np.int64)
        for d in range(9):
            rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
        rec[:, 11:11 + L] = qb[: m * L].reshape(m, L)
        rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
        rec[:, 14 + L:14 + 2 * L] = ord("I"); rec[:, 14 + 2 * L] = 10
        f.write(rec.tobytes())
