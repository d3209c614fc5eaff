"""Timing sanity of the long-sequence path (C4-like): sketch a few synthetic chromosome-scale references."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import rkmh_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(1)
t = time.time()
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n, dtype=np.uint8)]
lens = [n // 2, n // 3, n - n // 2 - n // 3]
offs = np.zeros(4, dtype=np.uint64); offs[1:] = np.cumsum(lens)
pad = np.concatenate([bases, np.zeros(16, np.uint8)])
print("generated %d bases in %.1f s" % (n, time.time() - t))
c = rkmh_amd.Context(0)
for rep in range(2):
    t = time.time()
    sk, ln = c.sketch_batch(pad, offs, [20], 2000)
    dt = time.time() - t
    print("sketch 3 sequences (%.0f Mb total), k=20 s=2000: %.3f s  (%.1f M windows/s incl. H2D)" % (n / 1e6, dt, n / dt / 1e6), ln)
if n <= 30_000_000:
    import oracle
    wsk, wln = oracle.sketch_refs(bases, offs, [20], 2000, threads=16)
    print("oracle agrees:", bool((wsk == sk).all() and (wln == ln).all()))
