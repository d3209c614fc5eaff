"""Long-read timing (general path: k_hash_tiles + k_sort_intersect, reads longer than the fused kernel's 1528 B)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rkmh_amd
from rkmh_amd import api, synth
pave = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = pave["bases"], pave["offsets"]
ctx = rkmh_amd.Context(0)
for L, n, k, S in ((7000, 25000, 12, 1000), (7000, 25000, 16, 1000), (3000, 60000, 16, 1000), (1400, 100000, 16, 1000)):
    ctx.set_references(rb, ro, [k], S)
    qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=16)
    for rep in range(3):
        t = time.time()
        out = ctx.classify(qb, qo)
        dt = time.time() - t
    print("L=%d n=%d k=%d: %.3f s = %.1f k reads/s = %.2f G bases/s (host path, PCIe included); frac hits>=5: %.3f"
          % (L, n, k, dt, n / dt / 1e3, n * L / dt / 1e9, float((out[:, 1] >= 5).mean())), flush=True)
