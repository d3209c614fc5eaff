"""Stress for the -M count pass: the same batch is counted REPS times into a large (default 200 M-slot) table and a
weighted checksum of the table is compared across repetitions.  Every increment is an atomic, so the checksum must
never change.  Written to demonstrate the stale-prefetch bug fixed in rk_classify.hip: with the
pre-fix kernel some repetitions differ, with the shipped one none do.

    python tools/ab_count_race.py [reads] [read_len] [slots] [reps]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rkmh_amd
from rkmh_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 200000000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40

dev = torch.device("cuda", 0)
ctx = rkmh_amd.Context(0)
refs = api.parse_files([os.path.join(ROOT, "tests", "golden", "data", "all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
ctx.set_references(rb, ro, [16], 1000)
qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=16)
d_b = torch.from_numpy(qb).to(dev)
d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
table = torch.zeros(slots, dtype=torch.int32, device=dev)
w = (torch.arange(slots, device=dev, dtype=torch.int64) % 1000003) + 1
cnt = rkmh_amd.Counter(ctx, slots=slots, device_ptr=table.data_ptr())
sums = []
for r in range(reps):
    table.zero_()
    torch.cuda.synchronize()
    ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt)
    ctx.synchronize()
    torch.cuda.synchronize()
    sums.append((int(table.sum(dtype=torch.int64).item()), int((table.to(torch.int64) * w).sum().item())))
vals, counts = np.unique(np.array([s[1] for s in sums]), return_counts=True)
mode = vals[np.argmax(counts)]
bad = [i for i, s in enumerate(sums) if s[1] != mode]
print("reads=%d len=%d slots=%d reps=%d: total increments %s; repetitions whose checksum differs from the mode: %d %s"
      % (n, L, slots, reps, sorted(set(s[0] for s in sums)), len(bad), bad[:10]))
sys.exit(1 if bad else 0)
