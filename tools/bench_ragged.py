"""Reads of UNEQUAL length (trimmed short reads: 100-150 bp) through the fused kernel: tiles whose reads differ in length
walk byte positions under a start bitmap instead of the compact window mapping.  Resident inputs, 1 M reads."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rkmh_amd
from rkmh_amd import api, synth
n = 1000000
dev = torch.device("cuda", 0)
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
ctx = rkmh_amd.Context(0)
ctx.set_references(rb, ro, [16], 1000)
qb, qo = synth.generate_reads_fast(rb, ro, 0, n, threads=16)
rng = np.random.default_rng(1)
for name, lens in (("all 150", np.full(n, 150)), ("uniform 100..150", rng.integers(100, 151, size=n)),
                   ("90 % 150, 10 % 100..149", np.where(rng.random(n) < 0.9, 150, rng.integers(100, 150, size=n)))):
    offs = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=offs[1:])
    idx = np.repeat(np.arange(n, dtype=np.int64) * 150 - offs[:-1], lens) + np.arange(offs[-1], dtype=np.int64)
    b = np.concatenate([qb[idx], np.zeros(16, np.uint8)])
    d_b = torch.from_numpy(b).to(dev); d_o = torch.from_numpy(offs).to(torch.int32).to(dev)
    d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        def step(): ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=st.cuda_stream)
        for _ in range(60): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): step()
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("%-28s %.3f ms per 1 M reads = %.3f G reads/s = %.1f G bases/s (flagged rows %d)"
          % (name, ms, n / ms / 1e6, offs[-1] / ms / 1e6, int((d_out[:, 0] < 0).sum().item())), flush=True)
