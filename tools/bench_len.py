"""Resident-input timing of the fused kernel for one read length (C2 references, k=16, s=1000).  Usage: python tools/bench_len.py <L> [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rkmh_amd
from rkmh_amd import api, synth
L = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
refs = api.parse_files([os.path.join(ROOT, "tests", "golden", "data", "all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
dev = torch.device("cuda", 0)
c = rkmh_amd.Context(0); K = int(os.environ.get("BENCH_K", "16")); c.set_references(rb, ro, [K], 1000)
qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=16)
d_b = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
f = lambda: c.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=L, stream=st.cuda_stream)
for _ in range(100): f()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("L=%d RKMH_TILE_T=%s: %.3f ms per %d reads" % (L, os.environ.get("RKMH_TILE_T", "auto"), e0.elapsed_time(e1) / 50, n))
