"""Resident-input timing of plain classification for a list of k-mer sizes (C2 references, s = 1000, 1 M reads of 150 bp).
Usage: python tools/bench_multik.py 12,14,16 [16 ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rkmh_amd
from rkmh_amd import api, synth
refs = api.parse_files([os.path.join(ROOT, "tests", "golden", "data", "all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
n, L = 1000000, 150
dev = torch.device("cuda", 0)
c = rkmh_amd.Context(0)
qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=16)
d_b = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
for arg in sys.argv[1:] or ["12,14,16"]:
    ks = [int(x) for x in arg.split(",")]
    S = int(os.environ.get("BENCH_S", "1000"))
    c.set_references(rb, ro, ks, S)
    f = lambda: c.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=L, stream=st.cuda_stream)
    for _ in range(50): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("k=%s s=%d RKMH_KF4_ENTRIES=%s: %.3f ms per 1 M reads (k-mer-space form: %s)" % (arg, S, os.environ.get("RKMH_KF4_ENTRIES", "auto"), e0.elapsed_time(e1) / 50, c.kmer_form()[0]), flush=True)
