"""Debug aid: multi-k classification on the GPU vs the CPU oracle, and its time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import oracle
import rkmh_amd
from rkmh_amd import api, synth
ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,16").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
refs = api.parse_files([os.path.join(ROOT, "tests", "golden", "data", "all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
qb, qo = synth.generate_reads_fast(rb, ro, 0, n)
ctx = rkmh_amd.Context(0)
ctx.set_references(rb, ro, ks, 1000)
sk, ln = ctx.get_reference_sketches()
dev = torch.device("cuda", 0)
d_b = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
f = lambda: ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=0)
f(); torch.cuda.synchronize()
got = d_out.cpu().numpy()
want = oracle.classify_stream(qb, qo, ks, 1000, sk, ln, threads=8)
bad = np.nonzero((got != want).any(axis=1))[0]
print("ks", ks, "kmer form:", ctx.kmer_form(), "mismatching rows:", len(bad), "of", n, " flagged(-2):", int((got[:, 0] == -2).sum()))
for i in bad[:10]:
    print(i, "got", got[i].tolist(), "want", want[i].tolist())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): f()
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("%.3f ms per %d reads" % (e0.elapsed_time(e1) / 20, n))
