"""Debug aid: classify a synthetic batch on the GPU and print the rows that differ from the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle
import rkmh_amd
from rkmh_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
refs = api.parse_files([os.path.join(ROOT, "tests", "golden", "data", "all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=L)
ctx = rkmh_amd.Context(0)
ctx.set_references(rb, ro, [k], 1000)
sk, ln = ctx.get_reference_sketches()
import torch
dev = torch.device("cuda", 0)
d_b = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=L, stream=0)
torch.cuda.synchronize()
got = d_out.cpu().numpy()
want = oracle.classify_stream(qb, qo, [k], 1000, sk, ln, threads=8)
bad = np.nonzero((got != want).any(axis=1))[0]
print("kmer form:", ctx.kmer_form(), "mismatching rows:", len(bad), "of", n, " flagged(-2):", int((got[:, 0] == -2).sum()))
for i in bad[:24]:
    print(i, "tile", i // 6, "pos", i % 6, "got", got[i].tolist(), "want", want[i].tolist())
