import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle as orc
import rkmh_amd
rng = np.random.default_rng(3)
def rand_dna(rng, n): return bytes(b"ACGT"[i] for i in rng.integers(0, 4, n))
refs = [rand_dna(rng, 450) for _ in range(5)]
refs.append(rand_dna(rng, 20) * 24)
reads = []
for i in range(400):
    r = refs[i % 6]
    reads += [r[:400], r[100:150], r[200:250], r[300:350]] if i % 2 else [r[50:100], r[:430], r[10:60], r[5:45]]
rb, ro = orc.pack(refs); qb, qo = orc.pack(reads)
qb = np.concatenate([qb, np.zeros(64, np.uint8)]); rb = np.concatenate([rb, np.zeros(64, np.uint8)])
ctx = rkmh_amd.Context(0)
ctx.set_references(rb, ro, [16], 1000)
sk, ln = ctx.get_reference_sketches()
want = orc.classify_stream(qb, qo, [16], 1000, sk, ln, threads=8)
n = len(reads)
d_b = torch.from_numpy(qb).cuda(); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).cuda()
d_out = torch.empty((n, 4), dtype=torch.int32, device="cuda")
for hint in (150, 120, 200, 450):
    d_out.zero_()
    ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=hint, stream=0)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    ok = (out[:, 0] == -2) | (out == want).all(axis=1)
    print("hint", hint, "kmer", ctx.kmer_form(), "flagged", int((out[:,0]==-2).sum()), "of", n, "bad", int((~ok).sum()))
    print(out[:12].tolist()); print([len(r) for r in reads[:12]])
