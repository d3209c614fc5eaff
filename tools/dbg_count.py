"""Debug aid: the two forms of the -M count pass against a bincount of rk_hash_batch, one / two batches, one / two streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rkmh_amd
from rkmh_amd import api, synth
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 200000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
L = 100
dev = torch.device("cuda", 0)
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
c = rkmh_amd.Context(0)
c.set_references(rb, ro, [k], 1000)
def pad(b):
    return np.concatenate([b, np.zeros(64, np.uint8)])
batches, wants, dbs = [], [], []
for seed in (0, 1):
    qb, qo = synth.generate_reads_fast(rb, ro, seed * n, (seed + 1) * n, read_len=L, threads=8)
    h, ho = c.hash_batch(pad(qb), qo, [k])
    wants.append(torch.bincount(torch.from_numpy((h % np.uint64(slots)).astype(np.int64)).to(dev), minlength=slots).to(torch.int32))
    dbs.append((torch.from_numpy(pad(qb)).to(dev), torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)))
table = torch.zeros(slots, dtype=torch.int32, device=dev)
cnt = rkmh_amd.Counter(c, slots=slots, device_ptr=table.data_ptr())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
def run(name, plan, want):
    for form in ("1", "0"):
        os.environ["RKMH_COUNT_BINS"] = form
        res = []
        for rep in range(4):
            table.zero_(); torch.cuda.synchronize()
            for bi, st, sync in plan:
                c.count_device(dbs[bi][0].data_ptr(), dbs[bi][1].data_ptr(), n, cnt, stream=st.cuda_stream)
                if sync: torch.cuda.synchronize()
            torch.cuda.synchronize()
            d = table != want
            res.append(int(d.sum().item()))
            if res[-1] and rep == 0:
                idx = torch.nonzero(d)[:8, 0]
                print("   e.g. slots", idx.tolist(), "got", table[idx].tolist(), "want", want[idx].tolist())
        print("%-40s form %s: mismatching slots per repetition %s" % (name, form, res), flush=True)
run("batch 0 alone", [(0, s1, True)], wants[0])
run("batch 1 alone", [(1, s1, True)], wants[1])
run("two batches, one stream, sync between", [(0, s1, True), (1, s1, True)], wants[0] + wants[1])
run("two batches, one stream", [(0, s1, False), (1, s1, False)], wants[0] + wants[1])
run("two batches, two streams", [(0, s1, False), (1, s2, False)], wants[0] + wants[1])
# which reads' windows do the device forms not count?  (chunks of reads, then single reads)
os.environ["RKMH_COUNT_BINS"] = sys.argv[4] if len(sys.argv) > 4 else "0"
qb, qo = synth.generate_reads_fast(rb, ro, n, 2 * n, read_len=L, threads=8)
h, ho = c.hash_batch(pad(qb), qo, [k])
sl = torch.from_numpy((h % np.uint64(slots)).astype(np.int64)).to(dev)
d_b, d_o = dbs[1]
def check(a, b):
    table.zero_(); torch.cuda.synchronize()
    c.count_device(d_b.data_ptr(), d_o.data_ptr() + 4 * a, b - a, cnt, stream=s1.cuda_stream); torch.cuda.synchronize()
    want = torch.bincount(sl[int(ho[a]):int(ho[b])], minlength=slots).to(torch.int32)
    return int((table != want).sum().item())
CH = 5000
for a in range(0, n, CH):
    nd = check(a, min(n, a + CH))
    if nd:
        print("reads [%d, %d): %d slots differ" % (a, a + CH, nd), flush=True)
        for r in range(a, min(n, a + CH)):
            if check(r, r + 1):
                print("  read", r, "offset", int(qo[r]), bytes(qb[int(qo[r]):int(qo[r + 1])]), flush=True)
