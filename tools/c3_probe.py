"""Where the time of BASELINE config 3's panel (every bundled FASTA: 266 sequences, among them 61 near-identical Zika genomes and 21 HPV16 variants)
goes: the same kernel on reads drawn from the whole panel, from the PaVE part only, from the Zika family only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rkmh_amd
from rkmh_amd import api, synth
data = os.path.join(ROOT, "tests/golden/data")
files = ["all_pave_ref.fa.gz", "zika.refs.fa.gz", "dengue.fa.gz", "new_refs.fa.gz", "hpv_16.fa.gz", "zika.fa.gz", "yellow_fever.fa.gz", "hpv_16_allFasta.fa.gz"]
panel = api.parse_files([os.path.join(data, f) for f in files])
pb, po = panel["bases"], panel["offsets"]
K = int(os.environ.get("K", "16"))   # K=20: the same probe through the hash-space kernel (k_classify_tile)
ctx = rkmh_amd.Context(0)
ctx.set_references(pb, po, [K], 1000)
print("references", panel["nseq"], "k-mer-space form:", ctx.kmer_form())
n = 1000000
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
def sub(lo, hi):
    b0, b1 = int(po[lo]), int(po[hi])
    return np.concatenate([pb[b0:b1], np.zeros(16, np.uint8)]), (po[lo:hi + 1] - po[lo])
for tag, (lo, hi) in (("whole panel", (0, panel["nseq"])), ("PaVE part (182)", (0, 182)), ("Zika family (60)", (182, 242)), ("HPV16 variants (new_refs, 10)", (243, 253))):
    sb, so = sub(lo, hi)
    qb, qo = synth.generate_reads_fast(sb, so, 0, n, read_len=150, threads=16)
    d_b = torch.from_numpy(qb).cuda(); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).cuda()
    d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    f = lambda: ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=st.cuda_stream)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    print("%-32s %.3f ms per 1 M reads; mean max_shared %.1f, rerouted %d" % (tag, e0.elapsed_time(e1) / 20, out[:, 1].mean(), int((out[:, 0] < 0).sum())), flush=True)
