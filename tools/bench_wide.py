"""k = 17 .. 20: the hash-space kernel (every window hashed: k_classify_tile) against the k-mer-space kernel on wide k-mers (rk_kmer.hip KT = 32) on one
resident 1 M-read batch, with the one-off cost of the 4^k enumeration beside it.  Usage: [KS=17,18,19,20] [S=1000] python tools/bench_wide.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rkmh_amd
from rkmh_amd import api, synth
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
n = 1000000
S = int(os.environ.get("S", "1000"))
qb, qo = synth.generate_reads_fast(rb, ro, 0, n)
d_b = torch.from_numpy(qb).cuda(); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).cuda()
st = torch.cuda.Stream()
os.environ["RKMH_KMER_ENUM_MAXK"] = "20"


def timed(ctx, d_out, reps=20):
    f = lambda: ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=st.cuda_stream)  # noqa: E731
    for _ in range(5):
        f()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(reps):
            f()
        e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps


for k in [int(x) for x in os.environ.get("KS", "17,18,19,20").split(",")]:
    outs = []
    for form in os.environ.get("FORMS", "hash-space,k-mer-space").split(","):
        ctx = rkmh_amd.Context(0)
        ctx._lib.rk_set_kmer_form(ctx._h, 1 if form == "k-mer-space" else 0)
        t = time.perf_counter()
        ctx.set_references(rb, ro, [k], S)
        setup = time.perf_counter() - t
        assert ctx.kmer_form()[0] == (form == "k-mer-space")
        d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        ms = timed(ctx, d_out)
        outs.append(d_out.cpu().numpy())
        print("k=%d s=%d %-12s %.3f ms per 1 M reads (%.2f G reads/s); set_references %.2f s" % (k, S, form, ms, n / ms / 1e6, setup), flush=True)
        ctx.close()
    assert len(outs) < 2 or (outs[0] == outs[1]).all(), "the two forms disagree at k = %d" % k
