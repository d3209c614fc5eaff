"""Resident-input timings of the fused kernel for workload variants beside the bench.py headline (C2):
more references, other k / several k, the -M two-pass path, other read lengths.  Prints one line per variant.
Usage (GPU box): python tools/bench_configs.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rkmh_amd
from rkmh_amd import api, synth

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(tag, refs_b, refs_o, ks, S, n, L, depth=None, steps=50):
    ctx = rkmh_amd.Context(0)
    t = time.time()
    ctx.set_references(refs_b, refs_o, ks, S)
    t_ref = time.time() - t
    qb, qo = synth.generate_reads_fast(refs_b, refs_o, 0, n, read_len=L, threads=16)
    d_b = torch.from_numpy(qb).to(dev)
    d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
    d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    cnt = None
    if depth is not None:
        cnt = api.Counter(ctx, 200000000)
        ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)
        st.synchronize()   # pass 1 must be complete before the keep bitmap is derived (rk_set_depth_filter also drains the device itself)
        ctx.set_depth_filter(cnt, depth)

    def step():
        ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=L, stream=st.cuda_stream)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    out = d_out.cpu().numpy()
    flagged = int((out[:, 0] < 0).sum())
    if depth is not None:
        e0.record()
        for _ in range(3):
            cnt.clear()
            ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        cms = e0.elapsed_time(e1) / 3
    else:
        cms = 0.0
    print("%-44s refs=%-4d k=%-10s S=%-5d L=%-5d n=%-8d classify %.3f ms = %.3f G reads/s%s  (rerouted rows %d, refs sketched in %.2f s)"
          % (tag, len(refs_o) - 1, ",".join(map(str, ks)), S, L, n, ms, n / ms / 1e6,
             "  count pass %.3f ms" % cms if depth is not None else "", flagged, t_ref), flush=True)
    ctx.set_depth_filter(None, 0) if depth is not None else None


pave = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = pave["bases"], pave["offsets"]
run("C2 headline", rb, ro, [16], 1000, 1000000, 150)
p300 = synth.synthetic_panel(300)
run("300 synthetic ~7.6 kb references", p300[0], p300[1], [16], 1000, 1000000, 150)
p1000 = synth.synthetic_panel(1000)
run("1000 synthetic references", p1000[0], p1000[1], [16], 1000, 1000000, 150)
p4000 = synth.synthetic_panel(4000)
run("4000 synthetic references", p4000[0], p4000[1], [16], 1000, 1000000, 150)
run("k=20 s=2000 (filter config)", rb, ro, [20], 2000, 1000000, 150)
run("k=20 s=1000", rb, ro, [20], 1000, 1000000, 150)
run("k=16 s=2000", rb, ro, [16], 2000, 1000000, 150)
run("k=21 s=1000", rb, ro, [21], 1000, 1000000, 150)
run("k=31 s=1000", rb, ro, [31], 1000, 1000000, 150)
run("k=24 s=1000 (runtime-k kernel)", rb, ro, [24], 1000, 1000000, 150)
run("k=12", rb, ro, [12], 1000, 1000000, 150)
run("multi-k 12,14,16", rb, ro, [12, 14, 16], 1000, 1000000, 150)
run("-M 2 (count pass + masked classify)", rb, ro, [16], 1000, 1000000, 150, depth=2)
run("100 bp reads", rb, ro, [16], 1000, 1000000, 100)
run("250 bp reads", rb, ro, [16], 1000, 1000000, 250)
run("1000 bp reads (200 k)", rb, ro, [16], 1000, 200000, 1000)
run("k=10", rb, ro, [10], 1000, 1000000, 150)
run("k=15", rb, ro, [15], 1000, 1000000, 150)
p2000 = synth.synthetic_panel(2000)
run("2000 synthetic references", p2000[0], p2000[1], [16], 1000, 1000000, 150)
run("1000 bp reads (fused, T=1)", rb, ro, [16], 1000, 200000, 1000)
