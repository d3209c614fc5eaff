"""How fast is the -M count pass when its atomics stay inside a small table?  (the premise of a slot-partitioned count pass)
Usage: python tools/bench_count_slots.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rkmh_amd
from rkmh_amd import api, synth
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
n = 1000000
qb, qo = synth.generate_reads_fast(rb, ro, 0, n)
ctx = rkmh_amd.Context(0)
d_b = torch.from_numpy(qb).cuda(); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).cuda()
d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
st = torch.cuda.Stream()
for k in (16, 20):
    ctx.set_references(rb, ro, [k], 1000)
    for slots in (200000000, 25000000, 3125000, 781250, 97656):
        cnt = api.Counter(ctx, slots)
        for _ in range(3):
            ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(st)
            for _ in range(10):
                ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)
            e1.record(st)
        st.synchronize()
        cms = e0.elapsed_time(e1) / 10
        ctx.set_depth_filter(cnt, 2)
        for _ in range(3):
            ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=st.cuda_stream)
        st.synchronize()
        with torch.cuda.stream(st):
            e0.record(st)
            for _ in range(10):
                ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=st.cuda_stream)
            e1.record(st)
        st.synchronize()
        print("k=%d slots=%-10d table %7.1f MB: count pass %.3f ms, masked classify %.3f ms" % (k, slots, slots * 4 / 1e6, cms, e0.elapsed_time(e1) / 10), flush=True)
        ctx.set_depth_filter(None, 0)
        cnt.destroy()
