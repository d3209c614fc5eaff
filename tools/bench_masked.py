"""The masked classification of -M (pass 2, rkmh.cpp:911-948) on one resident 1 M-read batch: time per launch at several table sizes.
Usage: [SLOTS=200000000,...] [REPS=10] [K=16] [BOUNDS=-1,0,1,6] python tools/bench_masked.py
BOUNDS: rk_set_min_num_bound values (-1 = exact min_num: every window looked up by slot)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rkmh_amd
from rkmh_amd import api, synth
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
n = 1000000
k = int(os.environ.get("K", "16"))
S = int(os.environ.get("S", "1000"))
qb, qo = synth.generate_reads_fast(rb, ro, 0, n)
ctx = rkmh_amd.Context(0)
d_b = torch.from_numpy(qb).cuda(); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).cuda()
d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
st = torch.cuda.Stream()
reps = int(os.environ.get("REPS", "10"))
ctx.set_references(rb, ro, [k], S)
def timed(f):
    for _ in range(3):
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


for slots in [int(x) for x in os.environ.get("SLOTS", "200000000,10000000,1000000").split(",")]:
    cnt = api.Counter(ctx, slots)
    ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)
    st.synchronize()
    ctx.set_depth_filter(cnt, 2)
    for bound in [int(x) for x in os.environ.get("BOUNDS", "-1,0,1,6").split(",")]:
        ctx.set_min_num_bound(bound)
        ms = timed(lambda: ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=st.cuda_stream))
        print("k=%d s=%d slots=%-10d bound=%-3d masked classify %.3f ms per 1 M reads (keep bitmap %.1f MB)" % (k, S, slots, bound, ms, slots / 8e6), flush=True)
    ctx.set_min_num_bound(-1)
    ctx.set_depth_filter(None, 0)
    cnt.destroy()
