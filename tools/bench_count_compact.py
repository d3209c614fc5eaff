"""Pass 1 of -M on one resident 1 M-read batch: the full table (slot-partitioned form, rk_count.hip) against the compact depth map
(rk_counter_create_compact: only the slots of index keys).  Usage: [K=16] [SLOTS=200000000] [REPS=10] python tools/bench_count_compact.py"""
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
reps = int(os.environ.get("REPS", "10"))
qb, qo = synth.generate_reads_fast(rb, ro, 0, n)
ctx = rkmh_amd.Context(0)
d_b = torch.from_numpy(qb).cuda(); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).cuda()
st = torch.cuda.Stream()
ctx.set_references(rb, ro, [k], S)
for slots in [int(x) for x in os.environ.get("SLOTS", "200000000,10000000").split(",")]:
    for compact in (False, True):
        cnt = api.Counter(ctx, slots, compact=compact)
        f = lambda: ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)  # noqa: E731
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
        print("k=%d s=%d slots=%-10d %-8s count pass %.3f ms per 1 M reads (%d entries, %.1f MB)" %
              (k, S, slots, "compact" if compact else "full", e0.elapsed_time(e1) / reps, cnt.entries, cnt.entries * 4 / 1e6), flush=True)
        cnt.destroy()
