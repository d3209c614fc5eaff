"""The -M count pass (pass 1, rkmh.cpp:904-910) in its two device forms, 1 M reads of 150 bp, tables of several sizes.
Usage: [SLOTS=200000000,...] [FORMS=0,1] [REPS=10] python tools/bench_count_forms.py [k ...]"""
import os, sys
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
st = torch.cuda.Stream()
reps = int(os.environ.get("REPS", "10"))
for k in [int(a) for a in sys.argv[1:]] or [16, 20]:
    ctx.set_references(rb, ro, [k], 1000)
    for slots in [int(x) for x in os.environ.get("SLOTS", "200000000,25000000,10000000").split(",")]:
        row = []
        for form in os.environ.get("FORMS", "0,1").split(","):
            os.environ["RKMH_COUNT_BINS"] = form
            cnt = api.Counter(ctx, slots)
            for _ in range(3):
                ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                e0.record(st)
                for _ in range(reps):
                    ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=st.cuda_stream)
                e1.record(st)
            st.synchronize()
            row.append(e0.elapsed_time(e1) / reps)
            cnt.destroy()
        print("k=%d slots=%-10d: %s ms per 1 M reads (forms %s; 0 = one atomic per window, 1 = slot-partitioned)" % (k, slots, ", ".join("%.3f" % r for r in row), os.environ.get("FORMS", "0,1")), flush=True)
os.environ.pop("RKMH_COUNT_BINS", None)
