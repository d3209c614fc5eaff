"""One large synthetic panel through the fused kernel (resident inputs); for ablation runs (RKMH_DBG with an RK_ABLATE build)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rkmh_amd
from rkmh_amd import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = 1000000
dev = torch.device("cuda", 0)
pb, po = synth.synthetic_panel(R)
import time
ctx = rkmh_amd.Context(0)
t0 = time.time()
ctx.set_references(pb, po, [16], 1000)
t_set = time.time() - t0
qb, qo = synth.generate_reads_fast(pb, po, 0, n, threads=16)
d_b = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
def step(): ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=150, stream=st.cuda_stream)
for _ in range(100): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): step()
e1.record(); torch.cuda.synchronize()
print("R=%d RKMH_DBG=%s: %.3f ms  (set_references %.2f s)" % (R, os.environ.get("RKMH_DBG", "0"), e0.elapsed_time(e1) / 50, t_set))
