"""Long-read timing (general path: k_hash_tiles + k_sort_intersect, reads longer than the fused kernel's 1528 B)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rkmh_amd
from rkmh_amd import api, synth
pave = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = pave["bases"], pave["offsets"]
ctx = rkmh_amd.Context(0)
for L, n, k, S in ((7000, 25000, 12, 1000), (7000, 25000, 16, 1000), (3000, 60000, 16, 1000), (1400, 100000, 16, 1000)):
    ctx.set_references(rb, ro, [k], S)
    qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=16)
    for rep in range(3):
        t = time.time()
        out = ctx.classify(qb, qo)
        dt = time.time() - t
    import torch
    d_b = torch.from_numpy(qb).cuda()
    d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).cuda()
    d_out = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.time()
        ctx.classify_device_all(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=L,
                                stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        dr = time.time() - t
    same = bool((d_out.cpu().numpy() == out).all())
    print("L=%d n=%d k=%d: host path %.3f s = %.2f G bases/s (PCIe included); resident %.4f s = %.1f G bases/s (rows equal: %s); frac hits>=5: %.3f"
          % (L, n, k, dt, n * L / dt / 1e9, dr, n * L / dr / 1e9, same, float((out[:, 1] >= 5).mean())), flush=True)
