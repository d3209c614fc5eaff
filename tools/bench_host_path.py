"""PCIe-inclusive rk_classify_batch from page-locked and from pageable host buffers (C2, 4 M reads).  Usage: python tools/bench_host_path.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, rkmh_amd
from rkmh_amd import api, synth
refs = api.parse_files([os.path.join(ROOT, "tests", "golden", "data", "all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
n, L = int(os.environ.get("N", "4000000")), 150
ctx = rkmh_amd.Context(0)
ctx.set_references(rb, ro, [16], 1000)
qb, _ = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=16)
hb = np.concatenate([qb[: n * L], np.zeros(16, np.uint8)])
ho = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
pb, po = api.pinned_array(hb.shape, np.uint8), api.pinned_array((n, 4), np.int32)
pb.array[:] = hb
out = np.zeros((n, 4), np.int32)
for name, f in (("page-locked", lambda: ctx.classify(pb.array, ho, out=po.array)), ("pageable", lambda: ctx.classify(hb, ho, out=out))):
    f(); best = 1e9
    for _ in range(5):
        t = time.perf_counter(); f(); best = min(best, time.perf_counter() - t)
    print("%-12s RKMH_CHUNK_READS=%s: %.2f ms for %d reads = %.1f M reads/s = %.1f GB/s" % (name, os.environ.get("RKMH_CHUNK_READS", "default"), best * 1e3, n, n / best / 1e6, n * (L + 4) / best / 1e9), flush=True)
assert (out == po.array).all()
