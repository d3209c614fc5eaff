import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import rkmh_amd
from rkmh_amd import api
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for i in range(3):
    t=time.perf_counter(); refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")]); print("parse %.1f ms" % ((time.perf_counter()-t)*1e3))
for env in ("0", "1"):
    os.environ["RKMH_GZ_BLOCKS"] = env
    t=time.perf_counter(); refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")]); print("parse RKMH_GZ_BLOCKS=%s %.1f ms" % (env, (time.perf_counter()-t)*1e3))
rb, ro = refs["bases"], refs["offsets"]
ctx = rkmh_amd.Context(0)
t=time.perf_counter(); ctx.set_references(rb, ro, [16], 1000); print("set_references, no cache, first %.1f ms" % ((time.perf_counter()-t)*1e3))
t=time.perf_counter(); ctx.set_references(rb, ro, [16], 1000); print("set_references, no cache, second %.1f ms" % ((time.perf_counter()-t)*1e3))
ctx.set_kmer_cache("/tmp/t.kmers")
for i in range(4):
    t=time.perf_counter(); ctx.set_references(rb, ro, [16], 1000); print("set_references %.1f ms (cache state %d)" % ((time.perf_counter()-t)*1e3, ctx.kmer_cache_state()))
sk, ln = ctx.get_reference_sketches()
for i in range(3):
    t=time.perf_counter(); ctx.set_reference_sketches(sk, ln, [16], 1000); print("set_reference_sketches %.1f ms" % ((time.perf_counter()-t)*1e3))
