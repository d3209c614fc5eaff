"""End-to-end timing of bin/rkmh stream on a generated FASTQ (parser + host pipeline + formatting), and a big-batch bench."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rkmh_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, read_len=L, threads=16)
path = "/tmp/reads_%d_%d.fq" % (n, L)
t = time.time()
with open(path, "wb") as f:
    qual = b"+" * L
    chunk = []
    for i in range(n):
        chunk.append(b"@r%09d\n%s\n+\n%s\n" % (i, bytes(qb[i * L:(i + 1) * L]), qual))
        if len(chunk) == 100000:
            f.write(b"".join(chunk)); chunk = []
    f.write(b"".join(chunk))
print("wrote", path, os.path.getsize(path) / 1e6, "MB in", round(time.time() - t, 1), "s")
for rep in range(3):
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "bin/rkmh"), "stream", "-r", os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz"),
                        "-f", path, "-k", "16", "-s", "1000"], stdout=open("/dev/null" if rep == 0 else "/tmp/out.tsv", "wb"), stderr=subprocess.PIPE, env=dict(os.environ, RKMH_TIMING="1"))
    print(r.stderr.decode())
    dt = time.time() - t
    print("bin/rkmh stream: rc", r.returncode, "%.2f s" % dt, "=> %.2f M reads/s end to end" % (n / dt / 1e6), "lines", sum(1 for _ in open("/tmp/out.tsv")) if rep else "(to /dev/null)")

# cross-check the CLI's lines against the resident-device entry point on the same reads
import numpy as np
import rkmh_amd
ctx = rkmh_amd.Context(0)
ctx.set_references(refs["bases"], refs["offsets"], [16], 1000)
m = min(n, 2000000)
res = ctx.classify(qb[:m * L + 16], qo[:m + 1])
bad = 0
with open("/tmp/out.tsv", "rb") as f:
    for i in range(m):
        parts = f.readline().rstrip(b"\n").split(b"\t")
        if parts[0] != refs["names"][int(res[i, 0])] or int(parts[2]) != int(res[i, 1]) or parts[1] != b"r%09d" % i:
            bad += 1
print("cross-check of the first", m, "lines against Context.classify:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
