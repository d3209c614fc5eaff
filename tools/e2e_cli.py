"""End-to-end timing of bin/rkmh stream on a generated FASTQ (parser + host pipeline + formatting), and a big-batch bench."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rkmh_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, threads=16)
path = "/tmp/reads_%d.fq" % n
t = time.time()
with open(path, "wb") as f:
    L = 150
    qual = b"+" * L
    chunk = []
    for i in range(n):
        chunk.append(b"@r%09d\n%s\n+\n%s\n" % (i, bytes(qb[i * L:(i + 1) * L]), qual))
        if len(chunk) == 100000:
            f.write(b"".join(chunk)); chunk = []
    f.write(b"".join(chunk))
print("wrote", path, os.path.getsize(path) / 1e6, "MB in", round(time.time() - t, 1), "s")
for rep in range(2):
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "bin/rkmh"), "stream", "-r", os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz"),
                        "-f", path, "-k", "16", "-s", "1000"], stdout=open("/tmp/out.tsv", "wb"), stderr=subprocess.PIPE, env=dict(os.environ, RKMH_TIMING="1"))
    print(r.stderr.decode())
    dt = time.time() - t
    print("bin/rkmh stream: rc", r.returncode, "%.2f s" % dt, "=> %.2f M reads/s end to end" % (n / dt / 1e6), "lines", sum(1 for _ in open("/tmp/out.tsv")))
