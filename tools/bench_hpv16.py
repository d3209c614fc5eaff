"""Wall time of `bin/rkmh hpv16` on N synthetic 150 bp reads drawn from the HPV16 sublineage genomes (tests/golden/data/new_refs.fa.gz).
Usage (GPU box): python tools/bench_hpv16.py [nreads]"""
import gzip, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "golden", "data")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
seqs = [b"".join(l.strip() for l in rec.split(b"\n")[1:]).upper() for rec in gzip.open(os.path.join(D, "new_refs.fa.gz")).read().split(b">")[1:]]
rng = np.random.default_rng(1)
out = bytearray()
for i in range(n):
    s = seqs[int(rng.integers(0, len(seqs)))]
    a = int(rng.integers(0, len(s) - 150))
    out += b">r%d\n" % i + s[a:a + 150] + b"\n"
p = "/tmp/hpv16_bench.fa"
open(p, "wb").write(out)
for rep in range(2):
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "bin", "rkmh"), "hpv16", "-f", p, "-R", D, "-k", "16"], capture_output=True, cwd="/tmp",
                       env=dict(os.environ, RKMH_TIMING="1"))
    dt = time.time() - t
    assert r.returncode == 0, r.stderr[-500:]
    print("hpv16: %d reads in %.2f s = %.0f reads/s; %d output lines" % (n, dt, n / dt, r.stdout.count(b"\n")))
    print("  " + " | ".join(l for l in r.stderr.decode().splitlines() if "kmer table" not in l and not l.startswith("\t"))[:400])
