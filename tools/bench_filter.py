"""C4-like: `rkmh filter` of mixed reads (90 % synthetic genome, 10 % HPV) against a large synthetic 24-sequence genome
+ the HPV panel is NOT the reference: as in SURVEY C4 the reference is the genome; k=20, s=2000, -M depth filter.
Usage: python tools/bench_filter.py [genome_Mb=240] [reads=2500000]"""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rkmh_amd import api, synth
gmb = int(sys.argv[1]) if len(sys.argv) > 1 else 240
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2500000
rng = np.random.default_rng(3)
t = time.time()
chrom = gmb * 1000000 // 24
acgt = np.frombuffer(b"ACGT", np.uint8)
fa = "/tmp/genome_%d.fa" % gmb
goffs = [0]
with open(fa, "wb") as f:
    parts = []
    for c in range(24):
        s = acgt[rng.integers(0, 4, size=chrom, dtype=np.uint8)]
        f.write(b">chr%d\n" % (c + 1)); f.write(s.tobytes()); f.write(b"\n")
        parts.append(s); goffs.append(goffs[-1] + chrom)
gb = np.concatenate(parts + [np.zeros(16, np.uint8)]); go = np.array(goffs, dtype=np.uint64)
hpv = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
nh = n // 10
q1, o1 = synth.generate_reads_fast(gb, go, 0, n - nh, threads=16)
q2, o2 = synth.generate_reads_fast(hpv["bases"], hpv["offsets"], 0, nh, threads=16)
fq = "/tmp/mixed_%d.fq" % n
with open(fq, "wb") as f:
    L = 150; qual = b"I" * L
    for src, m, tag in ((q1, n - nh, b"g"), (q2, nh, b"v")):
        chunk = []
        for i in range(m):
            chunk.append(b"@%s%08d\n%s\n+\n%s\n" % (tag, i, bytes(src[i * L:(i + 1) * L]), qual))
            if len(chunk) == 100000: f.write(b"".join(chunk)); chunk = []
        f.write(b"".join(chunk))
print("inputs: genome %d Mb (%.0f MB fasta), %d reads (%.0f MB fastq) generated in %.0f s" % (gmb, os.path.getsize(fa) / 1e6, n, os.path.getsize(fq) / 1e6, time.time() - t), flush=True)
import hashlib
for args in (["-k", "20", "-s", "2000"], ["-k", "20", "-s", "2000", "-M", "2"]):
    digests = []
    for label, env in (("host parser (RKMH_RAW=0)", {"RKMH_RAW": "0"}), ("device front end", {})):
        out = "/tmp/filter_%d.out" % len(digests)
        if os.path.exists(out): os.remove(out)          # (deleting the previous output is not part of the run)
        fo = open(out, "wb")
        t = time.time()
        r = subprocess.run([os.path.join(ROOT, "bin/rkmh"), "filter", "-r", fa, "-f", fq] + args, stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, RKMH_TIMING="1", **env))
        dt = time.time() - t
        fo.close()
        print(r.stderr.decode()[-1100:])
        h = hashlib.sha256()
        kept = 0
        with open(out, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk); kept += blk.count(b">")
        digests.append(h.hexdigest())
        print("rkmh filter %s [%s]: rc %d, %.2f s wall = %.2f M reads/s end to end; %d reads pass" % (" ".join(args), label, r.returncode, dt, n / dt / 1e6, kept), flush=True)
        if r.returncode: print(r.stderr.decode()[-500:])
    print("outputs identical: %s" % (digests[0] == digests[1]), flush=True)
