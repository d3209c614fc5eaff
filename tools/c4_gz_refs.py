"""GPU box: python tools/c4_gz_refs.py [genome_mb=3100] [reads=1000000] -- BASELINE config 4's reference as it is distributed: genome.fa.gz (ordinary gzip, ONE
deflate stream).  bin/rkmh filter -k 20 -s 2000 against the plain FASTA (stripped on the GPU), against the .gz inflated on the GPU (rk_fasta_load_put_gzip) and
against the .gz read by zlib + the host parser (RKMH_RAW_REFS=0): wall clock of the whole command and the reference stage; the outputs must be the same bytes."""
import hashlib, os, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rkmh_amd import api, synth
genome_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
exe = os.path.join(ROOT, "bin", "rkmh")
fa, fagz, fq, out = "/tmp/c4g.fa", "/tmp/c4g.fa.gz", "/tmp/c4g.fq", "/tmp/c4g.out"
hg38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
        114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
scale = genome_mb * 1e6 / sum(hg38)
rng = np.random.default_rng(3)
lut = np.frombuffer(b"ACGT" * 64, dtype=np.uint8)
t0 = time.perf_counter()
parts, offs = [], [0]
with open(fa, "wb") as f:
    for c, full in enumerate(hg38):
        n = max(1000, int(full * scale))
        s_ = lut[np.frombuffer(rng.bytes(n), dtype=np.uint8)]
        f.write(b">chr%d synthetic\n" % (c + 1))
        for lo in range(0, n, 6000000):            # 60-column lines, as genomes are distributed
            f.write(np.concatenate([s_[lo:lo + 6000000].reshape(-1, 60) if (min(n, lo + 6000000) - lo) % 60 == 0 else s_[lo:lo + 6000000][: (min(n, lo + 6000000) - lo) // 60 * 60].reshape(-1, 60),
                                    np.full(((min(n, lo + 6000000) - lo) // 60, 1), 10, np.uint8)], axis=1).tobytes())
            rest = (min(n, lo + 6000000) - lo) % 60
            if rest:
                f.write(s_[min(n, lo + 6000000) - rest:min(n, lo + 6000000)].tobytes() + b"\n")
        parts.append(s_); offs.append(offs[-1] + n)
gb = np.concatenate(parts + [np.zeros(16, np.uint8)]); del parts
qb, qo = synth.generate_reads_fast(gb, np.array(offs, dtype=np.uint64), 0, nreads, read_len=150, threads=16)
synth.write_fastq(fq, qb, qo, synth.read_names(0, nreads))
del gb
# one deflate stream written by 16 threads: independent pieces, each ended by a sync flush, the last one finished (what pigz -i does)
text = open(fa, "rb").read()
P = 32 << 20
pieces = [text[i:i + P] for i in range(0, len(text), P)]
def comp(a):
    i, piece = a
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return co.compress(piece) + co.flush(zlib.Z_FINISH if i == len(pieces) - 1 else zlib.Z_SYNC_FLUSH)
with ThreadPoolExecutor(16) as ex:
    raw = list(ex.map(comp, enumerate(pieces)))
with open(fagz, "wb") as f:
    f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
    for r in raw:
        f.write(r)
    f.write((zlib.crc32(text) & 0xFFFFFFFF).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little"))
fabz = "/tmp/c4g.bgzf.fa.gz"
with open(fabz, "wb") as f:       # and bgzip'd (independent 64 KB members)
    for i in range(0, len(text), 256 << 20):
        img = synth.bgzf_compress(text[i:i + (256 << 20)], level=6, threads=16)
        f.write(img[:-28] if i + (256 << 20) < len(text) else img)
del text, raw, pieces
print("genome %.2f GB of FASTA, %.2f GB as gzip, %d reads: made in %.0f s" % (os.path.getsize(fa) / 1e9, os.path.getsize(fagz) / 1e9, nreads, time.perf_counter() - t0))

def settle():
    for _ in range(80):
        if not any(p.isdigit() and os.path.exists("/proc/%s/comm" % p) and open("/proc/%s/comm" % p).read().strip() == "rkmh" for p in os.listdir("/proc")):
            return
        time.sleep(0.05)

def run(ref, env):
    best = None
    for _ in range(2):
        settle()
        if os.path.exists(out):
            os.remove(out)
        with open(out, "wb") as fo:
            t = time.perf_counter()
            r = subprocess.run([exe, "filter", "-r", ref, "-f", fq, "-k", "20", "-s", "2000", "--no-kmer-cache"], stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, RKMH_TIMING="1", **env))
            d = time.perf_counter() - t
        assert r.returncode == 0, r.stderr[-400:]
        if best is None or d < best[0]:
            best = (d, [l for l in r.stderr.decode().splitlines() if "references" in l])
    return best[0], hashlib.sha256(open(out, "rb").read()).hexdigest()[:16], best[1]

for label, ref, env in (("genome.fa, stripped on the GPU", fa, {}), ("genome.fa.gz, inflated on the GPU", fagz, {}), ("genome.fa.gz, zlib + host parser", fagz, {"RKMH_RAW_REFS": "0"}),
                        ("genome.fa.gz (bgzip), inflated on the GPU", fabz, {}), ("genome.fa.gz (bgzip), zlib + host parser", fabz, {"RKMH_RAW_REFS": "0"})):
    d, h, st = run(ref, env)
    print("%-42s wall %.2f s  output %s" % (label, d, h))
    for l in st:
        print("      " + l)
