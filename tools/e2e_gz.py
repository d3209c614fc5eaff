"""bin/rkmh stream on compressed reads (src/rkmh.cpp:238-263 opens everything with gzopen): whole-process wall time and reads/s for the same
N reads as plain FASTQ, as BGZF (bgzip: independent members, inflated by the device front end's workers) and as ordinary single-member
gzip (one deflate stream: the sequential zlib scanner).  Usage (GPU box): [N=16000000] [GZ_N=2000000] python tools/e2e_gz.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rkmh_amd import api, synth

n = int(os.environ.get("N", "16000000"))
gz_n = int(os.environ.get("GZ_N", "2000000"))
L = 150
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
tmp = os.environ.get("TMPDIR", "/tmp")
fq, bg, gz = (os.path.join(tmp, "e2e_gz." + x) for x in ("fq", "bgzf.fq.gz", "plain.fq.gz"))
t = time.time()
with open(fq, "wb") as f, open(bg, "wb") as fb:
    for lo in range(0, n, 1000000):
        m = min(1000000, n - lo)
        qb, _ = synth.generate_reads_fast(rb, ro, lo, lo + m, read_len=L, threads=16)
        rec = np.empty((m, 11 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 11 + L] = 10; rec[:, 10] = 10
        idx = np.arange(lo, lo + m, dtype=np.int64)
        for d in range(9):
            rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
        rec[:, 11:11 + L] = qb[: m * L].reshape(m, L)
        rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
        q = np.random.default_rng(lo).integers(35, 75, size=(m, L), dtype=np.uint8)     # qualities that do not compress to nothing
        rec[:, 14 + L:14 + 2 * L] = q; rec[:, 14 + 2 * L] = 10
        raw = rec.tobytes()
        f.write(raw)
        img = synth.bgzf_compress(raw, level=1, threads=16)
        fb.write(img[:-28] if lo + m < n else img)     # (one end-of-file member, at the end)
        if lo == 0:
            import gzip
            with open(gz, "wb") as fg:
                fg.write(gzip.compress(raw[: gz_n * (14 + 2 * L + 1)] if gz_n < m else raw, 1))
print("generated %d reads: FASTQ %.2f GB, BGZF %.2f GB, plain gzip of the first %d reads %.2f GB; %.1f s" %
      (n, os.path.getsize(fq) / 1e9, os.path.getsize(bg) / 1e9, min(gz_n, 1000000), os.path.getsize(gz) / 1e9, time.time() - t), flush=True)
exe = os.path.join(ROOT, "bin", "rkmh")
ref = os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")
out = os.path.join(tmp, "e2e_gz.out")


def run(files, env=None):
    best, lines, stages = None, 0, []
    for _ in range(2):
        if os.path.exists(out):
            os.remove(out)
        fo = open(out, "wb")
        t0 = time.perf_counter()
        r = subprocess.run([exe, "stream", "-r", ref, "-k", "16", "-s", "1000"] + sum((["-f", x] for x in files), []), stdout=fo, stderr=subprocess.PIPE,
                           env=dict(os.environ, RKMH_TIMING="1", **(env or {})))
        dt = time.perf_counter() - t0
        fo.close()
        assert r.returncode == 0, r.stderr.decode()[-500:]
        if best is None or dt < best:
            best = dt
            stages = [l for l in r.stderr.decode().splitlines() if "device front end:" in l]
    import hashlib
    h = hashlib.sha256()
    with open(out, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk); lines += blk.count(b"\n")
    return best, lines, h.hexdigest(), stages


for tag, files, reads, env in (("plain FASTQ", [fq], n, {}), ("plain FASTQ x4", [fq] * 4, 4 * n, {}),
                               ("plain FASTQ, RKMH_RAW_MMAP=1", [fq], n, {"RKMH_RAW_MMAP": "1"}), ("plain FASTQ x4, RKMH_RAW_MMAP=1", [fq] * 4, 4 * n, {"RKMH_RAW_MMAP": "1"}),
                               ("BGZF (device inflate)", [bg], n, {"RKMH_BGZF_DEVICE": "1"}), ("BGZF x4 (device inflate)", [bg] * 4, 4 * n, {"RKMH_BGZF_DEVICE": "1"}),
                               ("BGZF (host inflate)", [bg], n, {}), ("BGZF x4 (host inflate)", [bg] * 4, 4 * n, {}),
                               ("single-member gzip", [gz], min(gz_n, 1000000), {}), ("single-member gzip x4", [gz] * 4, 4 * min(gz_n, 1000000), {})):
    dt, lines, dig, stages = run(files, env)
    assert lines == reads, (tag, lines, reads)
    print("%-34s %9d reads  wall %.3f s  %.1f M reads/s whole process  sha %s" % (tag, reads, dt, reads / dt / 1e6, dig[:12]), flush=True)
    for s in stages:
        print("    " + s.strip())
