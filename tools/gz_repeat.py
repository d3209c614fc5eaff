"""GPU box: python tools/gz_repeat.py [reps] -- the bench's compressed-input comparison repeated: 4 M reads as plain FASTQ, as BGZF (device route) and as ordinary
gzip (device route), one file and four, every run started on an idle GPU; medians and spreads of the walls and of the marginal rates.
(One bench line holds one sample of each -- and the samples move by +-30 % between runs on this pool.)"""
import os, subprocess, sys, time, statistics, gzip
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rkmh_amd import api, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
exe, ref = os.path.join(ROOT, "bin", "rkmh"), os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")
n, L = 4000000, 150
refs = api.parse_files([ref])
fq, bg, sg, out = "/tmp/rep.fq", "/tmp/rep.bgzf.fq.gz", "/tmp/rep.single.fq.gz", "/tmp/rep.out"
with open(fq, "wb") as fo, open(bg, "wb") as fb:
    for lo in range(0, n, 1000000):
        m = 1000000
        qb, _ = synth.generate_reads_fast(refs["bases"], refs["offsets"], lo, lo + m, read_len=L, threads=16)
        rec = np.empty((m, 11 + L + 3 + L + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 11 + L] = 10; rec[:, 10] = 10
        idx = np.arange(lo, lo + m, dtype=np.int64)
        for d in range(9):
            rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
        rec[:, 11:11 + L] = qb[: m * L].reshape(m, L)
        rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
        rec[:, 14 + L:14 + 2 * L] = np.random.default_rng(lo).integers(35, 75, size=(m, L), dtype=np.uint8); rec[:, 14 + 2 * L] = 10
        raw = rec.tobytes()
        fo.write(raw)
        img = synth.bgzf_compress(raw, level=1, threads=16)
        fb.write(img[:-28] if lo + m < n else img)
        if lo == 0:
            open(sg, "wb").write(gzip.compress(raw, 1))

def settle():
    for _ in range(80):
        if not any(p.isdigit() and open("/proc/%s/comm" % p).read().strip() == "rkmh" for p in os.listdir("/proc") if os.path.exists("/proc/%s/comm" % p)):
            return
        time.sleep(0.05)

def wall(files, env=None):
    settle()
    if os.path.exists(out):
        os.remove(out)
    with open(out, "wb") as fo:
        t = time.perf_counter()
        r = subprocess.run([exe, "stream", "-r", ref, "-k", "16", "-s", "1000"] + sum((["-f", f] for f in files), []), stdout=fo, stderr=subprocess.PIPE,
                           env=dict(os.environ, **(env or {})))
        d = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-300:]
    return d

rows = {"plain": (fq, n), "BGZF, device": (bg, n), "gzip, device": (sg, 1000000)}
# GZ_REPEAT_WORKERS="3,4": the device-text worker counts to compare, interleaved run by run (one box, one page cache, one afternoon)
variants = [w for w in os.environ.get("GZ_REPEAT_WORKERS", "").split(",") if w] or [""]
res = {(v, k): ([], []) for v in variants for k in rows}
for rep in range(reps):
    for v in variants:
        env = {"RKMH_BGZF_DEVICE_WORKERS": v} if v else None
        for k, (f, _) in rows.items():
            res[(v, k)][0].append(wall([f], env)); res[(v, k)][1].append(wall([f] * 4, env))
for v in variants:
    if v:
        print("== RKMH_BGZF_DEVICE_WORKERS=%s" % v)
    for k, (f, nr) in rows.items():
        w1, w4 = res[(v, k)]
        marg = sorted(3 * nr / (b - a) / 1e6 for a, b in zip(w1, w4) if b > a) or [float("nan")]
        print("%-14s 1 file: median %.3f s (%.3f .. %.3f); 4 files: median %.3f s (%.3f .. %.3f); marginal M reads/s: median %.1f (%.1f .. %.1f); from the median walls %.1f"
              % (k, statistics.median(w1), min(w1), max(w1), statistics.median(w4), min(w4), max(w4), statistics.median(marg), marg[0], marg[-1],
                 3 * nr / (statistics.median(w4) - statistics.median(w1)) / 1e6))
