"""bin/rkmh stream end to end on a generated FASTQ (16 M reads of 150 bp = 5 GB, four times over with -f x 4) under several settings of
the device front end: block size, worker count, output mode.  Prints the wall clock and the marginal rate per setting.
Usage (GPU box): python tools/e2e_sweep.py [reads]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rkmh_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000000
L = 150
refs = api.parse_files([os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
tmp = tempfile.mkdtemp(prefix="rkmh_sweep_")
fq, tsv = os.path.join(tmp, "reads.fq"), os.path.join(tmp, "out.tsv")
with open(fq, "wb") as f:
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
        rec[:, 14 + L:14 + 2 * L] = ord("I"); rec[:, 14 + 2 * L] = 10
        f.write(rec.tobytes())
exe = os.path.join(ROOT, "bin", "rkmh")
ref = os.path.join(ROOT, "tests/golden/data/all_pave_ref.fa.gz")


def run(env, nfiles):
    t = time.perf_counter()
    r = subprocess.run([exe, "stream", "-r", ref] + ["-f", fq] * nfiles + ["-k", "16", "-s", "1000"], stdout=open(tsv, "wb"), stderr=subprocess.PIPE,
                       env=dict(os.environ, RKMH_TIMING="1", **env))
    dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr.decode()[-500:]
    return dt, r.stderr.decode()


if os.environ.get("SWEEP_AB"):
    # A/B with repetitions: slots per worker x workers; medians of three runs, output to a file and to /dev/null
    import statistics

    def run_to(env, nfiles, sink):
        if sink != os.devnull and os.path.exists(sink):
            os.remove(sink)                        # a fresh output file, opened before the clock starts
        fo = open(sink, "wb")
        t = time.perf_counter()
        r = subprocess.run([exe, "stream", "-r", ref] + ["-f", fq] * nfiles + ["-k", "16", "-s", "1000"], stdout=fo, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        dt = time.perf_counter() - t
        fo.close()
        assert r.returncode == 0, r.stderr.decode()[-500:]
        return dt
    for slots, ow in (("1", "1"), ("1", "3"), ("2", "3")):
        for w in ("4", "6", "8", "10"):
            env = {"RKMH_RAW_SLOTS": slots, "RKMH_RAW_WORKERS": w, "RKMH_OUT_WRITERS": ow}
            row = []
            for sink in (tsv, os.devnull):
                t1 = statistics.median(run_to(env, 1, sink) for _ in range(3))
                t4 = statistics.median(run_to(env, 4, sink) for _ in range(3))
                row.append((t1, t4, 3 * n / (t4 - t1) / 1e6))
            print("slots %s writers %s workers %2s: file 1x %.3f s 4x %.3f s marginal %.1f M reads/s | /dev/null 1x %.3f s 4x %.3f s marginal %.1f M reads/s"
                  % (slots, ow, w, row[0][0], row[0][1], row[0][2], row[1][0], row[1][1], row[1][2]), flush=True)
    settings = []
elif os.environ.get("SWEEP_SMALL"):
    settings = [{}, {}, {"RKMH_OUT_DIRECT": "1"}, {"RKMH_OUT_DIRECT": "0"}, {"RKMH_RAW_WORKERS": "6"}, {"RKMH_RAW_WORKERS": "10"}, {"RKMH_RAW_WORKERS": "12"},
                {"RKMH_RAW_BLOCK_KB": "8192"}, {"RKMH_RAW_BLOCK_KB": "8192", "RKMH_RAW_WORKERS": "12"}, {"RKMH_RAW_BLOCK_KB": "4096", "RKMH_RAW_WORKERS": "12"}, {"RKMH_RAW": "0"}]
else:
  settings = [{}] + [{"RKMH_RAW_BLOCK_KB": str(b), "RKMH_RAW_WORKERS": str(w)} for b in (8192, 16384, 32768, 65536) for w in (8, 12, 14, 16, 20)] + \
           [{"RKMH_OUT_DIRECT": "0"}, {"RKMH_RAW": "0"}]
for env in settings:
    best1 = min(run(env, 1)[0] for _ in range(2))
    dt4, err = run(env, 4)
    line = [l for l in err.splitlines() if "device front end:" in l]
    print("%-60s 1 file %.3f s, 4 files %.3f s, marginal %.1f M reads/s  %s" % (env, best1, dt4, 3 * n / (dt4 - best1) / 1e6, line[0].split(";", 1)[1].strip() if line else ""), flush=True)
for x in (fq, tsv):
    os.remove(x)
os.rmdir(tmp)
