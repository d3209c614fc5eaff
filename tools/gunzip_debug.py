"""GPU box: python tools/gunzip_debug.py [level] [nrec] [chunk_kb] -- one gzip file through rk_fastq_slot_load_gzip with RKMH_GZIP_DUMP; where the
device's text first differs from the true one, and in which chunk"""
import gzip, os, sys, re, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
level, nrec, ckb = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 30000, sys.argv[3] if len(sys.argv) > 3 else "32"
if os.environ.get("GUNZIP_DEBUG_CHILD") != "1":
    r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, GUNZIP_DEBUG_CHILD="1", RKMH_GZIP_DUMP="/tmp/gz_dump.bin", RKMH_BGZF_TIMING="1", RKMH_GZIP_CHUNK_KB=ckb),
                       capture_output=True)
    print(r.stdout.decode()[-3000:])
    err = r.stderr.decode()
    text = open("/tmp/gz_text.bin", "rb").read()
    got = open("/tmp/gz_dump.bin", "rb").read() if os.path.exists("/tmp/gz_dump.bin") else b""
    print("text", len(text), "device", len(got))
    a, b = np.frombuffer(text[:len(got)], np.uint8), np.frombuffer(got[:len(text)], np.uint8)
    bad = np.nonzero(a != b)[0]
    print("differing bytes:", len(bad), "first at", bad[:10])
    chunks = [tuple(map(int, m.groups())) for m in re.finditer(r"chunk (\d+): bits (\d+) \.\. (\d+), text (\d+) \+ (\d+), (\d+) entries, (\d+) literals", err)]
    print(len(chunks), "chunks")
    if len(bad):
        per = {}
        for c in chunks:
            n = int(((bad >= c[3]) & (bad < c[3] + c[4])).sum())
            if n: per[c[0]] = (n, c)
        for k in list(per)[:12]:
            n, c = per[k]
            rel = bad[(bad >= c[3]) & (bad < c[3] + c[4])] - c[3]
            print("chunk", k, c, "bad bytes", n, "first rel", rel[:6], "last rel", rel[-3:])
        i = int(bad[0])
        print("want", text[i - 40:i + 40]); print("got ", got[i - 40:i + 40])
    print("\n".join(l for l in err.splitlines() if "chunk " not in l)[-2500:])
    sys.exit(0)
from rkmh_amd import api
import rkmh_amd
rng = np.random.default_rng(level * 31 + nrec % 13 + 32)
recs = []
for i in range(nrec):
    L = int(rng.integers(30, 400))
    s = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[0.3, 0.2, 0.2, 0.29, 0.01]))
    q = bytes(rng.integers(33, 75, size=L, dtype=np.uint8))
    recs.append(b"@read%d/%d comment\n" % (i, i % 7) + s + b"\n+\n" + q + b"\n")
text = b"".join(recs)
open("/tmp/gz_text.bin", "wb").write(text)
open("/tmp/gz_t.fq.gz", "wb").write(gzip.compress(text, level))
if os.path.exists("/tmp/gz_dump.bin"):
    os.remove("/tmp/gz_dump.bin")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
refs = api.parse_files([os.path.join(root, "tests/golden/data/hpv_16.fa.gz")])
c = rkmh_amd.Context(0)
c.set_references(np.concatenate([refs["bases"], np.zeros(16, np.uint8)]), refs["offsets"], [16], 1000)
gz = api.Gzip.open("/tmp/gz_t.fq.gz")
slot = api.FastqSlot(c, max_bytes=32 << 20, device_text=True)
n = gz.plan(32 << 20)
for call in range(n):
    try:
        print("call", call, slot.load_gzip(gz, call))
    except Exception as e:
        print("call", call, "error", e)
        break
