"""C5 at full scale: rkmh call on 1000x coverage of HPV16 (52 700 x 150 bp reads with planted variants), k=12."""
import os, sys, time, subprocess, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
import test_gpu_parity as T
tmp = pathlib.Path("/tmp/callbench"); tmp.mkdir(exist_ok=True)
rec, reads, fa, fq = T._call_fixture(orc, os.path.join(ROOT, "tests/golden/data"), tmp, cov=1000)
print("reads:", len(reads))
for rep in range(2):
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "bin/rkmh"), "call", "-r", str(fa), "-f", str(fq), "-k", "12"], capture_output=True)
    dt = time.time() - t
    print("bin/rkmh call: rc %d, %.2f s wall, %d VCF rows" % (r.returncode, dt, sum(1 for l in r.stdout.decode().splitlines() if not l.startswith("#"))))
t = time.time()
rows = orc.call_rows([rec[0].decode()], [rec[1]], reads[:5270], 12, 100)
print("oracle (python, literal main_call) on 10 %% of the reads: %.1f s" % (time.time() - t))
