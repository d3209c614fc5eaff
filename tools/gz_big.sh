#!/bin/bash
# Usage (GPU box): [N=24000000] bash tools/gz_big.sh -- ONE BGZF file of N reads through bin/rkmh stream, host / device inflate: wall, main-loop time
cd ${GRAFT_REPO_ROOT:-.}
N=${N:-24000000}
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
rb, ro = refs["bases"], refs["offsets"]
n, L = $N, 150
with open("/tmp/big.fq.gz", "wb") as fb:
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
        rec[:, 14 + L:14 + 2 * L] = np.random.default_rng(lo).integers(35, 75, size=(m, L), dtype=np.uint8); rec[:, 14 + 2 * L] = 10
        img = synth.bgzf_compress(rec.tobytes(), level=1, threads=16)
        fb.write(img[:-28] if lo + m < n else img)
PY
ls -la /tmp/big.fq.gz
R="-r tests/golden/data/all_pave_ref.fa.gz -k 16"
run() {
  local label=$1; shift
  for rep in 1 2; do
    S=$(date +%s.%N); env "$@" RKMH_TIMING=1 timeout -s ABRT 120 bin/rkmh stream $R -f /tmp/big.fq.gz > /tmp/big.out 2>/tmp/big.err || { echo "$label failed"; tail -3 /tmp/big.err; return; }; E=$(date +%s.%N)
    python3 -c "print('%-40s %d M reads: wall %.3f s = %.1f M reads/s;  %s' % ('$label', $N // 1000000, $E - $S, $N / ($E - $S) / 1e6, '$(grep "main loop" /tmp/big.err | tr -s " ")'))"
  done
}
run "host inflate" RKMH_BGZF_DEVICE=0
run "device inflate, 8 workers" RKMH_BGZF_DEVICE=1
run "device inflate, 12 workers" RKMH_BGZF_DEVICE=1 RKMH_BGZF_DEVICE_WORKERS=12
run "device inflate, 8 workers, two slots" RKMH_BGZF_DEVICE=1 RKMH_BGZF_DEVICE_SLOTS=2
run "both (6 device workers)" RKMH_BGZF_DEVICE=2
run "by size" RKMH_X=1
