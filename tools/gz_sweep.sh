cd $GRAFT_REPO_ROOT
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from rkmh_amd import api, synth
refs = api.parse_files(["tests/golden/data/all_pave_ref.fa.gz"])
rb, ro = refs["bases"], refs["offsets"]
n, L = 8000000, 150
with open("/tmp/sw.fq.gz", "wb") as fb:
    for lo in range(0, n, 1000000):
        m = 1000000
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
IFS=";" read -ra LIST <<< "${CFGS:-16384 8;65536 8;131072 8;131072 12;262144 8;262144 12;524288 8}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  for rep in 1 2; do
    S=$(date +%s.%N)
    RKMH_BGZF_DEVICE=1 RKMH_RAW_BLOCK_KB=$1 RKMH_RAW_WORKERS=$2 bin/rkmh stream -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz -k 16 > /tmp/sw.out 2>/dev/null
    E=$(date +%s.%N)
    python3 -c "print(\"device inflate block_kb=$1 workers=$2: 16 M reads in %.3f s\" % ($E - $S))"
  done
done
S=$(date +%s.%N); bin/rkmh stream -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/sw.fq.gz -f /tmp/sw.fq.gz -k 16 > /tmp/sw.out 2>/dev/null; E=$(date +%s.%N); python3 -c "print(\"host inflate (default): 16 M reads in %.3f s\" % ($E - $S))"
