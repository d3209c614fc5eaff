#!/bin/bash
# Usage (GPU box): bash tools/e2e_timing.sh [reads=16000000]   -- stage timings of bin/rkmh stream on a generated FASTQ (RKMH_TIMING=1)
N=${1:-16000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python3 - <<PY
import os, sys, time, numpy as np
sys.path.insert(0, "$ROOT")
from rkmh_amd import api, synth
import bench
refs = api.parse_files([os.path.join("$ROOT", "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
n, L = $N, 150
t = time.time()
with open("/tmp/e2e_reads.fq", "wb") as f:
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
print("generated", n, "reads in %.1f s" % (time.time() - t))
PY
# E2E_VARIANTS="NAME=V,NAME2=V2 NAME=W ...": one timed pair of runs per entry (comma-separated environment settings)
for v in ${E2E_VARIANTS:-default}; do
  echo "== $v"
  envs=$(echo $v | tr ',' ' '); [ "$v" = default ] && envs=""
  for rep in 1 2; do
    time env RKMH_TIMING=1 $envs bin/rkmh stream -r tests/golden/data/all_pave_ref.fa.gz -f /tmp/e2e_reads.fq -k 16 -s 1000 > /tmp/e2e_out.tsv
  done
done
wc -l /tmp/e2e_out.tsv; nproc; cat /sys/fs/cgroup/cpu.max
rm -f /tmp/e2e_reads.fq /tmp/e2e_out.tsv
