#!/bin/bash
# Usage (GPU box): bash tools/profile_inflate.sh   -> gpurun_out/r05_inflate_kernel_stats.csv: kernel times of bin/rkmh stream on a BGZF file with RKMH_BGZF_DEVICE=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, "$ROOT")
from rkmh_amd import api, synth
refs = api.parse_files([os.path.join("$ROOT", "tests/golden/data/all_pave_ref.fa.gz")])
rb, ro = refs["bases"], refs["offsets"]
n, L = 2000000, 150
qb, _ = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=16)
rec = np.empty((n, 11 + L + 3 + L + 1), dtype=np.uint8)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 11 + L] = 10; rec[:, 10] = 10
idx = np.arange(n, dtype=np.int64)
for d in range(9):
    rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
rec[:, 11:11 + L] = qb[: n * L].reshape(n, L)
rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
rec[:, 14 + L:14 + 2 * L] = np.random.default_rng(0).integers(35, 75, size=(n, L), dtype=np.uint8); rec[:, 14 + 2 * L] = 10
open("/tmp/pi.fq.gz", "wb").write(synth.bgzf_compress(rec.tobytes(), level=1, threads=16))
PY
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pinf
RKMH_BGZF_DEVICE=1 RKMH_SLOW_EXIT=1 RKMH_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/pinf -o pinf --output-format csv -- $ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -f /tmp/pi.fq.gz -k 16 > /dev/null 2> /tmp/pinf.err
grep "rkmh timing" /tmp/pinf.err | tail -3
cp /tmp/pinf/*kernel_stats.csv $ROOT/gpurun_out/r05_inflate_kernel_stats.csv 2>/dev/null || cp /tmp/pinf/*/*kernel_stats.csv $ROOT/gpurun_out/r05_inflate_kernel_stats.csv
head -12 $ROOT/gpurun_out/r05_inflate_kernel_stats.csv | cut -c1-200
