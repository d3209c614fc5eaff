#!/bin/bash
# Usage (GPU box): bash tools/profile_inflate_pmc.sh  -> gpurun_out/r05_inflate_pmc.txt: instruction and cycle counters of the device inflate kernels
# (bin/rkmh stream on a BGZF file, RKMH_BGZF_DEVICE=1, one worker so that launches do not overlap)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
[ -f /tmp/pi.fq.gz ] || python3 - <<PY
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
: > $ROOT/gpurun_out/r05_inflate_pmc.txt
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pinf
  RKMH_BGZF_DEVICE=1 RKMH_RAW_WORKERS=1 RKMH_SLOW_EXIT=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pinf -o pinf --output-format csv -- $ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -f /tmp/pi.fq.gz -k 16 > /dev/null 2> /tmp/pinf.err
  python3 - >> $ROOT/gpurun_out/r05_inflate_pmc.txt <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/pinf/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_inflate" in n:
            acc[(n.split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (n, c), v in sorted(acc.items()):
    print("%-42s %-24s launches=%3d mean per launch %.5g" % (n, c, len(v), sum(v) / len(v)))
PY
done
cat $ROOT/gpurun_out/r05_inflate_pmc.txt
