#!/bin/bash
# Usage (GPU box): [N=16000000] bash tools/profile_inflate_pmc.sh [tag] -> gpurun_out/<tag>_inflate_pmc.txt, <tag>_inflate_kernel_stats.csv
# bin/rkmh stream on ONE BGZF file of N reads with one device worker (launches do not overlap): kernel times (rocprofv3 --kernel-trace --stats) and, in
# separate --pmc passes, waves / instructions / cycles per launch of the two inflate kernels -- SQ_WAVES per launch is the evidence that a launch fills the chip.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
TAG=${1:-r06}
N=${N:-16000000}
[ -f /tmp/big.fq.gz ] || N=$N QUICK=1 bash tools/gz_e2e.sh pmc_prep > /dev/null 2>&1
OUT=$ROOT/gpurun_out/${TAG}_inflate_pmc.txt
cd /tmp; export TMPDIR=/tmp
CMD="$ROOT/bin/rkmh stream -r $ROOT/tests/golden/data/all_pave_ref.fa.gz -f /tmp/big.fq.gz -k 16"
rm -rf /tmp/pinf
RKMH_BGZF_DEVICE_WORKERS=1 RKMH_BGZF_JOB_KB=1048576 RKMH_SLOW_EXIT=1 RKMH_BGZF_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/pinf -o p --output-format csv -- $CMD > /dev/null 2> /tmp/pinf.err
f=$(find /tmp/pinf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $ROOT/gpurun_out/${TAG}_inflate_kernel_stats.csv
echo "# one device worker, jobs of up to 1 GiB of text ($(ls -la /tmp/big.fq.gz | awk '{print $5}') compressed bytes, $N reads); [bgzf device] lines of the traced run:" > $OUT
grep "bgzf device" /tmp/pinf.err | head -6 >> $OUT
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pinf
  RKMH_BGZF_DEVICE_WORKERS=1 RKMH_BGZF_JOB_KB=1048576 RKMH_SLOW_EXIT=1 timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pinf -o p --output-format csv -- $CMD > /dev/null 2> /tmp/pinf.err
  python3 - >> $OUT <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/pinf/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_inflate" in n or "k_crc32" in n:
            acc[(n.split("(")[0][-44:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (n, c), v in sorted(acc.items()):
    print("%-46s %-24s launches=%3d mean per launch %.5g  max %.5g" % (n, c, len(v), sum(v) / len(v), max(v)))
PY
done
cat $OUT
cut -c1-160 $ROOT/gpurun_out/${TAG}_inflate_kernel_stats.csv | head -8
