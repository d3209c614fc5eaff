#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/profile.sh <tag>
# kernel-trace stats + separate PMC passes for bench.py; summaries land in gpurun_out/<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs --e2e-reads 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
