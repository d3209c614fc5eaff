#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_fastq.py -x -q > gpurun_out/r04_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t8.log
tail -5 gpurun_out/r04_t8.log
df -h /tmp | tail -1; mount | grep -E " /tmp | / " | head -3
SWEEP_SMALL=1 timeout 1500 python3 tools/e2e_sweep.py > gpurun_out/r04_e2e_sweep2.txt 2>&1
cat gpurun_out/r04_e2e_sweep2.txt
