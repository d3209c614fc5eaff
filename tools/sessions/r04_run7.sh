#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_fastq.py -x -q > gpurun_out/r04_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t7.log
tail -5 gpurun_out/r04_t7.log
timeout 1500 python3 tools/e2e_sweep.py > gpurun_out/r04_e2e_sweep.txt 2>&1
cat gpurun_out/r04_e2e_sweep.txt
bash tools/profile_masked.sh r04 > /dev/null 2>&1
cat gpurun_out/r04_M2_times.txt gpurun_out/r04_M2_pmc.txt
