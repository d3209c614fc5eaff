#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
SWEEP_AB=1 timeout 1500 python3 tools/e2e_sweep.py > gpurun_out/r04_e2e_ab.txt 2>&1
cat gpurun_out/r04_e2e_ab.txt
