#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r04_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t4.log
tail -40 gpurun_out/r04_t4.log
