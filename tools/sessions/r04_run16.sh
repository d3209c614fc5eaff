#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python3 tools/c3_probe.py 2>/dev/null | tee gpurun_out/r04_c3_probe3.txt
timeout 1500 python3 -m pytest tests -m gpu -q -x > gpurun_out/r04_t16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t16.log
tail -4 gpurun_out/r04_t16.log
