#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
free -g | head -2
timeout 1500 python3 tools/bench_filter.py 3100 10000000 2>&1 | tee gpurun_out/r04_c4_fullsize.txt
