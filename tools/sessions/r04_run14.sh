#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python3 tools/c3_probe.py 2>/dev/null | tee gpurun_out/r04_c3_probe2.txt
python3 bench.py --steps 50 --warmup 10 --cpu-seconds 3 --no-host-path --no-depth-filter --e2e-reads 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['kernel_ms'], d['c3_panel'])"
timeout 1500 python3 -m pytest tests -m gpu -q -x > gpurun_out/r04_t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t14.log
tail -4 gpurun_out/r04_t14.log
