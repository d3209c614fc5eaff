#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_fastq.py -x -q > gpurun_out/r04_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t5.log
tail -30 gpurun_out/r04_t5.log
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs > gpurun_out/r04_e2e_1.json 2> gpurun_out/r04_e2e_1.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_e2e_1.json').read().strip().split('\n')[-1]); print(json.dumps(d.get('e2e'), indent=1)); print(d['value'], d['roofline']['kernel_ms'])"
tail -5 gpurun_out/r04_e2e_1.err
