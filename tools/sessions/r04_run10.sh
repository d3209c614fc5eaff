#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_fastq.py -x -q > gpurun_out/r04_t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t10.log
tail -5 gpurun_out/r04_t10.log
for rep in 1 2; do
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs > gpurun_out/r04_e2e_4.json 2> gpurun_out/r04_e2e_4.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_e2e_4.json').read().strip().split('\n')[-1]); print(json.dumps(d.get('e2e'), indent=1))" | grep -E "wall_s|value|marginal|front end:|main loop"
done
for W in 6 10 12; do echo "workers $W"; RKMH_RAW_WORKERS=$W python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(json.dumps(d.get('e2e'), indent=1))" | grep -E "wall_s|marginal|front end:"; done
