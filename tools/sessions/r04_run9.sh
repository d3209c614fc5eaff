#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for rep in 1 2; do
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs > gpurun_out/r04_e2e_3.json 2> gpurun_out/r04_e2e_3.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_e2e_3.json').read().strip().split('\n')[-1]); print(json.dumps(d.get('e2e'), indent=1))"
done
