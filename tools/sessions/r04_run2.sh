#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_rates.hip -o /tmp/ubench_valu && /tmp/ubench_valu > gpurun_out/r04_valu_rates.txt 2>&1
cat gpurun_out/r04_valu_rates.txt
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "bench or rccl or minhashes" > gpurun_out/r04_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t2.log
tail -15 gpurun_out/r04_t2.log
