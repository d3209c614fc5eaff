#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_fastq.py -x -q > gpurun_out/r04_t13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_t13.log
tail -4 gpurun_out/r04_t13.log
python3 tools/c3_probe.py 2>/dev/null | tee gpurun_out/r04_c3_probe.txt
for rep in 1 2; do
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(json.dumps(d.get('e2e'), indent=1))" | grep -E "wall_s|marginal|front end:|main loop"
done
for W in 4 6 10; do echo "workers $W"; RKMH_RAW_WORKERS=$W python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(json.dumps(d.get('e2e'), indent=1))" | grep -E "wall_s|marginal|front end:"; done
python3 tools/make_fastq.py /tmp/prof4m.fq 4000000
cd /tmp; export TMPDIR=/tmp RKMH_SLOW_EXIT=1
rm -rf /tmp/fqprof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fqprof -o fq -- $GRAFT_REPO_ROOT/bin/rkmh stream -r $GRAFT_REPO_ROOT/tests/golden/data/all_pave_ref.fa.gz -f /tmp/prof4m.fq -k 16 -s 1000 > /tmp/prof4m.tsv 2> /tmp/prof4m.err
cd $GRAFT_REPO_ROOT
find /tmp/fqprof -name "*kernel_stats.csv" | head; cp $(find /tmp/fqprof -name "*kernel_stats.csv" | head -1) gpurun_out/r04_fastq_kernel_stats.csv
head -16 gpurun_out/r04_fastq_kernel_stats.csv | cut -c1-150
