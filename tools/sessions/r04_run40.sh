#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1500 python3 -m pytest tests/test_gpu_fastq.py -q -m gpu -k "cli" 2>&1 | tail -3
for rep in 1 2 3; do python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); e=d['e2e']; print('e2e wall %.3f s' % e['wall_s'], e.get('x4',{}).get('marginal_reads_per_s'), e['stages'][-3:])"; done
