#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_json_contract" 2>&1 | tail -15
( time python3 bench.py > gpurun_out/r04_c_bench.json 2> gpurun_out/r04_c_bench.err ) 2>&1 | tail -4
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_c_bench.json').read().strip().split('\n')[-1])
print('value %.4g  kernel_ms %.4f  frac %.4f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))
print(json.dumps(d['c4_filter'].get('full_size'), indent=1))"
