#!/bin/bash
# Usage: bash tools/sweep_env.sh VAR v1 v2 ...   -- bench.py (kernel ms) for each value of one env knob
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  printf "%s=%s " $VAR $v
  timeout 120 python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g ms/step %.4f kernel_frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
done
