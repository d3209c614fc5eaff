#!/bin/bash
# hpv16 with the emit spread over the CPUs: parity tests, then 2 M reads end to end
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hpv16" 2>&1 | tail -4
python3 tools/make_fastq.py /tmp/h2m.fq 2000000
t0=$(date +%s.%N); RKMH_TIMING=1 bin/rkmh hpv16 -f /tmp/h2m.fq -R tests/golden/data > /tmp/h.out 2> /tmp/h.err; t1=$(date +%s.%N)
python3 -c "print('hpv16: 2000000 reads in %.2f s = %.2f M reads/s' % ($t1 - $t0, 2.0 / ($t1 - $t0)))"; grep "rkmh timing" /tmp/h.err | tr '\n' '|'; echo; wc -l /tmp/h.out
