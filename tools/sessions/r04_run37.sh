#!/bin/bash
# hunt for the rare failure of test_every_kernel_form_at_scale[ks5-False--150]: the test in a loop while another process keeps the GPU busy
cd "$(dirname "$0")/../.."
setsid bash -c 'for j in 1 2 3 4 5 6; do python3 bench.py --steps 2000 --warmup 5 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs --e2e-reads 0 > /dev/null 2>&1; done' &
BG=$!   # (its own process group: ended as a group below, by number)
fails=0
for i in $(seq 1 20); do
  out=$(timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "test_every_kernel_form_at_scale and (ks5 or ks9 or ks0 or ks6)" 2>&1)
  if echo "$out" | grep -q "failed"; then fails=$((fails+1)); echo "$out" | grep -E "^E  |failed|FAILED" | head -30; fi
done
echo "loops with a failure: $fails of 20"
kill -- -$BG 2>/dev/null
