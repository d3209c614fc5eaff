#!/bin/bash
# is test_every_kernel_form_at_scale[ks5-False--150] flaky?  (it failed once under unrelated environment settings)
cd "$(dirname "$0")/../.."
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "test_every_kernel_form_at_scale and ks5" 2>&1 | grep -E "passed|failed|assert|AssertionError|E  " | head -12
done
echo "--- with the environment of the failing run"
for i in 1 2 3 4; do
  RKMH_RAW_SLOTS=2 RKMH_OUT_DIRECT=0 RKMH_RAW_BLOCK_KB=512 timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "test_every_kernel_form_at_scale and ks5" 2>&1 | grep -E "passed|failed|assert|AssertionError|E  " | head -12
done
