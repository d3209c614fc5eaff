#!/bin/bash
# the whole GPU suite while another process keeps the GPU busy: timing-dependent faults (in the tests or the product) show under contention
cd "$(dirname "$0")/../.."
setsid bash -c 'for j in $(seq 1 40); do python3 bench.py --steps 3000 --warmup 5 --cpu-seconds 0 --no-host-path --no-depth-filter --no-configs --e2e-reads 0 > /dev/null 2>&1; done' &
BG=$!   # (its own process group: ended as a group below, by number)
timeout 2700 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -30
kill -- -$BG 2>/dev/null
